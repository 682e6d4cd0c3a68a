#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3q; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*"; ( env "$@" timeout -k 10 300 python $ROOT/tools/train_graph_check.py --sync none --steps 12 --fast > $OUT/$name.log 2>&1; echo "rc=$?" >> $OUT/$name.log ); grep -v "amdgpu\|Warning\|run_backward" $OUT/$name.log | grep "replay\|rc=\|Error\|error\|sync=" | cut -c60-190 | tr '\n' '|'; echo; }
run no_wrw_gtc_nhwc MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0
run no_wrw_gtc_both MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS=0
run no_igemm MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0
