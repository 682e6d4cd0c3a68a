#!/bin/bash
# round-5 session 1 (GPU box): consumer operand-read placement variants of conv_split_kernel, isolated kernels then the loop
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export ANYSTEREO_ALLOW_STALE_LIB=1
K="gru08_zr_bs gru08_q gru16_zr_bs gru16_q gru04_zr gru04_q head_conv1 enc_conv enc_c2d2"
for r in 1 2; do
  tools/ab_kbench.sh "$K" x_base x_e1 cur x_e2x x_e2xx
done > gpurun_out/r05_s1_kbench.txt 2>&1
echo kbench done
L=$ROOT/any-stereo_amd/anystereo/lib
tools/ab_env_bench.sh 2 "ANYSTEREO_LIB=$L/x_base.so" "ANYSTEREO_LIB=$L/libanystereo_hip.so" "ANYSTEREO_LIB=$L/x_e2x.so" "ANYSTEREO_LIB=$L/x_e2xx.so" > gpurun_out/r05_s1_bench.txt 2>&1
echo bench done
