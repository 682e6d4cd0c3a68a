#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
tools/ab_env_bench.sh 3 "AS_X=0" "AS_CONV_LEAN=3" > gpurun_out/r05_s7_bench.txt 2>&1
cut -c1-400 gpurun_out/r05_s7_bench.txt
