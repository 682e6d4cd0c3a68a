#!/bin/bash
# round-5 session 5 (GPU box): issue order of the forked pre-loop (trunk_first); cache policy of the epilogue's result stores
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 400 python tools/ab_loop.py trunk_first=0,1 5 cfg2 > gpurun_out/r05_s5_trunk_first.txt 2>&1
echo "rc=$?"; tail -3 gpurun_out/r05_s5_trunk_first.txt
export ANYSTEREO_ALLOW_STALE_LIB=1
K="gru04_zr gru04_q head_conv1 gru08_zr_bs gru08_q enc_conv enc_c2d2"
for r in 1 2; do tools/ab_kbench.sh "$K" cur x_nt x_wt; done > gpurun_out/r05_s5_kbench.txt 2>&1
L=$ROOT/any-stereo_amd/anystereo/lib
tools/ab_env_bench.sh 3 "ANYSTEREO_LIB=$L/libanystereo_hip.so" "ANYSTEREO_LIB=$L/x_nt.so" "ANYSTEREO_LIB=$L/x_wt.so" > gpurun_out/r05_s5_bench.txt 2>&1
echo bench done
