#!/bin/bash
# round-5 session 2 (GPU box): one operand read per MFMA gap (x_il*) against the round-4 consumer loop; traffic-only lookup
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export ANYSTEREO_ALLOW_STALE_LIB=1
K="gru08_zr_bs gru08_q gru16_zr_bs gru16_q gru04_zr gru04_q head_conv1 enc_conv enc_c2d2"
for r in 1 2; do
  tools/ab_kbench.sh "$K" x_base x_il x_il4 x_il5
done > gpurun_out/r05_s2_kbench.txt 2>&1
echo kbench done
L=$ROOT/any-stereo_amd/anystereo/lib
for r in 1 2 3; do
for v in cur lk_traffic; do
  if [ "$v" = cur ]; then unset ANYSTEREO_LIB; else export ANYSTEREO_LIB=$L/$v.so; fi
  echo "== $v"
  python tools/kbench.py lookup_convc1 --reps 200 --graph 2>&1 | grep "us/launch"
  python tools/kbench.py lookup_convc1 --reps 200 --graph --cfg 5 2>&1 | grep "us/launch"
done; done > gpurun_out/r05_s2_lookup.txt 2>&1
unset ANYSTEREO_LIB
echo lookup done
tools/ab_env_bench.sh 3 "ANYSTEREO_LIB=$L/x_base.so" "ANYSTEREO_LIB=$L/x_il.so" "ANYSTEREO_LIB=$L/x_il4.so" > gpurun_out/r05_s2_bench.txt 2>&1
echo bench done
