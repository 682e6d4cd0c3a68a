#!/bin/bash
# round-5 session 3 (GPU box): loader-wave epilogue operand prefetch + interleaved operand reads against the round-4 library
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "conv or gru or update or resampl or head" > gpurun_out/r05_s3_pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r05_s3_pytest.txt
export ANYSTEREO_ALLOW_STALE_LIB=1
K="gru04_zr gru04_q head_conv1 gru08_zr_bs gru08_q gru16_zr_bs gru16_q enc_conv enc_c2d2"
for r in 1 2; do
  tools/ab_kbench.sh "$K" r4 x_nopf cur
  echo "== cur AS_CONV_LEAN=0"
  AS_CONV_LEAN=0 python tools/kbench.py $K --reps 30 2>&1 | grep "us/launch"
done > gpurun_out/r05_s3_kbench.txt 2>&1
echo kbench done
L=$ROOT/any-stereo_amd/anystereo/lib
tools/ab_env_bench.sh 3 "ANYSTEREO_LIB=$L/r4.so" "ANYSTEREO_LIB=$L/libanystereo_hip.so" "AS_CONV_LEAN=0" > gpurun_out/r05_s3_bench.txt 2>&1
echo bench done
python bench.py --no-cpu-baseline --no-batched 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['host'])" > gpurun_out/r05_s3_telemetry.txt 2>&1
ls /sys/class/drm/ >> gpurun_out/r05_s3_telemetry.txt 2>&1
