#!/bin/bash
# round-5 session 10 (GPU box): 64-channel tiles without K split for fp32-source convolutions (AS_CONV_PREFER64)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for r in 1 2; do
for v in 1 0; do
  AS_CONV_PREFER64=$v timeout -k 10 300 python bench.py --mode train --train-quick --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PREFER64=$v', d['value'], d['ms_per_step'], d['loss_first_last'])"
done; done > gpurun_out/r05_s10_train.txt 2>&1
cat gpurun_out/r05_s10_train.txt
tools/ab_env_bench.sh 2 "AS_CONV_PREFER64=1" "AS_CONV_PREFER64=0" > gpurun_out/r05_s10_bench.txt 2>&1
cut -c1-60 gpurun_out/r05_s10_bench.txt
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "conv or gru or update" > gpurun_out/r05_s10_pytest.txt 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r05_s10_pytest.txt
