#!/bin/bash
# round-5 session 8 (GPU box): depthwise 3x3 training kernels — parity, G8 step, cfg-4 step time with and without them
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "dwconv3x3_backward or training_step_vs_reference or training_step_is_bit" > gpurun_out/r05_s8_pytest.txt 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r05_s8_pytest.txt
for r in 1 2; do
for v in 1 0; do
  ANYSTEREO_TRAIN_DWCONV=$v timeout -k 10 300 python bench.py --mode train --train-quick --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TRAIN_DWCONV=$v', d['value'], d['ms_per_step'], d['loss_first_last'], d.get('grad_bytes'), d.get('exchange_ms'))"
done; done > gpurun_out/r05_s8_train.txt 2>&1
cat gpurun_out/r05_s8_train.txt
