#!/bin/bash
# round-5 session 9 (GPU box): frozen-BatchNorm fold in the training step — parity, G8, step time with and without
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "conv_frozen_bn or training_step_vs_reference or training_step_is_bit or trainer_graphed" > gpurun_out/r05_s9_pytest.txt 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r05_s9_pytest.txt
for r in 1 2; do
for v in 1 0; do
  ANYSTEREO_TRAIN_FOLD_BN=$v timeout -k 10 300 python bench.py --mode train --train-quick --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TRAIN_FOLD_BN=$v', d['value'], d['ms_per_step'], d['loss_first_last'])"
done; done > gpurun_out/r05_s9_train.txt 2>&1
cat gpurun_out/r05_s9_train.txt
