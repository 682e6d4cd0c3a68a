#!/bin/bash
# round-5 session 12 (GPU box): data-gradient weight pack straight from the forward weight — parity, G8, cfg-4 step time vs the round's previous build
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 700 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "dgrad_pack or backward or wgrad or training_step or trainer_graphed or mlp_tail" > gpurun_out/r05_s12_pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r05_s12_pytest.txt
for r in 1 2 3; do
  timeout -k 10 300 python bench.py --mode train --train-quick --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dgrad-pack', d['value'], d['ms_per_step'], d['loss_first_last'])"
done > gpurun_out/r05_s12_train.txt 2>&1
cat gpurun_out/r05_s12_train.txt
