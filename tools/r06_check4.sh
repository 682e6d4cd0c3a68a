#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "reduced_precision_vs_reference or conv_frozen_bn or training_step_vs_reference or trainer_graphed" > gpurun_out/r06_check4_tests.log 2>&1
echo "pytest rc=$?"; grep -n "G8 .* reduced\|conv_frozen_bn.*d_gamma\|passed\|failed\|Error\|stable tensors\|sensitive tensors" gpurun_out/r06_check4_tests.log | tail -30
timeout -k 10 600 python bench.py --mode train --steps 5 --warmup 4 --train-quick --no-cpu-baseline > gpurun_out/r06_check4_train.json 2> gpurun_out/r06_check4_train.err
echo "train rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r06_check4_train.json')); print(d['value'], d['ms_per_step'], d['reduced_precision'], d['exchange_ms'])"
