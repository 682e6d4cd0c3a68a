#!/bin/bash
# Run ON THE GPU BOX: the tests touched this round, then the default bench line (wall clock noted)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "autocast_training or trainer_graphed" > gpurun_out/r06_check1_tests.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r06_check1_tests.log
timeout -k 10 600 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "drivers_launcher" > gpurun_out/r06_check1_tests2.log 2>&1
echo "pytest2 rc=$?"; tail -15 gpurun_out/r06_check1_tests2.log
t0=$(date +%s)
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_check1_line.json 2> gpurun_out/r06_check1_line.err
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python -c "
import json
d=json.load(open('gpurun_out/r06_check1_line.json'))
print(d['value'], d['ms_per_step'], d['ms_per_gru_iter'], d['value_spread']['pairs_per_s'])
print(d['cpu_baseline'])
print((d.get('train_mode') or {}).get('ms_per_step'))"
