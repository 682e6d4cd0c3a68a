#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "whole_model or preloop or model_options or batch_consistency or reduced_precision_mode" > gpurun_out/r06_check6_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r06_check6_tests.log
for i in 1 2; do
timeout -k 10 200 python3 tools/pass_phases.py --reps 7 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['pass_us'], d['pre_loop_us'], d['us_per_iter'], d['post_loop_us'], d['markers_us'])"
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --no-batched --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_gru_iter'], d['value_spread']['pairs_per_s'])"
