#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "whole_model or preloop or model_options or batch_consistency" > gpurun_out/r06_check10_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r06_check10_tests.log
for v in "ANYSTEREO_EARLY_GATES=0" "ANYSTEREO_EARLY_GATES=1" "ANYSTEREO_EARLY_GATES=0" "ANYSTEREO_EARLY_GATES=1"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 tools/pass_phases.py --reps 7 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['pass_us'], d['pre_loop_us'], d['us_per_iter'], {k: d['markers_us'][k] for k in ('context_end','trunk_end','cost_agg_end')})"
done
timeout -k 10 200 python3 tools/pass_phases.py --reps 5 --stages 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(sorted(d['markers_us'].items(), key=lambda kv: kv[1]))"
