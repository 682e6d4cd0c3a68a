#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "dual or paired or preloop or whole_model or residual_block or conv_blocked" > gpurun_out/r06_check3_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r06_check3_tests.log
for v in "ANYSTEREO_PAIRED_HEADS=0" "ANYSTEREO_PAIRED_HEADS=1" "ANYSTEREO_PAIRED_HEADS=0" "ANYSTEREO_PAIRED_HEADS=1"; do
  echo "== $v"; env $v timeout -k 10 200 python3 tools/pass_phases.py --reps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['pass_us'], d['pre_loop_us'], d['us_per_iter'], {k: d['markers_us'][k] for k in ('stems_end','context_end','trunk_end','cost_agg_end')})"
done
