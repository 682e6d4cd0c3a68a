#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "whole_model or preloop or model_options or batch_consistency" > gpurun_out/r06_check12_tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r06_check12_tests.log
for v in "ANYSTEREO_FUSED_IR=0" "ANYSTEREO_FUSED_IR=1" "ANYSTEREO_FUSED_IR=0" "ANYSTEREO_FUSED_IR=1"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 tools/pass_phases.py --reps 7 --stages 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d['markers_us']; print(d['pass_us'], d['pre_loop_us'], d['us_per_iter'], {k: m[k] for k in ('trunk_block0','trunk_block1','trunk_block2','trunk_block3','trunk_block4','trunk_end','context_end','cost_agg_end')})"
done
