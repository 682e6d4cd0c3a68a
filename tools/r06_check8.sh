#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "conv3d or whole_model or preloop or backbone_blocks" > gpurun_out/r06_check8_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r06_check8_tests.log
for v in "ANYSTEREO_CONV3D_MFMA=0" "ANYSTEREO_CONV3D_MFMA=1" "ANYSTEREO_CONV3D_MFMA=1 ANYSTEREO_CONV3D_MFMA_MAX_VOXELS=200000" "ANYSTEREO_CONV3D_MFMA=0" "ANYSTEREO_CONV3D_MFMA=1"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 tools/pass_phases.py --reps 7 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['pass_us'], d['pre_loop_us'], d['us_per_iter'], {k: d['markers_us'][k] for k in ('context_end','trunk_end','cost_agg_end')})"
done
