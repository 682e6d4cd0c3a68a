#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -x -q -k "liif or whole_model or whole_forward" > gpurun_out/r06_check5_tests.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r06_check5_tests.log
for v in 0 1 0 1; do
  echo "== ANYSTEREO_LIIF_PATCH_ORDER=$v"
  ANYSTEREO_LIIF_PATCH_ORDER=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --no-batched --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_gru_iter'], d['kernel_times_us']['liif_tail'])"
done
for c in cfg3 cfg5; do for v in 0 1; do
  echo "== $c ANYSTEREO_LIIF_PATCH_ORDER=$v"
  ANYSTEREO_LIIF_PATCH_ORDER=$v timeout -k 10 300 python3 bench.py --config $c --no-cpu-baseline --no-extras --no-batched --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_times_us']['liif_tail'])"
done; done
