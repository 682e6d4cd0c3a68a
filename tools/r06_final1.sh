#!/bin/bash
# round-6 final run 1 (GPU box): run-to-run determinism probe, the whole GPU suite, then the smoke entry point
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/race_hunt.sh > gpurun_out/r06_race_hunt.log 2>&1; tail -4 gpurun_out/r06_race_hunt.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r06_gputest.log
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/r06_smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/r06_smoke.log
