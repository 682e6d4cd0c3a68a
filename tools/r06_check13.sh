#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 300 python3 tools/kbench_ir.py 2>/dev/null
for i in 1 2 3; do
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "model_options" 2>&1 | tail -2
done
