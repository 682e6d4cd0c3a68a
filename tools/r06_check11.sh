#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ir_block or backbone_blocks" > gpurun_out/r06_check11_tests.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/r06_check11_tests.log
