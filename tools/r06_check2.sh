#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "autocast_training or preloop_tensors" > gpurun_out/r06_check2_tests.log 2>&1
echo "pytest rc=$?"; grep -n "preloop hip\|passed\|failed\|Error" gpurun_out/r06_check2_tests.log | tail -20
timeout -k 10 300 python tools/pass_phases.py --reps 5 > gpurun_out/r06_phases_fine.json 2> gpurun_out/r06_phases_fine.txt
echo "phases rc=$?"; cat gpurun_out/r06_phases_fine.txt | grep -v amdgpu.ids; cat gpurun_out/r06_phases_fine.json | cut -c1-600
timeout -k 10 600 python tools/g8_margins.py gpurun_out/r06_g8_margins.json > gpurun_out/r06_g8_margins.txt 2>&1
echo "g8 rc=$?"; grep -v amdgpu.ids gpurun_out/r06_g8_margins.txt | tail -8
