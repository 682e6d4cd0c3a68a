#!/bin/bash
# round-5 session 4 (GPU box): schedule knobs of the window between gru04 q(i) and gru04 z|r(i+1); library convs of the training step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
tools/ab_env_bench.sh 3 "AS_X=0" "AS_CONV_LEAN=2" "AS_CONV_KSPLIT_MAX=1" "AS_CONV_KSPLIT_MAX=2" > gpurun_out/r05_s4_bench.txt 2>&1
echo bench done
timeout -k 10 400 python tools/train_lib_convs.py --steps 2 > gpurun_out/r05_s4_train_lib_convs.txt 2>&1
echo "train_lib_convs rc=$?"
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "model_options or resamplers" > gpurun_out/r05_s4_pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r05_s4_pytest.txt
