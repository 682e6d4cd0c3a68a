#!/bin/bash
# round-5 session 6 (GPU box): MFMA shape under power limit with both consumer loops software-pipelined; naive MIOpen solvers of cfg 4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape2 tools/experiments/mfma_shape2.hip 2>/dev/null
timeout -k 10 120 /tmp/mfma_shape2 > gpurun_out/r05_s6_mfma_shape2.txt 2>&1
echo "mfma_shape2 rc=$?"; cat gpurun_out/r05_s6_mfma_shape2.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape tools/experiments/mfma_shape.hip 2>/dev/null
timeout -k 10 120 /tmp/mfma_shape > gpurun_out/r05_s6_mfma_shape_r03form.txt 2>&1
cat gpurun_out/r05_s6_mfma_shape_r03form.txt
tools/find_naive_convs.sh > gpurun_out/r05_s6_find_naive.log 2>&1
echo "find_naive rc=$?"; tail -30 gpurun_out/r05_s6_find_naive.log
