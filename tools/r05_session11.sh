#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape2 tools/experiments/mfma_shape2.hip 2>/dev/null
timeout -k 10 200 /tmp/mfma_shape2 > gpurun_out/r05_s11_mfma_shape2.txt 2>&1
echo "rc=$?"; cat gpurun_out/r05_s11_mfma_shape2.txt
