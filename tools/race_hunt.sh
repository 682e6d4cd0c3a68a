#!/bin/bash
# Run ON THE GPU BOX: is the eager RAFT forward run-to-run deterministic on THIS box?  If not: which stage, which switch?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 120 python3 tools/kernel_repeat.py 2>/dev/null
out=$(timeout -k 10 120 python3 tools/race_probe.py raft fp32 2>/dev/null; timeout -k 10 120 python3 tools/race_probe.py raft split 2>/dev/null)
echo "$out"
if echo "$out" | grep -q "e-0[1-9]"; then
  echo "== this box shows the difference: localising"
  timeout -k 10 200 python3 tools/race_probe2.py fp32 2>/dev/null
  timeout -k 10 200 python3 tools/race_probe2.py split 2>/dev/null
  for v in "AMD_SERIALIZE_KERNEL=3" "HIP_LAUNCH_BLOCKING=1" "ANYSTEREO_GRID_CACHE=0" "ANYSTEREO_PAIRED_HEADS=0" "ANYSTEREO_LIIF_PATCH_ORDER=0" "SYNC_BETWEEN=1" "PYTORCH_NO_CUDA_MEMORY_CACHING=1"; do
    env $v timeout -k 10 120 python3 tools/race_probe.py raft fp32 2>/dev/null
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_rh -- python3 tools/race_probe.py raft fp32 > /dev/null 2>&1
  cut -d, -f1,2 gpurun_out/_rh/*/*kernel_stats.csv | cut -c1-110 | head -60
  rm -rf gpurun_out/_rh
else
  echo "== deterministic on this box"
fi
