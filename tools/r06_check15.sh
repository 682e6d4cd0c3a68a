#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in "X=1" "ANYSTEREO_PIPELINED_LOOP=0" "SERIAL=1" "ANYSTEREO_EARLY_GRU16=0" "ANYSTEREO_PARALLEL_CONTEXT=0" "ANYSTEREO_LIIF_EARLY_STATIC=0" "PYTORCH_NO_CUDA_MEMORY_CACHING=1"; do
  env $v timeout -k 10 120 python3 tools/race_probe.py raft fp32 2>/dev/null
done
env X=1 timeout -k 10 120 python3 tools/race_probe.py raft split 2>/dev/null
env X=1 timeout -k 10 120 python3 tools/race_probe.py igev fp32 2>/dev/null
