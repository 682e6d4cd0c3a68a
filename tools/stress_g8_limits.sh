#!/bin/bash
# Run ON THE GPU BOX: how much of the G8 test's per-tensor limits a fresh lease uses (both models, both modes, fresh process each).
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/g8_limits_$TAG.txt
cd $ROOT
: > $OUT
for nm in "raft split" "raft fp32" "igev split" "igev fp32"; do
  set -- $nm
  timeout -k 10 400 python3 tools/stress_g8.py --name $1 --mode $2 --reps 1 --limits 2>&1 | grep -a "fraction of the G8\|convd1.weight" | sed "s/^/$1 $2: /" >> $OUT
done
cat $OUT
