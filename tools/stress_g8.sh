#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the G8 stress matrix of VERDICT r3 item 1 -> gpurun_out/stress_<tag>/
#   tools/stress_g8.sh <tag> [full]     ("full" adds the MIOpen solver-family matrix and the solver log of lease a)
set -u
TAG=${1:-lease}
FULL=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stress_$TAG
mkdir -p $OUT
cd $ROOT
run() {  # run <label> <env...> -- <args...>
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  echo "=== $label: ${envs[*]:-} $*" | tee -a $OUT/summary.txt
  env "${envs[@]}" timeout -k 10 500 python3 tools/stress_g8.py "$@" --out $OUT/$label.json > $OUT/$label.log 2>&1
  echo "exit $?" >> $OUT/$label.log
  grep -a "^\[stress\]" $OUT/$label.log | grep -v '"host"' | tee -a $OUT/summary.txt
}
rocm-smi --showproductname > $OUT/box.txt 2>&1
hostname >> $OUT/box.txt
lscpu | grep -i "model name" >> $OUT/box.txt
# 30 repetitions in ONE process, then 5 fresh processes
run raft_split -- --name raft --mode split --reps 30 --truth
for i in 1 2 3 4 5; do run raft_split_fresh$i -- --name raft --mode split --reps 1; done
run raft_fp32 -- --name raft --mode fp32 --reps 3
run igev_split -- --name igev --mode split --reps 3
run igev_fp32 -- --name igev --mode fp32 --reps 3
# discriminators: uninitialised reads (NaN-filled allocations), races (every kernel launch serialised and blocking)
run raft_split_nanfill -- --name raft --mode split --reps 2 --nanfill
run igev_split_nanfill -- --name igev --mode split --reps 2 --nanfill
run raft_split_serial AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 -- --name raft --mode split --reps 2
# what an eps-level change of the forward does to the gradients (the fp64 CPU oracle shows the same quantised jumps)
run raft_split_perturb1e-6 -- --name raft --mode split --reps 12 --perturb 1e-6
if [ "$FULL" = "full" ]; then
  run raft_split_nowino MIOPEN_DEBUG_CONV_WINOGRAD=0 -- --name raft --mode split --reps 2
  run raft_split_nodirect MIOPEN_DEBUG_CONV_DIRECT=0 -- --name raft --mode split --reps 2
  run raft_split_nogemm MIOPEN_DEBUG_CONV_GEMM=0 -- --name raft --mode split --reps 2
  run raft_split_noigemm MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0 -- --name raft --mode split --reps 2
  MIOPEN_LOG_LEVEL=6 MIOPEN_ENABLE_LOGGING=1 timeout -k 10 500 python3 tools/stress_g8.py --name raft --mode split --reps 1 2>&1 | grep -a -i "solver\|solution\|algorithm" | cut -c1-300 | sort | uniq -c | sort -rn | head -200 > $OUT/miopen_solvers.txt
fi
echo done | tee -a $OUT/summary.txt
