#!/bin/bash
# Run ON THE GPU BOX: in-graph phase markers of the default forward under the pre-loop scheduling switches (same binary, same box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in "X=1" "ANYSTEREO_PARALLEL_CONTEXT=0" "ANYSTEREO_TRUNK_FIRST=1" "ANYSTEREO_PARALLEL_STEMS=0" "X=1"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 tools/pass_phases.py --reps 5 2>/dev/null
done
