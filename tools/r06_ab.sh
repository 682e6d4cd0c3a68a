#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of environment switches through the in-graph phase markers.
#   tools/r06_ab.sh "VAR=a" "VAR=b" ...        (each setting measured twice, interleaved)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 tools/pass_phases.py --reps 7 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d['markers_us']; print(d['pass_us'], d['pre_loop_us'], d['us_per_iter'], {k: m[k] for k in ('context_end','trunk_end','cost_agg_end') if k in m})"
done; done
