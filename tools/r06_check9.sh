#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in "X=1" "ANYSTEREO_FUSED_FRONT=full" "ANYSTEREO_FUSED_FRONT=lite" "X=1" "ANYSTEREO_FUSED_FRONT=full" "ANYSTEREO_EARLY_GRU16=0" "ANYSTEREO_EARLY_INTERP16=0"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 tools/pass_phases.py --reps 7 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['pass_us'], d['pre_loop_us'], d['us_per_iter'], d['post_loop_us'])"
done
