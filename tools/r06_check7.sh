#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in "X=1" "ANYSTEREO_TRUNK_FIRST=1 ANYSTEREO_CONTEXT_AFTER_STAGE=block0" "ANYSTEREO_TRUNK_FIRST=1 ANYSTEREO_CONTEXT_AFTER_STAGE=block1" "ANYSTEREO_TRUNK_FIRST=1 ANYSTEREO_CONTEXT_AFTER_STAGE=block2" "ANYSTEREO_TRUNK_FIRST=1 ANYSTEREO_CONTEXT_AFTER_STAGE=block3" "ANYSTEREO_TRUNK_FIRST=1 ANYSTEREO_CONTEXT_AFTER_STAGE=block4" "X=1"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 tools/pass_phases.py --reps 7 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['pass_us'], d['pre_loop_us'], d['us_per_iter'], d['markers_us'])"
done
