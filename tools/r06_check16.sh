#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in "ANYSTEREO_FUSED_IR=0" "ANYSTEREO_FUSED_IR=auto" "ANYSTEREO_FUSED_IR=0" "ANYSTEREO_FUSED_IR=auto"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 tools/pass_phases.py --reps 7 --stages 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d['markers_us']; print(d['pass_us'], d['pre_loop_us'], d['us_per_iter'], {k: m[k] for k in ('trunk_block0','trunk_block1','trunk_block2','trunk_block3','trunk_block4','trunk_end','context_end','cost_agg_end')})"
done
bash tools/race_hunt.sh | tail -3
