#!/bin/bash
# Run ON THE GPU BOX: alternate bench.py between library builds on the same box.  tools/ab_bench.sh <rounds> <name|cur> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
export ANYSTEREO_ALLOW_STALE_LIB=1  # variant libraries carry their revision, not the tree's source hash
for i in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = cur ]; then unset ANYSTEREO_LIB; else export ANYSTEREO_LIB=$ROOT/any-stereo_amd/anystereo/lib/$v.so; fi
    python $ROOT/bench.py --no-cpu-baseline --no-extras --no-batched --steps 10 --warmup 3 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['ms_per_gru_iter'], {k: v['avg_us'] for k, v in d['rooflines'].items() if 'conv' in k})"
  done
done
