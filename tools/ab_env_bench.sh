#!/bin/bash
# Run ON THE GPU BOX: alternate bench.py between environment settings on the same box.
#   tools/ab_env_bench.sh <rounds> "VAR=a" "VAR=b" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for v in "$@"; do
    env $v python $ROOT/bench.py --no-cpu-baseline --no-batched 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['ms_per_gru_iter'], {k: v['avg'] for k, v in d['kernel_times_us'].items() if 'conv' in k})"
  done
done
