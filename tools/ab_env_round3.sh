#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of this round's two loop changes through their environment switches (same binary):
#   base = AS_LOOKUP_DIRECT=0 AS_CONV_WIDE64=0 (round-2 behaviour), lookup = direct lookup only, both = default
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do
  for v in base lookup both; do
    case $v in
      base) export AS_LOOKUP_DIRECT=0 AS_CONV_WIDE64=0;;
      lookup) export AS_LOOKUP_DIRECT=1 AS_CONV_WIDE64=0;;
      both) export AS_LOOKUP_DIRECT=1 AS_CONV_WIDE64=1;;
    esac
    python $ROOT/bench.py --no-cpu-baseline --no-extras --no-batched --steps 10 --warmup 3 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_times_us']
print('$v', d['value'], d['ms_per_step'], d['ms_per_gru_iter'], {n: k[n]['avg'] for n in ('lookup_convc1', 'enc_convc2', 'gru04_zr_conv') if n in k})"
  done
done
