ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export ANYSTEREO_ALLOW_STALE_LIB=1
for v in base cur; do
  if [ "$v" = cur ]; then unset ANYSTEREO_LIB; else export ANYSTEREO_LIB=$ROOT/any-stereo_amd/anystereo/lib/$v.so; fi
  python $ROOT/bench.py --no-cpu-baseline --no-extras --no-batched --steps 10 --warmup 3 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['ms_per_gru_iter'])
print({k: v['avg'] for k, v in d['kernel_times_us'].items()})"
done
