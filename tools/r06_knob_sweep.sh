#!/bin/bash
# bench headline under schedule knobs, alternating with the default on one box:  bash tools/r06_knob_sweep.sh "VAR=val VAR2=val" ...
mkdir -p gpurun_out
run() {
  env $1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { echo "$1 failed"; tail -5 gpurun_out/sweep.err; return 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/sweep.json')); print('%-44s' % '$1', d['value'], d['ms_per_step'], d['ms_per_gru_iter'], d['value_spread']['pairs_per_s'])"
}
run AS_X=0 || exit 1
for k in "$@"; do
  run "$k" || exit 1
  run AS_X=0 || exit 1
done
