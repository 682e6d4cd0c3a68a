#!/bin/bash
# Same-box comparison of the headline: the round-5 tree (git archive 3363358 extracted to _r05_tree/ and built there) against this
# tree, bench.py alternated, side measurements off.  bash tools/r06_vs_r05.sh [rounds]
mkdir -p gpurun_out
n=${1:-3}
run() {  # $1 = tree dir, $2 = label
  (cd $1 && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras) > gpurun_out/vs.json 2> gpurun_out/vs.err || { echo "$2 failed"; tail -5 gpurun_out/vs.err; return 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/vs.json')); h=d.get('host',{}); print('%-8s' % '$2', d['value'], d['ms_per_step'], d['ms_per_gru_iter'], h.get('gpu_sclk_mhz'), h.get('gpu_power_w'))"
}
for i in $(seq 1 $n); do
  run _r05_tree r05 || exit 1
  run . r06 || exit 1
done
