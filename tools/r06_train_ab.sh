#!/bin/bash
# cfg-4 training step A/B on one box over an environment switch:  bash tools/r06_train_ab.sh VAR [rounds]   (VAR=0 vs VAR=1)
set -o pipefail
mkdir -p gpurun_out
var=$1; n=${2:-2}
for i in $(seq 1 $n); do
  for v in 0 1; do
    env $var=$v timeout -k 10 400 python bench.py --mode train --steps 10 --warmup 4 > gpurun_out/train_ab_$v.json 2> gpurun_out/train_ab_$v.err || { echo "$var=$v failed"; tail -20 gpurun_out/train_ab_$v.err; exit 1; }
    python3 - $var $v <<'P'
import json, sys
d = json.loads(open(f"gpurun_out/train_ab_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]}={sys.argv[2]}  ms_per_step {d['ms_per_step']}  samples/s {d['value']}  loss {d.get('loss_first_last')}")
P
  done
done
