#!/bin/bash
# N-rank bench.py protocol on a ONE-GPU box: all ranks on device 0 over gloo (bench.py REHEARSAL).  Not a measurement.
#   bash tools/r06_rehearse.sh [N ...]      (default: 2 4; at most 6 processes may share the card)
set -o pipefail
mkdir -p gpurun_out
export ANYSTEREO_BENCH_ONE_GPU_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0
ns="${@:-2 4}"
port=29541
for n in $ns; do
  t0=$(date +%s)
  timeout -k 10 560 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus $n --steps 5 --warmup 2 > gpurun_out/r06_rehearse_n$n.json 2> gpurun_out/r06_rehearse_n$n.err
  rc=$?
  echo "N=$n rc=$rc wall=$(( $(date +%s) - t0 ))s stdout_lines=$(wc -l < gpurun_out/r06_rehearse_n$n.json)"
  [ $rc -eq 0 ] || { tail -30 gpurun_out/r06_rehearse_n$n.err; exit $rc; }
  python3 - $n <<'P'
import json, sys
n = int(sys.argv[1])
d = json.loads(open(f"gpurun_out/r06_rehearse_n{n}.json").read())
t = d.get("train_mode") or {}
print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "value_spread", "rehearsal")})
print({k: t.get(k) for k in ("n_gpus", "global_batch", "value", "ms_per_step", "exchange_ms", "one_rank_ms_per_step", "scaling", "distinct_devices", "loss_first_last", "error")})
assert d["n_gpus"] == n and t.get("n_gpus") == n and t.get("global_batch") == 4 * n, "line does not carry the N-rank legs"
P
  [ $? -eq 0 ] || exit 1
  port=$((port + 1))
done
