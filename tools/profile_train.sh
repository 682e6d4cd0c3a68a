#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of a few cfg-4 training steps -> gpurun_out/<tag>_train_kernel_stats.csv
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_kt -- python $ROOT/bench.py --mode train --steps 2 --warmup 1 > $OUT/${TAG}_train_bench_under_rocprof.json 2>/dev/null
cp $OUT/_kt/*/*kernel_stats.csv $OUT/${TAG}_train_kernel_stats.csv
rm -rf $OUT/_kt
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/${TAG}_train_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:40]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% n={r["Calls"]:>6} avg={float(r["AverageNs"])/1e3:8.1f}us  {r["Name"][:110]}')
PY
