#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of tools/time_model.py <args...> (top kernels by total time)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
out=$ROOT/gpurun_out/prof_model
rm -rf $out; mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $ROOT/tools/time_model.py "$@" > $out.log 2>&1
grep "ms / pair" $out.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" "${TOPN:-25}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2])]:
    print(f"{int(r['TotalDurationNs']) / 1e6:9.2f} ms {100 * int(r['TotalDurationNs']) / tot:5.1f}% x{r['Calls']:>5} {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:100]}")
PY
rm -rf $out $out.log
