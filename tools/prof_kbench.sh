#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of tools/kbench.py.   tools/prof_kbench.sh <tag> <kbench args...>
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
out=$ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $ROOT/tools/kbench.py "$@" > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then python - "$f" "${TOPN:-8}" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "at::native" not in r["Name"] and "elementwise" not in r["Name"]]
for r in rows[:int(sys.argv[2])]:
    print(f"{float(r['AverageNs']) / 1e3:10.2f} us x{r['Calls']:>4}  {r['Name'][:90]}")
PY
else tail -5 $out.log; fi
rm -rf $out $out.log
