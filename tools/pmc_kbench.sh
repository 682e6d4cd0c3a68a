#!/bin/bash
# Run ON THE GPU BOX: SQ-counter passes over tools/kbench.py (each `--pmc` group in its own rocprofv3 run, no API tracing).
#   tools/pmc_kbench.sh "<kernel-name substring>" "<CTR CTR ...>;<CTR ...>" <kbench args...>
pat=$1; groups=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra G <<< "$groups"
i=0
for g in "${G[@]}"; do
  out=/tmp/pmc_$i; rm -rf $out
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -- python $ROOT/tools/kbench.py "$@" > /tmp/pmc_$i.log 2>&1
  python - "$out" "$pat" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
src, pat = sys.argv[1], sys.argv[2]
fs = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
if not fs:
    print("no counter file"); sys.exit(0)
acc, n = defaultdict(float), defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    if pat in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in acc:
    print(f"{k:32s} {acc[k] / max(1, len(n[k])):16.1f}  per launch ({len(n[k])} launches)")
PY
  i=$((i+1))
done
