#!/bin/bash
# FETCH_SIZE of the calibration pair lin384_64 / lin384_256 (the gru04 z|r operands, linear epilogue, one / four channel tiles)
cd /tmp && export TMPDIR=/tmp
for m in 1 2; do
  rm -rf /tmp/pf_$m
  AS_CONV_XCD=$m timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$m -- python3 $GRAFT_REPO_ROOT/tools/kbench.py lin384_64 lin384_256 --reps 5 > /dev/null 2>&1
  python3 - /tmp/pf_$m $m <<'P'
import csv, glob, os, sys
fs = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
rows = [r for r in csv.DictReader(open(fs[0])) if "conv_split" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
half = len(rows) // 2
for name, rs in (("lin384_64", rows[:half]), ("lin384_256", rows[half:])):
    v = [float(r["Counter_Value"]) for r in rs][3:]   # skip the warm-up launches
    print(f"AS_CONV_XCD={sys.argv[2]} {name}: FETCH_SIZE x2 = {sum(v) / len(v) / 512:.1f} MB per launch ({len(v)} launches, grid {rs[-1].get('Grid_Size')})")
P
done
