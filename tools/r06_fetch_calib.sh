#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/experiments/fetch_calib.hip -o /tmp/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fc
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -- /tmp/fetch_calib
python3 - <<'P'
import csv, glob, os
from collections import defaultdict
fs = glob.glob("/tmp/fc/**/*counter_collection.csv", recursive=True)
acc, n = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(fs[0])):
    if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("read"):
        k = r["Kernel_Name"].split("(")[0]
        acc[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(acc):
    kb = acc[k] / n[k]
    print(f"{k:14s} FETCH_SIZE {kb:12.1f} KiB raw per launch ({n[k]} launches) = {kb * 1024 / (1 << 30):.3f} of the 1 GiB read")
P
