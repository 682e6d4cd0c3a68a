#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of cfg-4 training steps; per-kernel totals of the TIMED steps only (everything
# after the first `spin_kernel` marker bench.py launches per timed step: MIOpen's solver search runs in the warm-up).
#   -> gpurun_out/<tag>_train_kernel_stats.csv (+ the top of the table on stdout)
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/_kt -- python $ROOT/bench.py --mode train --steps 3 --warmup 2 --no-extras --no-cpu-baseline > $OUT/${TAG}_train_bench_under_rocprof.json 2>/dev/null
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/_kt/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "spin_kernel" in r["Kernel_Name"]]
assert marks, "no step markers in the trace"
steps = len(marks)
sel = rows[marks[0] + 1:]
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    if "spin_kernel" in r["Kernel_Name"]:
        continue
    a = agg[r["Kernel_Name"]]
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
tot = sum(v[0] for v in agg.values())
span = (int(sel[-1]["End_Timestamp"]) - int(rows[marks[0]]["Start_Timestamp"])) / steps / 1e6
with open("$OUT/${TAG}_train_kernel_stats.csv", "w") as o:
    w = csv.writer(o)
    w.writerow(["Name", "CallsPerStep", "TotalMsPerStep", "AverageUs", "Percent"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        w.writerow([k, round(v[1] / steps, 2), round(v[0] / steps / 1e6, 4), round(v[0] / v[1] / 1e3, 2), round(100 * v[0] / tot, 2)])
# the library's own kernels by launch shape (grid in workgroups): which layers a template instance's time belongs to
def wg(r):
    try:
        return (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * max(1, int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1)))) * max(1, int(r.get("Grid_Size_Z", 1)) // max(1, int(r.get("Workgroup_Size_Z", 1))))
    except Exception:
        return -1
shape = collections.defaultdict(lambda: [0, 0])
for r in sel:
    n = r["Kernel_Name"]
    if "spin_kernel" in n or not ("anonymous namespace" in n and "at::native" not in n and "ck::" not in n or n.startswith("_ZN12_GLOBAL__N_1")):
        continue
    a = shape[(n, wg(r))]
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
with open("$OUT/${TAG}_train_kernel_stats_by_grid.csv", "w") as o:
    w = csv.writer(o)
    w.writerow(["Name", "Workgroups", "CallsPerStep", "TotalMsPerStep", "AverageUs"])
    for (k, g), v in sorted(shape.items(), key=lambda kv: -kv[1][0]):
        w.writerow([k, g, round(v[1] / steps, 2), round(v[0] / steps / 1e6, 4), round(v[0] / v[1] / 1e3, 2)])
print(f"{steps} timed steps: {span:.1f} ms wall per step on the GPU timeline, {tot / steps / 1e6:.1f} ms of kernel time per step, {sum(v[1] for v in agg.values()) // steps} launches per step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:36]:
    print(f"{v[0] / steps / 1e6:8.2f} ms {100 * v[0] / tot:5.1f}% n={v[1] // steps:>5} avg={v[0] / v[1] / 1e3:8.1f}us  {k[:120]}")
PY
rm -rf $OUT/_kt
