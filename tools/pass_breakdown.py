"""Steady-state per-kernel breakdown of ONE forward pass from a rocprofv3 --kernel-trace CSV.

    python tools/pass_breakdown.py <dir with *kernel_trace.csv> <out.json>
A pass = the dispatches between the ends of two consecutive liif_tail_kernel (or softmax_convex_kernel) launches (the last
kernel of a forward); the LAST complete window is used, so MIOpen's first-call solver search is excluded.  Phases:
pre (before the first lookup), loop (first lookup .. last GRU-iteration kernel), post.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    src, out = sys.argv[1], sys.argv[2]
    f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    last = "liif_tail_kernel" if any("liif_tail_kernel" in r[2] for r in rows) else "softmax_convex_kernel"
    ends = [i for i, r in enumerate(rows) if last in r[2]]
    assert len(ends) >= 2, "need two complete passes"
    win = rows[ends[-2] + 1: ends[-1] + 1]
    look = [i for i, r in enumerate(win) if "lookup_fwd" in r[2] or "lookup_convc1_kernel" in r[2] or "loop_front_kernel" in r[2]]
    last_loop = max(i for i, r in enumerate(win) if "tap_shift_sum" in r[2] or "conv3x3_to1" in r[2])
    phases = {"pre": win[:look[0]], "loop": win[look[0]:last_loop + 1], "post": win[last_loop + 1:]}
    res = {"wall_ms": (win[-1][1] - win[0][0]) / 1e6, "n_lookups": len(look)}
    for name, ks in phases.items():
        agg = defaultdict(lambda: [0, 0])
        for s, e, k in ks:
            agg[k[:100]][0] += e - s
            agg[k[:100]][1] += 1
        top = sorted(agg.items(), key=lambda kv: -kv[1][0])
        res[name] = {"kernel_ms": sum(v[0] for v in agg.values()) / 1e6, "launches": len(ks),
                     "wall_ms": (ks[-1][1] - ks[0][0]) / 1e6 if ks else 0.0,
                     "top": [{"kernel": k, "ms": v[0] / 1e6, "n": v[1]} for k, v in top[:40]]}
    json.dump(res, open(out, "w"), indent=1)
    for name in phases:
        print(name, "kernel_ms=%.3f wall_ms=%.3f launches=%d" % (res[name]["kernel_ms"], res[name]["wall_ms"], res[name]["launches"]))


if __name__ == "__main__":
    main()
