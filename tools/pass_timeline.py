"""One hipGraph replay of the forward as a timeline: rocprofv3 --kernel-trace CSV -> the kernels of the LAST complete pass
(the FASTEST window between two liif_tail_kernel ends = a graph replay) with start / end relative to the pass start, duration and queue; phase totals.
    python tools/pass_timeline.py <dir with *kernel_trace.csv> <out.json>"""
import csv
import glob
import json
import os
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "liif_tail_kernel" in r[2]]
    # the fastest complete pass = a hipGraph replay (bench.py also runs eager passes for its per-kernel event timing)
    j = min(range(1, len(ends)), key=lambda i: rows[ends[i]][1] - rows[ends[i - 1]][1])
    win = rows[ends[j - 1] + 1: ends[j] + 1]
    t0 = win[0][0]
    first_lookup = next(i for i, r in enumerate(win) if "lookup_convc1" in r[2] or "lookup_fwd" in r[2])
    pre = win[:first_lookup]
    res = {"pass_ms": (win[-1][1] - t0) / 1e6, "pre_loop_wall_ms": (win[first_lookup][0] - t0) / 1e6,
           "pre_loop_kernel_ms": sum(e - s for s, e, _, _ in pre) / 1e6, "pre_loop_launches": len(pre),
           "pre_loop": [{"kernel": k[:70], "start_us": round((s - t0) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1), "queue": q} for s, e, k, q in pre]}
    # post-loop = the upsampler: from the end of the last GRU-iteration kernel (tap_shift_sum) to the end of the pass
    last_loop = max(i for i, r in enumerate(win) if "tap_shift_sum" in r[2])
    post = win[last_loop + 1:]
    res["post_loop_wall_us"] = (win[-1][1] - win[last_loop][1]) / 1e3
    res["post_loop_kernel_us"] = sum(e - s for s, e, _, _ in post) / 1e3
    res["post_loop"] = [{"kernel": k[:70], "start_us": round((s - win[last_loop][1]) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1), "queue": q} for s, e, k, q in post]
    # one GRU iteration from the middle of the loop: kernels between two consecutive gru04 z|r convolutions
    zr = [i for i, r in enumerate(win) if "conv_split_kernel<3, 16, 64, 1, 2, 1" in r[2]]
    if len(zr) >= 4:
        a, b = zr[len(zr) // 2], zr[len(zr) // 2 + 1]
        t1 = win[a][0]
        res["iteration_period_us"] = (win[b][0] - t1) / 1e3
        res["iteration"] = [{"kernel": k[:60], "start_us": round((s - t1) / 1e3, 1), "end_us": round((e - t1) / 1e3, 1), "queue": q}
                            for s, e, k, q in win[a:b]]
    json.dump(res, open(out, "w"), indent=0)
    print({k: v for k, v in res.items() if k not in ("pre_loop", "post_loop", "iteration")})
    for r in res["post_loop"]:
        print("  post", r)
    for r in res.get("iteration", []):
        print("  iter", r["start_us"], r["end_us"], r["queue"], r["kernel"])
    # busy time per queue and the gaps on the timeline
    last = t0
    idle = 0
    for s, e, k, q in pre:
        if s > last:
            idle += s - last
        last = max(last, e)
    print("pre-loop: no kernel running for %.1f us" % (idle / 1e3))
    agg = {}
    for s, e, k, q in pre:
        a = agg.setdefault(k[:60], [0, 0])
        a[0] += e - s
        a[1] += 1
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
        print("%8.1f us n=%3d %s" % (v[0] / 1e3, v[1], k))


if __name__ == "__main__":
    main()
