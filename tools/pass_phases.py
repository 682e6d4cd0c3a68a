"""Phase boundaries of a REPLAYED forward graph from in-graph timeline markers (as_stamp; no profiler attached).
    python tools/pass_phases.py [--config cfg2] [--reps 5]
Prints {marker: us since pass_begin} of the median replay (by pass length) and the derived phases."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "any-stereo_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from anystereo import _lib, ops  # noqa: E402
from anystereo.harness import workloads as WL  # noqa: E402


def phases(model, inputs, iters, reps=5):
    i1, i2, coord, scale = inputs
    model.stamps = ops.Stamps(i1.device)
    model.enable_graph(True)
    runs = []
    with torch.no_grad():
        for _ in range(2):
            model(i1, i2, iters=iters, test_mode=True, hr_coord=coord, scale=scale)
        torch.cuda.synchronize()
        for _ in range(reps):
            for _ in range(3):  # back-to-back replays: the host is ahead of the GPU, as in the timed loop
                model(i1, i2, iters=iters, test_mode=True, hr_coord=coord, scale=scale)
            runs.append(model.stamps.read())
    model.stamps = None
    model.enable_graph(True)
    runs.sort(key=lambda r: r["pass_end"] - r["pass_begin"])
    r = runs[len(runs) // 2]
    t0 = r["pass_begin"]
    r = {k: round(v - t0, 1) for k, v in r.items()}
    fine = sorted(((v, k) for k, v in r.items() if k.startswith("it")))
    if fine:
        f0 = fine[0][0]
        print("loop stages (us since the first marked stage):", file=sys.stderr)
        for v, k in fine:
            print("  %8.1f  %s" % (v - f0, k), file=sys.stderr)
    r = {k: v for k, v in r.items() if not k.startswith("it")}
    out = {"loop_stages_us": {k: round(v - fine[0][0], 1) for v, k in fine} if fine else None, "markers_us": r, "pass_us": r["pass_end"], "pre_loop_us": r["loop_begin"], "loop_us": round(r["loop_end"] - r["loop_begin"], 1),
           "post_loop_us": round(r["pass_end"] - r["loop_end"], 1), "us_per_iter": round((r["loop_end"] - r["loop_begin"]) / iters, 2),
           "all_pass_us": [round(x["pass_end"] - x["pass_begin"], 1) for x in runs]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    _lib.load()
    dev = torch.device("cuda", 0)
    wl = WL.WORKLOADS[a.config]
    model, _ = WL.build_model(wl, device=dev)
    inputs = WL.build_inputs(wl, seed=1234, device=dev)
    print(json.dumps(phases(model, inputs, wl.iters, a.reps)))


if __name__ == "__main__":
    main()
