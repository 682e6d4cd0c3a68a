"""Phase boundaries of a REPLAYED forward graph from in-graph timeline markers (ops.Stamps / as_stamp): no profiler attached.

rocprofv3's queue interception changes how the parallel branches of a replayed hipGraph are fed (under --kernel-trace the two
pre-loop branches of the forward run one after the other: 5.6 ms, profiles/r06_base_pass_timeline.json; without it they overlap:
3.9 ms, profiles/r06_pass_phases_preloop_ab.txt), so the phases of a pass are measured here with marker kernels that are ordinary
nodes of the captured graph and read the device's constant-rate wall clock."""
from __future__ import annotations

import sys

import torch

from .. import ops


def phases(model, inputs, iters, reps=5, fine=False, verbose=False, stages=False):
    """-> {"pre_loop_us", "loop_us", "post_loop_us", "us_per_iter", "pass_us", "markers_us", "loop_stages_us", "all_pass_us"} of
    the median of `reps` replays of the captured forward of `model` on `inputs` = (image1, image2, hr_coord, scale).
    fine=True also places the operator-level markers of two consecutive GRU iterations (models/base.py::stamp_iters: head,
    lookup, 7x7, branch convs, merge conv | pool, gru08, interp | gru04 z|r, q); each marker is a one-thread kernel on its stream,
    so the fine form lengthens the marked iterations by ~50 us — the coarse form (8 markers per pass) is what bench.py reports."""
    i1, i2, coord, scale = inputs
    was_graph = bool(getattr(model, "_use_graph", False))
    keep_iters = model.stamp_iters
    model.stamp_iters = type(model).stamp_iters if fine else ()
    model.stamps = ops.Stamps(i1.device)
    model.stamps.stages = bool(stages)  # + a marker after every stage of the feature trunk
    model.enable_graph(True)
    runs = []
    try:
        with torch.no_grad():
            for _ in range(2):
                model(i1, i2, iters=iters, test_mode=True, hr_coord=coord, scale=scale)
            torch.cuda.synchronize()
            for _ in range(reps):
                for _ in range(3):  # back-to-back replays: the host is ahead of the GPU, as in the timed loop
                    model(i1, i2, iters=iters, test_mode=True, hr_coord=coord, scale=scale)
                runs.append(model.stamps.read())
    finally:
        model.stamps = None
        model.stamp_iters = keep_iters
        model.enable_graph(was_graph)
    runs.sort(key=lambda r: r["pass_end"] - r["pass_begin"])
    r = runs[len(runs) // 2]
    t0 = r["pass_begin"]
    r = {k: round(v - t0, 1) for k, v in r.items()}
    stages = sorted(((v, k) for k, v in r.items() if k.startswith("it")))
    if stages and verbose:
        print("loop stages (us since the first marked stage):", file=sys.stderr)
        for v, k in stages:
            print("  %8.1f  %s" % (v - stages[0][0], k), file=sys.stderr)
    r = {k: v for k, v in r.items() if not k.startswith("it")}
    return {"pre_loop_us": r["loop_begin"], "loop_us": round(r["loop_end"] - r["loop_begin"], 1),
            "post_loop_us": round(r["pass_end"] - r["loop_end"], 1), "us_per_iter": round((r["loop_end"] - r["loop_begin"]) / iters, 2),
            "pass_us": r["pass_end"], "markers_us": r,
            "loop_stages_us": {k: round(v - stages[0][0], 1) for v, k in stages} if stages else None,
            "all_pass_us": [round(x["pass_end"] - x["pass_begin"], 1) for x in runs],
            "how": "marker kernels (as_stamp: device wall clock, 10 ns ticks) captured into the forward's hipGraph at the phase boundaries; "
                   f"median of {reps} replays, each the last of 3 back-to-back replays; no profiler attached"}
