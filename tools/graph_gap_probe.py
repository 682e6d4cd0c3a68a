"""Where the time between two replays of the forward graph goes: ms per step of (a) the model call as bench.py issues it
(4 input copies + replay + result clone), (b) bare replays, against (c) the in-graph markers' pass length.
    python tools/graph_gap_probe.py [--config cfg2] [--steps 20]"""
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


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(steps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / steps, 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    _lib.load()
    dev = torch.device("cuda", 0)
    wl = WL.WORKLOADS[a.config]
    model, _ = WL.build_model(wl, device=dev)
    i1, i2, coord, scale = WL.build_inputs(wl, seed=1234, device=dev)
    res = {}
    for marked in (False, True):
        model.stamps = ops.Stamps(dev) if marked else None
        model.stamp_iters = ()
        model.enable_graph(True)
        with torch.no_grad():
            call = lambda: model(i1, i2, iters=wl.iters, test_mode=True, hr_coord=coord, scale=scale)  # noqa: E731
            call()
            g, st, out = next(reversed(model._graphs.values()))
            tag = "marked" if marked else "plain"
            res[tag + "_call_ms"] = [timed(call, a.steps) for _ in range(3)]
            res[tag + "_replay_only_ms"] = [timed(g.replay, a.steps) for _ in range(3)]

            def copies_only():
                st[0].copy_(i1), st[1].copy_(i2), st[2].copy_(coord), st[3].copy_(scale)
                out.clone()
            res[tag + "_copies_clone_only_ms"] = timed(copies_only, a.steps)
            if marked:
                for _ in range(3):
                    call()
                r = model.stamps.read()
                res["marked_pass_ms_from_markers"] = round((r["pass_end"] - r["pass_begin"]) / 1e3, 4)
        model.stamps = None
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
