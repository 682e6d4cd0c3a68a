"""Throughput of single-pair forwards with TWO pairs in flight: two captured forward graphs (own static buffers) replayed
alternately on two streams, so one pair's pre-loop (latency-bound, under-filled) runs beside the other pair's GRU loop.
Each pair is still its own batch-1 forward; only the issue order differs from the serial evaluation loop.
    python tools/experiments/two_in_flight.py [--config cfg2] [--steps 20]"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "any-stereo_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from anystereo import _lib  # noqa: E402
from anystereo.harness import workloads as WL  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    _lib.load()
    dev = torch.device("cuda", 0)
    wl = WL.WORKLOADS[a.config]
    m0, _ = WL.build_model(wl, device=dev)
    m1 = copy.deepcopy(m0)
    i1, i2, coord, scale = WL.build_inputs(wl, seed=1234, device=dev)
    models, streams = [m0, m1], [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    outs = [None, None]
    with torch.no_grad():
        for m in models:
            m.enable_graph(True)
            outs[0] = m(i1, i2, iters=wl.iters, test_mode=True, hr_coord=coord, scale=scale)  # capture
        torch.cuda.synchronize()
        ref = outs[0].clone()

        def serial(n):
            for _ in range(n):
                outs[0] = m0(i1, i2, iters=wl.iters, test_mode=True, hr_coord=coord, scale=scale)

        def piped(n):
            for k in range(n):
                s = streams[k & 1]
                with torch.cuda.stream(s):
                    outs[k & 1] = models[k & 1](i1, i2, iters=wl.iters, test_mode=True, hr_coord=coord, scale=scale)

        res = {}
        for name, fn in (("serial", serial), ("two_in_flight", piped), ("serial_again", serial), ("two_in_flight_again", piped)):
            fn(4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(a.steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[name] = {"pairs_per_s": round(a.steps / dt, 3), "ms_per_pair": round(dt / a.steps * 1e3, 3)}
        res["max_abs_diff_vs_serial"] = [float((o - ref).abs().max()) for o in outs]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
