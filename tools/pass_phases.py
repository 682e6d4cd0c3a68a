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

from anystereo import _lib  # noqa: E402
from anystereo.harness import workloads as WL  # noqa: E402
from anystereo.harness.phases import phases  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--stages", action="store_true", help="also a marker after every stage of the feature trunk")
    ap.add_argument("--fine", action="store_true", help="also the operator-level markers of two GRU iterations")
    a = ap.parse_args()
    _lib.load()
    dev = torch.device("cuda", 0)
    wl = WL.WORKLOADS[a.config]
    model, _ = WL.build_model(wl, device=dev)
    inputs = WL.build_inputs(wl, seed=1234, device=dev)
    print(json.dumps(phases(model, inputs, wl.iters, a.reps, fine=a.fine, verbose=True, stages=a.stages)))


if __name__ == "__main__":
    main()
