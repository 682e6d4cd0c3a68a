"""Same-process A/B of schedule options of the inference loop (model class attributes), interleaved rounds on one box:
    python tools/ab_loop.py early_gru16=0,1 [rounds] [cfg]
Each setting re-captures the forward's hipGraph; prints ms per pair per round and the medians."""
import os
import statistics
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch  # noqa: E402

from anystereo.harness import workloads as WL  # noqa: E402

name, vals = sys.argv[1].split("=")
vals = [v for v in vals.split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = WL.WORKLOADS[sys.argv[3] if len(sys.argv) > 3 else "cfg2"]
dev = "cuda:0"
model, args = WL.build_model(wl, device=dev)
i1, i2, coord, sc = WL.build_inputs(wl, device=dev)


def target(obj, attr):
    for o in (obj, obj.update_block, obj.update_block.encoder, obj.liif_up):
        if hasattr(type(o), attr) or hasattr(o, attr):
            return o
    raise AttributeError(attr)


owner = target(model, name)
res = {v: [] for v in vals}
outs = {}
with torch.no_grad():
    for r in range(rounds):
        for v in vals:
            setattr(owner, name, type(getattr(owner, name))(int(v)) if isinstance(getattr(owner, name), (bool, int)) else v)
            model.enable_graph(True)
            for _ in range(2):
                out = model(i1, i2, iters=wl.iters, test_mode=True, hr_coord=coord, scale=sc)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(10):
                out = model(i1, i2, iters=wl.iters, test_mode=True, hr_coord=coord, scale=sc)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t) / 10 * 1e3)
            outs[v] = out.clone()
ref = outs[vals[0]]
for v in vals:
    print(f"{name}={v}: ms/pair {[round(x, 3) for x in res[v]]} median {statistics.median(res[v]):.3f} "
          f"max|out - out[{vals[0]}]| = {(outs[v] - ref).abs().max().item():.2e}")
