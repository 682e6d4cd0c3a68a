"""Many replays of the default (graphed) training step on alternating batches: every loss finite and non-zero, gradients finite,
the loss level falling (run on the GPU box).   python tools/experiments/graphed_long_run.py [--steps 400]"""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo.harness.synthetic import fill_module_deterministic  # noqa: E402
from anystereo.harness.train import Trainer, synthetic_train_batch  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
a = ap.parse_args()
args = default_args("continuous_IGEVStereo")
m = __models__["continuous_IGEVStereo"](args)
fill_module_deterministic(m, base_seed=1)
tr = Trainer(m.to("cuda:0"), train_iters=16, max_disp=args.max_disp, num_steps=20000)
batches = [synthetic_train_batch(4, 160, 320, seed=s, device="cuda:0") for s in range(4)]
losses = []
t0 = time.perf_counter()
for i in range(a.steps):
    loss, met = tr.step(batches[i % 4])
    losses.append(loss)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
vals = [float(v) for v in losses]
bad = [i for i, v in enumerate(vals) if not (v == v) or v <= 0.0 or v > 1e5]
gfin = all(bool(torch.isfinite(p.grad).all()) for p in tr.model.parameters() if p.grad is not None)
pfin = all(bool(torch.isfinite(p).all()) for p in tr.model.parameters())
k = max(1, a.steps // 8)
print(f"{a.steps} steps in {dt:.1f} s ({dt / a.steps * 1e3:.1f} ms per step incl. warm-up); graph {tr.use_graph}, memset nodes {getattr(tr, 'graph_memsets', None)}")
print("mean loss per eighth of the run:", [round(sum(vals[j:j + k]) / len(vals[j:j + k]), 2) for j in range(0, a.steps, k)])
print("non-finite / zero losses at steps:", bad, "| gradients finite:", gfin, "| parameters finite:", pfin, "| overflow events:", tr.overflow_events)
sys.exit(1 if (bad or not gfin or not pfin) else 0)
