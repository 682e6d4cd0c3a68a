"""GPU: time forward+backward of every Conv3d / ConvTranspose3d (and optionally Conv2d) of the IGEV model at cfg-4 shapes,
one module at a time, to find which layers MIOpen serves with slow solvers.  usage: conv_bwd_times.py [--bench] [--2d]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch
import torch.nn as nn

from anystereo.harness.synthetic import fill_module_deterministic
from anystereo.harness.train import synthetic_train_batch
from anystereo.models import __models__, default_args

torch.backends.cudnn.benchmark = "--bench" in sys.argv
dev = torch.device("cuda", 0)
args = default_args("continuous_IGEVStereo")
model = __models__["continuous_IGEVStereo"](args)
fill_module_deterministic(model, base_seed=1)
model = model.to(dev).train()
model.freeze_bn()
kinds = (nn.Conv3d, nn.ConvTranspose3d) + ((nn.Conv2d, nn.ConvTranspose2d) if "--2d" in sys.argv else ())
shapes = {}
def _mk(n):
    def hook(m, i, o):
        shapes.setdefault(n, (m, tuple(i[0].shape)))   # returns None: a hook's return value would replace the output
    return hook


hooks = [m.register_forward_hook(_mk(n)) for n, m in model.named_modules() if isinstance(m, kinds)]
b = synthetic_train_batch(4, 160, 320, device=dev)
res = model(b[0], b[1], iters=1, hr_coord=b[2], scale=b[4])
for h in hooks:
    h.remove()
rows = []
for n, (m, shp) in shapes.items():
    x = torch.randn(shp, device=dev, requires_grad=True)
    def run():
        y = m(x)
        y.backward(torch.ones_like(y))
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record()
    torch.cuda.synchronize()
    rows.append((e0.elapsed_time(e1) / 3, n, type(m).__name__, shp, tuple(m.weight.shape), m.stride))
for r in sorted(rows, reverse=True)[:25]:
    print(f"{r[0]:8.2f} ms  {r[1]:40s} {r[2]:16s} in={r[3]} w={r[4]} s={r[5]}")
print("total", sum(r[0] for r in rows))
