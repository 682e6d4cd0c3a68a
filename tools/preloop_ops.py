"""Attribute the one-shot (pre-loop) kernels of an inference forward to their call sites: torch.profiler over one eager
forward with 1 GRU iteration; per (operator, input shapes, first anystereo frame) the summed device time.
    python tools/preloop_ops.py [--iters 1]"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
from anystereo.harness.synthetic import synthetic_pair  # noqa: E402
from anystereo.models import __models__  # noqa: E402
from anystereo.models.base import default_args  # noqa: E402
from anystereo.nn.liif import make_coord  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=1)
a = ap.parse_args()
torch.manual_seed(0)
model = __models__["continuous_IGEVStereo"](default_args("continuous_IGEVStereo")).eval().cuda()
H, W = 544, 960
img1, img2 = synthetic_pair(1, H, W, shift=8, seed=1234)
img1, img2 = img1.cuda(), img2.cuda()
coord = make_coord([540, 960]).unsqueeze(0).cuda()
scale = torch.tensor([[1.0]], device="cuda")
with torch.no_grad():
    for _ in range(2):
        model(img1, img2, iters=a.iters, test_mode=True, hr_coord=coord.clone(), scale=scale)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        model(img1, img2, iters=a.iters, test_mode=True, hr_coord=coord.clone(), scale=scale)
        torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    dt = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if dt <= 0:
        continue
    frame = next((f for f in (e.stack or []) if "anystereo" in f and "ops.py" not in f and "_lib.py" not in f), "")
    frame = frame.split("anystereo/")[-1][:70]
    shapes = str(e.input_shapes)[:60] if e.input_shapes else ""
    k = (e.name[:40], shapes, frame)
    agg[k][0] += dt
    agg[k][1] += 1
print("---- copies / elementwise operators by call site")
for ev in sorted(prof.key_averages(group_by_input_shape=True, group_by_stack_n=12), key=lambda e: -e.self_device_time_total):
    if ev.self_device_time_total < 8 or not any(k in ev.key for k in ("copy_", "cat", "add", "mul", "clamp", "relu", "sigmoid", "div", "sub", "contiguous", "clone", "fill", "zero")):
        continue
    fr = [f for f in (ev.stack or []) if "anystereo" in f]
    print(f"{ev.self_device_time_total:8.1f} us n={ev.count:3d} {ev.key[:28]:28s} {str(ev.input_shapes)[:70]:70s} {' <- '.join(x.split('anystereo/')[-1][:48] for x in fr[:3])}")
tot = sum(v[0] for v in agg.values())
print(f"device time {tot / 1e3:.2f} ms")
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{t:8.1f} us n={n:3d}  {k[0]:40s} {k[1]:60s} {k[2]}")
