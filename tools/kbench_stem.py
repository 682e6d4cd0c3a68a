"""Time the 7x7 stem at cfg-2 size: library path (MIOpen conv + bias + ReLU) vs the MFMA kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "any-stereo_amd"))
from anystereo import _lib as L  # noqa: E402
from anystereo import ops  # noqa: E402


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (b, h, w) in [(1, 544, 960), (2, 256, 512)]:
    x = torch.randn(b, 3, h, w, device="cuda")
    wt = torch.randn(64, 3, 7, 7, device="cuda") * 0.1
    bias = torch.randn(64, device="cuda")
    pk = ops.Stem7x7Pack()
    us_lib = t(lambda: torch.nn.functional.conv2d(x, wt, bias, 1, 3).relu_())
    us_hip = t(lambda: ops.conv7x7_c3(x, pk, wt, bias, act=L.ACT_RELU))
    fl = 2.0 * 147 * 64 * b * h * w
    print(f"stem 7x7 3->64 {b}x{h}x{w}: library {us_lib:7.1f} us   MFMA kernel {us_hip:7.1f} us ({fl * 3 / us_hip / 1e6:6.1f} TFLOP/s issued)")
