"""Time the 3-D convolution kernels of the cost aggregation at cfg-2 sizes (events around 20 launches each).
    [ANYSTEREO_LIB=...] python tools/kbench_conv3d.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "any-stereo_amd"))
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


dev = "cuda"
for (cin, cout, d, h, w, s) in [(8, 8, 48, 136, 240, 1), (8, 16, 48, 136, 240, 2), (16, 16, 24, 68, 120, 1), (16, 32, 24, 68, 120, 2),
                                (32, 32, 12, 34, 60, 1), (8, 1, 48, 136, 240, 1)]:
    x = torch.randn(1, cin, d, h, w, device=dev)
    wp = torch.randn(cin, 27, cout, device=dev) * 0.05
    us = t(lambda: ops.conv3d_k3(x, wp, None, s, 5))
    do, ho, wo = (d - 1) // s + 1, (h - 1) // s + 1, (w - 1) // s + 1
    fl = 2.0 * cin * cout * 27 * do * ho * wo
    print(f"conv3d {cin}->{cout} {d}x{h}x{w} s{s}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
for (cin, cout, d, h, w) in [(16, 8, 24, 68, 120), (32, 16, 12, 34, 60), (48, 32, 6, 17, 30)]:
    x = torch.randn(1, cin, d, h, w, device=dev)
    wp = torch.randn(cin, 4, 4, 4, cout, device=dev) * 0.05
    us = t(lambda: ops.deconv3d_k4s2(x, wp, None, 5))
    fl = 2.0 * cin * cout * 8 * (2 * d) * (2 * h) * (2 * w)
    print(f"deconv3d {cin}->{cout} {d}x{h}x{w}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
