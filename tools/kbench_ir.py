"""Time every inverted-residual block shape of the feature trunk at cfg 2 (B = 2): one launch (csrc/irblock.hip) vs the three-launch
path, each as 20 launches captured into one hipGraph between one event pair.   python tools/kbench_ir.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "any-stereo_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from anystereo import _lib, ops  # noqa: E402
from anystereo.harness.synthetic import det_uniform, fill_module_deterministic  # noqa: E402
from anystereo.nn.encoders import _InvRes  # noqa: E402

_lib.load()
dev = torch.device("cuda", 0)
shapes = [(16, 24, 2, 272, 480), (24, 24, 1, 136, 240), (24, 32, 2, 136, 240), (32, 32, 1, 68, 120), (32, 64, 2, 68, 120), (64, 64, 1, 34, 60),
          (64, 96, 1, 34, 60), (96, 96, 1, 34, 60), (96, 160, 2, 34, 60), (160, 160, 1, 17, 30)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


with torch.no_grad():
    for cin, cout, st, h, w in shapes:
        m = _InvRes(cin, cout, st).eval()
        fill_module_deterministic(m, 3)
        m = m.to(dev)
        x = det_uniform((2, cin, h, w), 5, -2, 2).to(dev)
        _InvRes.fused_ir = True
        t1 = timed(lambda: m(x))
        _InvRes.fused_ir = False
        t3 = timed(lambda: m(x))
        print("%3d -> %4d -> %3d s%d @ 2x%dx%d : one launch %7.1f us   three launches %7.1f us" % (cin, 6 * cin, cout, st, h, w, t1, t3))
