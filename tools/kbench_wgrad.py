"""Weight-gradient micro-benchmark at the cfg-4 training shapes (16 iterations stacked along the batch axis):
as_conv2d_wgrad (bf16 hi/lo split MFMA) vs the library's fp32 wgrad (aten.convolution_backward)."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch  # noqa: E402

from anystereo import ops  # noqa: E402

dev = "cuda:0"
SHAPES = [("gru04 zr", 64, 384, 256, 40, 80, 3), ("gru04 q", 64, 384, 128, 40, 80, 3), ("head conv1", 64, 128, 256, 40, 80, 3),
          ("enc conv", 64, 128, 127, 40, 80, 3), ("enc c2", 64, 64, 64, 40, 80, 3), ("enc c1", 64, 162, 64, 40, 80, 1),
          ("gru08 zr", 64, 384, 256, 20, 40, 3), ("gru16 zr", 64, 256, 256, 10, 20, 3)]


def timeit(f, n=10):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, b, cin, cout, h, w, k in SHAPES:
    x = torch.randn(b, cin, h, w, device=dev)
    dy = torch.randn(b, cout, h, w, device=dev) * 1e-3
    wt = torch.zeros(cout, cin, k, k, device=dev)
    t_hip = timeit(lambda: ops.conv2d_wgrad(x, dy, k))
    t_lib = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, wt, [cout], [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1, [False, True, True]))
    dw, db = ops.conv2d_wgrad(x, dy, k)
    _, rw, rb = torch.ops.aten.convolution_backward(dy, x, wt, [cout], [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1, [False, True, True])
    gf = 2.0 * b * h * w * cin * k * k * cout / 1e9
    print(f"{name:11s} hip {t_hip:8.1f} us ({gf / t_hip * 1e3:6.1f} TFLOP/s algorithmic)  library {t_lib:8.1f} us   rel diff {float((dw - rw).norm() / rw.norm()):.2e}  "
          f"bias {float((db - rb).norm() / rb.norm()):.2e}", flush=True)
