"""Which layers of an inference forward still run on library (MIOpen / rocBLAS / ATen) kernels: logs every torch conv /
linear / norm / pooling / interpolate call with its shapes (cfg 2 by default)."""
import collections
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from anystereo.harness import workloads as WL  # noqa: E402

wl = WL.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
dev = "cuda:0"
model, args = WL.build_model(wl, device=dev)
i1, i2, coord, sc = WL.build_inputs(wl, device=dev)
log = collections.Counter()


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        x = a[0] if torch.is_tensor(a[0]) else a[0][0]
        w = a[1] if len(a) > 1 and torch.is_tensor(a[1]) else None
        extra = {kk: vv for kk, vv in k.items() if kk in ("stride", "padding", "dilation", "groups", "size", "scale_factor", "mode", "kernel_size")}
        log[(name, tuple(x.shape), None if w is None else tuple(w.shape), str(extra), tuple(a[3:7]) if name.startswith("conv") and len(a) > 3 else ())] += 1
        return orig(*a, **k)
    setattr(mod, name, f)


for n in ("conv2d", "conv3d", "conv_transpose2d", "conv_transpose3d", "linear", "batch_norm", "instance_norm", "layer_norm",
          "avg_pool2d", "max_pool2d", "interpolate", "pixel_unshuffle", "leaky_relu", "gelu", "relu", "adaptive_avg_pool2d"):
    wrap(F, n)
for n in ("cat", "tanh", "sigmoid", "relu"):
    wrap(torch, n)
with torch.no_grad():
    model(i1, i2, iters=2, test_mode=True, hr_coord=coord, scale=sc)
for k, v in sorted(log.items(), key=lambda kv: str(kv[0])):
    print(v, k)
