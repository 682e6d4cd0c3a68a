"""Bitwise repeatability of as_conv2d over the conv shapes of the model (race detector): runs each case N times."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo import _lib as L  # noqa: E402
from anystereo import ops  # noqa: E402
from anystereo.harness.synthetic import det_uniform  # noqa: E402

dev = "cuda:0"
cases = [  # (cins, cout, ks, h, w, epilogue)
    ([128, 128, 128], 256, 3, 136, 240, L.EPI_GRU_ZR), ([128, 128, 128], 128, 3, 136, 240, L.EPI_GRU_Q),
    ([128, 128, 128], 256, 3, 68, 120, L.EPI_GRU_ZR), ([128, 128], 256, 3, 34, 60, L.EPI_GRU_ZR),
    ([128], 256, 3, 136, 240, L.EPI_LINEAR), ([64], 64, 3, 136, 240, L.EPI_LINEAR), ([162], 64, 1, 136, 240, L.EPI_LINEAR),
    ([256], 9, 1, 136, 240, L.EPI_LINEAR), ([64], 64, 3, 544, 960, L.EPI_LINEAR), ([96], 96, 3, 272, 480, L.EPI_LINEAR),
    ([16], 96, 1, 272, 480, L.EPI_LINEAR), ([24], 144, 1, 136, 240, L.EPI_LINEAR), ([960], 160, 1, 17, 30, L.EPI_LINEAR),
    ([228], 128, 1, 1, 518400, L.EPI_LINEAR), ([128], 384, 3, 136, 240, L.EPI_LINEAR),
]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for cins, cout, ks, h, w, epi in cases:
    srcs = [det_uniform((1, c, h, w), 10 + i).to(dev) for i, c in enumerate(cins)]
    wt = (det_uniform((cout, sum(cins), ks, ks), 30) * 0.05).to(dev)
    bias = det_uniform((cout,), 31).to(dev)
    pk = ops.PackedConv().get([wt], [bias])
    kw = {}
    if epi == L.EPI_GRU_ZR:
        kw = dict(epilogue=epi, h=srcs[0])
    elif epi == L.EPI_GRU_Q:
        kw = dict(epilogue=epi, h=srcs[0], z=torch.sigmoid(srcs[1]))
    ref = ops.conv2d(srcs, pk, **kw)
    ref = [t.clone() for t in (ref if isinstance(ref, tuple) else (ref,))]
    bad = 0
    worst = 0.0
    for _ in range(n):
        out = ops.conv2d(srcs, pk, **kw)
        out = out if isinstance(out, tuple) else (out,)
        for a, b in zip(out, ref):
            if not torch.equal(a, b):
                bad += 1
                worst = max(worst, (a - b).abs().max().item())
    print(f"{cins} -> {cout} k{ks} {h}x{w} epi{epi}: {bad} mismatching runs of {n}, worst {worst:.3e}", flush=True)
