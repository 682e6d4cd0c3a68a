"""Standalone launches of single hot kernels at a BASELINE config's sizes, for rocprofv3 / PMC passes.

    python tools/kbench.py corr_build lookup --reps 20 [--cfg 2]
Prints HIP-event time per launch (back-to-back launches on one stream).
"""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo import ops  # noqa: E402
from anystereo.harness.synthetic import det_uniform  # noqa: E402

CFG = {1: (1, 64, 128, 256, 4, 0), 2: (1, 136, 240, 96, 2, 8), 3: (1, 96, 312, 96, 2, 8), 5: (1, 336, 480, 96, 2, 8)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernels", nargs="+")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--precision", default=None)
    a = ap.parse_args()
    if a.precision:
        ops.set_precision(a.precision)
    b, h, w, c, L, g = CFG[a.cfg]
    dev = "cuda:0"
    f1 = det_uniform((b, c, h, w), 1).to(dev)
    f2 = det_uniform((b, c, h, w), 2).to(dev)
    gev = det_uniform((b, 8, 48, h, w), 3).to(dev) if g else None
    disp = det_uniform((b, 1, h, w), 4, 0.0, 40.0).to(dev)
    corr = ops.corr_build_pyramid(f1, f2, L)
    geo = ops.geo_pyramid(gev, L) if g else None
    x128 = [det_uniform((b, 128, h, w), 10 + i).to(dev) for i in range(3)]
    ctx = det_uniform((b, 384, h, w), 20).to(dev)
    wzr = det_uniform((256, 384, 3, 3), 30, -0.02, 0.02).to(dev)
    bzr = det_uniform((256,), 31).to(dev)
    pzr = ops.PackedConv().get([wzr], [bzr])
    from anystereo import _lib as Lb
    scoord = det_uniform((b, 2, h, w), 67, 0.0, float(w)).to(dev)
    sgrad = det_uniform((b, 9, h, w), 68).to(dev)
    w3d = det_uniform((8, 27, 8), 60, -0.1, 0.1).to(dev)
    xl1 = det_uniform((1, 64, 4 * h, 4 * w), 61).to(dev)
    pl1 = ops.PackedConv().get([det_uniform((64, 64, 3, 3), 62, -0.05, 0.05).to(dev)], [det_uniform((64,), 63).to(dev)])
    xq = det_uniform((1, 128, 1, 16 * h * w), 64).to(dev)
    pq2 = ops.PackedConv().get([det_uniform((64, 128, 1, 1), 65, -0.05, 0.05).to(dev)], [det_uniform((64,), 66).to(dev)])

    def zr_at(div):
        hh, ww = h // div, w // div
        xs = [det_uniform((b, 128, hh, ww), 40 + i).to(dev) for i in range(3)]
        cx = det_uniform((b, 384, hh, ww), 50).to(dev)
        return lambda: ops.conv2d(xs, pzr, add=cx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=xs[0])
    def to_bs(x):  # fp32 [B,C,H,W] -> ops.BS8 holding the same split the kernel's loaders compute
        bb, cc, hh, ww = x.shape
        hi = x.half()
        lo = ((x - hi.float()) * 2048.0).half()
        t = torch.stack([hi, lo], 1).view(bb, 2, cc // 8, 8, hh, ww).permute(0, 1, 2, 4, 5, 3).contiguous()
        return ops.BS8(t, cc)
    x128bs = [to_bs(x) for x in x128]
    fns = {
        "corr_build": lambda: ops.corr_build_pyramid(f1, f2, L),
        "geo_pyramid": (lambda: ops.geo_pyramid(gev, L)) if g else None,
        "lookup": lambda: ops.geo_corr_lookup(geo, corr, disp, 4),
        "gwc": lambda: ops.gwc_volume(f1, f2, 48, 8),
        "sampler_fwd": lambda: ops.corr_sampler_forward(corr[0], scoord, 4),
        "sampler_bwd": lambda: ops.corr_sampler_backward(corr[0], scoord, sgrad, 4),
        "cnet_l1": (lambda: ops.conv2d([xl1], pl1, act=Lb.ACT_RELU)),
        "liif_l2": (lambda: ops.conv2d([xq], pq2, act=Lb.ACT_RELU)),
        "conv3d_stem": (lambda: ops.conv3d_k3(gev, w3d, None, 1, 5)) if g else None,
        "gru08_zr": zr_at(2),
        "gru16_zr": zr_at(4),
        "gru_zr_bs": lambda: ops.conv2d(x128bs, pzr, add=ctx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=x128[0]),
        "gru_zr_bs1": lambda: ops.conv2d([x128bs[0], x128[1], x128[2]], pzr, add=ctx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=x128[0]),
        "gru_zr": lambda: ops.conv2d(x128, pzr, add=ctx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=x128[0]),
    }
    for k in a.kernels:
        fn = fns[k]
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        print(f"{k}: {s.elapsed_time(e) / a.reps * 1e3:.2f} us/launch (cfg {a.cfg}, precision {ops.get_precision()})")


if __name__ == "__main__":
    main()
