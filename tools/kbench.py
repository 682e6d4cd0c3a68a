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
    ap.add_argument("--graph", action="store_true", help="capture the reps into one hipGraph (kernels shorter than the ~20 us host launch path)")
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
    xl1bs = to_bs(xl1)
    xl1o = ops.BS8.empty(1, 64, 4 * h, 4 * w, dev)
    # the loop's other launches, operands as the model hands them over (blocked split-fp16 links)
    wc1 = (det_uniform((64, 162, 1, 1), 77) * (3.0 / 162) ** 0.5).to(dev)
    plc1 = ops.LookupConvPack().get(wc1, det_uniform((64,), 78, -0.1, 0.1).to(dev)) if g else None
    cor_bs = ops.BS8.empty(b, 64, h, w, dev)
    x64bs = [to_bs(det_uniform((b, 64, h, w), 90 + i).to(dev)) for i in range(2)]
    pc2 = ops.PackedConv().get([det_uniform((64, 64, 3, 3), 92, -0.05, 0.05).to(dev)], [det_uniform((64,), 93).to(dev)])
    pd2 = ops.PackedConv().get([det_uniform((64, 64, 3, 3), 94, -0.05, 0.05).to(dev)], [det_uniform((64,), 95).to(dev)])
    cd_bs = ops.BS8.empty(b, 128, h, w, dev)
    pcv = ops.PackedConv().get([det_uniform((127, 128, 3, 3), 96, -0.04, 0.04).to(dev)], [det_uniform((127,), 97).to(dev)])
    mf_bs = ops.BS8.empty(b, 128, h, w, dev)
    ph1 = ops.PackedConv().get([det_uniform((256, 128, 3, 3), 98, -0.04, 0.04).to(dev)], [det_uniform((256,), 99).to(dev)])
    tapw = det_uniform((256, 9), 100, -0.05, 0.05).to(dev)
    wq = det_uniform((128, 384, 3, 3), 101, -0.02, 0.02).to(dev)
    pq = ops.PackedConv().get([wq], [det_uniform((128,), 102).to(dev)])
    zt = det_uniform((b, 128, h, w), 103, 0.0, 1.0).to(dev)
    hq_bs = ops.BS8.empty(b, 128, h, w, dev)
    rh_bs = ops.BS8.empty(b, 128, h, w, dev)

    def lin384(cout):
        pk = ops.PackedConv().get([det_uniform((cout, 384, 3, 3), 110, -0.02, 0.02).to(dev)], [det_uniform((cout,), 111).to(dev)])
        ob = ops.BS8.empty(b, cout, h, w, dev)
        return lambda: ops.conv2d(x128bs, pk, out_bs=ob, bs_only=True)

    def gru_bs_at(div, q=False):
        hh, ww = h // div, w // div
        xs32 = [det_uniform((b, 128, hh, ww), 40 + i).to(dev) for i in range(3)]
        xs = [to_bs(t) for t in xs32]
        cx = det_uniform((b, 384, hh, ww), 50).to(dev)
        z_ = det_uniform((b, 128, hh, ww), 51, 0.0, 1.0).to(dev)
        o1, o2 = ops.BS8.empty(b, 128, hh, ww, dev), ops.BS8.empty(b, 128, hh, ww, dev)
        if q:
            return lambda: ops.conv2d(xs, pq, add=cx, add_coff=256, epilogue=Lb.EPI_GRU_Q, h=xs32[0], z=z_, out_bs=o2)
        return lambda: ops.conv2d(xs, pzr, add=cx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=xs32[0], out_bs=o1, bs_only=True)
    h8 = det_uniform((b, 128, h // 2, w // 2), 120).to(dev)
    h16 = det_uniform((b, 128, h // 4, w // 4), 121).to(dev)
    fns = {
        # the loop's resamplers (update.py:94-102) with blocked split-fp16 results
        "pool04_bs": lambda: ops.pool2x_bs(x128[0]), "pool08_bs": lambda: ops.pool2x_bs(h8),
        "interp08to04_bs": lambda: ops.interp_bs(h8, h, w), "interp16to08_bs": lambda: ops.interp_bs(h16, h // 2, w // 2),
        "corr_build": lambda: ops.corr_build_pyramid(f1, f2, L),
        "geo_pyramid": (lambda: ops.geo_pyramid(gev, L)) if g else None,
        "lookup": lambda: ops.geo_corr_lookup(geo, corr, disp, 4),
        "gwc": lambda: ops.gwc_volume(f1, f2, 48, 8),
        "sampler_fwd": lambda: ops.corr_sampler_forward(corr[0], scoord, 4),
        "sampler_bwd": lambda: ops.corr_sampler_backward(corr[0], scoord, sgrad, 4),
        "cnet_l1": (lambda: ops.conv2d([xl1], pl1, act=Lb.ACT_RELU)),
        "liif_l2": (lambda: ops.conv2d([xq], pq2, act=Lb.ACT_RELU)),
        # the context net's full-resolution 64 -> 64 layers with a blocked source (conv2 of a residual block: residual tail, fp32
        # result) and as a pure link (conv1: blocked in, blocked out)
        "cnet_l1_bs": (lambda: ops.conv2d([xl1bs], pl1, act=Lb.ACT_RELU, h=xl1)),
        "cnet_l1_bsbs": (lambda: ops.conv2d([xl1bs], pl1, act=Lb.ACT_RELU, out_bs=xl1o, bs_only=True)),
        "cnet_l1_bsboth": (lambda: ops.conv2d([xl1bs], pl1, act=Lb.ACT_RELU, h=xl1, out_bs=xl1o)),
        "conv3d_stem": (lambda: ops.conv3d_k3(gev, w3d, None, 1, 5)) if g else None,
        "lookup_convc1": (lambda: ops.lookup_convc1(geo, corr, disp, 4, plc1, out_bs=cor_bs)) if g else None,
        "enc_c2d2": lambda: ops.conv2d([x64bs[0]], pc2, act=Lb.ACT_RELU, out_bs=cd_bs, out_bs_coff=0, bs_only=True,
                                       dual={"src": x64bs[1], "pack": pd2, "out_coff": 64, "out_bs_coff": 64}),
        "enc_conv": lambda: ops.conv2d([cd_bs], pcv, act=Lb.ACT_RELU, out_bs=mf_bs, out_bs_coff=0, bs_only=True),
        "head_conv1": lambda: ops.conv2d([x128bs[0]], ph1, act=Lb.ACT_RELU, epilogue=Lb.EPI_RELU_TAPS, tap_w=tapw),
        "gru04_zr": gru_bs_at(1), "gru04_q": gru_bs_at(1, True),
        # FETCH_SIZE calibration: the same 3x3 conv over the same three blocked 128-channel sources with ONE
        # 64-channel output tile and with FOUR — if the four tiles of a pixel tile share their halo patches in one XCD's L2 the
        # patch part of the fetch traffic does not grow with the tile count
        "lin384_64": lin384(64), "lin384_256": lin384(256),
        "gru08_zr_bs": gru_bs_at(2), "gru08_q": gru_bs_at(2, True),
        "gru16_zr_bs": gru_bs_at(4), "gru16_q": gru_bs_at(4, True),
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
        if a.graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(a.reps):
                    fn()
            gr.replay()
            torch.cuda.synchronize()
            s.record()
            gr.replay()
            e.record()
        else:
            s.record()
            for _ in range(a.reps):
                fn()
            e.record()
        torch.cuda.synchronize()
        print(f"{k}: {s.elapsed_time(e) / a.reps * 1e3:.2f} us/launch (cfg {a.cfg}, precision {ops.get_precision()})")


if __name__ == "__main__":
    main()
