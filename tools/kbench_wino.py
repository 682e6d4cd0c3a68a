"""Direct vs Winograd F(4,3) launches of the 3x3 GRU-loop convolutions at cfg-2 size (graph-replayed, HIP events):
    python tools/kbench_wino.py [--reps 50]"""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo import _lib as Lb  # noqa: E402
from anystereo import ops  # noqa: E402
from anystereo.harness.synthetic import det_uniform  # noqa: E402


def to_bs(x):
    bb, cc, hh, ww = x.shape
    hi = x.half()
    lo = ((x - hi.float()) * 2048.0).half()
    return ops.BS8(torch.stack([hi, lo], 1).view(bb, 2, cc // 8, 8, hh, ww).permute(0, 1, 2, 4, 5, 3).contiguous(), cc)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    gr.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--h", type=int, default=136)
    ap.add_argument("--w", type=int, default=240)
    a = ap.parse_args()
    ops.set_precision("split")
    dev, b, h, w = "cuda:0", 1, a.h, a.w
    xs32 = [torch.tanh(det_uniform((b, 128, h, w), 40 + i, -2, 2)).to(dev) for i in range(3)]
    xs = [to_bs(t) for t in xs32]
    ctx = det_uniform((b, 384, h, w), 50).to(dev)
    z = det_uniform((b, 128, h, w), 51, 0.0, 1.0).to(dev)
    wzr, bzr = det_uniform((256, 384, 3, 3), 30, -0.02, 0.02).to(dev), det_uniform((256,), 31).to(dev)
    wq, bq = det_uniform((128, 384, 3, 3), 101, -0.02, 0.02).to(dev), det_uniform((128,), 102).to(dev)
    wh, bh = det_uniform((256, 128, 3, 3), 98, -0.04, 0.04).to(dev), det_uniform((256,), 99).to(dev)
    pzr, pq, ph = (ops.PackedConv().get([w_], [b_]) for w_, b_ in ((wzr, bzr), (wq, bq), (wh, bh)))
    pzr_w, pq_w, ph_w = (ops.PackedConv().get([w_], [b_], wino=True) for w_, b_ in ((wzr, bzr), (wq, bq), (wh, bh)))
    o1, o2, o3 = (ops.BS8.empty(b, c, h, w, dev) for c in (128, 128, 256))
    v = ops.wino_transform(xs)
    v1 = ops.wino_transform([xs[0]])
    runs = {
        "gru04_zr direct": lambda: ops.conv2d(xs, pzr, add=ctx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=xs32[0], out_bs=o1, bs_only=True),
        "gru04_zr winograd": lambda: ops.conv2d([v], pzr_w, add=ctx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=xs32[0], out_bs=o1, bs_only=True),
        "gru04_q direct": lambda: ops.conv2d(xs, pq, add=ctx, add_coff=256, epilogue=Lb.EPI_GRU_Q, h=xs32[0], z=z, out_bs=o2),
        "gru04_q winograd": lambda: ops.conv2d([v], pq_w, add=ctx, add_coff=256, epilogue=Lb.EPI_GRU_Q, h=xs32[0], z=z, out_bs=o2),
        "head_conv1(linear) direct": lambda: ops.conv2d([xs[0]], ph, act=Lb.ACT_RELU, out_bs=o3, bs_only=True),
        "head_conv1(linear) winograd": lambda: ops.conv2d([v1], ph_w, act=Lb.ACT_RELU, out_bs=o3, bs_only=True),
        "transform 384 ch (3 blocked sources)": lambda: ops.wino_transform(xs, out=v),
        "transform 128 ch (1 blocked source)": lambda: ops.wino_transform([xs[0]], out=v, c_off=0),
    }
    for k, fn in runs.items():
        print(f"{k}: {timed(fn, a.reps):.2f} us/launch ({h}x{w})", flush=True)
    # correctness of what was timed (same operands)
    a0 = ops.conv2d(xs, pzr, add=ctx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=xs32[0])
    a1 = ops.conv2d([ops.wino_transform(xs)], pzr_w, add=ctx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=xs32[0])
    print("z max |winograd - direct| =", (a0[0] - a1[0]).abs().max().item(), " r*h:", (a0[1] - a1[1]).abs().max().item())


if __name__ == "__main__":
    main()
