"""Read the in-kernel stamps of the diagnostic conv build (tools/conv_variant.sh stamps -DAS_CONV_STAMPS -> lib/stamps.so) on the GPU box:
    python tools/conv_stamps.py [gru_zr|gru_q|head1|liif_l2|cnet_l1]
Prints, per segment of conv_split_kernel's chunk loop, the mean cycles per chunk over all blocks (SHARES matter, not totals:
the stamps' fences forbid overlaps the product kernel has)."""
import ctypes as C
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
os.environ["ANYSTEREO_LIB"] = os.path.join(ROOT, "any-stereo_amd", "anystereo", "lib", "stamps.so")
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo import _lib as Lb, ops  # noqa: E402
from anystereo.harness.synthetic import det_uniform  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "gru_zr"
    dev, b, h, w = "cuda:0", 1, 136, 240
    U = lambda shape, seed, lo=-1.0, hi=1.0: det_uniform(shape, seed, lo, hi).to(dev)  # noqa: E731
    def to_bs(x):  # fp32 [B,C,H,W] -> ops.BS8 holding the split the kernel's loaders compute (all-DMA staging path)
        bb, cc, hh, ww = x.shape
        hi = x.half()
        lo = ((x - hi.float()) * 2048.0).half()
        return ops.BS8(torch.stack([hi, lo], 1).view(bb, 2, cc // 8, 8, hh, ww).permute(0, 1, 2, 4, 5, 3).contiguous(), cc)
    if which == "gru_zr_bs":
        xs = [U((b, 128, h, w), 10 + i) for i in range(3)]
        ctx = U((b, 384, h, w), 20)
        pk = ops.PackedConv().get([U((256, 384, 3, 3), 30, -0.02, 0.02)], [U((256,), 31)])
        xb = [to_bs(x) for x in xs]
        fn, chunks = (lambda: ops.conv2d(xb, pk, add=ctx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=xs[0])), 24
    elif which in ("enc_conv_bs", "enc_c2d2_bs", "gru_q_bs"):
        if which == "gru_q_bs":
            xs = [U((b, 128, h, w), 10 + i) for i in range(3)]
            ctx = U((b, 384, h, w), 20)
            z = U((b, 128, h, w), 21, 0.0, 1.0)
            pk = ops.PackedConv().get([U((128, 384, 3, 3), 30, -0.02, 0.02)], [U((128,), 31)])
            xb = [to_bs(x) for x in xs]
            ob = ops.BS8.empty(b, 128, h, w, dev)
            fn, chunks = (lambda: ops.conv2d(xb, pk, add=ctx, add_coff=256, epilogue=Lb.EPI_GRU_Q, h=xs[0], z=z, out_bs=ob)), 24
        elif which == "enc_conv_bs":
            xb = to_bs(U((b, 128, h, w), 10))
            pk = ops.PackedConv().get([U((127, 128, 3, 3), 30, -0.04, 0.04)], [U((127,), 31)])
            ob = ops.BS8.empty(b, 128, h, w, dev)
            fn, chunks = (lambda: ops.conv2d([xb], pk, act=Lb.ACT_RELU, out_bs=ob, out_bs_coff=0, bs_only=True)), 8
        else:
            xa, xb2 = to_bs(U((b, 64, h, w), 10)), to_bs(U((b, 64, h, w), 11))
            pk = ops.PackedConv().get([U((64, 64, 3, 3), 30, -0.05, 0.05)], [U((64,), 31)])
            pk2 = ops.PackedConv().get([U((64, 64, 3, 3), 32, -0.05, 0.05)], [U((64,), 33)])
            ob = ops.BS8.empty(b, 128, h, w, dev)
            fn, chunks = (lambda: ops.conv2d([xa], pk, act=Lb.ACT_RELU, out_bs=ob, out_bs_coff=0, bs_only=True,
                                             dual={"src": xb2, "pack": pk2, "out_coff": 64, "out_bs_coff": 64})), 4
    elif which == "gru_zr":
        xs = [U((b, 128, h, w), 10 + i) for i in range(3)]
        ctx = U((b, 384, h, w), 20)
        pk = ops.PackedConv().get([U((256, 384, 3, 3), 30, -0.02, 0.02)], [U((256,), 31)])
        fn, chunks = (lambda: ops.conv2d(xs, pk, add=ctx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=xs[0])), 24
    elif which == "gru_q":
        xs = [U((b, 128, h, w), 10 + i) for i in range(3)]
        ctx = U((b, 384, h, w), 20)
        z = U((b, 128, h, w), 21, 0.0, 1.0)
        pk = ops.PackedConv().get([U((128, 384, 3, 3), 30, -0.02, 0.02)], [U((128,), 31)])
        fn, chunks = (lambda: ops.conv2d(xs, pk, add=ctx, add_coff=256, epilogue=Lb.EPI_GRU_Q, h=xs[0], z=z)), 24
    elif which == "head1":
        x = U((b, 128, h, w), 10)
        pk = ops.PackedConv().get([U((256, 128, 3, 3), 30, -0.02, 0.02)], [U((256,), 31)])
        fn, chunks = (lambda: ops.conv2d([x], pk, act=Lb.ACT_RELU)), 8
    elif which == "cnet_l1":
        x = U((1, 64, 4 * h, 4 * w), 61)
        pk = ops.PackedConv().get([U((64, 64, 3, 3), 62, -0.05, 0.05)], [U((64,), 63)])
        fn, chunks = (lambda: ops.conv2d([x], pk, act=Lb.ACT_RELU)), 4
    else:
        x = U((1, 128, 1, 16 * h * w), 64)
        pk = ops.PackedConv().get([U((64, 128, 1, 1), 65, -0.05, 0.05)], [U((64,), 66)])
        fn, chunks = (lambda: ops.conv2d([x], pk, act=Lb.ACT_RELU)), 2
    lib = Lb.load()
    dbg = lib.as_debug_conv_stamps
    dbg.restype, dbg.argtypes = C.c_int, [C.c_void_p, C.c_int]
    n = 1024 * 16
    buf = (C.c_ulonglong * n)()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    assert dbg(buf, n) == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    assert dbg(buf, n) == 0
    t = torch.tensor(list(buf), dtype=torch.float64).view(1024, 16)
    used = t[(t.sum(1) > 0)]
    print(f"{which}: {used.shape[0]} blocks stamped, launch {s.elapsed_time(e) * 1e3:.1f} us (diagnostic build), {chunks} chunks/block")
    names = {0: "consumer: operand reads + MFMAs", 1: "consumer: barrier 1 (waits for loaders)", 2: "consumer: barrier 2 (patch commit)",
             13: "loader: weight loads issued", 8: "loader: patch loads issued+returned+split", 9: "loader: weight image store", 10: "loader: barrier 1 (waits for consumers)",
             11: "loader: patch commit", 12: "loader: barrier 2"}
    for role, slots in (("consumer", (0, 1, 2)), ("loader", (13, 8, 9, 10, 11, 12))):
        tot = sum(used[:, i].mean().item() for i in slots)
        if tot == 0:
            continue  # the interleaved loader pipeline carries no stamps
        for i in slots:
            m = used[:, i].mean().item()
            print(f"  {names[i]:45s} {m / chunks:9.0f} ticks/chunk  {100 * m / tot:5.1f} %   (min {used[:, i].min().item() / chunks:.0f}, max {used[:, i].max().item() / chunks:.0f})")
        print(f"  {role} total {tot / chunks:.0f} ticks/chunk")
    # block lifetime (thread 0 of each block): entry -> first unit staged -> chunk loop done -> tile parked -> end
    life = getattr(lib, "as_debug_conv_life", None)
    if life is not None:
        life.restype, life.argtypes = C.c_int, [C.c_void_p, C.c_int]
        lb = (C.c_ulonglong * (1024 * 8))()
        assert life(lb, 1024 * 8) == 0
        lt = torch.tensor(list(lb), dtype=torch.float64).view(1024, 8)
        ok = lt[(lt[:, 4] > lt[:, 0]) & (lt[:, 0] > 0)]
        if ok.shape[0]:
            seg = [("prologue (entry -> first unit staged)", 0, 1), ("chunk loop", 1, 2), ("bias + park + barriers", 2, 3), ("finish (operands, math, stores)", 3, 4)]
            tot = (ok[:, 4] - ok[:, 0]).mean().item()
            print(f"  block lifetime over {ok.shape[0]} blocks: {tot:.0f} ticks")
            for name, a, b in seg:
                d = (ok[:, b] - ok[:, a])
                print(f"    {name:42s} {d.mean().item():9.0f} ticks {100 * d.mean().item() / tot:5.1f} %  (min {d.min().item():.0f}, max {d.max().item():.0f})")
            rt = (ok[:, 6] - ok[:, 5])
            if (rt > 0).all():
                ghz = ((ok[:, 4] - ok[:, 0]) / rt * 0.1)
                print(f"    in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz): median {ghz.median().item():.3f} GHz "
                      f"(min {ghz.min().item():.3f}, max {ghz.max().item():.3f}); block life {rt.median().item() * 0.01:.1f} us")
            first = ok[:, 0].min().item()
            print(f"    block starts span {ok[:, 0].max().item() - first:.0f} ticks, ends span {ok[:, 4].max().item() - first:.0f} ticks after the first start")


if __name__ == "__main__":
    main()
