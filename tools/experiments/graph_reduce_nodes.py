"""Does a captured hipGraph replay ATen's multi-block reductions correctly?  (run on the GPU box)

A full `.sum()` of a large tensor is a two-stage reduction: the blocks of one output meet at a SEMAPHORE that ATen allocates and
zeroes with hipMemsetAsync in front of the kernel (aten/src/ATen/native/cuda/Reduce.cuh) — under capture a memset NODE.  The
graph below is x -> (a few elementwise kernels) -> several such reductions; every replay's results are compared with eager ones.

    python tools/experiments/graph_reduce_nodes.py [--replays 20] [--n 3276800] [--between none|eager]
"""
import argparse

import torch


def body(x, gt):
    valid = (gt < 512) & (gt > 0)
    zero = torch.zeros((), device=x.device)
    cnt = valid.sum().to(x.dtype)
    err = torch.where(valid, (x - gt).abs(), zero)
    per = err.view(16, -1).sum(1)
    loss = (per * 0.9).sum() / cnt
    epe = torch.where(valid, (x - gt) ** 2, zero).sum() / cnt
    px1 = (valid & ((x - gt).abs() > 1)).sum().to(x.dtype) / cnt
    return loss, epe, px1, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replays", type=int, default=20)
    ap.add_argument("--n", type=int, default=16 * 204800)
    ap.add_argument("--between", default="none")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g0 = torch.Generator(device=dev).manual_seed(1)
    gt = torch.rand(a.n, device=dev, generator=g0) * 600 - 20
    xs = [torch.rand(a.n, device=dev, generator=g0) * 300 for _ in range(4)]
    want = [tuple(float(v) for v in body(x, gt)) for x in xs]
    static = xs[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body(static, gt)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        outs = body(static, gt)
    bad = 0
    for r in range(a.replays):
        static.copy_(xs[r % 4])
        g.replay()
        got = tuple(float(v.clone()) for v in outs)
        if a.between == "eager":
            y = torch.randn(1 << 22, device=dev)
            (y * y).sum().item()
        w = want[r % 4]
        ok = all(abs(p - q) <= 1e-4 * max(1.0, abs(q)) for p, q in zip(got, w))
        bad += not ok
        if not ok or r < 3:
            print(f"replay {r}: got {tuple(round(v, 4) for v in got)}  want {tuple(round(v, 4) for v in w)}{'' if ok else '   <-- MISMATCH'}", flush=True)
    print(f"{bad} of {a.replays} replays differ (between={a.between})")


if __name__ == "__main__":
    main()
