"""Does a captured hipGraph that contains memcpy / memset NODES (what ATen's contiguous same-dtype copy_ and zero_ become
under capture) replay correctly while other eager work — including another thread's autograd backward — keeps the GPU busy?
(run on the GPU box)

    python tools/experiments/graph_copy_nodes.py [--links 300] [--replays 40] [--mode copy|kernel]

A dependent chain x -> kernel -> D2D copy -> memset + accumulate -> kernel ... is captured once; every replay's result is compared
with the eagerly computed value.  --mode kernel builds the same chain with the copy / zero steps as elementwise KERNELS.
"""
import argparse
import threading
import time

import torch


def chain(x, links, mode):
    acc = torch.empty_like(x)
    for i in range(links):
        y = x * 1.0009765625 + 0.5            # kernel
        z = y.clone() if mode == "copy" else y + 0.0   # D2D memcpy node | kernel
        if mode == "copy":
            acc.zero_()                        # memset node
        else:
            acc.mul_(0.0)
        acc += z                               # kernel
        x = torch.where(acc > 1e4, acc * 0.001, acc)
    return x


def churn(stop, dev):
    import torch.nn.functional as F
    torch.cuda.set_device(dev)
    net = torch.nn.Sequential(torch.nn.Conv2d(16, 64, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(64, 64, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.Conv2d(64, 16, 3, padding=1)).to(dev)
    x = torch.randn(8, 16, 96, 160, device=dev)
    while not stop.is_set():
        y = net(x)
        (y.square().mean() + F.avg_pool2d(y, 2).abs().mean()).backward()
        for p in net.parameters():
            p.grad = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--links", type=int, default=300)
    ap.add_argument("--replays", type=int, default=40)
    ap.add_argument("--mode", default="copy")
    ap.add_argument("--no-churn", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    x0 = torch.rand(1 << 20, device=dev)
    want = chain(x0.clone(), a.links, a.mode)
    torch.cuda.synchronize()
    static_in = x0.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain(static_in, 3, a.mode)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = chain(static_in, a.links, a.mode)
    stop = threading.Event()
    th = None
    if not a.no_churn:
        th = threading.Thread(target=churn, args=(stop, dev), daemon=True)
        th.start()
        time.sleep(1.0)
    bad = 0
    for r in range(a.replays):
        static_in.copy_(x0)
        g.replay()
        res = out.clone()
        torch.cuda.synchronize()
        err = (res - want).abs().max().item()
        if err != 0.0:
            bad += 1
            print(f"replay {r}: max |diff| {err:.3e}  ({(res != want).sum().item()} elements differ)", flush=True)
    stop.set()
    if th is not None:
        th.join(timeout=10)
    print(f"mode={a.mode} links={a.links} churn={not a.no_churn}: {bad} of {a.replays} replays differ from the eager result")


if __name__ == "__main__":
    main()
