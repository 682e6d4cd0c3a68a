"""Whole-step hipGraph of the cfg-4 training step against the eager step, step by step (run on the GPU box):

    python tools/train_graph_check.py [--steps 8] [--sync none|stream|device] [--lr 2e-4] [--batch 4]

Two trainers with identical weights and the same batch: one eager, one replaying the captured step (harness/train.py,
Trainer(graph=True)).  Prints the loss of every step of both, their relative difference, and ms per step.  `--sync device` puts a
device-wide torch.cuda.synchronize() between steps (round 2 saw a corrupted replay after one), `--sync stream` a stream
synchronisation, `--sync none` nothing.  Exit code 1 when a step's losses differ by more than --tol.
"""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo.harness.synthetic import fill_module_deterministic  # noqa: E402
from anystereo.harness.train import Trainer, synthetic_train_batch  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--sync", default="device", choices=["none", "stream", "device"])
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--tol", type=float, default=2e-3)
    ap.add_argument("--hw", type=int, nargs=2, default=[160, 320])
    a = ap.parse_args()
    dev = "cuda:0"
    args = default_args("continuous_IGEVStereo")
    torch.backends.cudnn.deterministic = True

    def fresh():
        m = __models__["continuous_IGEVStereo"](args)
        fill_module_deterministic(m, base_seed=1)
        return m.to(dev)

    batches = [synthetic_train_batch(a.batch, a.hw[0], a.hw[1], seed=s, device=dev) for s in range(2)]
    eager = Trainer(fresh(), lr=a.lr, num_steps=1000, train_iters=a.iters, max_disp=args.max_disp, graph=False)
    graphed = Trainer(fresh(), lr=a.lr, num_steps=1000, train_iters=a.iters, max_disp=args.max_disp, graph=True)
    assert graphed.use_graph and not eager.use_graph
    bad = 0
    t_e = t_g = 0.0
    for i in range(a.steps + graphed.graph_warmup):
        b = tuple(t.clone() for t in batches[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        le, _ = eager.step(tuple(t.clone() for t in b))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        lg, _ = graphed.step(tuple(t.clone() for t in b))
        torch.cuda.synchronize() if a.sync == "device" else (torch.cuda.current_stream().synchronize() if a.sync == "stream" else None)
        lgv = float(lg)
        t2 = time.perf_counter()
        lev = float(le)
        rel = abs(lev - lgv) / max(abs(lev), 1e-12)
        mode = "replay" if graphed._graph is not None and i >= graphed.graph_warmup else "eager warm-up"
        if i >= graphed.graph_warmup + 1:
            t_e += t1 - t0
            t_g += t2 - t1
        flag = "" if rel <= a.tol else "   <-- MISMATCH"
        bad += rel > a.tol
        print(f"step {i:2d} [{mode:13s}] eager loss {lev:.6f}  graphed {lgv:.6f}  rel diff {rel:.2e}{flag}", flush=True)
    n = max(1, a.steps - 1)
    # parameters after the run
    worst = 0.0
    for (n1, p1), (_, p2) in zip(eager.model.named_parameters(), graphed.model.named_parameters()):
        worst = max(worst, (p1 - p2).abs().max().item() / max(p1.abs().max().item(), 1e-12))
    print(f"sync={a.sync}: eager {t_e / n * 1e3:.1f} ms/step, graphed {t_g / n * 1e3:.1f} ms/step; worst parameter deviation after "
          f"{a.steps + graphed.graph_warmup} steps {worst:.2e} of the tensor's max; lr now {float(graphed.optimizer.param_groups[0]['lr']):.3e}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
