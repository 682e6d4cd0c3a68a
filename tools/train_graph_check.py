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


_CH = {}


def churn(a, dev, i):
    """Eager work between two replays (none of it touches the graphed model)."""
    import torch.nn.functional as F
    k = a.churn_kind
    if k == "alloc":
        junk = [torch.randn(a.churn * 1024 * 256 // 8, device=dev) for _ in range(8)]
        junk = [j * 2 for j in junk]
    elif k == "gemm":
        for n in (64, 300, 1000, 2048 + 64 * i):
            x = torch.randn(8, n, n // 2 + 7, device=dev)
            y = torch.bmm(x, x.transpose(1, 2))
            z = torch.matmul(y[0], torch.randn(n, 9, device=dev))
        del x, y, z
    elif k == "conv":
        for c, hw in ((3, 160), (32, 80), (96, 40), (128, 20 + i)):
            x = torch.randn(4, c, hw, hw * 2, device=dev, requires_grad=True)
            w = torch.randn(c + 8, c, 3, 3, device=dev, requires_grad=True)
            F.conv2d(x, w, stride=2, padding=1).sum().backward()
    elif k == "conv3d":
        for c, d in ((8, 24), (16, 12), (32, 6)):
            x = torch.randn(4, c, d, d * 2, d * 4, device=dev, requires_grad=True)
            w = torch.randn(c * 2, c, 3, 3, 3, device=dev, requires_grad=True)
            F.conv3d(x, w, stride=1, padding=1).sum().backward()
    elif k == "wgrad":
        from anystereo import ops
        for cin, cout, hw, ks in ((64, 64, 16, 3), (128, 127, 24, 3), (384, 256, 12, 3), (128, 64, 32, 1)):
            x = torch.randn(2, cin, hw, hw * 2, device=dev)
            d = torch.randn(2, cout, hw, hw * 2, device=dev) * 1e-3
            ops.conv2d_wgrad(x, d, ks)
    elif k == "convsame":
        from anystereo import grad as G, ops
        for cin, cout, hw, ks in ((64, 64, 16, 3), (128, 127, 24, 3), (384, 256, 12, 3)):
            x = torch.randn(2, cin, hw, hw * 2, device=dev, requires_grad=True)
            w = (torch.randn(cout, cin, ks, ks, device=dev) * 0.05).requires_grad_(True)
            G.Conv2dSame.apply(x, w, None, True, ops.PackedConv(), ops.PackedConv()).sum().backward()
    elif k == "bn":
        bn = _CH.setdefault("bn", torch.nn.BatchNorm3d(16).to(dev).train())
        x = torch.randn(4, 16, 12, 20, 40, device=dev, requires_grad=True)
        bn(x).sum().backward()
    elif k in ("infer", "trainfwd", "trainstep_small"):
        if "m" not in _CH:
            args = default_args("continuous_IGEVStereo")
            m = __models__["continuous_IGEVStereo"](args)
            fill_module_deterministic(m, base_seed=2)
            _CH["m"] = m.to(dev)
            _CH["b"] = synthetic_train_batch(1, 64, 128, n_query=2048, seed=9, device=dev)
            _CH["args"] = args
        m, b = _CH["m"], _CH["b"]
        if k == "infer":
            m.eval()
            with torch.no_grad():
                m(b[0], b[1], iters=2, test_mode=True, hr_coord=b[2].clone(), scale=b[4])
        elif k == "trainfwd":
            m.train()
            with torch.no_grad():
                m(b[0], b[1], iters=2, hr_coord=b[2].clone(), scale=b[4])
        else:
            m.train()
            m.zero_grad()
            _, preds = m(b[0], b[1], iters=2, hr_coord=b[2].clone(), scale=b[4])
            sum(p.mean() for p in preds).backward()


def install_preclip_probe(trainer):
    """With ANYSTEREO_TRAIN_GRAPH_SCOPE=grads the clip + optimizer run eagerly after the replay: list the gradients that are
    non-finite BEFORE clipping (one such tensor turns every gradient into NaN through the clip coefficient)."""
    import torch.nn.utils as U
    orig = U.clip_grad_norm_
    names = {id(p): n for n, p in trainer.model.named_parameters()}

    def probe(params, clip, *a, **k):
        params = list(params)
        if any(id(p) in names for p in params):
            torch.cuda.synchronize()
            bad = [(names[id(p)], int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for p in params
                   if id(p) in names and p.grad is not None and not torch.isfinite(p.grad).all()]
            mx = max((p.grad.abs().max().item() for p in params if id(p) in names and p.grad is not None and torch.isfinite(p.grad).all()), default=0.0)
            print(f"    pre-clip: {len(bad)} non-finite gradient tensors {bad[:8]}; largest finite |g| {mx:.3e}", flush=True)
        return orig(params, clip, *a, **k)
    U.clip_grad_norm_ = probe
    torch.nn.utils.clip_grad_norm_ = probe


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--sync", default="device", choices=["none", "stream", "device"])
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--tol", type=float, default=2e-3)
    ap.add_argument("--hw", type=int, nargs=2, default=[160, 320])
    ap.add_argument("--no-sort", action="store_true", help="model.sort_queries = False")
    ap.add_argument("--loss-scale", type=float, default=None)
    ap.add_argument("--fast", action="store_true", help="leave cudnn.deterministic off")
    ap.add_argument("--solo", action="store_true", help="no eager trainer beside the graphed one (nothing else allocates between replays)")
    ap.add_argument("--churn", type=int, default=0, help="with --solo: allocate and free this many MB of scratch tensors between replays")
    ap.add_argument("--one-batch", action="store_true", help="the same batch every step")
    ap.add_argument("--inspect", action="store_true", help="after every graphed step: which gradients / parameters / buffers are non-finite")
    ap.add_argument("--churn-kind", default="alloc", help="with --solo: what runs between replays: alloc | gemm | conv | conv3d | bn | infer | trainfwd | trainstep_small")
    a = ap.parse_args()
    dev = "cuda:0"
    args = default_args("continuous_IGEVStereo")
    torch.backends.cudnn.deterministic = not a.fast

    def fresh():
        m = __models__["continuous_IGEVStereo"](args)
        fill_module_deterministic(m, base_seed=1)
        if a.no_sort:
            m.sort_queries = False
        return m.to(dev)

    batches = [synthetic_train_batch(a.batch, a.hw[0], a.hw[1], seed=s, device=dev) for s in range(2)]
    eager = Trainer(fresh(), lr=a.lr, num_steps=1000, train_iters=a.iters, max_disp=args.max_disp, graph=False, loss_scale=a.loss_scale)
    graphed = Trainer(fresh(), lr=a.lr, num_steps=1000, train_iters=a.iters, max_disp=args.max_disp, graph=True, loss_scale=a.loss_scale)
    assert graphed.use_graph and not eager.use_graph
    if a.inspect and graphed.graph_scope != "step":
        install_preclip_probe(graphed)
    bad = 0
    t_e = t_g = 0.0
    for i in range(a.steps + graphed.graph_warmup):
        b = tuple(t.clone() for t in batches[0 if a.one_batch else i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if a.solo:
            le = torch.zeros(())
            if a.churn:
                churn(a, dev, i)
        else:
            le, _ = eager.step(tuple(t.clone() for t in b))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        lg, _ = graphed.step(tuple(t.clone() for t in b))
        torch.cuda.synchronize() if a.sync == "device" else (torch.cuda.current_stream().synchronize() if a.sync == "stream" else None)
        lgv = float(lg)
        t2 = time.perf_counter()
        if a.inspect:
            torch.cuda.synchronize()
            bad_g = [n for n, p in graphed.model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
            bad_p = [n for n, p in graphed.model.named_parameters() if not torch.isfinite(p).all()]
            bad_b = [n for n, b_ in graphed.model.named_buffers() if b_.is_floating_point() and not torch.isfinite(b_).all()]
            gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in graphed.model.parameters() if p.grad is not None)).item()
            print(f"    inspect: grad norm (after clip) {gn:.4e}; non-finite grads {len(bad_g)} {bad_g[:6]}; params {len(bad_p)} {bad_p[:4]}; buffers {len(bad_b)} {bad_b[:4]}", flush=True)
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
