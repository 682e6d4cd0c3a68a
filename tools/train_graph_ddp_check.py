"""The gradient half of the cfg-4 step as a captured graph + eager gradient exchange + eager update, step by step against the
eager step (run on the GPU box; one rank through RCCL unless --no-pg):

    python tools/train_graph_ddp_check.py [--steps 10] [--no-pg] [--no-allreduce] [--scope grads|step]

Runs the eager trainer first (fresh model, N steps on one batch), then a fresh graphed trainer on the same batch, and prints both
loss sequences.  Exit code 1 when a replayed step's loss differs by more than --tol.
"""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402
import torch.distributed as td  # noqa: E402

from anystereo.harness.synthetic import fill_module_deterministic  # noqa: E402
from anystereo.harness.train import Trainer, synthetic_train_batch  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-pg", action="store_true")
    ap.add_argument("--no-allreduce", action="store_true", help="process group alive, but the gradient exchange is skipped")
    ap.add_argument("--scope", default="grads")
    ap.add_argument("--tol", type=float, default=2e-3)
    ap.add_argument("--sync-before", action="store_true", help="device-wide synchronisation before every replay")
    ap.add_argument("--inspect", action="store_true", help="after every graphed step: static inputs, metrics, gradient norm, parameters")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ["ANYSTEREO_TRAIN_GRAPH_SCOPE"] = a.scope
    if not a.no_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29777")
        td.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    args = default_args("continuous_IGEVStereo")
    torch.backends.cudnn.deterministic = True

    def fresh(graph):
        m = __models__["continuous_IGEVStereo"](args)
        fill_module_deterministic(m, base_seed=1)
        tr = Trainer(m.to(dev), lr=2e-4, num_steps=1000, train_iters=16, max_disp=args.max_disp, graph=graph,
                     force_ddp=not a.no_pg, ddp_impl="flat")
        if a.no_allreduce:
            tr.allreduce_gradients_flat = lambda: 0
        return tr
    batch = synthetic_train_batch(4, 160, 320, seed=0, device=dev)
    n = a.steps + 3
    eager = fresh(False)
    le = [float(eager.step(tuple(t.clone() for t in batch))[0]) for _ in range(n)]
    del eager
    torch.cuda.synchronize()
    gr = fresh(True)
    print(f"graphed trainer: use_graph={gr.use_graph} scope={gr.graph_scope} exchange={gr.ddp_mode} fill={os.environ.get('ANYSTEREO_TRAIN_GRAPH_FILL', '1')}", flush=True)
    bad = 0
    for i in range(n):
        if a.sync_before:
            torch.cuda.synchronize()
        lg_t, met = gr.step(tuple(t.clone() for t in batch))
        lg = float(lg_t)
        if a.inspect and gr._graph is not None:
            torch.cuda.synchronize()
            ent = gr._graph
            same = [bool(torch.equal(s_, b_)) if j != 2 else bool(torch.equal(s_, b_.clamp(-1 + 1e-6, 1 - 1e-6))) for j, (s_, b_) in enumerate(zip(ent["batch"], batch))]
            gn = [p.grad for p in gr.model.parameters() if p.grad is not None]
            gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in gn)).item()
            nz = sum(int((g != 0).any()) for g in gn)
            pfin = all(bool(torch.isfinite(p).all()) for p in gr.model.parameters())
            ptrs = {"loss": ent["loss"].data_ptr(), **{k: v.data_ptr() for k, v in ent["metrics"].items()}}
            stat = {k: round(float(v), 4) for k, v in ent["metrics"].items()}
            seg = {}
            for sg in torch.cuda.memory_snapshot():
                for k, ptr in ptrs.items():
                    if sg["address"] <= ptr < sg["address"] + sg["total_size"]:
                        off, st_ = sg["address"], "?"
                        for blk in sg["blocks"]:
                            if off <= ptr < off + blk["size"]:
                                st_ = blk["state"]
                            off += blk["size"]
                        seg[k] = (sg.get("segment_pool_id"), st_, blk["size"])
            print(f"    static outputs read now {stat} (loss {float(ent['loss']):.4f}); returned clones at {lg_t.data_ptr():#x} / { {k: hex(v.data_ptr()) for k, v in met.items()} }; "
                  f"static at { {k: hex(v) for k, v in ptrs.items()} }; allocator view {seg}", flush=True)
            print(f"    static inputs intact {same}; metrics { {k: round(float(v), 4) for k, v in met.items()} }; |grad| {gnorm:.4e} ({nz}/{len(gn)} tensors non-zero); params finite {pfin}", flush=True)
        rel = abs(lg - le[i]) / max(abs(le[i]), 1e-12)
        mode = "replay" if (gr._graph is not None and i >= gr.graph_warmup) else "warm-up"
        flag = "" if rel <= a.tol else "   <-- MISMATCH"
        bad += rel > a.tol
        print(f"step {i:2d} [{mode:7s}] eager {le[i]:.6f}  graphed {lg:.6f}  rel {rel:.2e}{flag}", flush=True)
    if not a.no_pg:
        td.destroy_process_group()
    print("memset nodes (replaced, left):", getattr(gr, "graph_memsets", None))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
