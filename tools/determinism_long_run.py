"""Deterministic training mode over several optimizer steps (cfg 4 shapes): two Trainers from the same weights on the same batches,
ops.set_deterministic(True) + torch.backends.cudnn.deterministic — losses and final parameters must be the same BITS.
    python tools/determinism_long_run.py [steps] [graph:0|1]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch  # noqa: E402

from anystereo import ops  # noqa: E402
from anystereo.harness.synthetic import fill_module_deterministic  # noqa: E402
from anystereo.harness.train import Trainer, synthetic_train_batch  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
graph = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
dev = "cuda:0"
args = default_args("continuous_IGEVStereo")
torch.backends.cudnn.deterministic = True
batches = [synthetic_train_batch(4, 160, 320, seed=s, device=dev) for s in range(2)]


def run(det):
    ops.set_deterministic(det)
    m = __models__["continuous_IGEVStereo"](args)
    fill_module_deterministic(m, base_seed=1)
    tr = Trainer(m.to(dev), lr=1e-4, num_steps=1000, train_iters=16, max_disp=args.max_disp, graph=graph)
    losses = []
    for i in range(steps):
        loss, _ = tr.step(tuple(t.clone() for t in batches[i % 2]))
        losses.append(float(loss))
    torch.cuda.synchronize()
    return losses, torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()


for det in (True, False):
    la, pa = run(det)
    lb, pb = run(det)
    same = torch.equal(pa, pb)
    print(f"deterministic={det} graph={graph}: {steps} steps; losses equal: {la == lb}; parameters bit-equal after the last step: {same}; "
          f"max |dp| {(pa - pb).abs().max().item():.2e}; loss {la[0]:.4f} -> {la[-1]:.4f}", flush=True)
ops.set_deterministic(False)
