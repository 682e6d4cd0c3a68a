"""List the host synchronisations of one steady-state cfg-4 training step (torch.cuda.set_sync_debug_mode("warn")): each one is a
point where the host stops issuing work until the GPU has caught up."""
import os
import sys
import warnings

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch  # noqa: E402

from anystereo.harness.synthetic import fill_module_deterministic  # noqa: E402
from anystereo.harness.train import Trainer, synthetic_train_batch  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402

dev = "cuda:0"
args = default_args("continuous_IGEVStereo")
model = __models__["continuous_IGEVStereo"](args)
fill_module_deterministic(model, base_seed=1)
model = model.to(dev)
tr = Trainer(model, train_iters=16, max_disp=args.max_disp, graph=False)
batch = synthetic_train_batch(4, n_query=51200, device=dev)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    tr.step(batch)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print(len(w), "synchronising calls in one step")
seen = {}
for x in w:
    key = f"{x.filename.replace(ROOT, '.')}:{x.lineno}  {str(x.message)[:90]}"
    seen[key] = seen.get(key, 0) + 1
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
