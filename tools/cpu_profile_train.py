"""Where the host time of a cfg-4 training step goes (cProfile over a few steady-state steps): the step is launch-bound once
the kernels are fast, so Python / dispatcher overhead per launch is the quantity to watch."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch  # noqa: E402

from anystereo.harness import workloads as WL  # noqa: E402
from anystereo.harness.train import Trainer, synthetic_train_batch  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402
from anystereo.harness.synthetic import fill_module_deterministic  # noqa: E402

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
import time  # noqa: E402
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(batch)
pr.disable()
torch.cuda.synchronize()
print("wall per step (profiled) %.1f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
