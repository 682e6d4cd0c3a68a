"""Eager run-to-run determinism of the cfg-2 forward:  python tools/det_check.py [iters] [trials]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo.harness.query import pad_for_multi_train  # noqa: E402
from anystereo.harness.synthetic import fill_module_deterministic, synthetic_pair  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 32
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda:0"
torch.backends.cudnn.deterministic = os.environ.get("DET", "0") == "1"  # MIOpen: exclude atomics-based solvers
model = __models__["continuous_IGEVStereo"](default_args("continuous_IGEVStereo")).eval()
fill_module_deterministic(model, base_seed=1)
model = model.to(dev)
img1, img2 = synthetic_pair(1, 540, 960, shift=8, seed=1234)
i1, i2, coord, _ = pad_for_multi_train(img1, img2, 1.0, divis_by=32)
i1, i2, coord = i1.to(dev), i2.to(dev), coord.unsqueeze(0).to(dev)
sc = torch.tensor([[1.0]], device=dev)
with torch.no_grad():
    model(i1, i2, iters=2, test_mode=True, hr_coord=coord.clone(), scale=sc)
    ref = model(i1, i2, iters=iters, test_mode=True, hr_coord=coord.clone(), scale=sc)
    for t in range(trials):
        out = model(i1, i2, iters=iters, test_mode=True, hr_coord=coord.clone(), scale=sc)
        print(f"trial {t}: equal={torch.equal(out, ref)} maxdiff={(out - ref).abs().max().item():.3e}", flush=True)
