"""Wall time of a graphed forward of either model family:  python tools/time_model.py continuous_RAFTStereo [H W iters scale]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo.harness.query import pad_for_multi_train  # noqa: E402
from anystereo.harness.synthetic import fill_module_deterministic, synthetic_pair  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "continuous_RAFTStereo"
H, W, iters = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 540), (3, 960), (4, 32)))
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
dev = "cuda:0"
model = __models__[name](default_args(name)).eval()
fill_module_deterministic(model, base_seed=1)
model = model.to(dev)
img1, img2 = synthetic_pair(1, H, W, shift=8, seed=1234)
i1, i2, coord, _ = pad_for_multi_train(img1, img2, scale, divis_by=32 if "IGEV" in name else 16)
i1, i2, coord = i1.to(dev), i2.to(dev), coord.unsqueeze(0).to(dev)
sc = torch.tensor([[scale]], device=dev)
model.enable_graph(True)
with torch.no_grad():
    for _ in range(2):
        out = model(i1, i2, iters=iters, test_mode=True, hr_coord=coord, scale=sc)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        out = model(i1, i2, iters=iters, test_mode=True, hr_coord=coord, scale=sc)
    torch.cuda.synchronize()
print(f"{name} {W}x{H} x{scale} {iters} iters: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms / pair, finite={bool(torch.isfinite(out).all())}")
