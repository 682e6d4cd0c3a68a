"""Bit-repeatability of the RAFT-shaped volume / lookup kernels (C = 256, L = 4, G = 0) and the upsampler, launch after launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch  # noqa: E402

from anystereo import ops  # noqa: E402
from anystereo.harness.synthetic import det_uniform  # noqa: E402

DEV = "cuda:0"
for mode in ("fp32", "split"):
    ops.set_precision(mode)
    for (h, w) in ((16, 24), (64, 128)):
        f1, f2 = det_uniform((1, 256, h, w), 1).to(DEV), det_uniform((1, 256, h, w), 2).to(DEV)
        disp = det_uniform((1, 1, h, w), 3, 0.0, 20.0).to(DEV)
        ref = ops.corr_build_pyramid(f1, f2, 4)
        refl = ops.geo_corr_lookup(None, ref, disp, 4)
        bad_b = bad_l = 0
        for _ in range(50):
            lv = ops.corr_build_pyramid(f1, f2, 4)
            bad_b += int(any(not torch.equal(a, b) for a, b in zip(lv, ref)))
            bad_l += int(not torch.equal(ops.geo_corr_lookup(None, lv, disp, 4), refl))
        print(mode, (h, w), "corr_build differing launches:", bad_b, "/50; lookup:", bad_l, "/50")

# stage-level repeatability of the RAFT model (no synchronisation between repeats): fnet, cnet, volume + first lookup, update block
from anystereo.harness.synthetic import fill_module_deterministic, synthetic_pair  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402

for mode in ("fp32", "split"):
    ops.set_precision(mode)
    model = __models__["continuous_RAFTStereo"](default_args("continuous_RAFTStereo")).eval()
    fill_module_deterministic(model, base_seed=1)
    model = model.to(DEV)
    img1, img2 = (t.to(DEV) for t in synthetic_pair(1, 64, 96, shift=6, seed=99))
    i1, i2 = (2 * (img1 / 255.0) - 1.0).contiguous(), (2 * (img2 / 255.0) - 1.0).contiguous()
    with torch.no_grad():
        def flat(o):
            if torch.is_tensor(o):
                return [o]
            return [t for x in o for t in flat(x)]
        for name, fn in (("fnet", lambda: model.fnet([i1, i2])), ("cnet", lambda: model.cnet(i1, num_layers=3)),
                         ("context", lambda: model._context(i1))):
            ref = flat(fn())
            bad = 0
            for _ in range(30):
                bad += int(any(not torch.equal(a, b) for a, b in zip(flat(fn()), ref)))
            print(mode, name, "differing repeats:", bad, "/30")
