#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "model_options and raft" 2>&1 | tail -3
timeout -k 10 300 python - <<'PY'
import sys, torch
sys.path[:0] = [".", "any-stereo_amd", "tests"]
from anystereo import ops
from anystereo.harness.synthetic import fill_module_deterministic, synthetic_pair
from anystereo.models import __models__, default_args
from oracle import ops as O
DEV = "cuda:0"
ops.set_precision("fp32")
key = "continuous_RAFTStereo"
model = __models__[key](default_args(key)).eval()
fill_module_deterministic(model, base_seed=1)
model = model.to(DEV)
H, W = 64, 96
img1, img2 = (t.to(DEV) for t in synthetic_pair(1, H, W, shift=6, seed=99))
coord = O.make_coord([round(H * 1.5), round(W * 1.5)]).view(1, -1, 2).to(DEV)
sc = torch.tensor([[1.5]], device=DEV)
from anystereo.nn.encoders import MultiBasicEncoder
with torch.no_grad():
    outs = [model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=sc) for _ in range(4)]
    print("runs equal to run 0:", [bool(torch.equal(o, outs[0])) for o in outs], [(o - outs[0]).abs().max().item() for o in outs])
    # capture intermediate: context outputs, fnet
    def stage(m):
        i1 = (2 * (img1 / 255.0) - 1.0).contiguous(); i2 = (2 * (img2 / 255.0) - 1.0).contiguous()
        c = m.cnet(i1, num_layers=3)
        f = m.fnet([i1, i2]) if hasattr(m, "fnet") else None
        return c, f
    a, b = stage(model), stage(model)
    for lv in range(3):
        for k in range(2):
            print("cnet", lv, k, bool(torch.equal(a[0][lv][k], b[0][lv][k])))
    if a[1] is not None:
        print("fnet", [bool(torch.equal(x, y)) for x, y in zip(a[1], b[1])])
PY
