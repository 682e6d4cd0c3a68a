"""Run-to-run equality of the eager RAFT / IGEV forward in one precision mode under schedule switches (a difference between two
identical runs = a race or an uninitialised read).   python tools/race_probe.py raft fp32"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch  # noqa: E402

from anystereo import ops  # noqa: E402
from anystereo.harness.synthetic import fill_module_deterministic, synthetic_pair  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402
from oracle import ops as O  # noqa: E402

name, mode = sys.argv[1], sys.argv[2]
DEV = "cuda:0"
ops.set_precision(mode)
key = "continuous_RAFTStereo" if name == "raft" else "continuous_IGEVStereo"
model = __models__[key](default_args(key)).eval()
fill_module_deterministic(model, base_seed=1)
model = model.to(DEV)
H, W = 64, 96 if name == "raft" else 128
img1, img2 = (t.to(DEV) for t in synthetic_pair(1, H, W, shift=6, seed=99))
coord = O.make_coord([round(H * 1.5), round(W * 1.5)]).view(1, -1, 2).to(DEV)
sc = torch.tensor([[1.5]], device=DEV)
if os.environ.get("SERIAL") == "1":
    model.serial_streams = True
noise = os.environ.get("NOISE")
ns = torch.cuda.Stream() if noise else None
na = torch.randn(4096, 4096, device=DEV) if noise else None
with torch.no_grad():
    outs = []
    for _ in range(int(os.environ.get("RUNS", "6"))):
        if noise:  # a competing stream of large GEMMs / memory sweeps: different CU availability and timing for every launch
            with torch.cuda.stream(ns):
                for _k in range(int(noise)):
                    nb = na @ na if (_k & 1) else na * 1.0001
            del nb
        outs.append(model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=sc))
        if os.environ.get("SYNC_BETWEEN") == "1":
            torch.cuda.synchronize()
        if os.environ.get("SYNC_BETWEEN") == "main":
            torch.cuda.current_stream().synchronize()
    print(name, mode, {k: os.environ[k] for k in os.environ if k.startswith("ANYSTEREO") or k in ("SERIAL", "SYNC_BETWEEN", "NOISE", "RUNS")},
          "max |run_i - run_0|:", ["%.1e" % (o - outs[0]).abs().max().item() for o in outs])
