"""Which stage of the eager RAFT forward differs between two identical runs?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch  # noqa: E402

from anystereo import ops  # noqa: E402
from anystereo.harness.synthetic import fill_module_deterministic, synthetic_pair  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402
from oracle import ops as O  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
DEV = "cuda:0"
ops.set_precision(mode)
key = "continuous_RAFTStereo"
model = __models__[key](default_args(key)).eval()
fill_module_deterministic(model, base_seed=1)
model = model.to(DEV)
H, W = 64, 96
img1, img2 = (t.to(DEV) for t in synthetic_pair(1, H, W, shift=6, seed=99))
coord = O.make_coord([round(H * 1.5), round(W * 1.5)]).view(1, -1, 2).to(DEV)
sc = torch.tensor([[1.5]], device=DEV)

caps = []


def run():
    cap = {}
    it0, up0, lk0 = model._iterate, model.upsample_disp, model._hot_lookup_fn

    def lookup(*a):
        for i, t in enumerate(a):
            cap[f"lookup_in{i}"] = t.detach().clone()
        fn = lk0(*a)
        for i, t in enumerate(fn.init_corr_pyramid):
            cap[f"corr_lv{i}"] = t.detach().clone()
        return fn

    def iterate(fn, net, inp, disp, *a, **k):
        for i, t in enumerate(net):
            cap[f"net{i}"] = t.detach().clone()
        for i, cs in enumerate(inp):
            for j, c in enumerate(cs):
                cap[f"ctx{i}.{j}"] = c.detach().clone()
        r = it0(fn, net, inp, disp, *a, **k)
        cap["disp_final"] = r[0].detach().clone()
        return r

    def up(disp, hidden, s4, s2, s1, hr_coord=None, scale=1):
        cap["up_disp"] = disp.detach().clone()
        cap["up_hidden"] = hidden.detach().clone()
        for n, t in (("s4", s4), ("s2", s2), ("s1", s1)):
            if t is not None:
                cap["up_" + n] = t.detach().clone()
        r = up0(disp, hidden, s4, s2, s1, hr_coord=hr_coord, scale=scale)
        cap["up_out"] = r.detach().clone()
        return r

    model._iterate, model.upsample_disp, model._hot_lookup_fn = iterate, up, lookup
    try:
        with torch.no_grad():
            cap["out"] = model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=sc).clone()
    finally:
        del model._iterate, model.upsample_disp, model._hot_lookup_fn
    if os.environ.get("SYNC_BETWEEN", "0") == "1":
        torch.cuda.synchronize()
    return cap


runs = [run() for _ in range(6)]
torch.cuda.synchronize()
a = runs[0]
for k, b in enumerate(runs[1:]):
    print("run", k + 1, "vs 0:", {n: "%.1e" % (a[n].float() - b[n].float()).abs().max().item() for n in a if not torch.equal(a[n], b[n])} or "all equal")
