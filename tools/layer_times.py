"""Per-module GPU time of one eager forward (HIP events around every leaf module), grouped by top-level child.
    python tools/layer_times.py [--iters 2]
"""
import argparse
import os
import sys
from collections import defaultdict

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo.harness.query import pad_for_multi_train  # noqa: E402
from anystereo.harness.synthetic import fill_module_deterministic, synthetic_pair  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2)
    a = ap.parse_args()
    dev = "cuda:0"
    model = __models__["continuous_IGEVStereo"](default_args("continuous_IGEVStereo")).eval()
    fill_module_deterministic(model, base_seed=1)
    model = model.to(dev)
    img1, img2 = synthetic_pair(1, 540, 960, shift=8, seed=1234)
    i1, i2, coord, _ = pad_for_multi_train(img1, img2, 1.0, divis_by=32)
    i1, i2, coord = i1.to(dev), i2.to(dev), coord.unsqueeze(0).to(dev)
    sc = torch.tensor([[1.0]], device=dev)
    with torch.no_grad():
        for _ in range(2):
            model(i1, i2, iters=a.iters, test_mode=True, hr_coord=coord.clone(), scale=sc)
    recs = []

    def pre(m, inp):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        m._t0 = e

    def post(name):
        def f(m, inp, out):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            shp = tuple(inp[0].shape) if inp and torch.is_tensor(inp[0]) else None
            recs.append((name, type(m).__name__, m._t0, e, shp))
        return f
    for name, m in model.named_modules():
        if len(list(m.children())) == 0 and not name.startswith("update_block"):
            m.register_forward_pre_hook(pre)
            m.register_forward_hook(post(name))
    with torch.no_grad():
        model(i1, i2, iters=a.iters, test_mode=True, hr_coord=coord.clone(), scale=sc)
    torch.cuda.synchronize()
    rows = [(n, t, s.elapsed_time(e) * 1e3, shp) for n, t, s, e, shp in recs]
    grp, typ = defaultdict(float), defaultdict(float)
    for n, t, us, _ in rows:
        grp[n.split(".")[0]] += us
        typ[t] += us
    print("total leaf-module us: %.0f" % sum(r[2] for r in rows))
    print("by top-level:", {k: round(v) for k, v in sorted(grp.items(), key=lambda kv: -kv[1])})
    print("by type:", {k: round(v) for k, v in sorted(typ.items(), key=lambda kv: -kv[1])})
    for n, t, us, shp in sorted(rows, key=lambda r: -r[2])[:45]:
        print(f"{us:8.1f} us  {t:18s} {n:45s} {shp}")


if __name__ == "__main__":
    main()
