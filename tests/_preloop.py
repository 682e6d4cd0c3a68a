"""Capture the tensors the GRU loop starts from (continuous_IGEVstereo.py:245-276) out of a product / oracle model without touching
its code: the two hooks every forward passes them through are wrapped on the instance.  Compared against tests/golden/preloop_igev.npz
(the imported reference's values at two shapes; make_golden.py::golden_preloop)."""
import numpy as np
import torch


def capture_preloop(model, img1, img2, coord, scale):
    cap = {}
    lookup0, iterate0 = model._hot_lookup_fn, model._iterate

    def lookup(match_left, match_right, gev):
        cap["match_left"], cap["match_right"], cap["gev"] = (t.detach().float().clone() for t in (match_left, match_right, gev))
        return lookup0(match_left, match_right, gev)

    def iterate(geo_fn, net_list, inp_list, init_disp, *a, **k):
        for i, t in enumerate(net_list):
            cap[f"net{i}"] = t.detach().float().clone()
        for i, cs in enumerate(inp_list):
            cap[f"ctx{i}"] = torch.cat([c.detach().float() for c in cs], dim=1)  # cz | cr | cq = the conv's output (:273)
        return iterate0(geo_fn, net_list, inp_list, init_disp, *a, **k)

    model._hot_lookup_fn, model._iterate = lookup, iterate
    try:
        with torch.no_grad():
            init_disp, preds = model(img1, img2, iters=1, test_mode=False, hr_coord=coord.clone(), scale=scale)
    finally:
        del model._hot_lookup_fn, model._iterate
    cap["init_disp"], cap["pred_0"] = init_disp.detach().float(), preds[0].detach().float()
    return cap


def check_preloop(cap, g, key, rtol, what=""):
    """Every captured tensor against the reference's stored subsample (every `stride`-th element) and whole-tensor sums; returns
    {tensor: max |d| / max |ref|}."""
    stride = int(g["stride"])
    worst = {}
    for n, t in cap.items():
        want = g[f"{key}.{n}"]
        want = want if torch.is_tensor(want) else torch.from_numpy(np.asarray(want))
        assert list(t.shape) == [int(v) for v in g[f"{key}.{n}.shape"]], (what, key, n, tuple(t.shape))
        got = t.detach().cpu().reshape(-1)[::stride]
        scale = want.abs().max().item()
        e = (got.double() - want.double()).abs().max().item() / max(scale, 1e-30)
        worst[n] = e
        assert e <= rtol, f"{what} {key} {n}: max |d| / max |ref| = {e:.3e} > {rtol:.1e}"
        s_ref = [float(v) for v in g[f"{key}.{n}.sums"]]
        s_got = float(t.detach().cpu().double().abs().sum())
        assert abs(s_got - s_ref[1]) <= 10 * rtol * s_ref[1] + 1e-6, f"{what} {key} {n}: sum |t| {s_got} vs {s_ref[1]}"
    return worst
