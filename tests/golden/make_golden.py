"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (read-only) in the
build container and running its own functions on deterministic inputs.

    python tests/golden/make_golden.py            # needs /root/reference; writes tests/golden/*.npz

The reference ships no tests or golden vectors, so these files are the parity pin of
the oracle (oracle/ops.py, oracle/model.py) and, through it, of the HIP path.  Only inputs and
reference OUTPUTS are stored — no reference source travels.  Import recipe: SURVEY.md Appendix B
(five in-memory shims, no edits to the reference).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))

from anystereo.harness.synthetic import det_uniform, fill_module_deterministic, synthetic_pair, tiny_train_case  # noqa: E402
from anystereo.models.base import default_args  # noqa: E402
from anystereo.nn.encoders import MobileNetV2Trunk  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    for n in ("opt_einsum", "timm"):
        sys.modules[n] = types.ModuleType(n)
    sys.modules["opt_einsum"].contract = torch.einsum
    # timm is unavailable offline: the same MobileNetV2 trunk is injected on both sides
    sys.modules["timm"].create_model = lambda *a, **k: MobileNetV2Trunk()
    pkg = types.ModuleType("models")
    pkg.__path__ = [REF + "/models"]
    sys.modules["models"] = pkg
    import models.coreContinuous_IGEV.submodule as _s
    a = types.ModuleType("models.coreContinuous_A2A4IGEV")
    a.__path__ = []
    sys.modules["models.coreContinuous_A2A4IGEV"] = a
    sys.modules["models.coreContinuous_A2A4IGEV.submodule"] = _s
    torch.Tensor.cuda = lambda self, *a, **k: self  # liif.py hard-codes .cuda()


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


# parameters whose full gradient is stored (the rest: L2 norms only) — one per hot-path operator family + the
# backbone tensors that receive their gradient THROUGH the volume build / gwc / lookup backward
TRAIN_FULL = {
    "igev": ["update_block.encoder.convc1.bias", "update_block.encoder.convd1.weight", "update_block.gru04.convq.bias",
             "update_block.gru16.convz.bias", "update_block.disp_head.conv2.weight", "liif_up.imnet.layers.0.weight",
             "liif_up.imnet.layers.6.weight", "desc.bias", "corr_stem.bn.weight", "corr_stem.conv.weight",
             "context_zqr_convs.0.bias", "stem_2.head.1.bias"],
    "raft": ["update_block.encoder.convc1.bias", "update_block.encoder.convd1.weight", "update_block.gru04.convq.bias",
             "update_block.disp_head.conv2.weight", "liif_up.imnet.layers.0.weight", "liif_up.imnet.layers.6.weight",
             "context_zqr_convs.0.bias", "fnet.conv2.bias"],
}


def golden_train(RefIGEV, RefRAFT, ns2):
    """G8: one training forward/backward of the imported reference — loss and parameter gradients."""
    torch.set_grad_enabled(True)
    for name, Ref in (("igev", RefIGEV), ("raft", RefRAFT)):
        args = default_args("continuous_IGEVStereo" if name == "igev" else "continuous_RAFTStereo")
        model = Ref(args)
        fill_module_deterministic(model, base_seed=1)
        model.train()
        model.freeze_bn()  # train_continuous_IGEV.py:189
        H, W, img1, img2, coord, gt, scale = tiny_train_case(name)
        res = model(img1, img2, iters=3, hr_coord=coord.clone(), scale=scale)
        preds = res[1] if name == "igev" else res
        loss, met = ns2["sequence_loss_multiscale"](preds, gt, ((gt < 512) & (gt > 0)).float(), max_disp=args.max_disp)
        loss.backward()
        named = dict(model.named_parameters())
        names = sorted(n for n, p in named.items() if p.grad is not None)
        norms = np.array([float(named[n].grad.double().norm()) for n in names])
        full = {f"g{i}": named[n].grad for i, n in enumerate(TRAIN_FULL[name])}
        save(f"train_{name}", loss=loss, names=np.array(names), norms=norms, full_names=np.array(TRAIN_FULL[name]),
             last_pred=preds[-1], **full)
        print(name, "train loss", float(loss), "params with grad", len(names), "max norm", norms.max())
    torch.set_grad_enabled(False)


def golden_train_sensitivity(RefIGEV, RefRAFT, ns2, seeds=32, rel=1e-6, seed0=1000, merge=False):
    """G8 companion: how far the imported reference's OWN gradients move when its two input images are perturbed by a relative
    N(0, rel) — the size of the forward differences between two correct fp32 implementations (summation order, another library
    convolution algorithm).  The gradient of a ReLU network is piecewise constant in its activation pattern: a handful of the
    ~10^6 pre-activations of the tiny fixture lie within 1e-6 (relative) of zero, each one that changes side moves individual
    gradient tensors by a DISCRETE amount (RAFT convd1.weight: 5e-4 ... 1e-2 of its maximum per flip; disp_head.conv2.weight:
    1e-7).  Stored per parameter: the largest relative deviation of its gradient norm; per stored full tensor: the largest
    max-norm deviation relative to the tensor's maximum.  tests/test_hip_parity.py::test_training_step_vs_reference derives its
    per-tensor limits from these numbers instead of one fixed tolerance."""
    torch.set_grad_enabled(True)
    for name, Ref in (("igev", RefIGEV), ("raft", RefRAFT)):
        args = default_args("continuous_IGEVStereo" if name == "igev" else "continuous_RAFTStereo")
        model = Ref(args)
        fill_module_deterministic(model, base_seed=1)
        model.train()
        model.freeze_bn()
        H, W, img1, img2, coord, gt, scale = tiny_train_case(name)

        def run(seed):
            a, b = img1, img2
            if seed is not None:
                g = torch.Generator().manual_seed(seed0 + seed)
                a = a * (1.0 + rel * torch.randn(a.shape, generator=g))
                b = b * (1.0 + rel * torch.randn(b.shape, generator=g))
            model.zero_grad(set_to_none=True)
            res = model(a, b, iters=3, hr_coord=coord.clone(), scale=scale)
            preds = res[1] if name == "igev" else res
            loss, _ = ns2["sequence_loss_multiscale"](preds, gt, ((gt < 512) & (gt > 0)).float(), max_disp=args.max_disp)
            loss.backward()
            return {n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.grad is not None}

        g0 = run(None)
        names = sorted(g0)
        n0 = np.array([float(g0[n].norm()) for n in names])
        norm_dev = np.zeros(len(names))
        full_dev = np.zeros(len(TRAIN_FULL[name]))
        full_dev_p90 = np.zeros(len(TRAIN_FULL[name]))
        for s in range(seeds):
            g = run(s)
            nn_ = np.array([float(g[n].norm()) for n in names])
            norm_dev = np.maximum(norm_dev, np.abs(nn_ - n0) / (n0 + 1e-6 * n0.max()))
            for i, n in enumerate(TRAIN_FULL[name]):
                d = (g[n] - g0[n]).abs() / g0[n].abs().max()
                full_dev[i] = max(full_dev[i], float(d.max()))
                full_dev_p90[i] = max(full_dev_p90[i], float(d.flatten().kthvalue(max(1, int(0.9 * d.numel()))).values))
            print(name, "seed", s, "full_dev", " ".join(f"{v:.1e}" for v in full_dev), flush=True)
        runs = [f"{seeds} seeds from {seed0} at {rel:g} relative"]
        if merge:  # keep the larger deviation of this run and the stored one (more seeds, another noise level)
            old = np.load(os.path.join(HERE, f"train_{name}_sens.npz"))
            assert [str(n) for n in old["names"]] == names
            norm_dev, full_dev = np.maximum(norm_dev, old["norm_dev"]), np.maximum(full_dev, old["full_dev"])
            full_dev_p90 = np.maximum(full_dev_p90, old["full_dev_p90"])
            runs = ([str(r) for r in old["runs"]] if "runs" in old.files else [f"{int(old['seeds'])} seeds from 1000 at {float(old['rel']):g} relative"]) + runs
        save(f"train_{name}_sens", names=np.array(names), norm_dev=norm_dev, full_names=np.array(TRAIN_FULL[name]), full_dev=full_dev,
             full_dev_p90=full_dev_p90, seeds=seeds, rel=rel, runs=np.array(runs))
    torch.set_grad_enabled(False)


def margin_case(name, cand):
    """Inputs of candidate `cand` for the margin-screened G8 fixture: ONE small pair (a tenth of the main fixture's pre-activations,
    so a seed can be found whose ReLU pattern has a margin), 200 queries on the s = 1.5 grid, 2 GRU iterations."""
    from anystereo.nn.liif import make_coord
    h, w = 32, 64
    img1, img2 = synthetic_pair(1, h, w, shift=4, seed=500 + cand)
    s, nq = 1.5, 200
    grid = make_coord([round(h * s), round(w * s)])
    idx = (det_uniform((nq,), 900 + cand, 0.0, 1.0) * grid.shape[0]).long().clamp(max=grid.shape[0] - 1)
    coord = grid[idx].unsqueeze(0).contiguous()
    gt = det_uniform((1, 1, nq), 950 + cand, 0.5, 30.0)
    return img1, img2, coord, gt, torch.tensor([[s]])


class KinkMargin:
    """While active, every piecewise-linear activation the reference evaluates (relu / relu_ / leaky_relu / relu6 / hardtanh, as
    functions and as modules) reports how close its input comes to a kink, relative to the input's RMS: the smallest such ratio
    of a forward pass is the pass's MARGIN — a forward difference below it (summation order, another convolution algorithm,
    the 22-bit operand split) cannot change the activation pattern, so the gradient stays on the same linear piece."""

    def __init__(self):
        self.margin, self.count, self.where = float("inf"), 0, None

    def _see(self, x, kinks, tag):
        if not torch.is_tensor(x) or not x.is_floating_point() or x.numel() == 0:
            return
        xd = x.detach().double()
        rms = float(xd.pow(2).mean().sqrt())
        if rms == 0.0:
            return
        # exact hits are structural (zero padding, a ReLU of a ReLU's output): they stay on their side under any small perturbation
        m = min(float(d[d > 0].min()) if bool((d > 0).any()) else float("inf") for d in ((xd - k).abs() for k in kinks)) / rms
        self.count += x.numel()
        if m < self.margin:
            self.margin, self.where = m, f"{tag}{tuple(x.shape)}"

    def __enter__(self):
        import torch.nn.functional as F
        self._saved = []
        def wrap(mod, name, kinks):
            orig = getattr(mod, name)
            def f(x, *a, **k):
                self._see(x, kinks if not callable(kinks) else kinks(a, k), name)
                return orig(x, *a, **k)
            self._saved.append((mod, name, orig))
            setattr(mod, name, f)
        for mod in (F, torch):
            for name, kinks in (("relu", (0.0,)), ("relu_", (0.0,)), ("leaky_relu", (0.0,)), ("leaky_relu_", (0.0,)),
                                ("relu6", (0.0, 6.0))):
                if hasattr(mod, name):
                    wrap(mod, name, kinks)
        wrap(F, "hardtanh", lambda a, k: (float(k.get("min_val", a[0] if a else -1.0)), float(k.get("max_val", a[1] if len(a) > 1 else 1.0))))
        return self

    def __exit__(self, *exc):
        for mod, name, orig in self._saved:
            setattr(mod, name, orig)


def golden_train_margin(RefIGEV, RefRAFT, ns2, candidates=400, confirm=128, bar=3e-4, only_model=None, cand0=0):
    """G8, second fixture (review of round 4, item 4): a training step whose activation pattern has a MARGIN.  The gradient of a
    ReLU network is piecewise constant in its activation pattern; the main fixture has pre-activations within 1e-7 (relative)
    of zero, so eps-level forward differences move individual gradients in discrete steps (train_*_sens.npz: up to 5e-2) and
    its per-tensor limits are wide exactly where a regression would hide.  Here the imported reference screens `candidates`
    input seeds by the margin of their forward pass (KinkMargin) and keeps the one whose closest pre-activation is farthest from
    its kink; `confirm` perturbed runs of the reference (relative N(0, 2e-7) and N(0, 1e-6) on both images — the forward
    difference of two correct fp32 implementations and five times that) then show how far its own gradients move.  Stored: the
    inputs, loss, last prediction, all gradient norms, the TRAIN_FULL tensors, the margin and the deviations; the test's limit is
    a uniform 1e-3 for every tensor, which needs every deviation below `bar`."""
    torch.set_grad_enabled(True)
    for name, Ref in (("igev", RefIGEV), ("raft", RefRAFT)):
        if only_model and name != only_model:
            continue
        args = default_args("continuous_IGEVStereo" if name == "igev" else "continuous_RAFTStereo")
        model = Ref(args)
        fill_module_deterministic(model, base_seed=1)
        model.train()
        model.freeze_bn()

        def run(case, seed, rel, backward=True):
            img1, img2, coord, gt, scale = case
            a, b = img1, img2
            if seed is not None:
                g = torch.Generator().manual_seed(7000 + seed)
                a = a * (1.0 + rel * torch.randn(a.shape, generator=g))
                b = b * (1.0 + rel * torch.randn(b.shape, generator=g))
            model.zero_grad(set_to_none=True)
            res = model(a, b, iters=2, hr_coord=coord.clone(), scale=scale)
            preds = res[1] if name == "igev" else res
            loss, _ = ns2["sequence_loss_multiscale"](preds, gt, ((gt < 512) & (gt > 0)).float(), max_disp=args.max_disp)
            if not backward:
                return loss.detach(), preds[-1].detach(), None
            loss.backward()
            return loss.detach(), preds[-1].detach(), {n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.grad is not None}

        best = (-1.0, None, None, 0)
        for cand in range(cand0, cand0 + candidates):
            case = margin_case(name, cand)
            with KinkMargin() as km, torch.no_grad():
                run(case, None, 0.0, backward=False)
            if km.margin > best[0]:
                best = (km.margin, cand, km.where, km.count)
                print(name, "candidate", cand, "margin %.2e at %s over %d pre-activations" % (km.margin, km.where, km.count), flush=True)
        margin, cand, where, count = best
        print(name, "best margin %.2e (candidate %d) of %d candidates; %d pre-activations per pass" % (margin, cand, candidates, count), flush=True)
        if margin < 1e-5:
            # 2e-7 relative image noise (two correct fp32 forwards) reaches the pre-activations at ~1e-6 of their RMS: below 1e-5 a
            # pattern change, hence a discrete gradient step, stays likely and the fixture would be as fragile as the main one
            print(name, "no candidate has a margin >= 1e-5: no fixture written", flush=True)
            continue
        case = margin_case(name, cand)
        loss0, pred0, g0 = run(case, None, 0.0)
        names = sorted(g0)
        n0 = np.array([float(g0[n].norm()) for n in names])
        floor = 1e-3 * float(np.median(n0))   # tensors whose gradient is rounding noise (a bias in front of a normalisation)
        nd_all, fd_all = np.zeros(len(names)), np.zeros(len(TRAIN_FULL[name]))
        for s_ in range(confirm):
            _, _, g = run(case, 100 + s_, 2e-7 if s_ % 2 == 0 else 1e-6)
            nn_ = np.array([float(g[n].norm()) for n in names])
            nd_all = np.maximum(nd_all, np.abs(nn_ - n0) / np.maximum(n0, floor))
            fd_all = np.maximum(fd_all, np.array([float(((g[n] - g0[n]).abs() / g0[n].abs().max()).max()) for n in TRAIN_FULL[name]]))
            if s_ % 16 == 15:
                print(name, "candidate", cand, "after", s_ + 1, "perturbed runs: norm dev %.2e (%s) full dev %.2e" %
                      (nd_all.max(), names[int(nd_all.argmax())], fd_all.max()), flush=True)
        assert nd_all.max() < bar and fd_all.max() < bar, f"{name}: candidate {cand} (margin {margin:.2e}) still moves by {nd_all.max():.2e} / {fd_all.max():.2e}"
        img1, img2, coord, gt, scale = case
        full = {f"g{i}": g0[n].float() for i, n in enumerate(TRAIN_FULL[name])}
        save(f"train_margin_{name}", img1=img1, img2=img2, coord=coord, gt=gt, scale=scale, iters=2, candidate=cand, loss=loss0,
             last_pred=pred0, names=np.array(names), norms=n0, norm_floor=floor, full_names=np.array(TRAIN_FULL[name]), norm_dev=nd_all,
             full_dev=fd_all, margin=margin, margin_at=where, pre_activations=count,
             screen=f"best margin of {candidates} candidates from {cand0}; {confirm} perturbed reference runs alternating 2e-7 / 1e-6 relative", **full)
    torch.set_grad_enabled(False)


def golden_model_options(RefIGEV, RefRAFT, rliif):
    """G7 companion (review of round 4, item 6): the forward() branches evaluation.py can reach that the main G7 fixture does not
    take — `args.slow_fast_gru = True` (continuous_IGEVstereo.py:288-291, prune_raft_stereo.py:280-283: the low-resolution GRUs
    are pre-updated before every full update), RAFT's `output_raw=True` return tuple (prune_raft_stereo.py:293-296) and a
    `flow_init` argument (accepted and never read by either reference forward: outputs with and without it are stored)."""
    outs = {}
    for name, Ref, (H, W) in (("igev", RefIGEV, (64, 128)), ("raft", RefRAFT, (64, 96))):
        args = default_args("continuous_IGEVStereo" if name == "igev" else "continuous_RAFTStereo")
        args.slow_fast_gru = True
        model = Ref(args).eval()
        fill_module_deterministic(model, base_seed=1)
        img1, img2 = synthetic_pair(1, H, W, shift=6, seed=99)
        s = 1.5
        coord = rliif.make_coord([round(H * s), round(W * s)]).unsqueeze(0)
        sc = torch.tensor([[s]])
        outs[f"{name}_slowfast"] = model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=sc)
        model.args.slow_fast_gru = False
        base = model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=sc)
        fi = det_uniform((1, 1, H // 4, W // 4), 123, 0.0, 20.0)
        with_fi = model(img1, img2, iters=3, flow_init=fi, test_mode=True, hr_coord=coord.clone(), scale=sc)
        outs[f"{name}_base"], outs[f"{name}_flow_init_out"], outs[f"{name}_flow_init"] = base, with_fi, fi
        assert torch.equal(base, with_fi), "the reference reads flow_init after all: the fixture's premise is wrong"
        if name == "raft":
            raw = model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=sc, output_raw=True)
            assert isinstance(raw, tuple) and len(raw) == 2
            outs["raft_raw_disp"], outs["raft_raw_up"] = raw
        else:
            r2 = model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=sc, output_raw=True)
            assert torch.is_tensor(r2), "IGEV's forward ignores output_raw (continuous_IGEVstereo.py:303-305)"
            outs["igev_output_raw"] = r2
        outs[f"{name}_HW"] = np.array([H, W])
    save("model_opts", **outs)


PRELOOP_SIZES = ((64, 128), (96, 160))  # the G7 size and a non-square one whose 1/32 map is 3 x 5 (odd in both dimensions)
PRELOOP_STRIDE = 5                       # every 5th element of each flattened tensor is stored (plus fp64 sums of the whole)


def golden_preloop(RefIGEV, rliif):
    """Pre-loop tensors of the imported reference (continuous_IGEVstereo.py:245-276): the matching features, the geometry
    encoding volume, the initial disparity, the hidden states and the context terms the GRU loop starts from — what the backbone
    WIRING (stems, feature trunk, descriptor head, cost aggregation, context network, context_zqr_convs) produces.  The oracle's
    whole-model check shares the product's backbone modules; these vectors pin that wiring against the reference itself, at two
    shapes.  Captured with forward hooks (no edits to the reference)."""
    outs = {}
    for (H, W) in PRELOOP_SIZES:
        args = default_args("continuous_IGEVStereo")
        model = RefIGEV(args).eval()
        fill_module_deterministic(model, base_seed=1)
        img1, img2 = synthetic_pair(1, H, W, shift=6, seed=99)
        cap = {"desc": [], "zqr": []}
        hooks = [model.desc.register_forward_hook(lambda m, i, o: cap["desc"].append(o.detach().clone())),
                 model.cost_agg.register_forward_hook(lambda m, i, o: cap.__setitem__("gev", o.detach().clone())),
                 model.cnet.register_forward_hook(lambda m, i, o: cap.__setitem__("cnet", [[t.detach().clone() for t in lv] for lv in o]))]
        hooks += [c.register_forward_hook(lambda m, i, o: cap["zqr"].append(o.detach().clone())) for c in model.context_zqr_convs]
        coord = rliif.make_coord([H, W]).unsqueeze(0)
        init_disp, preds = model(img1, img2, iters=1, test_mode=False, hr_coord=coord.clone(), scale=torch.tensor([[1.0]]))
        for h_ in hooks:
            h_.remove()
        assert len(cap["desc"]) == 2 and len(cap["zqr"]) == 3 and len(cap["cnet"]) == 3
        tensors = {"match_left": cap["desc"][0], "match_right": cap["desc"][1], "gev": cap["gev"], "init_disp": init_disp,
                   "pred_0": preds[0]}
        for i in range(3):
            tensors[f"net{i}"] = torch.tanh(cap["cnet"][i][0])   # :271
            tensors[f"ctx{i}"] = cap["zqr"][i]                   # :273 before the split into cz | cr | cq
        key = f"{H}x{W}"
        for n, t in tensors.items():
            flat = t.reshape(-1)
            outs[f"{key}.{n}"] = flat[::PRELOOP_STRIDE].clone()
            outs[f"{key}.{n}.shape"] = np.array(t.shape)
            outs[f"{key}.{n}.sums"] = np.array([float(flat.double().sum()), float(flat.double().abs().sum())])
    save("preloop_igev", sizes=np.array(PRELOOP_SIZES), stride=PRELOOP_STRIDE, **outs)


SENS_ARGS = {}
MARGIN_ARGS = {}


def main(only=None):
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    import_reference()
    import models.coreContinuous_IGEV.geometry as rgeo
    import models.corePrune_RAFT.geometry as rgeo_raft
    import models.coreContinuous_IGEV.submodule as rsub
    import models.coreContinuous_IGEV.update as rupd
    import models.corePrune_RAFT.update as rupd_raft
    import models.coreContinuous_IGEV.liif as rliif
    from models.coreContinuous_IGEV.continuous_IGEVstereo import continuous_IGEVStereo as RefIGEV
    from models.corePrune_RAFT.prune_raft_stereo import continuous_RaftStereo as RefRAFT
    def gen_update():
        # ---- G5: update block (IGEV and RAFT encoders) ------------------------------------------------
        for tag, mod, model_name in (("igev", rupd, "continuous_IGEVStereo"), ("raft", rupd_raft, "continuous_RAFTStereo")):
            args = default_args(model_name)
            ub = mod.BasicMultiUpdateBlock(args, hidden_dims=args.hidden_dims).eval()
            fill_module_deterministic(ub, base_seed=5)
            h, w = 8, 12
            net = [torch.tanh(det_uniform((1, 128, h >> i, w >> i), 40 + i, -2, 2)) for i in range(3)]
            ctx = [det_uniform((1, 384, h >> i, w >> i), 50 + i) for i in range(3)]
            inp = [list(c.split(128, dim=1)) for c in ctx]
            cor_planes = ub.encoder.convc1.in_channels
            corr = det_uniform((1, cor_planes, h, w), 60, -3, 3)
            disp = det_uniform((1, 1, h, w), 61, 0, 12)
            mf = ub.encoder(disp, corr)
            g16 = ub.gru16(net[2], *inp[2], rupd.pool2x(net[1]))
            net_out, delta = ub([n.clone() for n in net], inp, corr, disp)
            net_lo = ub([n.clone() for n in net], inp, iter16=True, iter08=True, iter04=False, update=False)
            # every flag combination the forward loops use (continuous_IGEVstereo.py:288-293: the slow-fast pre-updates and the
            # main update) plus the remaining (iter16, iter08, iter04, update) patterns a caller can pass
            flags = {}
            for name, kw in (("f16", dict(iter16=True, iter08=False, iter04=False, update=False)),
                             ("f08_04", dict(iter16=False, iter08=True, iter04=True, update=True)),
                             ("f04", dict(iter16=False, iter08=False, iter04=True, update=True)),
                             ("fall_noup", dict(iter16=True, iter08=True, iter04=True, update=False))):
                res = ub([n.clone() for n in net], inp, corr if kw["iter04"] else None, disp if kw["iter04"] else None, **kw)
                nets = res[0] if kw["update"] else res
                for i in range(3):
                    flags[f"{name}_net{i}"] = nets[i]
                if kw["update"]:
                    flags[f"{name}_delta"] = res[1]
            if only == "update":
                save(f"update_flags_{tag}", **flags)
                continue
            save(f"update_flags_{tag}", **flags)
            save(f"update_{tag}", net0=net[0], net1=net[1], net2=net[2], ctx0=ctx[0], ctx1=ctx[1], ctx2=ctx[2], corr=corr,
                 disp=disp, motion=mf, gru16=g16, out0=net_out[0], out1=net_out[1], out2=net_out[2], delta=delta,
                 lo1=net_lo[1], lo2=net_lo[2], pool=rupd.pool2x(net[0]), interp=rupd.interp(net[2], net[1]),
                 head=ub.disp_head(net[0]))


    if only == "update":
        return gen_update()
    if only == "model_opts":
        return golden_model_options(RefIGEV, RefRAFT, rliif)
    if only == "preloop":
        return golden_preloop(RefIGEV, rliif)
    if only in ("train", "train_sens", "train_margin"):
        import ast
        import torch.nn.functional as F
        tree = ast.parse(open(os.path.join(REF, "train_continuous_IGEV.py")).read())
        fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "sequence_loss_multiscale"][0]
        ns2 = {"torch": torch, "F": F}
        exec(compile(ast.Module(body=[fn], type_ignores=[]), "train_continuous_IGEV.py", "exec"), ns2)
        if only == "train_sens":
            return golden_train_sensitivity(RefIGEV, RefRAFT, ns2, **SENS_ARGS)
        if only == "train_margin":
            return golden_train_margin(RefIGEV, RefRAFT, ns2, **MARGIN_ARGS)
        return golden_train(RefIGEV, RefRAFT, ns2)

    # ---- G1/G2/G3: correlation, pyramids, lookup (IGEV: L=2,G=8; RAFT: L=4,G=0) ------------------
    for tag, (b, c, h, w, w2) in {"even": (2, 96, 3, 20, 20), "odd": (1, 96, 2, 21, 21)}.items():
        f1 = det_uniform((b, c, h, w), 11)
        f2 = det_uniform((b, c, h, w2), 12)
        gev = det_uniform((b, 8, 48, h, w), 13)
        fn = rgeo.Combined_Geo_Encoding_Volume(f1, f2, gev, num_levels=2, radius=4)
        # disparities hitting both zero-pad edges, exact integers and x.5 positions
        disp = det_uniform((b, 1, h, w), 14, -6.0, w + 6.0)
        disp[:, :, 0, 0:4] = torch.tensor([0.0, 3.0, 7.5, -0.5])
        disp[:, :, -1, -3:] = torch.tensor([float(w - 1), 47.0, 48.5])
        coords = torch.arange(w).float().reshape(1, 1, w, 1).repeat(b, h, 1, 1)
        out = fn(disp.clone(), coords)
        save(f"lookup_igev_{tag}", f1=f1, f2=f2, gev=gev, disp=disp,
             corr0=fn.init_corr_pyramid[0].reshape(b, h, w, w2), corr1=fn.init_corr_pyramid[1].reshape(b, h, w, w2 // 2),
             geo0=fn.geo_volume_pyramid[0].reshape(b, h, w, 8, 48), geo1=fn.geo_volume_pyramid[1].reshape(b, h, w, 8, 24),
             out=out)
    b, c, h, w = 1, 256, 2, 37
    f1 = det_uniform((b, c, h, w), 21)
    f2 = det_uniform((b, c, h, w), 22)
    fn = rgeo_raft.CorrBlock1D(f1, f2, num_levels=4, radius=4)
    disp = det_uniform((b, 1, h, w), 23, -5.0, w + 5.0)
    coords = torch.arange(w).float().reshape(1, 1, w, 1).repeat(b, h, 1, 1)
    out = fn(disp.clone(), coords)
    save("lookup_raft", f1=f1, f2=f2, disp=disp, out=out,
         **{f"corr{i}": fn.init_corr_pyramid[i].reshape(b, h, w, w >> i) for i in range(4)})

    # ---- G4/A5: gwc volume, disparity regression ------------------------------------------------
    fl = det_uniform((1, 96, 3, 60), 31)
    fr = det_uniform((1, 96, 3, 60), 32)
    vol = rsub.build_gwc_volume(fl, fr, 48, 8)
    cost = det_uniform((2, 48, 3, 5), 33, -4.0, 4.0)
    prob = torch.softmax(cost, dim=1)
    save("gwc_dispreg", fl=fl, fr=fr, vol=vol, cost=cost, init_disp=rsub.disparity_regression(prob, 48))

    gen_update()

    # ---- G6: LIIF pieces --------------------------------------------------------------------------
    feat = det_uniform((2, 20, 5, 7), 71)
    aff = rliif.AffinityFeature(3, 3, 1, 0)(feat)
    liif = {}
    for s in (1.0, 1.5, 2.0, 2.95):
        hh, ww = round(5 * 4 * s), round(7 * 4 * s)
        coord = rliif.make_coord([hh, ww]).unsqueeze(0).repeat(2, 1, 1)
        coord[:, 0] = torch.tensor([-1.0, 1.0])       # exactly on the clamp boundary
        coord[:, 1] = torch.tensor([1.0, -1.0])
        rel, qf, _ = rliif.liif_feat_multiscale_train(feat, coord.clone(), torch.tensor([[s]]))
        key = str(s).replace(".", "p")
        liif[f"coord_{key}"], liif[f"rel_{key}"], liif[f"qfeat_{key}"] = coord, rel, qf
    args = default_args("continuous_IGEVStereo")
    aff_set = {"win_w": 3, "win_h": 3, "dilation": [1, 2, 4, 8]}
    up = rliif.liif_out_multi_scale_Training(encoder_dim=208, mlphidden_list=[128, 64, 64], pos_dim=0, unfold="with_v2ISU",
                                             affinity_settings=aff_set, number_input=2, chanels=[176, 32]).eval()
    fill_module_deterministic(up, base_seed=7, gain=2.0)
    x4 = det_uniform((1, 176, 4, 6), 81)
    x2 = det_uniform((1, 32, 8, 12), 82)
    coord = rliif.make_coord([24, 36]).unsqueeze(0)  # scale 1.5 over the 16x24 full-res grid
    mask = up([x4, x2], coord.clone(), torch.tensor([[1.5]]))
    dlow = det_uniform((1, 1, 4, 6), 83, 0, 20)
    sm = torch.softmax(mask, dim=1)
    cu = rsub.context_upsample_multiscale_train(dlow * 4.0 * 1.5, sm, coord.clone())
    save("liif", feat=feat, aff=aff, x4=x4, x2=x2, coord=coord, mask=mask, dlow=dlow, convex=cu, **liif)

    # ---- G7: whole models, tiny -------------------------------------------------------------------
    sd_keys = {}
    for name, Ref, (H, W) in (("igev", RefIGEV, (64, 128)), ("raft", RefRAFT, (64, 96))):
        args = default_args("continuous_IGEVStereo" if name == "igev" else "continuous_RAFTStereo")
        model = Ref(args).eval()
        fill_module_deterministic(model, base_seed=1)
        img1, img2 = synthetic_pair(1, H, W, shift=6, seed=99)
        outs = {}
        for s in (1.0, 1.5):
            coord = rliif.make_coord([round(H * s), round(W * s)]).unsqueeze(0)
            sc = torch.tensor([[s]])
            up_test = model(img1, img2, iters=3, test_mode=True, hr_coord=coord.clone(), scale=sc)
            key = str(s).replace(".", "p")
            outs[f"test_{key}"] = up_test
        coord = rliif.make_coord([H, W]).unsqueeze(0)
        res = model(img1, img2, iters=3, test_mode=False, hr_coord=coord.clone(), scale=torch.tensor([[1.0]]))
        preds = res[1] if name == "igev" else res
        if name == "igev":
            outs["init_disp"] = res[0]
        for i, p in enumerate(preds):
            outs[f"pred_{i}"] = p
        save(f"model_{name}", H=H, W=W, **outs)
        print(name, "disp_up range", float(up_test.min()), float(up_test.max()))
        sd_keys["continuous_IGEVStereo" if name == "igev" else "continuous_RAFTStereo"] = {
            k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(sd_keys, f, indent=0, sort_keys=True)
    golden_model_options(RefIGEV, RefRAFT, rliif)
    golden_preloop(RefIGEV, rliif)

    # ---- §8(f1): query grid of pad_for_multi_train (evaluation.py:67-89) ---------------------------
    # evaluation.py itself cannot be imported (missing tensorboardX / fvcore / a dangling model import,
    # SURVEY.md §0 item 3), so the single function is compiled from its AST node; InputPadder.get_pad_num
    # is undefined in the reference and is supplied with the only meaning its call site allows (item 4).
    import ast
    import math
    import torch.nn.functional as F
    from models.coreContinuous_IGEV.utils.utils import InputPadder as RefPadder
    RefPadder.get_pad_num = lambda self: [self._pad[2], self._pad[3], self._pad[0], self._pad[1]]
    tree = ast.parse(open(os.path.join(REF, "evaluation.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "pad_for_multi_train"][0]
    ns = {"math": math, "F": F, "torch": torch, "InputPadder": RefPadder, "make_coord": rliif.make_coord}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "evaluation.py", "exec"), ns)
    cases = []
    for (H, W, s, model_name) in [(540, 960, 1.0, "continuous_IGEVStereo"), (375, 1242, 2.0, "continuous_IGEVStereo"),
                                  (100, 150, 1.5, "continuous_IGEVStereo"), (97, 131, 2.95, "continuous_RAFTStereo"),
                                  (64, 96, 1.0, "continuous_RAFTStereo")]:
        a = argparse.Namespace(scale_test=s, model=model_name)
        img = torch.zeros(1, 3, H, W)
        i1, i2, coord = ns["pad_for_multi_train"](a, img, img)
        div = 32 if "IGEVStereo" in model_name else 16
        padder = RefPadder((1, 3, int(math.ceil(H / s)), int(math.ceil(W / s))), divis_by=div)
        cases.append(dict(H=H, W=W, scale=s, divis_by=div, padded=list(i1.shape[-2:]),
                          pad_num=[int(i * s) for i in padder.get_pad_num()], coord_shape=list(coord.shape),
                          first=coord[0].tolist(), last=coord[-1].tolist(), coord_sum=float(coord.double().sum())))
    with open(os.path.join(HERE, "query_grid.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("wrote state_dict_keys.json, query_grid.json")

    # ---- §8(f2)/(f3): loss + metrics --------------------------------------------------------------
    # train_continuous_IGEV.py cannot be imported (argparse + dataset imports at module scope need cv2 etc.),
    # so sequence_loss_multiscale is compiled from its AST node like pad_for_multi_train above.
    tree = ast.parse(open(os.path.join(REF, "train_continuous_IGEV.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "sequence_loss_multiscale"][0]
    ns2 = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "train_continuous_IGEV.py", "exec"), ns2)
    sys.modules.setdefault("metrics_utils", types.ModuleType("metrics_utils")).__path__ = [REF + "/metrics_utils"]
    if "torchvision" not in sys.modules:  # experiment.py:7 imports torchvision.utils for image logging only
        tv = types.ModuleType("torchvision"); tv.utils = types.ModuleType("torchvision.utils")
        sys.modules["torchvision"], sys.modules["torchvision.utils"] = tv, tv.utils
    import metrics_utils.metrics as rmet
    B, Q = 2, 500
    gt = det_uniform((B, 1, Q), 201, 0.0, 900.0)
    valid = (det_uniform((B, 1, Q), 202, 0.0, 1.0) > 0.2).float()
    preds = [gt + det_uniform((B, 1, Q), 210 + i, -6.0, 6.0) * (1.0 - 0.15 * i) for i in range(5)]
    loss, met = ns2["sequence_loss_multiscale"](preds, gt, valid, max_disp=700)
    est, g3 = preds[-1][:, 0].reshape(B, 20, 25), gt[:, 0].reshape(B, 20, 25)
    m3 = (g3 > 0) & (g3 < 192)
    save("loss_metrics", gt=gt, valid=valid, **{f"pred{i}": p for i, p in enumerate(preds)}, loss=loss,
         epe_m=met["epe"], px1=met["1px"], px3=met["3px"],
         EPE=rmet.EPE_metric(est, g3, m3), D1=rmet.D1_metric(est, g3, m3), Thres2=rmet.Thres_metric(est, g3, m3, 2.0))

    # ---- G8: training step (loss + parameter gradients) -------------------------------------------
    golden_train(RefIGEV, RefRAFT, ns2)
    golden_train_sensitivity(RefIGEV, RefRAFT, ns2)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["train", "train_sens", "train_margin", "update", "model_opts", "preloop"], default=None,
                    help="regenerate only the G8 training-step fixtures / their perturbation sensitivities / only the G5 flag-combination fixtures")
    ap.add_argument("--sens-seeds", type=int, default=32)
    ap.add_argument("--sens-rel", type=float, default=1e-6)
    ap.add_argument("--sens-seed0", type=int, default=1000)
    ap.add_argument("--sens-merge", action="store_true", help="train_sens: keep the maximum of this run and the stored file")
    ap.add_argument("--margin-model", choices=["igev", "raft"], default=None)
    ap.add_argument("--margin-candidates", type=int, default=400)
    ap.add_argument("--margin-cand0", type=int, default=0)
    ap.add_argument("--margin-confirm", type=int, default=128)
    a_ = ap.parse_args()
    MARGIN_ARGS.update(only_model=a_.margin_model, candidates=a_.margin_candidates, cand0=a_.margin_cand0, confirm=a_.margin_confirm)
    SENS_ARGS.update(seeds=a_.sens_seeds, rel=a_.sens_rel, seed0=a_.sens_seed0, merge=a_.sens_merge)
    main(a_.only)
