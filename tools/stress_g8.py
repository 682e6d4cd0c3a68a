"""Stress harness for the G8 training-step pin (tests/test_hip_parity.py::test_training_step_vs_reference).

Runs the test's body `--reps` times in ONE process (fresh model each time) and reports, per repetition:
  * loss, error of every stored full gradient against the reference's fixture (tests/golden/train_*.npz) and against the
    fp64 evaluation of the CPU oracle on the same inputs (computed once, `--truth`);
  * an exact (integer) checksum of every module output of the forward and of every parameter gradient, compared with
    repetition 0: the first forward module that differs, the gradients that differ (count, worst relative deviation).
Discriminators:
  --nanfill   every torch.empty / empty_like / new_empty float allocation on the GPU is filled with NaN first: a kernel that reads
              memory nobody wrote shows up as NaN in the loss / a gradient instead of as a box-dependent number
  --perturb R relative N(0, R) noise on the two images (what a different library convolution algorithm does to the forward)
  (a serialised run is the same command under AMD_SERIALIZE_KERNEL=3 / HIP_LAUNCH_BLOCKING=1 in a fresh process)
Writes a JSON record (checksums included, so two boxes can be diffed) to --out.

    python tools/stress_g8.py --name raft --mode split --reps 8 --out gpurun_out/stress/raft_split.json
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "any-stereo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

DEV = os.environ.get("STRESS_DEV", "cuda:0")


def csum(t: torch.Tensor) -> int:
    """Order-independent exact checksum of a tensor's bits."""
    t = t.detach().contiguous()
    if t.dtype == torch.float32:
        v = t.view(torch.int32)
    elif t.dtype in (torch.float16, torch.bfloat16):
        v = t.view(torch.int16)
    elif t.dtype == torch.float64:
        v = t.view(torch.int64)
    else:
        v = t
    return int(v.to(torch.int64).sum().item())


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in flat(x)]
    return []


def install_nanfill():
    real_empty, real_like, real_new = torch.empty, torch.empty_like, torch.Tensor.new_empty

    def poison(t):
        if t.is_cuda and t.is_floating_point() and t.numel():
            t.fill_(float("nan"))
        return t

    torch.empty = lambda *a, **k: poison(real_empty(*a, **k))
    torch.empty_like = lambda *a, **k: poison(real_like(*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: poison(real_new(self, *a, **k))


def oracle_truth(name):
    """fp64 evaluation of the CPU oracle model (oracle/model.py) on the fixture's inputs -> {param: grad (float64)}."""
    from oracle.model import OracleIGEV, OracleRAFT
    from anystereo.harness.metrics import sequence_loss_multiscale
    from anystereo.harness.synthetic import fill_module_deterministic, tiny_train_case
    from anystereo.models import default_args
    dt = torch.float64
    args = default_args("continuous_IGEVStereo" if name == "igev" else "continuous_RAFTStereo")
    model = (OracleIGEV if name == "igev" else OracleRAFT)(args)
    fill_module_deterministic(model, base_seed=1)
    model = model.to(dt).train()
    model.freeze_bn()
    model.hot_dtype = dt
    h, w, i1, i2, coord, gt, scale = tiny_train_case(name)
    res = model(i1.to(dt), i2.to(dt), iters=3, hr_coord=coord.to(dt), scale=scale.to(dt))
    preds = res[1] if name == "igev" else res
    gtd = gt.to(dt)
    loss, _ = sequence_loss_multiscale(preds, gtd, ((gtd < 512) & (gtd > 0)).to(dt), max_disp=args.max_disp)
    loss.backward()
    return float(loss.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def run_once(name, mode, perturb=0.0, seed=0, hooks=True):
    from anystereo import ops
    from anystereo.harness.metrics import sequence_loss_multiscale
    from anystereo.harness.synthetic import fill_module_deterministic, tiny_train_case
    from anystereo.models import __models__, default_args
    args = default_args("continuous_IGEVStereo" if name == "igev" else "continuous_RAFTStereo")
    model = __models__[args.model](args)
    fill_module_deterministic(model, base_seed=1)
    model = model.to(DEV).train()
    model.freeze_bn()
    h, w, img1, img2, coord, gt, scale = tiny_train_case(name)
    if perturb:
        g = torch.Generator().manual_seed(seed)
        img1 = img1 * (1.0 + perturb * torch.randn(img1.shape, generator=g))
        img2 = img2 * (1.0 + perturb * torch.randn(img2.shape, generator=g))
    fwd = []
    handles = []
    if hooks:
        for mn, m in model.named_modules():
            if mn:
                handles.append(m.register_forward_hook(lambda mod, inp, out, mn=mn: fwd.append((mn, [csum(t) for t in flat(out)]))))
    prev = torch.backends.cudnn.deterministic
    prev_mode = ops.get_precision()
    torch.backends.cudnn.deterministic = True
    ls = 4096.0 if mode == "split" else 1.0
    try:
        ops.set_precision(mode)
        res = model(img1.to(DEV), img2.to(DEV), iters=3, hr_coord=coord.to(DEV), scale=scale.to(DEV))
        preds = res[1] if name == "igev" else res
        gtd = gt.to(DEV)
        loss, _ = sequence_loss_multiscale(preds, gtd, ((gtd < 512) & (gtd > 0)).float(), max_disp=args.max_disp)
        (loss * ls).backward()
        torch.cuda.synchronize()
    finally:
        torch.backends.cudnn.deterministic = prev
        ops.set_precision(prev_mode)
        for hd in handles:
            hd.remove()
    grads = {n: (p.grad.detach().double().cpu() / ls) for n, p in model.named_parameters() if p.grad is not None}
    gsum = {n: csum(p.grad) for n, p in model.named_parameters() if p.grad is not None}
    return dict(loss=float(loss.item()), preds=[csum(p) for p in preds], last_pred=preds[-1].detach().cpu(), fwd=fwd, grads=grads, gsum=gsum)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="raft", choices=["raft", "igev"])
    ap.add_argument("--mode", default="split", choices=["split", "fp32"])
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--nanfill", action="store_true")
    ap.add_argument("--perturb", type=float, default=0.0)
    ap.add_argument("--truth", action="store_true", help="also compare with the fp64 CPU oracle (adds ~1 min of CPU time)")
    ap.add_argument("--limits", action="store_true", help="report every value as a fraction of the G8 test's per-tensor limit")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    z = np.load(os.path.join(ROOT, "tests", "golden", f"train_{a.name}.npz"))
    full = [str(n) for n in z["full_names"]]
    gold = {n: torch.from_numpy(z[f"g{i}"]).double() for i, n in enumerate(full)}
    prop = torch.cuda.get_device_properties(0)
    info = dict(host=socket.gethostname(), gpu=prop.name, cus=prop.multi_processor_count, torch=torch.__version__,
                env={k: v for k, v in os.environ.items() if k.startswith(("MIOPEN", "AMD_SERIALIZE", "HIP_LAUNCH", "ANYSTEREO"))},
                name=a.name, mode=a.mode, nanfill=a.nanfill, perturb=a.perturb)
    print("[stress]", json.dumps(info), flush=True)
    truth = None
    if a.truth:
        t0 = time.time()
        try:
            tl, truth = oracle_truth(a.name)
        except RuntimeError as e:  # IGEV: the library 3-D convolutions of the cost-volume stem have no fp64 path here
            print(f"[stress] no fp64 oracle for {a.name}: {str(e)[:100]}", flush=True)
            tl, truth = None, None
    if truth is not None:
        print(f"[stress] fp64 oracle: loss {tl:.9f} ({time.time() - t0:.0f} s); fixture vs fp64: "
              + ", ".join(f"{n.split('.', 1)[-1]} {((gold[n] - truth[n]).abs().max() / truth[n].abs().max()).item():.2e}" for n in full), flush=True)
    if a.nanfill:
        install_nanfill()
    lim_e = lim_n = None
    if a.limits:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.setdefault("PYTEST_DISABLE_PLUGIN_AUTOLOAD", "1")
        import importlib.util
        spec = importlib.util.spec_from_file_location("g8_limits_src", os.path.join(ROOT, "tests", "test_hip_parity.py"))
        src = open(spec.origin).read()
        ns = {"os": os, "__file__": spec.origin}
        start = src.index("G8_FLOOR_ELEM")
        end = src.index("def _g8_run")
        exec(compile(src[start:end], spec.origin, "exec"), ns)  # the limit constants and _g8_limits only
        lim_e, lim_n = ns["_g8_limits"](a.name)
    recs, first = [], None
    for r in range(a.reps):
        o = run_once(a.name, a.mode, perturb=a.perturb, seed=r)
        errs = {n: ((o["grads"][n] - gold[n]).abs().max() / gold[n].abs().max()).item() for n in full}
        terr = {n: ((o["grads"][n] - truth[n]).abs().max() / truth[n].abs().max()).item() for n in full} if truth else {}
        nonfinite = [n for n, g in o["grads"].items() if not torch.isfinite(g).all()]
        names = [str(n) for n in z["names"]]
        norms = np.array([float(o["grads"][n].norm()) for n in names])
        norm_rel = np.abs(norms - z["norms"]) / (z["norms"] + 1e-6 * z["norms"].max())
        rec = dict(rep=r, loss=o["loss"], loss_rel=abs(o["loss"] - float(z["loss"])) / abs(float(z["loss"])),
                   last_pred_abs=(o["last_pred"] - torch.from_numpy(z["last_pred"])).abs().mean().item(), err_vs_fixture=errs,
                   err_vs_fp64=terr, nonfinite=nonfinite, preds=o["preds"], norm_rel={n: float(v) for n, v in zip(names, norm_rel)})
        if first is None:
            first = o
            rec["fwd"] = o["fwd"]
            rec["gsum"] = o["gsum"]
        else:
            diff_mod = next((f"{m0} (call {i})" for i, ((m0, c0), (m1, c1)) in enumerate(zip(first["fwd"], o["fwd"])) if m0 != m1 or c0 != c1), None)
            dg = {}
            for n, c in o["gsum"].items():
                if c != first["gsum"][n]:
                    g0, g1 = first["grads"][n], o["grads"][n]
                    dg[n] = ((g0 - g1).abs().max() / g0.abs().max().clamp_min(1e-300)).item()
            rec["first_differing_forward_module"] = diff_mod
            rec["stored_diff_vs_rep0"] = {n: ((first["grads"][n] - o["grads"][n]).abs().max() / first["grads"][n].abs().max()).item() for n in full}
            rec["grads_differing_from_rep0"] = len(dg)
            rec["worst_grad_diffs"] = sorted(dg.items(), key=lambda kv: -kv[1])[:6]
        if lim_e is not None:
            re_ = {n: errs[n] / lim_e[n] for n in full}
            zero = z["norms"] < 1e-7 * z["norms"].max()
            rn_ = {n: (0.0 if zz else float(v) / lim_n[n]) for n, v, zz in zip(names, norm_rel, zero)}
            rec["worst_elem_ratio"] = max(re_.items(), key=lambda kv: kv[1])
            rec["worst_norm_ratio"] = max(rn_.items(), key=lambda kv: kv[1])
            print(f"[stress] rep {r}: fraction of the G8 limits used: elements {rec['worst_elem_ratio'][1]:.2f} ({rec['worst_elem_ratio'][0]}), "
                  f"norms {rec['worst_norm_ratio'][1]:.2f} ({rec['worst_norm_ratio'][0]})", flush=True)
        recs.append(rec)
        w = max(errs, key=errs.get)
        line = (f"[stress] rep {r}: loss rel {rec['loss_rel']:.1e} pred|d| {rec['last_pred_abs']:.1e} convd1.weight {errs['update_block.encoder.convd1.weight']:.3e}"
                f" worst {w.split('.', 1)[-1]} {errs[w]:.3e}")
        if truth:
            line += f" | vs fp64: convd1.weight {terr['update_block.encoder.convd1.weight']:.3e} worst {max(terr.values()):.3e}"
        if nonfinite:
            line += f" | NON-FINITE gradients: {len(nonfinite)} e.g. {nonfinite[:4]}"
        if r:
            line += f" | vs rep0: fwd differs at {rec['first_differing_forward_module']}, {rec['grads_differing_from_rep0']} grads differ"
            line += f", stored tensors differ by <= {max(rec['stored_diff_vs_rep0'].values()):.1e}"
            if rec["worst_grad_diffs"]:
                line += " (worst " + ", ".join(f"{n.split('.', 1)[-1]} {v:.1e}" for n, v in rec["worst_grad_diffs"][:3]) + ")"
        print(line, flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(dict(info=info, reps=recs), f)
        print("[stress] wrote", a.out)


if __name__ == "__main__":
    main()
