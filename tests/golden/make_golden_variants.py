"""G9: the off-by-default options of the implicit upsampler (SURVEY.md §8 f4 — liif.py:181-337, :448-572, :575-678;
submodule.py:375-399), captured from the IMPORTED reference like make_golden.py (same shims, read-only).

    python tests/golden/make_golden_variants.py      # needs /root/reference; writes liif_variants.npz / .json

For every option set the reference module is built, filled from the closed-form generator and run on deterministic
inputs; inputs and the reference OUTPUT are stored.  Option sets the reference itself cannot run (it raises) are
recorded by exception type in liif_variants.json — the build mirrors them as errors, not as features.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, npy  # noqa: E402
from anystereo.harness.synthetic import det_uniform, fill_module_deterministic  # noqa: E402

AFF = {"win_w": 3, "win_h": 3, "dilation": [1, 2, 4, 8]}

# name -> constructor options of liif_out_multi_scale_Training (two inputs [176 @ 4x6, 32 @ 8x12] unless number_input=3)
VARIANTS = {
    "default": dict(unfold="with_v2ISU"),
    "unfold_none": dict(unfold=None),
    "only_unfold": dict(unfold="only_unfold"),
    "with_ISU": dict(unfold="with_ISU"),
    "with_1_4ISU": dict(unfold="with_1_4ISU"),
    "with_embed_ISU": dict(unfold="with_embed_ISU"),
    "only_ISU": dict(unfold="only_ISU"),
    "with_1_43ISU": dict(unfold="with_1_43ISU"),
    "with_1_43v2ISU": dict(unfold="with_1_43v2ISU"),
    "with_3v2ISU": dict(unfold="with_3v2ISU"),
    "with_Dila_ISU": dict(unfold="with_Dila_ISU"),
    "only_Dila_ISU": dict(unfold="only_Dila_ISU"),
    "with_Dila_3ISU": dict(unfold="with_Dila_3ISU"),
    "only_Dila_3ISU": dict(unfold="only_Dila_3ISU"),
    "with_Dila_2ISU": dict(unfold="with_Dila_2ISU"),
    "only_Dila_2ISU": dict(unfold="only_Dila_2ISU"),
    "pos_enc": dict(unfold="with_v2ISU", pos_dim=24, pos_enconding=True, require_grad=False),
    "pos_enc_learned": dict(unfold="with_v2ISU", pos_dim=24, pos_enconding=True, require_grad=True),
    "pos_enc_new": dict(unfold="with_v2ISU", pos_dim=24, pos_enconding_new=True),
    "pos_dim_unused": dict(unfold="with_v2ISU", pos_dim=24),
    "decode_cell": dict(unfold="with_v2ISU", decode_cell=True),
    "quater_both": dict(unfold="with_v2ISU", quater_nearest="both"),
    "quater_single": dict(unfold="with_v2ISU", quater_nearest="single"),
    "local_ensemble": dict(unfold="with_v2ISU", local_ensemble=True),
    "three_inputs": dict(unfold="with_v2ISU", number_input=3),
    "pos_enc_cell_quater": dict(unfold="with_ISU", pos_dim=8, pos_enconding=True, require_grad=False, decode_cell=True,
                                quater_nearest="both"),
}


def inputs(n_in, batch=1, k=1):
    x4 = det_uniform((batch, 176, 4 * k, 6 * k), 81)
    x2 = det_uniform((batch, 32, 8 * k, 12 * k), 82)
    x1 = det_uniform((batch, 8, 16 * k, 24 * k), 84)
    return ([x1, x2, x4], [8, 32, 176]) if n_in == 3 else ([x4, x2], [176, 32])


def _err(e):
    msg = (str(e).splitlines() or [""])[0][:160]
    if isinstance(e, AssertionError) and not msg:
        msg = "assert False"
    return {"ok": False, "error": type(e).__name__, "message": msg}


# whole models with non-default option sets (tiny, 2 iterations, test mode, scale 1.5): name -> (model, size, options)
MODEL_VARIANTS = {
    "igev_type2": ("continuous_IGEVStereo", (64, 128), dict(agg_type="type2")),
    "igev_quater_posenc_cell": ("continuous_IGEVStereo", (64, 128), dict(quater_nearest="both", pos_enconding=True, pos_dim=8,
                                                                        decode_cell=True, unfold_similarity="with_ISU")),
    "igev_type1_norm": ("continuous_IGEVStereo", (64, 128), dict(agg_type="type1", disparity_norm=True)),
    "igev_type3_norm2": ("continuous_IGEVStereo", (64, 128), dict(agg_type="type3", disparity_norm2=True,
                                                                 unfold_similarity="with_embed_ISU")),
    "raft_type2_onlyISU_norm": ("continuous_RAFTStereo", (64, 96), dict(agg_type="type2", unfold_similarity="only_ISU",
                                                                       disparity_norm=True)),
    "raft_type1_unfold_quater": ("continuous_RAFTStereo", (64, 96), dict(agg_type="type1", unfold_similarity="only_unfold",
                                                                        quater_nearest="single")),
}


def model_variants(status):
    from anystereo.harness.synthetic import synthetic_pair
    from anystereo.models.base import default_args
    import models.coreContinuous_IGEV.liif as rliif
    from models.coreContinuous_IGEV.continuous_IGEVstereo import continuous_IGEVStereo as RefIGEV
    from models.corePrune_RAFT.prune_raft_stereo import continuous_RaftStereo as RefRAFT
    outs, keys = {}, {}
    for name, (mname, (H, W), opt) in MODEL_VARIANTS.items():
        args = default_args(mname, **opt)
        try:
            model = (RefIGEV if "IGEV" in mname else RefRAFT)(args).eval()
            fill_module_deterministic(model, base_seed=1)
            img1, img2 = synthetic_pair(1, H, W, shift=6, seed=99)
            coord = rliif.make_coord([round(H * 1.5), round(W * 1.5)]).unsqueeze(0)
            out = model(img1, img2, iters=2, test_mode=True, hr_coord=coord.clone(), scale=torch.tensor([[1.5]]))
            outs[name] = npy(out)
            keys[name] = {k: list(v.shape) for k, v in model.state_dict().items() if k.startswith(("liif_up.", "stem_"))}
            status["model:" + name] = {"ok": True}
            print("model", name, "ok", tuple(out.shape), float(out.min()), float(out.max()))
        except Exception as e:
            status["model:" + name] = _err(e)
            print("model", name, status["model:" + name])
    np.savez_compressed(os.path.join(HERE, "model_variants.npz"), **outs)
    with open(os.path.join(HERE, "model_variants_keys.json"), "w") as f:
        json.dump({"options": {k: [v[0], list(v[1]), v[2]] for k, v in MODEL_VARIANTS.items()}, "state_dict": keys}, f, indent=0,
                  sort_keys=True)
    print("wrote model_variants.npz (%.1f KiB)" % (os.path.getsize(os.path.join(HERE, "model_variants.npz")) / 1024))


def main():
    torch.set_grad_enabled(False)
    import_reference()
    import models.coreContinuous_IGEV.liif as rliif
    import models.coreContinuous_IGEV.submodule as rsub
    status, arrs = {}, {}
    coord = rliif.make_coord([24, 36]).unsqueeze(0)  # scale 1.5 over the 16x24 full-resolution grid
    coord[:, 0] = torch.tensor([-1.0, 1.0])          # on the clamp boundary
    coord[:, 5] = torch.tensor([0.999, -0.999])
    scale = torch.tensor([[1.5]])
    dlow = det_uniform((1, 1, 4, 6), 83, 0, 20)
    arrs["coord"], arrs["dlow"] = npy(coord), npy(dlow)
    for name, opt in VARIANTS.items():
        n_in = opt.get("number_input", 2)
        feats, chans = inputs(n_in)
        kw = dict(encoder_dim=sum(chans), mlphidden_list=[128, 64, 64], pos_dim=0, affinity_settings=AFF, number_input=n_in,
                  chanels=chans)
        kw.update(opt)
        try:
            up = rliif.liif_out_multi_scale_Training(**kw).eval()
            fill_module_deterministic(up, base_seed=7, gain=2.0)
            mask = up([f.clone() for f in feats], coord.clone(), scale)
            sm = torch.softmax(mask, dim=1)
            if opt.get("quater_nearest") is None:
                conv = rsub.context_upsample_multiscale_train(dlow * 4.0 * 1.5, sm, coord.clone())
            else:
                conv = rsub.context_upsample_multiscale_train_quaterp(dlow * 4.0 * 1.5, sm, coord.clone())
            status[name] = {"ok": True, "in_dim": int(up.imnet.layers[0].weight.shape[1]), "out_dim": int(mask.shape[1]),
                            "state_dict": {k: list(v.shape) for k, v in up.state_dict().items()}}
            arrs[f"{name}__mask"], arrs[f"{name}__convex"] = npy(mask), npy(conv)
        except Exception as e:  # the reference cannot run this option set
            status[name] = _err(e)
            if "sliding blocks" in status[name]["message"]:  # dilation larger than the tiny map: ask again at 32x48 / 64x96
                try:
                    up([f.clone() for f in inputs(n_in, k=8)[0]], coord.clone(), scale)
                    status[name] = {"ok": True, "note": "runs on larger maps only; no fixture"}
                except Exception as e2:
                    status[name] = _err(e2)
        print(name, status[name] if not status[name]["ok"] else ("ok", status[name]["in_dim"], status[name]["out_dim"]))
    # batch > 1 with decode_cell: the cell assignment broadcasts scale [B,1] against [B,Q] (liif.py:112-114)
    for name, batch in (("decode_cell_b2", 2),):
        feats, chans = inputs(2, batch)
        try:
            up = rliif.liif_out_multi_scale_Training(encoder_dim=208, mlphidden_list=[128, 64, 64], pos_dim=0, unfold="with_v2ISU",
                                                     decode_cell=True, affinity_settings=AFF, number_input=2, chanels=chans).eval()
            fill_module_deterministic(up, base_seed=7, gain=2.0)
            sc2 = torch.tensor([[1.5], [2.0]])
            mask = up(feats, coord.repeat(2, 1, 1), sc2)
            status[name] = {"ok": True}
            arrs[f"{name}__mask"] = npy(mask)
        except Exception as e:
            status[name] = _err(e)
        print(name, status[name])
    model_variants(status)
    np.savez_compressed(os.path.join(HERE, "liif_variants.npz"), **arrs)
    with open(os.path.join(HERE, "liif_variants.json"), "w") as f:
        json.dump({"options": {k: {a: (b if not isinstance(b, bool) else b) for a, b in v.items()} for k, v in VARIANTS.items()},
                   "reference": status}, f, indent=1, sort_keys=True)
    print("wrote liif_variants.npz (%.1f KiB), liif_variants.json" % (os.path.getsize(os.path.join(HERE, "liif_variants.npz")) / 1024))


if __name__ == "__main__":
    main()
