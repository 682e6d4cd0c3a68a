"""`torch.ops.anystereo.*` — the hot-path operators as registered PyTorch operators.

Each operator is a functional schema in the `anystereo` namespace whose ONLY kernel is the HIP one
(dispatch key `CUDA`, which is HIP on ROCm) — there is no CPU kernel, so a CPU tensor fails in the dispatcher
("Could not run 'anystereo::…' with arguments from the 'CPU' backend"), which is the product's no-fallback rule.
Differentiable operators additionally carry an `AutogradCUDA` kernel that routes through the
`torch.autograd.Function`s of `anystereo.grad` (HIP forward + HIP backward).

Stateless operators (reference call site):
    corr_build_pyramid(f1, f2, num_levels) -> Tensor[]                   geometry.py:63-72, :27-29
    geo_pyramid(geo_volume, num_levels) -> Tensor[]                      geometry.py:17-25
    geo_corr_lookup(geo_levels, corr_levels, disp, radius) -> Tensor     geometry.py:34-60, utils.py:59-73
    gwc_volume(fl, fr, maxdisp, groups) -> Tensor                        submodule.py:253-271
    disparity_regression(cost, apply_softmax) -> Tensor                  submodule.py:321-325
    structure_feature(x) -> Tensor                                       liif.py:432-446, :496-499
    convex_upsample(disp, mask, coord, scale?, mask_is_logits) -> Tensor submodule.py:357-372
    corr_sampler_forward / corr_sampler_backward                         sampler/sampler.cpp:24-45
Operators that carry weights take them as tensors, in the reference module's parameter order:
    motion_encoder(disp, corr, weights[5], biases[5]) -> Tensor          update.py:84-92  (convc1, convc2, convd1, convd2, conv)
    convgru_step(h, cz, cr, cq, x[], weights[3], biases[3]) -> Tensor    update.py:33-41  (convz, convr, convq)
    disp_head(x, weights[2], biases[2]) -> Tensor                        update.py:23-24  (conv1, conv2)
    liif_upsample(feats[], coord, weights[], biases[]) -> Tensor         liif.py:644-678  (MLP layers in order) -> [B,9,Q]
They run the same nn modules as the models (one cached module per weight set; the caller's tensors are substituted for the
module parameters by torch.func.functional_call — no copy, gradients flow to the caller's tensors).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import grad as G
from . import ops

_lib = torch.library.Library("anystereo", "DEF")

_lib.define("corr_build_pyramid(Tensor f1, Tensor f2, int num_levels) -> Tensor[]")
_lib.define("geo_pyramid(Tensor geo_volume, int num_levels) -> Tensor[]")
_lib.define("geo_corr_lookup(Tensor[] geo_levels, Tensor[] corr_levels, Tensor disp, int radius) -> Tensor")
_lib.define("gwc_volume(Tensor fl, Tensor fr, int maxdisp, int groups) -> Tensor")
_lib.define("disparity_regression(Tensor cost, bool apply_softmax) -> Tensor")
_lib.define("structure_feature(Tensor x) -> Tensor")
_lib.define("convex_upsample(Tensor disp, Tensor mask, Tensor coord, Tensor? scale, bool mask_is_logits) -> Tensor")
_lib.define("corr_sampler_forward(Tensor volume, Tensor coords, int radius) -> Tensor")
_lib.define("corr_sampler_backward(Tensor volume, Tensor coords, Tensor corr_grad, int radius) -> Tensor")
_lib.define("motion_encoder(Tensor disp, Tensor corr, Tensor[] weights, Tensor[] biases) -> Tensor")
_lib.define("convgru_step(Tensor h, Tensor cz, Tensor cr, Tensor cq, Tensor[] x, Tensor[] weights, Tensor[] biases) -> Tensor")
_lib.define("disp_head(Tensor x, Tensor[] weights, Tensor[] biases) -> Tensor")
_lib.define("liif_upsample(Tensor[] feats, Tensor coord, Tensor[] weights, Tensor[] biases) -> Tensor")

OPS = ("corr_build_pyramid", "geo_pyramid", "geo_corr_lookup", "gwc_volume", "disparity_regression", "structure_feature",
       "convex_upsample", "corr_sampler_forward", "corr_sampler_backward", "motion_encoder", "convgru_step", "disp_head",
       "liif_upsample")


def _f(t: torch.Tensor) -> torch.Tensor:
    return t.float().contiguous()


# ---- stateless operators ----------------------------------------------------------------------------------------
def _corr_build_pyramid(f1, f2, num_levels):
    f1, f2 = _f(f1), _f(f2)
    if G.needs_grad(f1, f2):
        return list(G.CorrBuildPyramid.apply(f1, f2, num_levels))
    return ops.corr_build_pyramid(f1, f2, num_levels)


def _geo_pyramid(gev, num_levels):
    gev = _f(gev)
    if G.needs_grad(gev):
        return list(G.GeoPyramid.apply(gev, num_levels))
    return ops.geo_pyramid(gev, num_levels)


def _geo_corr_lookup(geo_levels, corr_levels, disp, radius):
    disp = _f(disp)
    levels = list(geo_levels) + list(corr_levels)
    if G.needs_grad(*levels):
        return G.Lookup.apply(disp, radius, len(geo_levels), None, *levels)
    return ops.geo_corr_lookup(list(geo_levels), list(corr_levels), disp, radius)


def _gwc_volume(fl, fr, maxdisp, groups):
    fl, fr = _f(fl), _f(fr)
    if G.needs_grad(fl, fr):
        return G.GwcVolume.apply(fl, fr, maxdisp, groups)
    return ops.gwc_volume(fl, fr, maxdisp, groups)


def _disparity_regression(cost, apply_softmax):
    cost = _f(cost)
    if G.needs_grad(cost):
        return G.DisparityRegression.apply(cost, apply_softmax)
    return ops.disparity_regression(cost, apply_softmax)


def _structure_feature(x):
    x = _f(x)
    if G.needs_grad(x):
        return G.StructureFeature.apply(x)
    return ops.structure_feature(x)


def _convex_upsample(disp, mask, coord, scale, mask_is_logits):
    disp, mask, coord = _f(disp), _f(mask), _f(coord)
    scale = None if scale is None else _f(scale).reshape(-1)
    if G.needs_grad(disp, mask):
        return G.ConvexUpsample.apply(disp, mask, coord, scale, mask_is_logits)
    return ops.convex_upsample(disp, mask, coord, scale=scale, mask_is_logits=mask_is_logits)


def _corr_sampler_forward(volume, coords, radius):
    return ops.corr_sampler_forward(volume, coords, radius)


def _corr_sampler_backward(volume, coords, corr_grad, radius):
    return ops.corr_sampler_backward(volume, coords, corr_grad, radius)


# ---- operators with weights: the models' own nn modules, called with the caller's tensors as parameters ----------
_modules: dict = {}


def _cached(kind: str, tensors, build):
    """One module per (operator, weight set).  Weight sets are told apart by their data pointers: the modules' packed-weight
    caches (ops.PackedConv) key on pointer + version, so two alternating sets must not share a module."""
    key = (kind,) + tuple((t.data_ptr(), tuple(t.shape)) for t in tensors)
    m = _modules.get(key)
    if m is None:
        if len(_modules) > 256:
            _modules.clear()
        m = _modules[key] = build()
    return m


def _call(m, names, weights, biases, what, *args):
    """m(*args) with `<name>.weight` / `<name>.bias` replaced by the caller's tensors (torch.func.functional_call: no copy,
    and gradients flow to the caller's tensors)."""
    if len(weights) != len(names) or len(biases) != len(names):
        raise RuntimeError(f"anystereo::{what}: expected {len(names)} weights and biases, got {len(weights)} / {len(biases)}")
    params = {}
    own = dict(m.named_parameters())
    for n, w, b in zip(names, weights, biases):
        for suffix, t in ((".weight", w), (".bias", b)):
            if tuple(own[n + suffix].shape) != tuple(t.shape):
                raise RuntimeError(f"anystereo::{what}: {n}{suffix} must be {tuple(own[n + suffix].shape)}, got {tuple(t.shape)}")
            params[n + suffix] = t
    return torch.func.functional_call(m, params, args)


def _motion_encoder(disp, corr, weights, biases):
    import argparse

    from .nn.update import BasicMotionEncoder

    def build():
        # cor_planes = corr_levels * 9 * (geo_channels + 1): any factorisation builds the same module
        return BasicMotionEncoder(argparse.Namespace(corr_levels=1, corr_radius=4), geo_channels=weights[0].shape[1] // 9 - 1).to(disp.device)
    m = _cached("enc", list(weights) + list(biases), build)
    out = _call(m, ["convc1", "convc2", "convd1", "convd2", "conv"], weights, biases, "motion_encoder", disp, corr)
    return out.float() if isinstance(out, ops.BS8) else out  # inside the models the features stay a blocked link tensor


def _convgru_step(h, cz, cr, cq, x, weights, biases):
    from .nn.update import ConvGRU
    hid = weights[0].shape[0]

    def build():
        return ConvGRU(hid, weights[0].shape[1] - hid, weights[0].shape[2]).to(h.device)
    m = _cached("gru", list(weights) + list(biases), build)
    return _call(m, ["convz", "convr", "convq"], weights, biases, "convgru_step", h, cz, cr, cq, *x)


def _disp_head(x, weights, biases):
    from .nn.update import DispHead

    def build():
        return DispHead(weights[0].shape[1], weights[0].shape[0], weights[1].shape[0]).to(x.device)
    m = _cached("head", list(weights) + list(biases), build)
    return _call(m, ["conv1", "conv2"], weights, biases, "disp_head", x)


def _liif_upsample(feats, coord, weights, biases):
    from .nn.liif import liif_out_multi_scale_Training
    enc = [f.shape[1] for f in feats]

    def build():
        return liif_out_multi_scale_Training(
            pos_dim=0, encoder_dim=sum(enc), mlphidden_list=[w.shape[0] for w in weights[:-1]], unfold="with_v2ISU",
            affinity_settings={"win_w": 3, "win_h": 3, "dilation": [1, 2, 4, 8]}, number_input=len(enc), chanels=enc).to(coord.device)
    m = _cached("liif", list(weights) + list(biases), build)
    names = [f"imnet.layers.{i}" for i, l in enumerate(m.imnet.layers) if isinstance(l, torch.nn.Linear)]
    return _call(m, names, weights, biases, "liif_upsample", list(feats), coord)


_IMPLS = {
    "corr_build_pyramid": _corr_build_pyramid, "geo_pyramid": _geo_pyramid, "geo_corr_lookup": _geo_corr_lookup,
    "gwc_volume": _gwc_volume, "disparity_regression": _disparity_regression, "structure_feature": _structure_feature,
    "convex_upsample": _convex_upsample, "corr_sampler_forward": _corr_sampler_forward,
    "corr_sampler_backward": _corr_sampler_backward, "motion_encoder": _motion_encoder, "convgru_step": _convgru_step,
    "disp_head": _disp_head, "liif_upsample": _liif_upsample,
}
for _name, _fn in _IMPLS.items():
    _lib.impl(_name, _fn, "CUDA")
    # The implementations above pick the autograd.Function form themselves when a gradient is required, so the same
    # callable serves as the autograd kernel (it never re-enters the dispatcher: `ops` / `grad` call the C ABI directly).
    _lib.impl(_name, _fn, "AutogradCUDA")
