"""Hot-path free functions of models/coreContinuous_IGEV/submodule.py, HIP-backed, same signatures:
groupwise_correlation/build_gwc_volume :253-271, disparity_regression :321-325,
context_upsample_multiscale_train :357-372."""
from __future__ import annotations

import torch

from .. import grad as G
from .. import ops
from ..harness.timing import scope


def build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """[B,C,H,W] x2 -> [B,num_groups,maxdisp,H,W] group-wise correlation volume."""
    fl, fr = refimg_fea.float().contiguous(), targetimg_fea.float().contiguous()
    with scope("gwc_volume"):
        if G.needs_grad(fl, fr):
            return G.GwcVolume.apply(fl, fr, maxdisp, num_groups)
        return ops.gwc_volume(fl, fr, maxdisp, num_groups)


def disparity_regression(x, maxdisp):
    """sum_d d * x[:, d] for a probability volume x [B,D,H,W] -> [B,1,H,W]."""
    assert x.dim() == 4 and x.shape[1] == maxdisp
    x = x.float().contiguous()
    if G.needs_grad(x):
        return G.DisparityRegression.apply(x, False)
    return ops.disparity_regression(x, apply_softmax=False)


def softmax_disparity_regression(cost):
    """Fused F.softmax(cost, 1) + disparity_regression (continuous_IGEVstereo.py:267-268)."""
    cost = cost.float().contiguous()
    with scope("disparity_regression"):
        if G.needs_grad(cost):
            return G.DisparityRegression.apply(cost, True)
        return ops.disparity_regression(cost, apply_softmax=True)


def context_upsample_multiscale_train(disp_low, up_weights, hr_coord):
    """Convex 3x3 upsampling at arbitrary query coordinates -> [B,Q].
    disp_low [B,1,h,w] (already scaled), up_weights [B,9,Q] (already softmaxed), hr_coord [B,Q,2].
    Like the reference (submodule.py:366) this clamps `hr_coord` IN PLACE."""
    hr_coord.clamp_(-1 + 1e-6, 1 - 1e-6)
    d, m, c = disp_low.float().contiguous(), up_weights.float().contiguous(), hr_coord.float().contiguous()
    if G.needs_grad(d, m):
        return G.ConvexUpsample.apply(d, m, c, None, False)[:, 0]
    return ops.convex_upsample(d, m, c, scale=None, mask_is_logits=False)[:, 0]


def context_upsample_multiscale_train_quaterp(disp_low, up_weights, hr_coord):
    """Four-sample convex upsampling -> [B,Q] (submodule.py:375-399): disp_low [B,1,h,w] (already scaled), up_weights [B,4,Q]
    (already softmaxed), hr_coord [B,Q,2] — NOT clamped in place by this variant."""
    d, m, c = disp_low.float().contiguous(), up_weights.float().contiguous(), hr_coord.float().contiguous()
    if G.needs_grad(d, m):
        return G.ConvexUpsampleQuater.apply(d, m, c, None, False)[:, 0]
    return ops.convex_upsample_quater(d, m, c, scale=None, mask_is_logits=False)[:, 0]
