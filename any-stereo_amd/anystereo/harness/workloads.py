"""The BASELINE.json configurations as synthetic workloads: one place that bench.py and the
full-size parity tests share, so "cfg 3" means the same tensors everywhere.

    cfg1  corePrune_RAFT   256x512,  8 GRU iterations, scale 1.0                      (the reference's CPU-runnable case)
    cfg2  coreContinuous_IGEV 960x540 -> pad 544x960, 32 iterations, scale 1.0, Q = 518 400   (the headline metric)
    cfg3  coreContinuous_IGEV KITTI LR 375x1242 -> pad 384x1248, 32 iterations, fixed x2.0 protocol
          (evaluation_validate.py:92-106), Q = 750 x 2484 = 1 863 000
    cfg5  coreContinuous_IGEV Middlebury-F output 1988x2880 at x1.5 -> LR 1326x1920 -> pad 1344x1920, 48 iterations
          (evaluation.py:67-89), Q = 5 725 440
cfg 4 (training) lives in harness/train.py.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from .query import pad_for_multi_train, pad_for_multi_train_fixed
from .synthetic import synthetic_pair


@dataclass(frozen=True)
class Workload:
    name: str
    model: str
    height: int          # size handed to the padding protocol (cfg 3: the LOW-resolution pair; others: the wanted output)
    width: int
    scale: float
    iters: int
    protocol: str        # "downscale" = pad_for_multi_train, "fixed" = pad_for_multi_train_fixed
    divis_by: int
    what: str


WORKLOADS = {
    "cfg1": Workload("cfg1", "continuous_RAFTStereo", 256, 512, 1.0, 8, "downscale", 16,
                     "corePrune_RAFT, 256x512 synthetic pair"),
    "cfg2": Workload("cfg2", "continuous_IGEVStereo", 540, 960, 1.0, 32, "downscale", 32,
                     "coreContinuous_IGEV inference, 960x540 SceneFlow-shape synthetic pair"),
    "cfg3": Workload("cfg3", "continuous_IGEVStereo", 375, 1242, 2.0, 32, "fixed", 32,
                     "coreContinuous_IGEV inference, KITTI-shape 1242x375 low-resolution pair, fixed x2.0 implicit upsampling"),
    "cfg5": Workload("cfg5", "continuous_IGEVStereo", 1988, 2880, 1.5, 48, "downscale", 32,
                     "coreContinuous_IGEV inference, Middlebury-F 2880x1988 output at x1.5 (low-resolution pair 1920x1326)"),
}


def custom(height: int, width: int, scale: float, iters: int, model: str = "continuous_IGEVStereo") -> Workload:
    return Workload("custom", model, height, width, scale, iters, "downscale", 32 if "IGEV" in model else 16,
                    f"{model} inference, {width}x{height} synthetic pair")


def build_inputs(wl: Workload, seed: int = 1234, pairs: int = 1, device=None):
    """-> (image1_pad, image2_pad, hr_coord [B,Q,2], scale [B,1]) on `device` (CPU when None).  Images are the padded
    network inputs (0..255 float), hr_coord the (row, col) query grid of the wanted output inside the padded frame."""
    img1, img2 = synthetic_pair(1, wl.height, wl.width, shift=8, seed=seed)
    if wl.protocol == "fixed":
        i1, i2, coord, _ = pad_for_multi_train_fixed(img1, img2, int(wl.scale), divis_by=wl.divis_by)
    else:
        i1, i2, coord, _ = pad_for_multi_train(img1, img2, wl.scale, divis_by=wl.divis_by)
    coord = coord.unsqueeze(0)
    scale = torch.tensor([[float(wl.scale)]])
    if device is not None:
        i1, i2, coord, scale = (t.to(device) for t in (i1, i2, coord, scale))
    if pairs > 1:
        i1, i2 = i1.repeat(pairs, 1, 1, 1), i2.repeat(pairs, 1, 1, 1)
        coord, scale = coord.repeat(pairs, 1, 1), scale.repeat(pairs, 1)
    return i1, i2, coord.contiguous(), scale


def build_model(wl: Workload, base_seed: int = 1, device=None):
    """Random-init weights of the named architecture, deterministic (no checkpoints offline), eval mode."""
    from ..models import __models__, default_args
    from .synthetic import fill_module_deterministic
    args = default_args(wl.model)
    model = __models__[wl.model](args).eval()
    fill_module_deterministic(model, base_seed=base_seed)
    return (model if device is None else model.to(device)), args
