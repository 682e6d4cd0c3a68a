"""Deterministic, torch-version-independent synthetic data and weights.

Golden vectors must reproduce on any torch build, so nothing here touches a torch RNG:
values come from an integer hash of the element index (numpy uint64 arithmetic).
Used by tests/golden/make_golden.py (to fill the imported reference), by the parity
tests (to fill this package's modules identically) and by bench.py.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def det_uniform(shape, seed: int, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """Uniform-looking fp32 values in [lo, hi) from a splitmix-style hash of (seed, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    x = np.arange(n, dtype=np.uint64) + np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1), 24 bits
    out = (lo + (hi - lo) * u).astype(np.float32).reshape(shape)
    return torch.from_numpy(out)


def name_seed(name: str) -> int:
    return zlib.crc32(name.encode("utf-8")) & 0x7FFFFFFF


@torch.no_grad()
def fill_module_deterministic(module: torch.nn.Module, base_seed: int = 0, gain: float = 1.0) -> None:
    """Fill every parameter/buffer of `module` from det_uniform keyed by its state_dict name.

    * weights with >=2 dims: U(-1,1) * gain * sqrt(3 / fan_in)   (variance ~ gain^2 / fan_in)
    * 1-D `weight` (norm scales): 1 + 0.1 U ;  `bias`: 0.1 U
    * running_mean: 0.1 U ; running_var: 1 + 0.2 |U| ; num_batches_tracked untouched
    The same call on the imported reference model and on this package's model gives
    identical weights because the state_dict names are identical.
    """
    sd = module.state_dict()
    for name, t in sd.items():
        if not torch.is_floating_point(t):
            continue
        s = name_seed(name) + base_seed
        u = det_uniform(tuple(t.shape), s)
        if name.endswith("running_mean"):
            v = 0.1 * u
        elif name.endswith("running_var"):
            v = 1.0 + 0.2 * u.abs()
        elif t.dim() >= 2:
            fan_in = int(np.prod(t.shape[1:]))
            v = u * (gain * (3.0 / fan_in) ** 0.5)
        elif name.endswith("weight"):
            v = 1.0 + 0.1 * u
        else:
            v = 0.1 * u
        t.copy_(v.to(t.dtype))


def synthetic_pair(batch: int, height: int, width: int, shift: int = 8, seed: int = 1234):
    """A structured synthetic stereo pair in [0,255): image2 is image1 rolled left by `shift`
    plus small noise, so the correlation volume has a real ridge.
    Smooth low-frequency content + hash noise keeps the CNN features non-degenerate."""
    yy = torch.arange(height, dtype=torch.float32).view(1, 1, height, 1)
    xx = torch.arange(width, dtype=torch.float32).view(1, 1, 1, width)
    ch = torch.arange(3, dtype=torch.float32).view(1, 3, 1, 1)
    bb = torch.arange(batch, dtype=torch.float32).view(batch, 1, 1, 1)
    base = 127.5 + 60.0 * torch.sin(0.071 * xx + 0.9 * ch + 0.37 * bb) * torch.cos(0.053 * yy - 0.4 * ch) \
        + 40.0 * torch.sin(0.23 * xx + 0.19 * yy + 1.7 * ch)
    tex = det_uniform((batch, 3, height, width), seed, -25.0, 25.0)
    img1 = (base + tex).clamp(0.0, 254.999)
    noise = det_uniform((batch, 3, height, width), seed + 1, -2.0, 2.0)
    img2 = (torch.roll(img1, shifts=-shift, dims=3) + noise).clamp(0.0, 254.999)
    return img1.contiguous(), img2.contiguous()


def tiny_train_case(name: str):
    """Inputs of the G8 training-step fixture (tests/golden/train_{igev,raft}.npz): a batch of 2 tiny pairs, 300 random
    queries per sample from the s = 1.5 grid, ground truth U(0.5, 40) — the shape of a cfg-4 step
    (train_continuous_IGEV.py:214-239, stereo_datasets.py:190-193) at a size the CPU oracle runs in seconds."""
    from ..nn.liif import make_coord
    h, w = (64, 128) if name == "igev" else (64, 96)
    img1, img2 = synthetic_pair(2, h, w, shift=6, seed=77)
    s, nq = 1.5, 300
    grid = make_coord([round(h * s), round(w * s)])
    idx = [(det_uniform((nq,), 300 + b, 0.0, 1.0) * grid.shape[0]).long().clamp(max=grid.shape[0] - 1) for b in range(2)]
    coord = torch.stack([grid[i] for i in idx]).contiguous()
    gt = det_uniform((2, 1, nq), 310, 0.5, 40.0)
    return h, w, img1, img2, coord, gt, torch.tensor([[s], [s]])
