"""cfg 4: the data-parallel training step — one process per GPU, DDP over RCCL.

Mirrors the `multi_training` branch of train_continuous_IGEV.py:201-239 (model.train() + freeze_bn, AdamW +
linear OneCycleLR, sequence_loss_multiscale, clip 1.0); the reference shards the batch with nn.DataParallel
(:184), here every rank owns batch/world samples and the only collective is DDP's bucketed fp32 gradient
all-reduce (45.6 MB for IGEV), overlapped with backward.  Synthetic batches have the shapes of
stereo_datasets.py:71,116-123,190-193: 160x320 network input, 51 200 random HR queries per sample, one scale
per sample in [1, 2.95].
"""
from __future__ import annotations

import os

import torch
import torch.distributed as td

from ..nn.liif import make_coord
from .metrics import fetch_optimizer, train_step
from .synthetic import det_uniform, synthetic_pair


def synthetic_train_batch(batch: int, height: int = 160, width: int = 320, n_query: int | None = None, seed: int = 0,
                          scale_min: float = 1.0, scale_max: float = 2.95, device="cpu"):
    """(image1, image2, hr_coord [B,Q,2], hr_disp [B,1,Q], scale [B,1]) — Q = height*width by default
    (`sample_q = inp_size[0]*inp_size[1]`, stereo_datasets.py:71), queries drawn without replacement from the
    cell-centre grid of the round(size*scale) HR image (stereo_datasets.py:190-193)."""
    q = n_query or height * width
    img1, img2 = synthetic_pair(batch, height, width, shift=8, seed=1000 + seed)
    scale = det_uniform((batch, 1), 2000 + seed, scale_min, scale_max)
    gen = torch.Generator().manual_seed(3000 + seed)
    rows = []
    for b in range(batch):
        s = float(scale[b, 0])
        grid = make_coord([round(height * s), round(width * s)])
        n = grid.shape[0]
        idx = torch.randperm(n, generator=gen)[:q] if n >= q else torch.randint(n, (q,), generator=gen)
        rows.append(grid[idx])
    hr_coord = torch.stack(rows).contiguous()
    hr_disp = det_uniform((batch, 1, q), 4000 + seed, 0.5, 64.0)
    return tuple(t.to(device) for t in (img1, img2, hr_coord, hr_disp, scale))


def shard_batch(batch, rank: int, world: int):
    """Rank's contiguous slice of a global batch (the role of DataParallel's scatter, train_continuous_IGEV.py:184)."""
    n = batch[0].shape[0]
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return tuple(t[rank * per:(rank + 1) * per].contiguous() for t in batch)


class Trainer:
    """model.train() + freeze_bn + (DDP when a process group exists) + AdamW/OneCycleLR; `step(batch)` runs
    harness.metrics.train_step and returns (loss, metrics)."""

    def __init__(self, model, lr: float = 2e-4, wdecay: float = 1e-5, num_steps: int = 100000, train_iters: int = 16,
                 max_disp: int = 192, lr_fixed: bool = False, mixed_precision: bool = False, bucket_cap_mb: int = 25,
                 force_ddp: bool = False):
        model.train()
        model.freeze_bn()  # train_continuous_IGEV.py:203
        self.model = model
        self.module = model
        if td.is_available() and td.is_initialized() and (td.get_world_size() > 1 or force_ddp):
            p = next(model.parameters())
            ids = [p.device.index] if p.is_cuda else None
            # the IGEV classifier (and everything else that only feeds init_disp) gets no gradient from
            # sequence_loss_multiscale: DDP has to be told that some parameters stay unused (measured on one MI355X rank through
            # RCCL: 139 vs 129 ms per step without the wrapper).  ANYSTEREO_DDP_STATIC=1 tries `static_graph` instead of the
            # per-step graph walk; it works with gloo on CPU but raised on the GPU box in round 1 — left opt-in.
            static = os.environ.get("ANYSTEREO_DDP_STATIC", "0") != "0"
            self.module = torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, find_unused_parameters=not static,
                                                                    static_graph=static, bucket_cap_mb=bucket_cap_mb)
        self.optimizer, self.scheduler = fetch_optimizer(lr, wdecay, num_steps, model.parameters(), lr_fixed)
        self.scaler = torch.amp.GradScaler("cuda", enabled=True) if mixed_precision else None
        self.train_iters, self.max_disp = train_iters, max_disp

    def step(self, batch):
        # DDP wraps in train mode; BatchNorm2d must stay frozen even after a .train() from outside
        if not self.module.training:
            self.module.train()
            self.model.freeze_bn()
        return train_step(self.module, self.optimizer, self.scheduler, self.scaler, batch, self.train_iters,
                          max_disp=self.max_disp)
