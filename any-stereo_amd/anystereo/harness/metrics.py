"""§8(f2)/(f3): the training loss and the evaluation metrics that define "parity" for Any-Stereo.

sequence_loss_multiscale  train_continuous_IGEV.py:68-94   (exponentially weighted masked L1 over the GRU predictions)
fetch_optimizer           train_continuous_IGEV.py:125-134 (AdamW + linear OneCycleLR)
EPE / D1 / Thres          metrics_utils/metrics.py:66-90 with the per-image wrapper :22-42
Host-side tensor code (any device); pinned by tests/golden/loss_metrics.npz captured from the reference.
"""
from __future__ import annotations

import os

import torch


def sequence_loss_multiscale(disp_preds, disp_gt, valid, loss_gamma=0.9, max_disp=700, sync_free=False):
    """disp_preds: list of [B,1,Q]; disp_gt, valid [B,1,Q].  Returns (loss, {'epe','1px','3px'}).

    sync_free=False is the reference's statement line by line (train_continuous_IGEV.py:68-94): boolean-mask indexing and
    `.item()` — 16 + 3 host synchronisations per call, ~100 small launches.  sync_free=True evaluates the same quantities as
    masked sums over the stacked predictions (mean over the valid set = sum(err * valid) / count; same value up to the fp32
    summation order, pinned by the same golden numbers) in ~10 launches with NO synchronisation: the host keeps issuing the
    backward pass while the GPU still runs the forward.  The metrics then are 0-d tensors (convert when logging)."""
    n = len(disp_preds)
    assert n >= 1
    valid = (valid >= 0.5) & (disp_gt < max_disp)
    assert valid.shape == disp_gt.shape, [valid.shape, disp_gt.shape]
    gamma = loss_gamma ** (15 / (n - 1)) if n > 1 else loss_gamma
    if sync_free:
        # Masking is a SELECT, never a multiplication: Middlebury ground truth holds inf at invalid pixels
        # (frame_utils.readDispMiddlebury returns the PFM raw) and inf * 0 = NaN would poison the loss, the gradients and the
        # AdamW state.  The ground truth is sanitised first so that no inf / NaN enters the graph at all (the backward of
        # abs() multiplies the incoming zero by sign(x)); the reference excludes those pixels by boolean indexing (:84-86).
        zero = torch.zeros((), dtype=disp_gt.dtype, device=disp_gt.device)
        gt = torch.where(valid, disp_gt, zero)
        cnt = valid.sum().to(disp_gt.dtype)
        preds = torch.stack(list(disp_preds))                       # [n,B,1,Q]
        assert preds.shape[1:] == valid.shape
        per_pred = torch.where(valid, (preds - gt).abs(), zero).flatten(1).sum(1)  # [n] masked L1 sums
        # gamma^(n-1-i), built on the device (a torch.tensor(list, device=...) is a blocking host-to-device copy)
        w = torch.pow(torch.full((), gamma, dtype=torch.float64, device=per_pred.device),
                      torch.arange(n - 1, -1, -1, dtype=torch.float64, device=per_pred.device)).to(per_pred.dtype)
        loss = (w * per_pred).sum() / cnt
        with torch.no_grad():
            epe = torch.sum((disp_preds[-1] - gt) ** 2, dim=1).sqrt().view(-1)
            v = valid.view(-1)
            metrics = {"epe": torch.where(v, epe, zero).sum() / cnt,
                       "1px": (v & (epe > 1)).sum().to(cnt.dtype) / cnt,
                       "3px": (v & (epe > 3)).sum().to(cnt.dtype) / cnt}
        return loss, metrics
    loss = 0.0
    for i, pred in enumerate(disp_preds):
        w = gamma ** (n - i - 1)
        err = (pred - disp_gt).abs()
        assert err.shape == valid.shape
        loss = loss + w * err[valid.bool()].mean()
    epe = torch.sum((disp_preds[-1] - disp_gt) ** 2, dim=1).sqrt().view(-1)[valid.view(-1)]
    metrics = {"epe": epe.mean().item(), "1px": (epe > 1).float().mean().item(), "3px": (epe > 3).float().mean().item()}
    return loss, metrics


def fetch_optimizer(lr, wdecay, num_steps, params, lr_fixed=False, capturable=False):
    """AdamW + linear OneCycleLR (train_continuous_IGEV.py:125-134).  capturable=True: the optimizer's step counter and its
    learning rate live on the device (the scheduler writes the tensor), so `optimizer.step()` can be captured into a hipGraph."""
    params = list(params)
    if capturable:
        dev = params[0].device
        opt = torch.optim.AdamW(params, lr=torch.tensor(float(lr), device=dev), weight_decay=wdecay, eps=1e-8, capturable=True, foreach=True)
    else:
        # CUDA parameters: torch's fused multi-tensor AdamW (the same update in ~3 launches instead of ~60 foreach launches:
        # -0.5 ms per step on the GPU-bound graphed step; ANYSTEREO_FUSED_ADAMW=0 = the foreach form)
        fused = os.environ.get("ANYSTEREO_FUSED_ADAMW", "1") == "1" and params[0].is_cuda
        opt = torch.optim.AdamW(params, lr=lr, weight_decay=wdecay, eps=1e-8, **({"fused": True} if fused else {}))
    sched = None if lr_fixed else torch.optim.lr_scheduler.OneCycleLR(
        opt, lr, num_steps + 100, pct_start=0.01, cycle_momentum=False, anneal_strategy="linear")
    return opt, sched


def _per_image(fn, d_est, d_gt, mask, *args):
    assert d_est.dim() == 3 and d_est.shape == d_gt.shape == mask.shape
    return torch.stack([fn(d_est[i], d_gt[i], mask[i], *args) for i in range(d_gt.shape[0])]).mean()


@torch.no_grad()
def epe_metric(d_est, d_gt, mask):
    """mean over images of the masked mean |D_est - D_gt|  (metrics.py:84-90)."""
    return _per_image(lambda e, g, m: (e[m] - g[m]).abs().mean(), d_est, d_gt, mask)


@torch.no_grad()
def d1_metric(d_est, d_gt, mask):
    """fraction with error > 3 px AND > 5 % of |gt|  (metrics.py:66-72)."""
    def f(e, g, m):
        e, g = e[m], g[m]
        err = (g - e).abs()
        return ((err > 3) & (err / g.abs() > 0.05)).float().mean()
    return _per_image(f, d_est, d_gt, mask)


@torch.no_grad()
def thres_metric(d_est, d_gt, mask, thres):
    """fraction with error > thres  (metrics.py:74-81)."""
    assert isinstance(thres, (int, float))
    return _per_image(lambda e, g, m: ((g[m] - e[m]).abs() > thres).float().mean(), d_est, d_gt, mask)


def train_step(model, optimizer, scheduler, scaler, batch, train_iters, max_disp=192, clip=1.0, loss_scale=1.0, sync_free_loss=False,
               should_step=None, phase="all"):
    """One optimisation step with the reference's ordering (train_continuous_IGEV.py:214-239, multi_training branch):
    zero_grad -> forward(train mode) -> sequence_loss_multiscale with valid = (gt < 512) & (gt > 0) -> scaled backward ->
    unscale -> clip_grad_norm_(1.0) -> optimizer step -> scheduler step (unless fixed lr) -> scaler update.
    `batch` = (image1, image2, hr_coord, hr_disp_gt, scale); `scaler` may be None (no mixed precision).
    `loss_scale` (a power of two, no GradScaler): the backward pass runs on loss * loss_scale and the gradients are divided by it
    before clipping — exact in fp32, it only moves the 1e-6 .. 1e-9 activation gradients of this loss away from the fp16
    subnormal range of the split-precision dgrad kernels (x = hi + lo/2048 keeps 22 bits only above |x| ~ 6e-5).
    `should_step` (optional callable, evaluated after backward): False drops this step's optimizer update (the Trainer's
    split-precision overflow gate; the schedule still advances, as under GradScaler).
    `phase`: "all" (default) | "grads" (zero_grad .. unscaled gradients; returns (loss, metrics)) | "update" (clip .. scheduler
    on the gradients a "grads" call left; returns None) — the two halves of a step for per-segment graph capture.
    Model-agnostic host logic (any module with the reference's forward signature)."""
    loss = metrics = None
    if phase in ("all", "grads"):
        image1, image2, hr_coord, hr_disp_gt, scale = batch
        optimizer.zero_grad()
        assert model.training
        res = model(image1, image2, iters=train_iters, hr_coord=hr_coord, scale=scale)
        disp_preds = res[1] if isinstance(res, tuple) else res  # IGEV: (init_disp, preds); RAFT: preds (prune_raft_stereo.py:297)
        loss, metrics = sequence_loss_multiscale(disp_preds, hr_disp_gt, (hr_disp_gt < 512) & (hr_disp_gt > 0.0), max_disp=max_disp,
                                                 sync_free=sync_free_loss)
        if scaler is not None:
            scaler.scale(loss).backward()
            scaler.unscale_(optimizer)
        elif loss_scale != 1.0:
            (loss * loss_scale).backward()
            grads = [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
            torch._foreach_mul_(grads, 1.0 / loss_scale)
        else:
            loss.backward()
        loss = loss.detach()
        if phase == "grads":
            return loss, metrics
    torch.nn.utils.clip_grad_norm_([p for g in optimizer.param_groups for p in g["params"]], clip)  # = model.parameters(), without the module walk
    if should_step is not None and not should_step():
        optimizer.zero_grad()
    elif scaler is not None:
        scaler.step(optimizer)
    else:
        optimizer.step()
    if scheduler is not None:
        scheduler.step()
    if scaler is not None:
        scaler.update()
    return (loss, metrics) if phase == "all" else None
