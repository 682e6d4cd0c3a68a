"""Host-side operator layer: torch tensors in, device pointers out to the C ABI.

PyTorch is plumbing here (device memory through the caching allocator, the current HIP stream);
all arithmetic of the hot path happens in libanystereo_hip.so.  Every op demands CUDA (=HIP)
fp32 contiguous tensors and raises RuntimeError otherwise — no silent fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import List, Optional, Sequence

import torch

from . import _lib as L


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> C.c_void_p:
    """The current stream of the current device as a raw hipStream_t (the dispatcher-free lookup: ~0.3 us instead of ~10 us for
    torch.cuda.current_stream() — the training step issues ~1500 of these per step and is host-bound)."""
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return C.c_void_p(_RAW_STREAM(_CUR_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOGUARD = _NoGuard()


def _guard(dev):
    """Device guard for a launch: free when `dev` already is the current device (the normal case: one process per GPU)."""
    if _CUR_DEVICE is not None and dev.index is not None and _CUR_DEVICE() == dev.index:
        return _NOGUARD
    return torch.cuda.device(dev)


def _req(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name}: expected a tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (HIP) tensor — the anystereo hot path has no CPU fallback")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def set_precision(mode: str) -> None:
    """'fp32' = exact fp32 MFMA; 'split' = 3 x fp16 split-precision MFMA (see as_set_precision)."""
    L.check(L.load().as_set_precision({"fp32": 0, "split": 1}[mode]), "set_precision")


def set_fast_fp16(flag: bool) -> None:
    """Reduced-precision variant of split mode for the convolution kernels: fp16 operands, fp32 accumulate, one MFMA per
    product (the reference's autocast path, continuous_IGEVstereo.py:287).  Not the parity mode: its own tolerance."""
    L.check(L.load().as_set_fast16(1 if flag else 0), "set_fast16")


def get_fast_fp16() -> bool:
    return bool(L.load().as_get_fast16())


class fast_fp16:
    """with ops.fast_fp16(True): ...  (restores the previous setting)"""

    def __init__(self, flag: bool):
        self.flag = flag

    def __enter__(self):
        self.prev = get_fast_fp16()
        set_fast_fp16(self.flag)

    def __exit__(self, *exc):
        set_fast_fp16(self.prev)
        return False


def get_precision() -> str:
    return "fp32" if L.load().as_get_precision() == 0 else "split"


# ------------------------------------------------------------------------------------------------
# correlation volume / pyramids / lookup
# ------------------------------------------------------------------------------------------------


def corr_build_pyramid(f1: torch.Tensor, f2: torch.Tensor, num_levels: int) -> List[torch.Tensor]:
    """[B,C,H,W1],[B,C,H,W2] -> levels [B,H,W1,W2>>i] (geometry.py:63-72,:27-29)."""
    _req(f1, "fmap1"), _req(f2, "fmap2")
    if f1.dim() != 4 or f2.dim() != 4 or f1.shape[:3] != f2.shape[:3]:
        raise RuntimeError(f"corr_build_pyramid: incompatible shapes {tuple(f1.shape)} / {tuple(f2.shape)}")
    b, c, h, w1 = f1.shape
    w2 = f2.shape[3]
    lv = [torch.empty((b, h, w1, w2 >> i), device=f1.device, dtype=torch.float32) for i in range(num_levels)]
    pp, keep = L.ptr_array([t.data_ptr() for t in lv])
    with _guard(f1.device):
        L.check(L.load().as_corr_build_pyramid(_p(f1), _p(f2), pp, b, c, h, w1, w2, num_levels, _stream()),
                "corr_build_pyramid")
    return lv


def geo_pyramid(gev: torch.Tensor, num_levels: int) -> List[torch.Tensor]:
    """[B,G,D,H,W] -> levels [B,H,W,D>>i,G] (geometry.py:17-25)."""
    _req(gev, "geo_volume")
    if gev.dim() != 5:
        raise RuntimeError(f"geo_pyramid: expected [B,G,D,H,W], got {tuple(gev.shape)}")
    b, g, d, h, w = gev.shape
    lv = [torch.empty((b, h, w, d >> i, g), device=gev.device, dtype=torch.float32) for i in range(num_levels)]
    pp, keep = L.ptr_array([t.data_ptr() for t in lv])
    with _guard(gev.device):
        L.check(L.load().as_geo_pyramid(_p(gev), pp, b, g, d, h, w, num_levels, _stream()), "geo_pyramid")
    return lv


def geo_corr_lookup(geo: Optional[Sequence[torch.Tensor]], corr: Sequence[torch.Tensor], disp: torch.Tensor,
                    radius: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused pyramid lookup -> [B, L*(2r+1)*(G+1), H, W] (geometry.py:34-60)."""
    _req(disp, "disp")
    nl = len(corr)
    b, one, h, w = disp.shape
    if one != 1:
        raise RuntimeError(f"geo_corr_lookup: disp must be [B,1,H,W], got {tuple(disp.shape)}")
    for i, t in enumerate(corr):
        _req(t, f"corr[{i}]")
    w2 = corr[0].shape[3]
    for i, t in enumerate(corr):
        if tuple(t.shape) != (b, h, w, w2 >> i):
            raise RuntimeError(f"geo_corr_lookup: corr[{i}] has shape {tuple(t.shape)}, expected {(b, h, w, w2 >> i)}")
    g = d = 0
    if geo:
        if len(geo) != nl:
            raise RuntimeError("geo_corr_lookup: geo and corr pyramids differ in depth")
        d, g = geo[0].shape[3], geo[0].shape[4]
        for i, t in enumerate(geo):
            _req(t, f"geo[{i}]")
            if tuple(t.shape) != (b, h, w, d >> i, g):
                raise RuntimeError(f"geo_corr_lookup: geo[{i}] has shape {tuple(t.shape)}, expected {(b, h, w, d >> i, g)}")
    ch = nl * (2 * radius + 1) * (g + 1)
    if out is None:
        out = torch.empty((b, ch, h, w), device=disp.device, dtype=torch.float32)
    else:
        _req(out, "out")
        if tuple(out.shape) != (b, ch, h, w):
            raise RuntimeError("geo_corr_lookup: bad out shape")
    gp, k1 = L.ptr_array([t.data_ptr() for t in geo]) if geo else (None, None)
    cp, k2 = L.ptr_array([t.data_ptr() for t in corr])
    with _guard(disp.device):
        L.check(L.load().as_geo_corr_lookup_fwd(gp, cp, _p(disp), _p(out), b, h, w, w2, d, g, nl, radius, _stream()),
                "geo_corr_lookup_fwd")
    return out


class LookupConvPack:
    """MFMA fragment image of convc1's weight [64, cin, 1, 1] for the fused lookup + convc1 kernel (rebuilt when the weight
    or bias changes: identity through weak references + version counters)."""

    def __init__(self):
        self._key, self._refs, self.image, self.bias = None, None, None, None

    def get(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        ts = [weight, bias]
        key = tuple(None if t is None else (t.data_ptr(), t._version, t.device) for t in ts)
        alive = self._refs is not None and all((r is None) == (t is None) and (r is None or r() is t) for r, t in zip(self._refs, ts))
        if key != self._key or not alive:
            w = weight.detach().reshape(weight.shape[0], -1).float().contiguous()
            if w.shape[0] != 64:
                raise RuntimeError("LookupConvPack: convc1 must have 64 output channels")
            image = torch.empty(L.load().as_lookup_convc1_pack_bytes(w.shape[1]), device=w.device, dtype=torch.uint8)
            with _guard(w.device):
                L.check(L.load().as_lookup_convc1_pack(_p(w), w.shape[1], _p(image), _stream()), "lookup_convc1_pack")
            self.image, self.cin = image, w.shape[1]
            self.bias = None if bias is None else bias.detach().float().contiguous()
            self._key, self._refs = key, [None if t is None else weakref.ref(t) for t in ts]
        return self


def lookup_convc1_supported(geo, corr, radius: int) -> bool:
    g = geo[0].shape[4] if geo else 0
    return radius == 4 and ((g == 8 and len(corr) == 2) or (g == 0 and len(corr) == 4))


def lookup_convc1(geo: Optional[Sequence[torch.Tensor]], corr: Sequence[torch.Tensor], disp: torch.Tensor, radius: int,
                  pack: LookupConvPack, out_bs: Optional["BS8"] = None, out_bs_coff: int = 0, want_f32: bool = False, relu: bool = True):
    """relu(convc1(lookup(disp))) in one kernel (geometry.py:34-60 + update.py:84-85): -> fp32 [B,64,H,W] (want_f32) and / or
    the blocked split-fp16 tensor `out_bs` channels [out_bs_coff, +64)."""
    _req(disp, "disp")
    nl = len(corr)
    b, one, h, w = disp.shape
    w2 = corr[0].shape[3]
    for i, t in enumerate(corr):
        _req(t, f"corr[{i}]")
        if tuple(t.shape) != (b, h, w, w2 >> i):
            raise RuntimeError(f"lookup_convc1: corr[{i}] has shape {tuple(t.shape)}, expected {(b, h, w, w2 >> i)}")
    g = d = 0
    if geo:
        d, g = geo[0].shape[3], geo[0].shape[4]
        for i, t in enumerate(geo):
            _req(t, f"geo[{i}]")
            if tuple(t.shape) != (b, h, w, d >> i, g):
                raise RuntimeError(f"lookup_convc1: geo[{i}] has shape {tuple(t.shape)}")
    if nl * (2 * radius + 1) * (g + 1) != pack.cin:
        raise RuntimeError(f"lookup_convc1: the lookup yields {nl * (2 * radius + 1) * (g + 1)} channels, convc1 expects {pack.cin}")
    out = torch.empty((b, 64, h, w), device=disp.device, dtype=torch.float32) if want_f32 else None
    if out_bs is None and not want_f32:
        raise RuntimeError("lookup_convc1: no output requested")
    if out_bs is not None:
        _req(out_bs.t, "out_bs", torch.float16)
        if out_bs.shape[0] != b or tuple(out_bs.shape[2:]) != (h, w):
            raise RuntimeError("lookup_convc1: out_bs shape mismatch")
    gp, k1 = L.ptr_array([t.data_ptr() for t in geo]) if geo else (None, None)
    cp, k2 = L.ptr_array([t.data_ptr() for t in corr])
    with _guard(disp.device):
        L.check(L.load().as_lookup_convc1_fwd(gp, cp, _p(disp), _p(pack.image), _p(pack.bias), _p(None if out_bs is None else out_bs.t),
                                              0 if out_bs is None else out_bs.c, out_bs_coff, _p(out), 1 if relu else 0,
                                              b, h, w, w2, d, g, nl, radius, _stream()), "lookup_convc1_fwd")
    return out


def loop_front(geo, corr, taps: torch.Tensor, head_bias, disp_old: torch.Tensor, radius: int, pack: LookupConvPack,
               w7: torch.Tensor, b7, copy_out: Optional["BS8"] = None, copy_coff: int = 0):
    """The front of a GRU iteration in one launch (as_loop_front_fwd): disp_new = disp_old + tap_shift_sum(taps) + head_bias,
    cor = relu(convc1(lookup(disp_new))), d1 = relu(conv7x7(disp_new) + b7) -> (disp_new [B,1,H,W], cor BS8, d1 BS8); with
    `copy_out` (BS8) disp_new is also written to its channel `copy_coff`."""
    _req(taps, "taps"), _req(disp_old, "disp")
    b, one, h, w = disp_old.shape
    if taps.shape[0] != b or tuple(taps.shape[2:]) != (h, w) or taps.shape[1] % 9:
        raise RuntimeError("loop_front: taps must be [B, groups*9, H, W]")
    nl = len(corr)
    w2 = corr[0].shape[3]
    for i, t in enumerate(corr):
        _req(t, f"corr[{i}]")
        if tuple(t.shape) != (b, h, w, w2 >> i):
            raise RuntimeError(f"loop_front: corr[{i}] has shape {tuple(t.shape)}")
    g = d = 0
    if geo:
        d, g = geo[0].shape[3], geo[0].shape[4]
        for i, t in enumerate(geo):
            _req(t, f"geo[{i}]")
            if tuple(t.shape) != (b, h, w, d >> i, g):
                raise RuntimeError(f"loop_front: geo[{i}] has shape {tuple(t.shape)}")
    if nl * (2 * radius + 1) * (g + 1) != pack.cin or (w7 is not None and tuple(w7.shape) != (64, 1, 7, 7)):
        raise RuntimeError("loop_front: convc1 / convd1 shapes do not match the lookup")
    wt = tapmajor_7x7(w7) if w7 is not None else None  # None: finish + lookup + convc1 only (the caller runs the 7x7 conv)
    f = lambda t: None if t is None else (t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().float().contiguous())  # noqa: E731
    hb, bb7 = f(head_bias), f(b7)
    disp_new = torch.empty_like(disp_old)
    cor = BS8.empty(b, 64, h, w, disp_old.device)
    d1 = BS8.empty(b, 64, h, w, disp_old.device) if wt is not None else None
    if copy_out is not None:
        _req(copy_out.t, "copy_out", torch.float16)
        if copy_out.shape[0] != b or tuple(copy_out.shape[2:]) != (h, w):
            raise RuntimeError("loop_front: copy_out shape mismatch")
    gp, k1 = L.ptr_array([t.data_ptr() for t in geo]) if geo else (None, None)
    cp, k2 = L.ptr_array([t.data_ptr() for t in corr])
    with _guard(disp_old.device):
        L.check(L.load().as_loop_front_fwd(gp, cp, _p(taps), taps.shape[1] // 9, _p(hb), _p(disp_old), _p(disp_new), _p(pack.image),
                                           _p(pack.bias), _p(cor.t), _p(wt), 0 if wt is None else wt.shape[1], _p(bb7),
                                           _p(None if d1 is None else d1.t),
                                           _p(None if copy_out is None else copy_out.t), 0 if copy_out is None else copy_out.c, copy_coff,
                                           b, h, w, w2, d, g, nl, radius, _stream()), "loop_front_fwd")
    return disp_new, cor, d1


def geo_corr_lookup_backward(disp, d_out, geo_shapes, corr_shapes, radius, into=None):
    """Gradients w.r.t. the pyramid levels (transpose of the lookup).  `into` = (d_geo, d_corr) of an earlier call: the windows
    are ADDED to those buffers (one gradient per level for all GRU iterations of a step) instead of filling fresh ones."""
    _req(disp, "disp"), _req(d_out, "d_out")
    b, _, h, w = disp.shape
    nl = len(corr_shapes)
    if into is not None:
        d_geo, d_corr = into
    else:
        d_corr = [torch.zeros(s, device=disp.device, dtype=torch.float32) for s in corr_shapes]
        d_geo = [torch.zeros(s, device=disp.device, dtype=torch.float32) for s in geo_shapes] if geo_shapes else []
    g = d = 0
    if d_geo:
        d, g = geo_shapes[0][3], geo_shapes[0][4]
    w2 = corr_shapes[0][3]
    gp, k1 = L.ptr_array([t.data_ptr() for t in d_geo]) if d_geo else (None, None)
    cp, k2 = L.ptr_array([t.data_ptr() for t in d_corr])
    with _guard(disp.device):
        fn = L.load().as_geo_corr_lookup_bwd_accum if into is not None else L.load().as_geo_corr_lookup_bwd
        L.check(fn(_p(disp), _p(d_out), gp, cp, b, h, w, w2, d, g, nl, radius, _stream()), "geo_corr_lookup_bwd")
    return d_geo, d_corr


_DT = {torch.float32: L.AS_F32, torch.float16: L.AS_F16, torch.float64: L.AS_F64}


def corr_sampler_forward(volume: torch.Tensor, coords: torch.Tensor, radius: int) -> torch.Tensor:
    """`corr_sampler.forward` contract (sampler/sampler.cpp:24-32) -> corr [N,2r+1,H1,W1]."""
    if volume.dtype not in _DT:
        raise RuntimeError(f"corr_sampler: unsupported volume dtype {volume.dtype}")
    _req(volume, "volume", volume.dtype), _req(coords, "coords")
    if volume.dim() != 4 or coords.dim() != 4 or coords.shape[1] not in (1, 2):
        raise RuntimeError("corr_sampler: volume must be [N,H1,W1,W2] and coords [N,2,H1,W1]")
    n, h1, w1, w2 = volume.shape
    if (coords.shape[0], coords.shape[2], coords.shape[3]) != (n, h1, w1):
        raise RuntimeError("corr_sampler: coords shape does not match volume")
    out = torch.empty((n, 2 * radius + 1, h1, w1), device=volume.device, dtype=volume.dtype)
    with _guard(volume.device):
        L.check(L.load().as_corr_sampler_fwd(_p(volume), _p(coords), _p(out), n, h1, w1, w2, radius,
                                             coords.shape[1], _DT[volume.dtype], _stream()), "corr_sampler_fwd")
    return out


def corr_sampler_backward(volume: torch.Tensor, coords: torch.Tensor, corr_grad: torch.Tensor, radius: int) -> torch.Tensor:
    """`corr_sampler.backward` contract (sampler/sampler.cpp:34-45) -> volume_grad."""
    if volume.dtype not in _DT:
        raise RuntimeError(f"corr_sampler: unsupported volume dtype {volume.dtype}")
    _req(volume, "volume", volume.dtype), _req(coords, "coords"), _req(corr_grad, "corr_grad", volume.dtype)
    n, h1, w1, w2 = volume.shape
    if tuple(corr_grad.shape) != (n, 2 * radius + 1, h1, w1):
        raise RuntimeError("corr_sampler: corr_grad shape mismatch")
    grad = torch.empty_like(volume)
    with _guard(volume.device):
        L.check(L.load().as_corr_sampler_bwd(_p(coords), _p(corr_grad), _p(grad), n, h1, w1, w2, radius,
                                             coords.shape[1], _DT[volume.dtype], _stream()), "corr_sampler_bwd")
    return grad


def gwc_volume(fl: torch.Tensor, fr: torch.Tensor, maxdisp: int, groups: int) -> torch.Tensor:
    """build_gwc_volume (submodule.py:261-271) -> [B,G,D,H,W]."""
    _req(fl, "refimg_fea"), _req(fr, "targetimg_fea")
    if fl.shape != fr.shape or fl.dim() != 4:
        raise RuntimeError("gwc_volume: feature maps must share a [B,C,H,W] shape")
    b, c, h, w = fl.shape
    out = torch.empty((b, groups, maxdisp, h, w), device=fl.device, dtype=torch.float32)
    with _guard(fl.device):
        L.check(L.load().as_gwc_volume_fwd(_p(fl), _p(fr), _p(out), b, c, h, w, maxdisp, groups, _stream()), "gwc_volume_fwd")
    return out


def disparity_regression(cost: torch.Tensor, apply_softmax: bool) -> torch.Tensor:
    """(softmax over D then) sum_d d*p (continuous_IGEVstereo.py:267-268, submodule.py:321-325) -> [B,1,H,W]."""
    _req(cost, "cost")
    b, d, h, w = cost.shape
    out = torch.empty((b, 1, h, w), device=cost.device, dtype=torch.float32)
    with _guard(cost.device):
        L.check(L.load().as_disparity_regression(_p(cost), _p(out), b, d, h, w, 1 if apply_softmax else 0, _stream()),
                "disparity_regression")
    return out


# ------------------------------------------------------------------------------------------------
# convolutions (update block, MLP)
# ------------------------------------------------------------------------------------------------


class PackedConv:
    """Weights of one (possibly channel-concatenated) conv re-laid out for the implicit-GEMM kernel.

    `weights` are nn.Conv2d weights [Cout_i, Cin, K, K] (or nn.Linear [Cout_i, Cin]) concatenated
    along Cout; the pack is rebuilt whenever a weight's version counter, storage or device changes,
    so optimiser steps and load_state_dict are picked up."""

    def __init__(self):
        self._key = None
        self._refs = None
        self.wpack = None
        self.bias = None
        self.cin = self.cout = self.ks = 0
        self.split = False

    def invalidate(self):
        """Force a rebuild at the next use (after edits through `.data`, which do not bump version counters)."""
        self._key = None

    def _alive(self, ts) -> bool:
        # the key holds (address, version): a freed tensor's address (version 0 again) can be recycled by the caching
        # allocator, so the pack also remembers WHICH tensor objects it was built from
        return self._refs is not None and len(self._refs) == len(ts) and all(
            (r is None) == (t is None) and (r is None or r() is t) for r, t in zip(self._refs, ts))

    def _remember(self, ts):
        self._refs = [None if t is None else weakref.ref(t) for t in ts]

    def get_folded(self, conv, bn):
        """Pack of `bn(conv(x))` for an eval-mode BatchNorm: w' = w * g/sqrt(var+eps), b' = (b - mean) * g/sqrt(var+eps) + beta
        (the affine map BatchNorm applies with its running statistics).  Rebuilt when any of the six tensors changes."""
        split = L.load().as_get_precision() == 1
        ts = [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        key = ("bn", split) + tuple(None if t is None else (t.data_ptr(), t._version, t.device) for t in ts)
        if key != self._key or not self._alive(ts):
            w, b = fold_bn(conv, bn)
            self._build([w], [b], split, key)
            self._remember(ts)
        return self

    def get(self, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]], transform=None):
        split = L.load().as_get_precision() == 1
        key = (split,) + tuple((w.data_ptr(), w._version, w.device) for w in weights) + \
            tuple((None if b is None else (b.data_ptr(), b._version)) for b in biases)
        ts = list(weights) + list(biases)
        if key != self._key or not self._alive(ts):
            ws = [w.detach() if transform is None else transform(w.detach()) for w in weights]
            self._build(ws, biases, split, key)
            self._remember(ts)
        return self

    def get_dgrad(self, weight: torch.Tensor):
        """Pack of the DATA-GRADIENT convolution of `weight` [Cout, Cin, k, k] (or [Cout, Cin]): W' = W transposed over the channel
        axes with flipped taps, no bias.  In split precision the pack kernel reads W in that order directly
        (as_conv_pack_weights_split_t): no `transpose().flip().contiguous()` copies per layer and step."""
        split = L.load().as_get_precision() == 1
        key = ("dgrad", split, weight.data_ptr(), weight._version, weight.device)
        if key != self._key or not self._alive([weight]):
            w = weight.detach()
            w = w.reshape(w.shape[0], w.shape[1], *(w.shape[2:] if w.dim() == 4 else (1, 1)))
            if not split:
                self._build([w.transpose(0, 1).flip(2, 3).contiguous()], [None], split, key)
            else:
                w = w.contiguous().float()
                _req(w, "conv weight")
                cout_w, cin_w, ks, ks2 = w.shape
                if ks != ks2:
                    raise RuntimeError("PackedConv: non-square kernel")
                cin, cout = cout_w, cin_w   # of the data-gradient convolution
                n = L.load().as_conv_pack_size_split(cin, cout, ks)
                if n <= 0:
                    raise RuntimeError(f"PackedConv: unsupported conv Cin={cin} Cout={cout} K={ks}")
                wp = torch.empty(n, device=w.device, dtype=torch.float16)
                with _guard(w.device):
                    L.check(L.load().as_conv_pack_weights_split_t(_p(w), _p(wp), cin, cout, ks, _stream()), "conv_pack_weights_split_t")
                self.split, self.bias, self.wpack = True, None, wp
                self.cin, self.cout, self.ks = cin, cout, ks
                self._key = key
            self._remember([weight])
        return self

    def _build(self, ws, biases, split, key):
        ws = [w.reshape(w.shape[0], w.shape[1], *(w.shape[2:] if w.dim() == 4 else (1, 1))) for w in ws]
        w = torch.cat(ws, dim=0).contiguous().float() if len(ws) > 1 else ws[0].contiguous().float()
        _req(w, "conv weight")
        cout, cin, ks, ks2 = w.shape
        if ks != ks2:
            raise RuntimeError("PackedConv: non-square kernel")
        n = (L.load().as_conv_pack_size_split if split else L.load().as_conv_pack_size)(cin, cout, ks)
        if n <= 0:
            raise RuntimeError(f"PackedConv: unsupported conv Cin={cin} Cout={cout} K={ks}")
        wp = torch.empty(n, device=w.device, dtype=torch.float16 if split else torch.float32)
        with _guard(w.device):
            if split:
                L.check(L.load().as_conv_pack_weights_split(_p(w), _p(wp), cin, cout, ks, _stream()), "conv_pack_weights_split")
            else:
                L.check(L.load().as_conv_pack_weights(_p(w), _p(wp), cin, cout, ks, _stream()), "conv_pack_weights")
        self.split = split
        if all(b is None for b in biases):
            bias = None
        else:
            bias = torch.cat([(torch.zeros(wi.shape[0], device=w.device) if bi is None else bi.detach().float())
                              for wi, bi in zip(ws, biases)]).contiguous()
        self.wpack, self.bias, self.cin, self.cout, self.ks, self._key = wp, bias, cin, cout, ks, key


@torch.no_grad()
def fold_bn(conv, bn, out_dim: int = 0):
    """(weight, bias) of the conv that equals eval-mode `bn(conv(x))`; scale computed in fp64, returned fp32.
    out_dim: the weight's output-channel dimension (0 for Conv*, 1 for ConvTranspose*)."""
    var, mean = bn.running_var.double(), bn.running_mean.double()
    g = torch.ones_like(var) if bn.weight is None else bn.weight.double()
    beta = torch.zeros_like(var) if bn.bias is None else bn.bias.double()
    s = g / torch.sqrt(var + bn.eps)
    shape = [1] * conv.weight.dim()
    shape[out_dim] = -1
    w = (conv.weight.double() * s.view(*shape)).float()
    b0 = torch.zeros_like(var) if conv.bias is None else conv.bias.double()
    return w, ((b0 - mean) * s + beta).float()


class BS8:
    """Blocked split-fp16 activation tensor, the link format between two split-precision convolutions
    (include/anystereo_hip.h, as_conv_desc.src_bs / out_bs): t [B, 2, ceil(C/8), H, W, 8] float16 holds a plane set of hi parts
    and one of lo parts of x = hi + lo/2048, the 8 channels of a block contiguous per pixel — what the consuming kernel's
    loaders would compute from the fp32 tensor, so results are bit-identical to passing that tensor while a loader item is
    2 coalesced 16-B loads instead of 8 dword loads and the split."""

    __slots__ = ("t", "c")

    def __init__(self, t: torch.Tensor, c: int):
        self.t, self.c = t, c

    @staticmethod
    def empty(b: int, c: int, h: int, w: int, device) -> "BS8":
        return BS8(torch.empty((b, 2, (c + 7) // 8, h, w, 8), device=device, dtype=torch.float16), c)

    @property
    def shape(self):  # logical NCHW shape
        return (self.t.shape[0], self.c, self.t.shape[3], self.t.shape[4])

    @property
    def device(self):
        return self.t.device

    def record_stream(self, stream) -> None:
        self.t.record_stream(stream)

    def float(self) -> torch.Tensor:
        """The fp32 tensor the record pairs stand for (tests / debugging)."""
        b, _, c8, h, w, _ = self.t.shape
        v = self.t[:, 0].float() + self.t[:, 1].float() / 2048.0
        return v.permute(0, 1, 4, 2, 3).reshape(b, c8 * 8, h, w)[:, :self.c].contiguous()


def conv2d(srcs: Sequence[torch.Tensor], pack: PackedConv, act: int = L.ACT_NONE, add: Optional[torch.Tensor] = None,
           add_coff: int = 0, out: Optional[torch.Tensor] = None, out_coff: int = 0, epilogue: int = L.EPI_LINEAR,
           h: Optional[torch.Tensor] = None, z: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None,
           stride: int = 1, out_bs: Optional[BS8] = None, out_bs_coff: int = 0, bs_only: bool = False, dual: Optional[dict] = None,
           tap_w: Optional[torch.Tensor] = None):
    """Implicit-GEMM conv over the channel concat of `srcs` (never materialised) with fused epilogue.
    stride 2: 3x3 / padding 1 / LINEAR epilogue in split precision only; outputs are [(H-1)//2+1, (W-1)//2+1].
    Split precision only: a source may be a BS8 (blocked split-fp16 link tensor); `out_bs` receives such a copy of the result
    (LINEAR / GRU_Q: of out; GRU_ZR: of r*h) in channels [out_bs_coff, ...); with `bs_only` the fp32 form of that result is
    not written and None is returned in its place.
    dual = {"src": tensor | BS8, "pack": PackedConv, "out_coff": int, "out_bs_coff": int}: a second convolution of the same
    shape in the same launch, writing its own channel window of out / out_bs (LINEAR epilogue, one source, no add).
    Optional keys: "h" (its residual; then `h` must be given too), "act" (its own activation), and "out" / "out_bs": DENSE outputs
    of its own ([B,Cout,H,W] / BS8 of Cout channels, mirroring which of out / out_bs the first convolution writes) instead of a
    channel window; with "out": True a fresh tensor is allocated.  Returns (out, out_second) then."""
    b, _, hin, win = srcs[0].shape
    if stride not in (1, 2):
        raise RuntimeError("conv2d: stride must be 1 or 2")
    hh, ww = (hin, win) if stride == 1 else ((hin - 1) // 2 + 1, (win - 1) // 2 + 1)
    kc = 16 if pack.split else (8 if pack.ks == 3 else 32)  # channels per K chunk of the kernel (csrc/conv.hip)
    if any(s.shape[1] % kc for s in srcs[:-1]):
        # a K chunk must not straddle two tensors: materialise the concat for odd splits (never on the model path)
        srcs = [torch.cat([s.float() if isinstance(s, BS8) else s for s in srcs], dim=1)]
    if (out_bs is not None or any(isinstance(s, BS8) for s in srcs)) and not (pack.split and stride == 1):
        raise RuntimeError("conv2d: blocked split-fp16 tensors need the split-precision kernel at stride 1")
    d = L.ConvDesc()
    cin = 0
    if len(srcs) > L.AS_MAX_SRCS:
        raise RuntimeError(f"conv2d: at most {L.AS_MAX_SRCS} sources")
    for i, s in enumerate(srcs):
        if isinstance(s, BS8):
            _req(s.t, f"src[{i}]", torch.float16)
            d.src_bs[i] = 1
        else:
            _req(s, f"src[{i}]")
        if s.shape[0] != b or tuple(s.shape[2:]) != (hin, win):
            raise RuntimeError(f"conv2d: src[{i}] shape {tuple(s.shape)} does not match {(b, '*', hin, win)}")
        d.src[i] = s.t.data_ptr() if isinstance(s, BS8) else s.data_ptr()
        d.src_c[i] = s.shape[1]
        cin += s.shape[1]
    if cin != pack.cin:
        raise RuntimeError(f"conv2d: sources hold {cin} channels but the weights expect {pack.cin}")
    d.n_src = len(srcs)
    d.wpack = pack.wpack.data_ptr()
    d.bias = 0 if pack.bias is None else pack.bias.data_ptr()
    cout = pack.cout
    if add is not None:
        _req(add, "add")
        if add.shape[0] != b or tuple(add.shape[2:]) != (hh, ww):
            raise RuntimeError("conv2d: add shape mismatch")
        d.add, d.add_ctot, d.add_coff = add.data_ptr(), add.shape[1], add_coff
    dev = srcs[0].device
    if out_bs is not None:
        _req(out_bs.t, "out_bs", torch.float16)
        cres = cout // 2 if epilogue == L.EPI_GRU_ZR else cout
        if out_bs.shape[0] != b or tuple(out_bs.shape[2:]) != (hh, ww) or out_bs_coff % 8 or out_bs_coff + cres > (out_bs.c + 7) // 8 * 8:
            raise RuntimeError("conv2d: out_bs does not fit the result")
        d.out_bs, d.out_bs_ctot, d.out_bs_coff, d.bs_only = out_bs.t.data_ptr(), out_bs.c, out_bs_coff, 1 if bs_only else 0
    elif bs_only:
        raise RuntimeError("conv2d: bs_only without out_bs")
    if epilogue == L.EPI_RELU_TAPS:
        # act(conv) feeds a following 3x3, Cout -> 1 convolution with weights tap_w [Cout, 9]: the result is that convolution's
        # per-tap channel reductions, one set of 9 planes per 64-channel tile (finish with tap_shift_sum)
        _req(tap_w, "tap_w")
        if tuple(tap_w.shape) != (cout, 9) or not pack.split or pack.ks != 3 or stride != 1:
            raise RuntimeError("conv2d(RELU_TAPS): tap_w must be [Cout, 9]; 3x3 split-precision convolution at stride 1")
        out = torch.empty((b, (cout + 63) // 64 * 9, hh, ww), device=dev, dtype=torch.float32)
        d.out, d.tap_w = out.data_ptr(), tap_w.data_ptr()
    elif epilogue == L.EPI_LINEAR and bs_only:
        out = None
    elif epilogue == L.EPI_LINEAR:
        if out is None:
            out = torch.empty((b, cout, hh, ww), device=dev, dtype=torch.float32)
        _req(out, "out")
        d.out, d.out_ctot, d.out_coff = out.data_ptr(), out.shape[1], out_coff
    if epilogue == L.EPI_LINEAR:
        if h is not None:  # residual tail: out = relu(h + act(conv))
            _req(h, "h")
            if tuple(h.shape) != (b, cout, hh, ww):
                raise RuntimeError("conv2d(LINEAR): residual h must be [B,Cout,H,W]")
            d.h = h.data_ptr()
    elif epilogue == L.EPI_RELU_TAPS:
        pass
    elif epilogue == L.EPI_GRU_ZR:
        ch = cout // 2
        _req(h, "h")
        if out is None:
            out = torch.empty((b, ch, hh, ww), device=dev, dtype=torch.float32)
        if out2 is None and not bs_only:
            out2 = torch.empty((b, ch, hh, ww), device=dev, dtype=torch.float32)
        _req(out, "out")
        for t in (h, out) + (() if bs_only else (out2,)):
            _req(t, "h/out/out2")
            if tuple(t.shape) != (b, ch, hh, ww):
                raise RuntimeError("conv2d(GRU_ZR): h/out/out2 must be [B,Cout/2,H,W]")
        d.h, d.out, d.out2 = h.data_ptr(), out.data_ptr(), (0 if bs_only else out2.data_ptr())
        if bs_only:
            out2 = None
    else:
        _req(h, "h"), _req(z, "z")
        if out is None:
            out = torch.empty((b, cout, hh, ww), device=dev, dtype=torch.float32)
        _req(out, "out")
        for t in (h, z, out):
            if tuple(t.shape) != (b, cout, hh, ww):
                raise RuntimeError("conv2d(GRU_Q): h/z/out must be [B,Cout,H,W]")
        d.h, d.z, d.out = h.data_ptr(), z.data_ptr(), out.data_ptr()
    if dual is not None:
        s2, p2 = dual["src"], dual["pack"]
        if (epilogue != L.EPI_LINEAR or len(srcs) != 1 or add is not None or stride != 1
                or (p2.cin, p2.cout, p2.ks, p2.split) != (pack.cin, pack.cout, pack.ks, pack.split) or tuple(s2.shape) != tuple(srcs[0].shape)):
            raise RuntimeError("conv2d(dual): needs two LINEAR single-source convolutions of one shape")
        h2 = dual.get("h")
        if (h is None) != (h2 is None):
            raise RuntimeError("conv2d(dual): a residual for both convolutions or for neither")
        if h2 is not None:
            _req(h2, "dual h")
            if tuple(h2.shape) != (b, cout, hh, ww):
                raise RuntimeError("conv2d(dual): residual h must be [B,Cout,H,W]")
            d.h2 = h2.data_ptr()
        if "act" in dual:
            d.dual_act2, d.act2 = 1, int(dual["act"])
        o2, obs2 = dual.get("out"), dual.get("out_bs")
        if o2 is not None or obs2 is not None:
            if (obs2 is not None) != (out_bs is not None) or (o2 is not None) != (out is not None):
                raise RuntimeError("conv2d(dual): separate second outputs mirror the first convolution's (fp32 and / or blocked)")
            if o2 is True:
                o2 = torch.empty((b, cout, hh, ww), device=dev, dtype=torch.float32)
            if o2 is not None:
                _req(o2, "dual out")
                if tuple(o2.shape) != (b, cout, hh, ww):
                    raise RuntimeError("conv2d(dual): out must be [B,Cout,H,W]")
                d.out_b = o2.data_ptr()
            if obs2 is not None:
                _req(obs2.t, "dual out_bs", torch.float16)
                if tuple(obs2.shape) != (b, cout, hh, ww):
                    raise RuntimeError("conv2d(dual): out_bs must hold [B,Cout,H,W]")
                d.out_bs_b = obs2.t.data_ptr()
            dual_second_out = o2
        else:
            dual_second_out = None
        if isinstance(s2, BS8):
            _req(s2.t, "dual src", torch.float16)
        else:
            _req(s2, "dual src")
        d.dual, d.src2, d.src2_bs = 1, (s2.t if isinstance(s2, BS8) else s2).data_ptr(), 1 if isinstance(s2, BS8) else 0
        d.wpack2, d.bias2 = p2.wpack.data_ptr(), (0 if p2.bias is None else p2.bias.data_ptr())
        d.out_coff2, d.out_bs_coff2 = int(dual.get("out_coff", 0)), int(dual.get("out_bs_coff", 0))
        if out is not None and d.out_coff2 + cout > out.shape[1]:
            raise RuntimeError("conv2d(dual): second output window outside out")
        if out_bs is not None and (d.out_bs_coff2 % 8 or d.out_bs_coff2 + cout > (out_bs.c + 7) // 8 * 8):
            raise RuntimeError("conv2d(dual): second output window outside out_bs")
    d.B, d.H, d.W, d.Cin, d.Cout, d.KS = b, hin, win, cin, cout, pack.ks
    d.stride = stride
    d.act, d.epilogue = act, epilogue
    d.precision = 1 if pack.split else 0
    ws = None
    if pack.split and stride == 1:
        n_ws = L.load().as_conv_ws_elems(b, cout, hh, ww)
        if n_ws > 0:  # small feature map: give the kernel split-K scratch (caching allocator: no sync)
            ws = torch.empty(n_ws, device=dev, dtype=torch.float32)
            d.ws, d.ws_elems = ws.data_ptr(), n_ws
    with _guard(dev):
        L.check(L.load().as_conv2d(C.byref(d), _stream()), "conv2d")
    if dual is not None and (dual.get("out") is not None or dual.get("out_bs") is not None):
        return out, dual_second_out
    return (out, out2) if epilogue == L.EPI_GRU_ZR else out


def conv2d_plain(x: torch.Tensor, pack: "PackedConv", act: int = L.ACT_NONE) -> torch.Tensor:
    """conv2d([x], pack, act) for the plain case — one fp32 source, LINEAR epilogue, stride 1, fresh output — with the host
    work cut to the descriptor fields that case needs (the training step issues ~450 of these per step and is host-bound)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()) or x.shape[1] != pack.cin:
        return conv2d([x], pack, act=act)  # full checks and error messages
    b, cin, hh, ww = x.shape
    cout = pack.cout
    lib = L.load()
    out = torch.empty((b, cout, hh, ww), device=x.device, dtype=torch.float32)
    d = L.ConvDesc()
    d.src[0], d.src_c[0], d.n_src = x.data_ptr(), cin, 1
    d.wpack = pack.wpack.data_ptr()
    if pack.bias is not None:
        d.bias = pack.bias.data_ptr()
    d.out, d.out_ctot = out.data_ptr(), cout
    d.B, d.H, d.W, d.Cin, d.Cout, d.KS = b, hh, ww, cin, cout, pack.ks
    d.stride, d.act, d.epilogue, d.precision = 1, act, L.EPI_LINEAR, 1 if pack.split else 0
    ws = None
    if pack.split:
        n_ws = lib.as_conv_ws_elems(b, cout, hh, ww)
        if n_ws > 0:
            ws = torch.empty(n_ws, device=x.device, dtype=torch.float32)
            d.ws, d.ws_elems = ws.data_ptr(), n_ws
    with _guard(x.device):
        L.check(lib.as_conv2d(C.byref(d), _stream()), "conv2d")
    return out


_TAPMAJOR = {}  # (data_ptr, version, device) of a [Cout,1,7,7] weight -> its [49,Cout] transpose


def tapmajor_7x7(weight: torch.Tensor) -> torch.Tensor:
    """[Cout,1,7,7] -> tap-major [49, Cout padded to 64] fp32, cached per PARAMETER object + version.  (Pass the module's
    parameter, not a `.detach()` made at the call site: that is a new object every time and would rebuild the copy — two
    elementwise kernels, 10 us on the GRU loop's critical stream — at every call.)"""
    cout = weight.shape[0]
    key = (weight.data_ptr(), weight._version, weight.device)
    ent = _TAPMAJOR.get(key)
    # the entry must belong to THIS tensor: a freed weight's address (and version 0) is reused by the caching allocator
    wt = ent[1] if (ent is not None and ent[0]() is weight) else None
    if wt is None:
        if len(_TAPMAJOR) > 64:
            _TAPMAJOR.clear()
        wt = torch.zeros((49, (cout + 63) // 64 * 64), device=weight.device, dtype=torch.float32)
        wt[:, :cout] = weight.detach().float().reshape(cout, 49).t()
        _TAPMAJOR[key] = (weakref.ref(weight), wt)
    return wt


def conv7x7_c1_relu(x, weight, bias, out=None, out_coff=0, copy_out=None, copy_coff=0):
    """relu(conv7x7(x [B,1,H,W]) + bias) into channels [out_coff, out_coff+Cout) of `out` (update.py:81,87);
    with `copy_out` [B,C,H,W], x is also written to its channel `copy_coff`."""
    _req(x, "x")
    if not weight.is_cuda:
        raise RuntimeError("conv7x7_c1_relu: weight must be a CUDA (HIP) tensor")
    b, one, h, w = x.shape
    cout = weight.shape[0]
    if one != 1 or tuple(weight.shape[1:]) != (1, 7, 7):
        raise RuntimeError("conv7x7_c1_relu: expects x [B,1,H,W] and weight [Cout,1,7,7]")
    if out is None:
        out = torch.empty((b, cout, h, w), device=x.device, dtype=torch.float32)
    obs = isinstance(out, BS8)  # blocked split-fp16 result (feeds a split-precision convolution only)
    _req(out.t if obs else out, "out", torch.float16 if obs else torch.float32)
    wt = tapmajor_7x7(weight)
    if bias is not None:
        bias = bias.detach()
        bias = bias if (bias.dtype == torch.float32 and bias.is_contiguous()) else bias.float().contiguous()
    with _guard(x.device):
        cbs = isinstance(copy_out, BS8)
        if copy_out is not None:
            _req(copy_out.t if cbs else copy_out, "copy_out", torch.float16 if cbs else torch.float32)
            if copy_out.shape[0] != b or tuple(copy_out.shape[2:]) != (h, w):
                raise RuntimeError("conv7x7_c1_relu: copy_out shape mismatch")
        L.check(L.load().as_conv7x7_c1_relu(_p(x), _p(wt), _p(bias), _p(out.t if obs else out), b, h, w, cout, out.shape[1], out_coff, 1,
                                            _p(copy_out.t if cbs else copy_out), 0 if copy_out is None else copy_out.shape[1], copy_coff,
                                            1 if cbs else 0, 1 if obs else 0, _stream()),
                "conv7x7_c1_relu")
    return out


def conv3x3_few(x, wpack, bias=None, stride: int = 1, act: int = L.ACT_NONE):
    """act(conv3x3(x [B,Cin<=8,H,W], padding 1, stride) + bias) with wpack [Cin,9,Cout] (= weight.permute(1,2,3,0)): the
    3-channel image stems (extractor.py:331-336)."""
    _req(x, "x"), _req(wpack, "wpack")
    b, cin, h, w = x.shape
    if wpack.dim() != 3 or wpack.shape[0] != cin or wpack.shape[1] != 9:
        raise RuntimeError("conv3x3_few: wpack must be [Cin,9,Cout]")
    cout = wpack.shape[2]
    if bias is not None:
        _req(bias, "bias")
    out = torch.empty((b, cout, (h - 1) // stride + 1, (w - 1) // stride + 1), device=x.device, dtype=torch.float32)
    with _guard(x.device):
        L.check(L.load().as_conv3x3_few(_p(x), _p(wpack), _p(bias), _p(out), b, cin, cout, h, w, stride, act, _stream()), "conv3x3_few")
    return out


class IrBlockPack:
    """Fragments + fp32 parameters of one MobileNetV2 inverted-residual block for as_ir_block (csrc/irblock.hip), BatchNorm folded
    in fp64 (fold_bn); rebuilt when any of the block's 15 tensors changes (object identity + version)."""

    def __init__(self):
        self._key, self._refs = None, None
        self.pack = self.fparams = None
        self.dims = None

    def get(self, conv_pw, bn1, conv_dw, bn2, conv_pwl, bn3):
        ts = [conv_pw.weight, conv_dw.weight, conv_pwl.weight]
        for bn in (bn1, bn2, bn3):
            ts += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        key = tuple((t.data_ptr(), t._version, t.device) for t in ts)
        alive = self._refs is not None and all(r() is t for r, t in zip(self._refs, ts))
        if key != self._key or not alive:
            with torch.no_grad():
                w1, b1 = fold_bn(conv_pw, bn1)
                wd, b2 = fold_bn(conv_dw, bn2)
                w3, b3 = fold_bn(conv_pwl, bn3)
                mid, cin = w1.shape[:2]
                cout = w3.shape[0]
                mp = (mid + 31) // 32 * 32
                f = torch.zeros(11 * mp + cout, device=w1.device, dtype=torch.float32)
                f[:mid] = b1
                f[mp:mp + mid] = b2
                f[2 * mp:2 * mp + 9 * mid] = wd.reshape(mid, 9).reshape(-1)
                f[11 * mp:] = b3
                lib = L.load()
                pack = torch.empty(int(lib.as_ir_block_pack_bytes(cin, mid, cout)), device=w1.device, dtype=torch.uint8)
                w1c, w3c = w1.reshape(mid, cin).contiguous(), w3.reshape(cout, mid).contiguous()
                with _guard(w1.device):
                    L.check(lib.as_ir_block_pack(_p(w1c), _p(w3c), cin, mid, cout, pack.data_ptr(), _stream()), "ir_block_pack")
            self.pack, self.fparams, self.dims = pack, f, (cin, mid, cout)
            self._key, self._refs = key, [weakref.ref(t) for t in ts]
        return self


def ir_block(x: torch.Tensor, pk: "IrBlockPack", stride: int, residual: bool) -> torch.Tensor:
    """One MobileNetV2 inverted-residual block (extractor.py:327-342) as ONE launch: expand 1x1 + ReLU6 -> depthwise 3x3 + ReLU6 ->
    project 1x1 (+ x), eval-mode BatchNorm folded; the expanded tensor never leaves LDS (as_ir_block)."""
    _req(x, "x")
    b, c, h, w = x.shape
    cin, mid, cout = pk.dims
    if c != cin:
        raise RuntimeError(f"ir_block: input has {c} channels, the block expects {cin}")
    out = torch.empty((b, cout, (h - 1) // stride + 1, (w - 1) // stride + 1), device=x.device, dtype=torch.float32)
    with _guard(x.device):
        L.check(L.load().as_ir_block(_p(x), pk.pack.data_ptr(), _p(pk.fparams), _p(out), b, cin, mid, cout, h, w, stride,
                                     1 if residual else 0, _stream()), "ir_block")
    return out


class Stem7x7Pack:
    """MFMA fragments of a [64,3,7,7] weight (csrc/stem7x7.hip), rebuilt when the weight tensor object or its version changes."""

    def __init__(self):
        self._key = None
        self._ref = None
        self.wpack = None

    def invalidate(self):
        """Force a rebuild at the next use (after edits through `.data`, which do not bump version counters)."""
        self._key = None

    def get(self, weight: torch.Tensor) -> torch.Tensor:
        key = (weight._version, weight.data_ptr(), weight.device)
        # (address, version) alone can alias a freed tensor whose storage the allocator recycled: remember the object too
        if key != self._key or self._ref is None or self._ref() is not weight:
            if tuple(weight.shape) != (64, 3, 7, 7) or not weight.is_cuda:
                raise RuntimeError("conv7x7_c3: weight must be a CUDA tensor [64,3,7,7]")
            w = weight.detach().float().contiguous()
            lib = L.load()
            self.wpack = torch.empty((int(lib.as_conv7x7_c3_pack_bytes()),), device=weight.device, dtype=torch.uint8)
            with _guard(weight.device):
                L.check(lib.as_conv7x7_c3_pack(_p(w), self.wpack.data_ptr(), _stream()), "conv7x7_c3_pack")
            self._key, self._ref = key, weakref.ref(weight)
        return self.wpack


def conv7x7_c3(x, pack: "Stem7x7Pack", weight, bias=None, act: int = L.ACT_NONE):
    """act(conv7x7(x [B,3,H,W], weight [64,3,7,7], padding 3) + bias) -> [B,64,H,W] (the encoders' stem, extractor.py:127)."""
    _req(x, "x")
    b, c, h, w = x.shape
    if c != 3:
        raise RuntimeError("conv7x7_c3: x must be [B,3,H,W]")
    wp = pack.get(weight)
    if bias is not None:
        bias = bias.detach()
        bias = bias if (bias.dtype == torch.float32 and bias.is_contiguous()) else bias.float().contiguous()
    out = torch.empty((b, 64, h, w), device=x.device, dtype=torch.float32)
    with _guard(x.device):
        L.check(L.load().as_conv7x7_c3(_p(x), wp.data_ptr(), _p(bias), _p(out), b, h, w, act, _stream()), "conv7x7_c3")
    return out


def conv3x3_to1(x, weight, bias):
    """conv3x3(x [B,Cin,H,W]) + bias -> [B,1,H,W] (DispHead.conv2, update.py:19,24)."""
    _req(x, "x"), _req(weight, "weight")
    b, cin, h, w = x.shape
    if tuple(weight.shape) != (1, cin, 3, 3):
        raise RuntimeError("conv3x3_to1: weight must be [1,Cin,3,3]")
    out = torch.empty((b, 1, h, w), device=x.device, dtype=torch.float32)
    with _guard(x.device):
        L.check(L.load().as_conv3x3_to1(_p(x), _p(weight), _p(bias), _p(out), b, cin, h, w, _stream()), "conv3x3_to1")
    return out


def tap_shift_sum(s, bias, addend=None):
    """out[b,0,y,x] = (addend +) bias + sum_t s[b,t,y+ky-1,x+kx-1] (zero padded) — see as_tap_shift_sum."""
    _req(s, "s")
    if addend is not None:
        _req(addend, "addend")
        if tuple(addend.shape) != (s.shape[0], 1, s.shape[2], s.shape[3]):
            raise RuntimeError("tap_shift_sum: addend must be [B,1,H,W]")
    b, planes, h, w = s.shape
    if planes % 9:
        raise RuntimeError("tap_shift_sum: expects [B, groups*9, H, W]")
    out = torch.empty((b, 1, h, w), device=s.device, dtype=torch.float32)
    with _guard(s.device):
        L.check(L.load().as_tap_shift_sum(_p(s), _p(bias), _p(addend), _p(out), b, h, w, planes // 9, _stream()), "tap_shift_sum")
    return out


def pool2x(x):
    """avg_pool2d(x, 3, stride=2, padding=1) (update.py:94-95)."""
    _req(x, "x")
    b, c, h, w = x.shape
    out = torch.empty((b, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), device=x.device, dtype=torch.float32)
    with _guard(x.device):
        L.check(L.load().as_pool2x(_p(x), _p(out), b, c, h, w, _stream()), "pool2x")
    return out


def pool2x_bs(x) -> "BS8":
    """pool2x with a blocked split-fp16 result (for maps that only feed split-precision convolutions)."""
    _req(x, "x")
    b, c, h, w = x.shape
    out = BS8.empty(b, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, x.device)
    with _guard(x.device):
        L.check(L.load().as_pool2x_bs(_p(x), _p(out.t), b, c, h, w, _stream()), "pool2x_bs")
    return out


def interp_bs(x, ho: int, wo: int) -> "BS8":
    """Bilinear align_corners=True resize with a blocked split-fp16 result."""
    _req(x, "x")
    b, c, h, w = x.shape
    out = BS8.empty(b, c, ho, wo, x.device)
    with _guard(x.device):
        L.check(L.load().as_interp_bilinear_ac_bs(_p(x), _p(out.t), b, c, h, w, ho, wo, _stream()), "interp_bs")
    return out


def dwconv3x3(x, weight, bias=None, stride: int = 1, act: int = L.ACT_NONE, residual=None):
    """act(depthwise conv3x3(x, padding 1, stride) + bias) (+ residual) — conv_dw of a MobileNetV2 block."""
    _req(x, "x"), _req(weight, "weight")
    b, c, h, w = x.shape
    if tuple(weight.shape) != (c, 1, 3, 3):
        raise RuntimeError("dwconv3x3: weight must be [C,1,3,3]")
    out = torch.empty((b, c, (h - 1) // stride + 1, (w - 1) // stride + 1), device=x.device, dtype=torch.float32)
    if residual is not None:
        _req(residual, "residual")
        if residual.shape != out.shape:
            raise RuntimeError("dwconv3x3: residual shape mismatch")
    if bias is not None:
        _req(bias, "bias")
    with _guard(x.device):
        L.check(L.load().as_dwconv3x3(_p(x), _p(weight), _p(bias), _p(residual), _p(out), b, c, h, w, stride, act, _stream()),
                "dwconv3x3")
    return out


def dwconv3x3_backward_data(d_out, weight, h: int, w: int, stride: int):
    """d_x [B,C,h,w] of the depthwise 3x3 convolution (padding 1, stride 1|2) with output gradient d_out."""
    _req(d_out, "d_out"), _req(weight, "weight")
    b, c, ho, wo = d_out.shape
    if (ho, wo) != ((h - 1) // stride + 1, (w - 1) // stride + 1) or tuple(weight.shape) != (c, 1, 3, 3):
        raise RuntimeError("dwconv3x3_backward_data: shapes do not belong to one convolution")
    if stride == 1:  # the forward kernel on the flipped taps
        return dwconv3x3(d_out, weight.flip(2, 3).contiguous())
    d_x = torch.empty((b, c, h, w), device=d_out.device, dtype=torch.float32)
    with _guard(d_out.device):
        L.check(L.load().as_dwconv3x3_s2_bwd_data(_p(d_out), _p(weight), _p(d_x), b, c, h, w, _stream()), "dwconv3x3_s2_bwd_data")
    return d_x


def dwconv3x3_wgrad(x, d_out, stride: int):
    """d_weight [C,1,3,3]: nine products summed over (batch, output pixels) per channel, in a fixed order (block partials per
    slice of the positions, then the slices in order)."""
    _req(x, "x"), _req(d_out, "d_out")
    b, c, h, w = x.shape
    if tuple(d_out.shape) != (b, c, (h - 1) // stride + 1, (w - 1) // stride + 1):
        raise RuntimeError("dwconv3x3_wgrad: d_out shape does not match x and the stride")
    lib = L.load()
    n = lib.as_dwconv3x3_wgrad_slices(b, c, h, w, stride)
    if n < 1:
        raise RuntimeError("dwconv3x3_wgrad: bad size / stride")
    part = torch.empty((c, n, 9), device=x.device, dtype=torch.float32)
    with _guard(x.device):
        L.check(lib.as_dwconv3x3_wgrad(_p(x), _p(d_out), _p(part), n, b, c, h, w, stride, _stream()), "dwconv3x3_wgrad")
    return (part[:, 0] if n == 1 else part.sum(dim=1)).view(c, 1, 3, 3)


def conv3d_k3(x, wpack, bias=None, stride: int = 1, act: int = L.ACT_NONE, gate=None):
    """act(Conv3d 3x3x3(x, padding 1, stride) + bias) [* gate [B,Cout,Ho,Wo] broadcast over depth: FeatureAtt, submodule.py:328-341];
    wpack = weight.permute(1,2,3,4,0) as [Cin,27,Cout]."""
    _req(x, "x"), _req(wpack, "wpack")
    b, cin, d, h, w = x.shape
    if wpack.dim() != 3 or wpack.shape[0] != cin or wpack.shape[1] != 27:
        raise RuntimeError("conv3d_k3: wpack must be [Cin,27,Cout]")
    cout = wpack.shape[2]
    if bias is not None:
        _req(bias, "bias")
    out = torch.empty((b, cout, (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1), device=x.device,
                      dtype=torch.float32)
    if gate is not None:
        _req(gate, "gate")
        if tuple(gate.shape) != (b, cout, out.shape[3], out.shape[4]):
            raise RuntimeError(f"conv3d_k3: gate must be [B,Cout,Ho,Wo] = {(b, cout, out.shape[3], out.shape[4])}, got {tuple(gate.shape)}")
    with _guard(x.device):
        L.check(L.load().as_conv3d_k3_gated(_p(x), _p(wpack), _p(bias), _p(gate), _p(out), b, cin, cout, d, h, w, stride, act, _stream()),
                "conv3d_k3")
    return out


def deconv3d_k4s2(x, wpack, bias=None, act: int = L.ACT_NONE):
    """act(ConvTranspose3d(k=4, s=2, p=1)(x) + bias); wpack = weight.permute(0,2,3,4,1) as [Cin,4,4,4,Cout]."""
    _req(x, "x"), _req(wpack, "wpack")
    b, cin, d, h, w = x.shape
    if wpack.dim() != 5 or wpack.shape[0] != cin or tuple(wpack.shape[1:4]) != (4, 4, 4):
        raise RuntimeError("deconv3d_k4s2: wpack must be [Cin,4,4,4,Cout]")
    cout = wpack.shape[4]
    if bias is not None:
        _req(bias, "bias")
    out = torch.empty((b, cout, 2 * d, 2 * h, 2 * w), device=x.device, dtype=torch.float32)
    with _guard(x.device):
        L.check(L.load().as_deconv3d_k4s2(_p(x), _p(wpack), _p(bias), _p(out), b, cin, cout, d, h, w, act, _stream()),
                "deconv3d_k4s2")
    return out


def instance_norm_act(x, eps: float = 1e-5, act: int = L.ACT_NONE, residual=None):
    """act(InstanceNorm(x)) over the trailing spatial dims of x [B,C,*], affine = False (BasicConv_IN tail); with
    `residual` (same shape): relu(residual + act(InstanceNorm(x))), the tail of an InstanceNorm ResidualBlock."""
    _req(x, "x")
    if residual is not None:
        _req(residual, "residual")
        if residual.shape != x.shape:
            raise RuntimeError("instance_norm_act: residual shape mismatch")
    b, c = x.shape[:2]
    hw = x[0, 0].numel()
    out = torch.empty_like(x)
    ws = torch.empty(L.load().as_instance_norm_ws_bytes(b * c) // 8, device=x.device, dtype=torch.float64)
    with _guard(x.device):
        L.check(L.load().as_instance_norm_act(_p(x), _p(residual), _p(out), _p(ws), b * c, hw, eps, act, _stream()), "instance_norm_act")
    return out


def layernorm2d_act(x, weight, bias, eps: float = 1e-6, act: int = L.ACT_NONE):
    """act(LayerNorm2d(x)): per-pixel normalisation over channels of NCHW x with affine weight / bias."""
    _req(x, "x"), _req(weight, "weight"), _req(bias, "bias")
    b, c, h, w = x.shape
    out = torch.empty_like(x)
    with _guard(x.device):
        L.check(L.load().as_layernorm2d_act(_p(x), _p(weight), _p(bias), _p(out), b, c, h, w, eps, act, _stream()), "layernorm2d_act")
    return out


class FoldedConv:
    """Cache of the (BatchNorm-folded) weight / bias of a conv that runs on a direct kernel or stays on MIOpen;
    `layout` = None keeps the module's layout, 'c3d' gives the [Cin,27,Cout] pack of as_conv3d_k3."""

    def __init__(self, layout=None):
        self._key, self._wb, self.layout, self._refs = None, None, layout, None

    @torch.no_grad()
    def get(self, conv, bn=None):
        ts = [conv.weight, conv.bias] + ([] if bn is None else [bn.weight, bn.bias, bn.running_mean, bn.running_var])
        key = tuple(None if t is None else (t.data_ptr(), t._version, t.device) for t in ts)
        alive = self._refs is not None and all((r is None) == (t is None) and (r is None or r() is t) for r, t in zip(self._refs, ts))
        if key != self._key or not alive:
            self._refs = [None if t is None else weakref.ref(t) for t in ts]
            if bn is None:
                w, b = conv.weight.detach().float(), None if conv.bias is None else conv.bias.detach().float().contiguous()
            else:
                w, b = fold_bn(conv, bn, out_dim=1 if self.layout == "d3d" else 0)
            if self.layout == "c3d":
                w = w.permute(1, 2, 3, 4, 0).reshape(w.shape[1], 27, w.shape[0])
            elif self.layout == "c2d":  # [Cout,Cin,3,3] -> [Cin,9,Cout] (as_conv3x3_few)
                w = w.permute(1, 2, 3, 0).reshape(w.shape[1], 9, w.shape[0])
            elif self.layout == "d3d":  # ConvTranspose3d weight [Cin,Cout,4,4,4] -> [Cin,4,4,4,Cout]
                w = w.permute(0, 2, 3, 4, 1)
            self._wb, self._key = (w.contiguous(), b), key
        return self._wb


def interp(x, ho: int, wo: int):
    """F.interpolate(x, (ho, wo), mode='bilinear', align_corners=True) (update.py:100-102)."""
    _req(x, "x")
    b, c, h, w = x.shape
    out = torch.empty((b, c, ho, wo), device=x.device, dtype=torch.float32)
    with _guard(x.device):
        L.check(L.load().as_interp_bilinear_ac(_p(x), _p(out), b, c, h, w, ho, wo, _stream()), "interp_bilinear_ac")
    return out


def conv2d_wgrad(x, dy, ks: int, want_bias: bool = True):
    """(dW [Cout,Cin,ks,ks], db [Cout] | None) of a stride-1 "same" convolution from its input x [B,Cin,H,W] and output gradient
    dy [B,Cout,H,W] (as_conv2d_wgrad: bf16 hi/lo split MFMA, deterministic split-K) — update.py:16-92 under autograd.
    x / dy may be equally long LISTS of such tensors (one pair per GRU iteration, same shapes): reduced in one launch without a
    stacking copy (as_conv2d_wgrad_multi, up to 32 pairs)."""
    xs, dys = (list(x), list(dy)) if isinstance(x, (list, tuple)) else ([x], [dy])
    if len(xs) != len(dys) or not xs:
        raise RuntimeError("conv2d_wgrad: x and dy must be lists of the same length")
    if len(xs) > 32:
        xs, dys = [torch.cat(xs, 0)], [torch.cat(dys, 0)]
    for t in xs + dys:
        _req(t, "x / dy")
    per, cin, h, w = xs[0].shape
    cout = dys[0].shape[1]
    if any(tuple(t.shape) != (per, cin, h, w) for t in xs) or any(tuple(t.shape) != (per, cout, h, w) for t in dys):
        raise RuntimeError(f"conv2d_wgrad: x {[tuple(t.shape) for t in xs][:2]} and dy {[tuple(t.shape) for t in dys][:2]} do not match")
    lib = L.load()
    dev = xs[0].device
    nbytes = int(lib.as_conv2d_wgrad_ws_bytes(per * len(xs), cin, cout, h, w, ks))
    if nbytes < 0:
        raise RuntimeError(f"conv2d_wgrad: unsupported problem (ks={ks}, x {tuple(xs[0].shape)})")
    ws = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
    dw = torch.empty((cout, cin, ks, ks), device=dev, dtype=torch.float32)
    db = torch.empty((cout,), device=dev, dtype=torch.float32) if want_bias else None
    xp, k1 = L.ptr_array([t.data_ptr() for t in xs])
    dp, k2 = L.ptr_array([t.data_ptr() for t in dys])
    with _guard(dev):
        L.check(lib.as_conv2d_wgrad_multi(xp, dp, len(xs), per, _p(dw), _p(db) if db is not None else None, cin, cout, h, w, ks, _p(ws),
                                          nbytes, _stream()), "conv2d_wgrad")
    return dw, db


# ------------------------------------------------------------------------------------------------
# LIIF upsampler stages
# ------------------------------------------------------------------------------------------------


def structure_feature(x: torch.Tensor) -> torch.Tensor:
    """cat(x, cosine affinity to the 8 neighbours) -> [B,C+8,H,W] (liif.py:432-446, :496-499)."""
    _req(x, "x")
    b, c, h, w = x.shape
    out = torch.empty((b, c + 8, h, w), device=x.device, dtype=torch.float32)
    ws = torch.empty((b, h, w), device=x.device, dtype=torch.float32)
    with _guard(x.device):
        L.check(L.load().as_structure_feature(_p(x), _p(out), _p(ws), b, c, h, w, _stream()), "structure_feature")
    return out


def liif_gather(feat: torch.Tensor, coord: torch.Tensor, latent: torch.Tensor, lat_coff: int) -> None:
    """Nearest gather + relative coordinates into channels [lat_coff, lat_coff+C+2) of latent [B,Ctot,Q]
    (liif.py:108-137)."""
    _req(feat, "feat"), _req(coord, "coord"), _req(latent, "latent")
    b, c, h, w = feat.shape
    q = coord.shape[1]
    if tuple(coord.shape) != (b, q, 2) or latent.shape[0] != b or latent.shape[2] != q:
        raise RuntimeError("liif_gather: coord must be [B,Q,2] and latent [B,Ctot,Q]")
    with _guard(feat.device):
        L.check(L.load().as_liif_gather(_p(feat), _p(coord), _p(latent), b, c, h, w, q, latent.shape[1], lat_coff, _stream()),
                "liif_gather")


def liif_gather_mlp1(u0, u1, coord, wrel, bias):
    """relu(u0[nearest0(q)] + u1[nearest1(q)] + wrel·rel(q) + bias) -> [B,C,Q]: gather + first MLP layer (see the header)."""
    _req(u0, "u0"), _req(coord, "coord"), _req(wrel, "wrel")
    b, c, h0, w0 = u0.shape
    q = coord.shape[1]
    h1 = w1 = 0
    if u1 is not None:
        _req(u1, "u1")
        if u1.shape[0] != b or u1.shape[1] != c:
            raise RuntimeError("liif_gather_mlp1: u1 must be [B,C,H1,W1]")
        h1, w1 = u1.shape[2], u1.shape[3]
    if tuple(coord.shape) != (b, q, 2) or tuple(wrel.shape) != (c, 2 * (1 if u1 is None else 2)):
        raise RuntimeError("liif_gather_mlp1: coord must be [B,Q,2] and wrel [C,2*n_src]")
    if bias is not None:
        _req(bias, "bias")
    out = torch.empty((b, c, q), device=u0.device, dtype=torch.float32)
    with _guard(u0.device):
        L.check(L.load().as_liif_gather_mlp1(_p(u0), _p(u1), _p(coord), _p(wrel), _p(bias), _p(out), b, c, h0, w0, h1, w1, q,
                                             _stream()), "liif_gather_mlp1")
    return out


def liif_rel_key(coord, sizes, want_rel=True, want_key=False):
    """Relative coordinates [B, 2*n_src, Q] of the queries w.r.t. the nearest cell of each source (liif.py:127-129) and/or the
    int32 sort key [B,Q] of the training path.  sizes = [(H0,W0)] or [(H0,W0),(H1,W1)]."""
    _req(coord, "coord")
    b, q = coord.shape[:2]
    n = len(sizes)
    if n not in (1, 2) or tuple(coord.shape) != (b, q, 2):
        raise RuntimeError("liif_rel_key: coord must be [B,Q,2] and 1 or 2 sources")
    rel = torch.empty((b, 2 * n, q), device=coord.device, dtype=torch.float32) if want_rel else None
    key = torch.empty((b, q), device=coord.device, dtype=torch.int32) if want_key else None
    (h0, w0), (h1, w1) = sizes[0], (sizes[1] if n > 1 else (0, 0))
    with _guard(coord.device):
        L.check(L.load().as_liif_rel_key(_p(coord), _p(rel), _p(key), b, q, n, h0, w0, h1, w1, _stream()), "liif_rel_key")
    return rel, key


# Deterministic training (ANYSTEREO_DETERMINISTIC=1 / set_deterministic(True)): the three places where gradients are summed with
# float atomics — the upsampler's two scatter-adds onto the low-resolution maps and the convex upsampling's scatter onto the
# disparity — run in a gather form with a fixed summation order instead (as_liif_gather_bwd_det over the queries sorted once
# per forward), and the fused first-layer backward (in-kernel atomics) is not used.  With torch.backends.cudnn.deterministic = True
# for the library layers the parameter gradients of a training step are then the same bits on every run.
_DETERMINISTIC = [os.environ.get("ANYSTEREO_DETERMINISTIC", "0") == "1"]
_SEGMENTS = {}


def set_deterministic(on: bool) -> None:
    _DETERMINISTIC[0] = bool(on)
    _SEGMENTS.clear()


def get_deterministic() -> bool:
    return _DETERMINISTIC[0]


_NONDET_WARNED = set()


def warn_nondeterministic(what: str) -> None:
    """Deterministic mode covers the default option set's three float-atomic sums; the off-by-default upsampler options
    (`liif_latent_backward`, `ConvexUpsampleQuater.backward`) still scatter with float atomics.  Say so once per path
    instead of handing out non-repeatable gradients silently."""
    if _DETERMINISTIC[0] and what not in _NONDET_WARNED:
        _NONDET_WARNED.add(what)
        import warnings
        warnings.warn(f"anystereo deterministic mode: {what} accumulates with float atomics (not covered by the gather-form "
                      "scatters); its gradients are not bit-repeatable", RuntimeWarning, stacklevel=3)


def query_segments(coord, h: int, w: int):
    """(order int32 [B,Q], starts int32 [B,h*w+1]) of the queries' nearest pixels on an h x w map: `order` sorts each batch
    element's queries by pixel (stable), pixel p's queries sit at sorted positions [starts[p], starts[p+1]).  Cached for the
    coordinate tensor at hand (one entry per map size)."""
    key = (coord.data_ptr(), tuple(coord.shape), coord._version, h, w, coord.device)
    slot = (str(coord.device), h, w)   # one entry per (device, map size): two devices alternating at one size do not evict each other
    ent = _SEGMENTS.get(slot)
    if ent is not None and ent[0] == key and ent[1]() is coord:
        return ent[2], ent[3]
    _, k = liif_rel_key(coord, [(h, w)], want_rel=False, want_key=True)
    pix = (k >> 2).to(torch.int64)
    sp, order = torch.sort(pix, dim=1, stable=True)
    bounds = torch.arange(h * w + 1, device=coord.device, dtype=torch.int64).unsqueeze(0).expand(coord.shape[0], -1).contiguous()
    starts = torch.searchsorted(sp.contiguous(), bounds).to(torch.int32).contiguous()
    order = order.to(torch.int32).contiguous()
    if len(_SEGMENTS) > 32:
        _SEGMENTS.clear()
    _SEGMENTS[slot] = (key, weakref.ref(coord), order, starts)
    return order, starts


def liif_scatter_add(d_rows, coord, c, h, w, coff=0):
    """Transpose of the nearest gather: d_rows [B,Ctot,Q] channels [coff, coff+c) summed into [B,c,h,w]."""
    _req(d_rows, "d_rows"), _req(coord, "coord")
    b, ctot, q = d_rows.shape
    if _DETERMINISTIC[0]:
        order, starts = query_segments(coord, h, w)
        out = torch.empty((b, c, h, w), device=d_rows.device, dtype=torch.float32)
        with _guard(d_rows.device):
            L.check(L.load().as_liif_gather_bwd_det(_p(d_rows), _p(order), _p(starts), _p(out), b, c, h * w, q, ctot, coff, _stream()),
                    "liif_gather_bwd_det")
        return out
    out = torch.empty((b, c, h, w), device=d_rows.device, dtype=torch.float32)
    with _guard(d_rows.device):
        L.check(L.load().as_liif_gather_bwd(_p(d_rows), _p(coord), _p(out), b, c, h, w, q, ctot, coff, _stream()), "liif_gather_bwd")
    return out


def convex_upsample(disp, mask, coord, scale=None, mask_is_logits=False):
    """(softmax(mask) ·) disp[3x3 nbr of the nearest low-res pixel] -> [B,1,Q]; with `scale` [B] the
    disparity is multiplied by 4*scale_b on the fly
    (continuous_IGEVstereo.py:204,212-214; submodule.py:357-372)."""
    _req(disp, "disp"), _req(mask, "mask"), _req(coord, "coord")
    b, one, h, w = disp.shape
    q = coord.shape[1]
    if one != 1 or tuple(mask.shape) != (b, 9, q) or tuple(coord.shape) != (b, q, 2):
        raise RuntimeError("convex_upsample: shape mismatch")
    if scale is not None:
        _req(scale, "scale")
        if scale.numel() != b:
            raise RuntimeError("convex_upsample: scale must hold one value per batch element")
    out = torch.empty((b, 1, q), device=disp.device, dtype=torch.float32)
    with _guard(disp.device):
        L.check(L.load().as_convex_upsample(_p(disp), _p(scale), _p(mask), _p(coord), _p(out), b, h, w, q,
                                            1 if mask_is_logits else 0, _stream()), "convex_upsample")
    return out


# ---- §8 f4: off-by-default upsampler options (csrc/liif_variants.hip) ---------------------------------------------------
def liif_latent_width(c, unfold9=False, n_samp=1, n_enc=0, cell=False):
    """Channels one source occupies in the MLP input (see as_liif_latent in the header)."""
    return (9 * c if unfold9 else c) * n_samp + 2 + 2 * n_enc + (2 if cell else 0)


def liif_latent(feat, coord, latent, lat_coff, unfold9=False, n_samp=1, emb=None, cell=None) -> None:
    """One source's block [features | rel | sin/cos encoding | cell] of the MLP input, channel-major (liif.py:652-676 with the
    options of :108-176, :339-370)."""
    _req(feat, "feat"), _req(coord, "coord"), _req(latent, "latent")
    b, c, h, w = feat.shape
    q = coord.shape[1]
    if tuple(coord.shape) != (b, q, 2) or latent.shape[0] != b or latent.shape[2] != q:
        raise RuntimeError("liif_latent: coord must be [B,Q,2] and latent [B,Ctot,Q]")
    n_enc = 0
    if emb is not None:
        _req(emb, "emb")
        if emb.dim() != 2 or emb.shape[1] != 2:
            raise RuntimeError("liif_latent: emb must be [n,2]")
        n_enc = emb.shape[0]
    if cell is not None:
        _req(cell, "cell")
        if tuple(cell.shape) != (b, q, 2):
            raise RuntimeError("liif_latent: cell must be [B,Q,2]")
    with _guard(feat.device):
        L.check(L.load().as_liif_latent(_p(feat), _p(coord), _p(emb), _p(cell), _p(latent), b, c, h, w, q, latent.shape[1], lat_coff,
                                        1 if unfold9 else 0, n_samp, n_enc, _stream()), "liif_latent")


def liif_latent_backward(d_latent, coord, lat_coff, c, h, w, unfold9=False, n_samp=1):
    """d_feat [B,C,H,W]: scatter-add of the feature channels of one source's block."""
    _req(d_latent, "d_latent"), _req(coord, "coord")
    b, ctot, q = d_latent.shape
    d_feat = torch.empty((b, c, h, w), device=d_latent.device, dtype=torch.float32)
    warn_nondeterministic("liif_latent_backward (upsampler option sets other than the default)")
    with _guard(d_latent.device):
        L.check(L.load().as_liif_latent_bwd(_p(d_latent), _p(coord), _p(d_feat), b, c, h, w, q, ctot, lat_coff, 1 if unfold9 else 0,
                                            n_samp, _stream()), "liif_latent_bwd")
    return d_feat


def convex_upsample_quater(disp, mask, coord, scale=None, mask_is_logits=False):
    """Four-sample convex combination -> [B,1,Q] (context_upsample_multiscale_train_quaterp, submodule.py:375-399)."""
    _req(disp, "disp"), _req(mask, "mask"), _req(coord, "coord")
    b, one, h, w = disp.shape
    q = coord.shape[1]
    if one != 1 or tuple(mask.shape) != (b, 4, q) or tuple(coord.shape) != (b, q, 2):
        raise RuntimeError("convex_upsample_quater: disp [B,1,h,w], mask [B,4,Q], coord [B,Q,2] expected")
    if scale is not None:
        _req(scale, "scale")
        if scale.numel() != b:
            raise RuntimeError("convex_upsample_quater: scale must hold one value per batch element")
    out = torch.empty((b, 1, q), device=disp.device, dtype=torch.float32)
    with _guard(disp.device):
        L.check(L.load().as_convex_upsample_quater(_p(disp), _p(scale), _p(mask), _p(coord), _p(out), b, h, w, q,
                                                   1 if mask_is_logits else 0, _stream()), "convex_upsample_quater")
    return out


def affinity_backward(x, sf_out, d_out, with_x: bool):
    """Gradient w.r.t. x of cat(x, affinity(x)) (with_x) or of affinity(x) alone, the affinity taken on the LIVE map.
    sf_out / d_out: the forward output and its gradient ([B,C+8,H,W] or [B,8,H,W])."""
    _req(x, "x"), _req(sf_out, "sf_out"), _req(d_out, "d_out")
    b, c, h, w = x.shape
    ctot = c + 8 if with_x else 8
    if tuple(sf_out.shape) != (b, ctot, h, w) or tuple(d_out.shape) != (b, ctot, h, w):
        raise RuntimeError("affinity_backward: forward output / gradient shape mismatch")
    plane = h * w
    skip = c * plane * 4 if with_x else 0
    dx = torch.empty_like(x)
    ws = torch.empty((b, h, w), device=x.device, dtype=torch.float32)
    with _guard(x.device):
        L.check(L.load().as_affinity_bwd(_p(x), sf_out.data_ptr() + skip, ctot * plane, d_out.data_ptr() + skip, ctot * plane,
                                         _p(d_out) if with_x else None, ctot * plane, _p(dx), _p(ws), b, c, h, w, _stream()),
                "affinity_bwd")
    return dx


# ------------------------------------------------------------------------------------------------
# LIIF upsampler, fused inference pipeline (csrc/liif_fused.hip)
# ------------------------------------------------------------------------------------------------


def _src_arrays(srcs):
    ptrs, keep = L.ptr_array([t.data_ptr() for t in srcs])
    ch = (C.c_int * len(srcs))(*[int(t.shape[1]) for t in srcs])
    return ptrs, ch, keep


def liif_affinity(srcs: Sequence[torch.Tensor]) -> torch.Tensor:
    """AffinityFeature(cat(srcs, dim=1)) -> [B,8,H,W] without materialising the concat or copying it (liif.py:432-446)."""
    for i, t in enumerate(srcs):
        _req(t, f"src[{i}]")
    b, _, h, w = srcs[0].shape
    if any(t.shape[0] != b or tuple(t.shape[2:]) != (h, w) for t in srcs) or not 1 <= len(srcs) <= 3:
        raise RuntimeError("liif_affinity: 1..3 sources of one [B,*,H,W] shape")
    aff = torch.empty((b, 8, h, w), device=srcs[0].device, dtype=torch.float32)
    ctot = sum(int(t.shape[1]) for t in srcs)
    ws = torch.empty(L.load().as_liif_affinity_ws_bytes(b, h, w, (ctot + 7) // 8) // 4, device=srcs[0].device, dtype=torch.float32)
    ptrs, ch, keep = _src_arrays(srcs)
    with _guard(aff.device):
        L.check(L.load().as_liif_affinity(ptrs, ch, len(srcs), _p(aff), _p(ws), b, h, w, _stream()), "liif_affinity")
    return aff


class LiifLowresPack:
    """Split-fp16 MFMA fragments of a column block of the first Linear weight (as_liif_lowres_pack), rebuilt when the weight
    changes (identity + version counter through a weak reference)."""

    def __init__(self):
        self._key, self._ref, self.image = None, None, None

    def get(self, weight: torch.Tensor, koff: int, k: int):
        key = (weight.data_ptr(), weight._version, weight.device, koff, k)
        if key != self._key or self._ref is None or self._ref() is not weight:
            w = weight.detach()
            w = w if (w.dtype == torch.float32 and w.is_contiguous()) else w.float().contiguous()
            if w.dim() != 2 or w.shape[0] != 128:
                raise RuntimeError("LiifLowresPack: weight must be [128, in_dim]")
            image = torch.empty(L.load().as_liif_lowres_pack_bytes(k), device=w.device, dtype=torch.uint8)
            with _guard(w.device):
                L.check(L.load().as_liif_lowres_pack(_p(w), w.shape[1], koff, k, _p(image), _stream()), "liif_lowres_pack")
            self.image, self._key, self._ref = image, key, weakref.ref(weight)
        return self


def liif_lowres_cl(srcs: Sequence[torch.Tensor], pack: LiifLowresPack) -> torch.Tensor:
    """out[b, y*W+x, :] = W1[:, koff:koff+K] @ cat(srcs)[b, :, y, x] -> [B, H*W, 128] channels-last (first MLP layer at low
    resolution, liif.py:9-25 applied before the nearest gather of liif.py:108-137); pack = LiifLowresPack.get(W1, koff, K)."""
    for i, t in enumerate(srcs):
        _req(t, f"src[{i}]")
    b, _, h, w = srcs[0].shape
    if sum(int(t.shape[1]) for t in srcs) != pack._key[4]:
        raise RuntimeError("liif_lowres_cl: sources do not hold the packed column count")
    out = torch.empty((b, h * w, 128), device=srcs[0].device, dtype=torch.float32)
    ptrs, ch, keep = _src_arrays(srcs)
    with _guard(out.device):
        L.check(L.load().as_liif_lowres_cl(ptrs, ch, len(srcs), _p(pack.image), _p(out), b, h, w, _stream()), "liif_lowres_cl")
    return out


def liif_rows_cl(srcs: Sequence[torch.Tensor]) -> torch.Tensor:
    """Channels-last copy of a small input's structure feature: [B, H*W, 48] = cat(srcs)[b, :, y, x], zero padded — the rows
    the tail gathers when it takes that input's first-layer product per query (as_liif_tail_direct)."""
    for i, t in enumerate(srcs):
        _req(t, f"src[{i}]")
    b, _, h, w = srcs[0].shape
    pitch = int(L.load().as_liif_rows_pitch())
    if sum(int(t.shape[1]) for t in srcs) > pitch:
        raise RuntimeError(f"liif_rows_cl: more than {pitch} channels")
    out = torch.empty((b, h * w, pitch), device=srcs[0].device, dtype=torch.float32)
    ptrs, ch, keep = _src_arrays(srcs)
    with _guard(out.device):
        L.check(L.load().as_liif_rows_cl(ptrs, ch, len(srcs), _p(out), b, h, w, _stream()), "liif_rows_cl")
    return out


class LiifTailPack:
    """LDS weight image of the fused tail kernel (layers 2-4 + the relative-coordinate columns / bias of layer 1), rebuilt
    when a source tensor changes (identity + version counter; weak references, so a recycled address cannot alias)."""

    def __init__(self):
        self._key = None
        self._refs = None
        self.image = None
        self.wrel = None

    def get(self, lin, rel_cols):
        """lin: the four nn.Linear of the MLP; rel_cols: column index of (rel_row, rel_col) per source in lin[0].weight."""
        ts = [lin[0].weight, lin[0].bias] + [t for m in lin[1:] for t in (m.weight, m.bias)]
        key = tuple(None if t is None else (t.data_ptr(), t._version, t.device) for t in ts) + (tuple(rel_cols),)
        alive = self._refs is not None and all((r is None) == (t is None) and (r is None or r() is t) for r, t in zip(self._refs, ts))
        if key != self._key or not alive:
            w1 = lin[0].weight.detach()
            wrel = torch.cat([w1[:, o:o + 2] for o in rel_cols], dim=1).float().contiguous()
            f = lambda t: None if t is None else t.detach().float().contiguous()  # noqa: E731
            w2, w3, w4 = (f(m.weight) for m in lin[1:])
            if tuple(w2.shape) != (64, 128) or tuple(w3.shape) != (64, 64) or tuple(w4.shape) != (9, 64) or w1.shape[0] != 128:
                raise RuntimeError("LiifTailPack: the fused tail is built for the default MLP 128-64-64-9")
            b1, b2, b3, b4 = f(lin[0].bias), f(lin[1].bias), f(lin[2].bias), f(lin[3].bias)
            image = torch.empty(L.load().as_liif_tail_image_bytes(), device=w1.device, dtype=torch.uint8)
            with _guard(w1.device):
                L.check(L.load().as_liif_tail_pack(_p(wrel), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(w4), _p(b4),
                                                   len(rel_cols), _p(image), _stream()), "liif_tail_pack")
            self.image, self.wrel, self._key = image, wrel, key
            self._refs = [None if t is None else weakref.ref(t) for t in ts]
        return self


PATCH_ORDER = os.environ.get("ANYSTEREO_LIIF_PATCH_ORDER", "1") != "0"  # A/B: 0 = 32-query runs of one row (round-5 order)


def liif_query_rows(coord: torch.Tensor) -> torch.Tensor:
    """Device int32 [1]: the row length of a raster-ordered query grid [B,Q,2] (0: none) — as_liif_query_rows; no host sync."""
    _req(coord, "coord")
    out = torch.empty(1, device=coord.device, dtype=torch.int32)
    with _guard(coord.device):
        L.check(L.load().as_liif_query_rows(_p(coord), coord.shape[0], coord.shape[1], _p(out), _stream()), "liif_query_rows")
    return out


def liif_tail(u0, u1, sizes, coord, pack: LiifTailPack, disp, scale=None, clamp_inplace=True, want_logits=False,
              direct1: Optional[LiifLowresPack] = None, row_len: Optional[torch.Tensor] = None):
    """The per-query tail: gather + first-layer finish + MLP + softmax + convex upsampling -> [B,1,Q] (and the mask logits
    [B,9,Q] when asked).  u0 / u1: liif_lowres_cl results; sizes = [(H0,W0)] or [(H0,W0),(H1,W1)].
    direct1 = the LiifLowresPack of the second input's columns: u1 then is liif_rows_cl's [B, H1*W1, 48] raw rows and the tail
    takes that input's first-layer product per query (as_liif_tail_direct)."""
    _req(u0, "u0"), _req(coord, "coord"), _req(disp, "disp")
    b, q = coord.shape[:2]
    (h0, w0), (h1, w1) = sizes[0], (sizes[1] if u1 is not None else (0, 0))
    if tuple(coord.shape) != (b, q, 2) or tuple(u0.shape) != (b, h0 * w0, 128) or disp.shape[0] != b or disp.shape[1] != 1:
        raise RuntimeError("liif_tail: shape mismatch")
    if u1 is not None:
        _req(u1, "u1")
        if tuple(u1.shape) != (b, h1 * w1, 128 if direct1 is None else int(L.load().as_liif_rows_pitch())):
            raise RuntimeError("liif_tail: u1 shape mismatch")
    elif direct1 is not None:
        raise RuntimeError("liif_tail: direct1 without a second input")
    if scale is not None:
        _req(scale, "scale")
        if scale.numel() != b:
            raise RuntimeError("liif_tail: scale must hold one value per batch element")
    out = torch.empty((b, 1, q), device=coord.device, dtype=torch.float32)
    logits = torch.empty((b, 9, q), device=coord.device, dtype=torch.float32) if want_logits else None
    if row_len is None and PATCH_ORDER:
        row_len = liif_query_rows(coord)  # order hint only (reads the coordinates before the tail clamps them in place)
    if row_len is not None and (row_len.dtype != torch.int32 or not row_len.is_cuda):
        raise RuntimeError("liif_tail: row_len must be a CUDA int32 tensor")
    with _guard(coord.device):
        if direct1 is not None:
            L.check(L.load().as_liif_tail_direct(_p(u0), _p(u1), _p(coord), _p(pack.image), _p(direct1.image), _p(disp), _p(scale),
                                                 _p(out), _p(logits), b, q, h0, w0, h1, w1, disp.shape[2], disp.shape[3],
                                                 1 if clamp_inplace else 0, _p(row_len), _stream()), "liif_tail_direct")
        else:
            L.check(L.load().as_liif_tail(_p(u0), _p(u1), _p(coord), _p(pack.image), _p(disp), _p(scale), _p(out), _p(logits), b, q,
                                          h0, w0, h1, w1, disp.shape[2], disp.shape[3], 1 if clamp_inplace else 0, _p(row_len), _stream()),
                    "liif_tail")
    return (out, logits) if want_logits else out


def conv7x7_c1_wgrad(xs, dys, want_bias: bool = True):
    """(dW [Cout,1,7,7], db [Cout] | None) of the motion encoder's 7x7 convolution of the one-channel disparity map from its inputs
    x [B,1,H,W] and output gradients dy [B,Cout,H,W] — lists of equally shaped pairs (one per GRU iteration) are reduced in one
    launch (as_conv7x7_c1_wgrad_multi; update.py:81,87)."""
    xs, dys = (list(xs), list(dys)) if isinstance(xs, (list, tuple)) else ([xs], [dys])
    if len(xs) != len(dys) or not xs:
        raise RuntimeError("conv7x7_c1_wgrad: x and dy must be lists of the same length")
    if len(xs) > 32:
        xs, dys = [torch.cat(xs, 0)], [torch.cat(dys, 0)]
    for t in xs + dys:
        _req(t, "x / dy")
    per, one, h, w = xs[0].shape
    cout = dys[0].shape[1]
    if one != 1 or any(tuple(t.shape) != (per, 1, h, w) for t in xs) or any(tuple(t.shape) != (per, cout, h, w) for t in dys):
        raise RuntimeError("conv7x7_c1_wgrad: expects x [B,1,H,W] and dy [B,Cout,H,W] pairs of one shape")
    lib = L.load()
    dev = xs[0].device
    nbytes = int(lib.as_conv7x7_c1_wgrad_ws_bytes(per * len(xs), cout, h, w))
    if nbytes < 0:
        raise RuntimeError(f"conv7x7_c1_wgrad: unsupported problem (Cout={cout})")
    ws = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
    dw = torch.empty((cout, 1, 7, 7), device=dev, dtype=torch.float32)
    db = torch.empty((cout,), device=dev, dtype=torch.float32) if want_bias else None
    xp, k1 = L.ptr_array([t.data_ptr() for t in xs])
    dp, k2 = L.ptr_array([t.data_ptr() for t in dys])
    with _guard(dev):
        L.check(lib.as_conv7x7_c1_wgrad_multi(xp, dp, len(xs), per, _p(dw), _p(db), cout, h, w, _p(ws), nbytes, _stream()), "conv7x7_c1_wgrad")
    return dw, db


class LiifMlpBwdPack:
    """Fragments of the TRANSPOSED layers 2-4 of the MLP for as_liif_mlp_bwd, rebuilt when a weight changes."""

    def __init__(self):
        self._key = None
        self._refs = None
        self.image = None

    def get(self, w2, w3, w4):
        ts = [w2, w3, w4]
        key = tuple((t.data_ptr(), t._version, t.device) for t in ts)
        alive = self._refs is not None and all(r() is t for r, t in zip(self._refs, ts))
        if key != self._key or not alive:
            if tuple(w2.shape) != (64, 128) or tuple(w3.shape) != (64, 64) or tuple(w4.shape) != (9, 64):
                raise RuntimeError("LiifMlpBwdPack: built for the default MLP 128-64-64-9")
            f = lambda t: t.detach().float().contiguous()  # noqa: E731
            a, b_, c = f(w2), f(w3), f(w4)
            image = torch.empty(L.load().as_liif_mlp_bwd_image_bytes(), device=w2.device, dtype=torch.uint8)
            with _guard(w2.device):
                L.check(L.load().as_liif_mlp_bwd_pack(_p(a), _p(b_), _p(c), _p(image), _stream()), "liif_mlp_bwd_pack")
            self.image, self._key = image, key
            self._refs = [weakref.ref(t) for t in ts]
        return self


def liif_mlp_fwd(u0, u1, sizes, coord, pack: LiifTailPack):
    """Mask logits [B,9,Q] of the per-query MLP from the channels-last first-layer rows (training forward: no activation is kept,
    the coordinates are not clamped).  u0 [B, H0*W0, 128]; u1 [B1, H1*W1, 128] with B % B1 == 0 (query batch b reads b % B1)."""
    _req(u0, "u0"), _req(u1, "u1"), _req(coord, "coord")
    b, q = coord.shape[:2]
    (h0, w0), (h1, w1) = sizes
    if tuple(coord.shape) != (b, q, 2) or tuple(u0.shape) != (b, h0 * w0, 128) or tuple(u1.shape[1:]) != (h1 * w1, 128) or b % u1.shape[0]:
        raise RuntimeError("liif_mlp_fwd: shape mismatch")
    logits = torch.empty((b, 9, q), device=coord.device, dtype=torch.float32)
    with _guard(coord.device):
        L.check(L.load().as_liif_mlp_fwd(_p(u0), _p(u1), _p(coord), _p(pack.image), _p(logits), b, u1.shape[0], q, h0, w0, h1, w1,
                                         _stream()), "liif_mlp_fwd")
    return logits


def liif_mlp_bwd(u0, u1, sizes, coord, pack: LiifTailPack, pack_t: LiifMlpBwdPack, d_logits, fuse_first: bool = False):
    """-> (h1 [B,128,Q], h2, h3 [B,64,Q], d3, d2 [B,64,Q], d1 [B,128,Q]): the MLP's post-ReLU activations (recomputed) and the
    gradients w.r.t. the pre-activations of layers 3, 2, 1 for d_logits [B,9,Q] (as_liif_mlp_bwd).
    fuse_first: d1 is consumed inside the kernel instead of written — the last element of the result then is the tuple
    (d_u0 [B,128,H0,W0], d_u1 [B1,128,H1,W1], d_wrel [128,4]) (scatter-adds of d1 into the two first-layer maps, its reduction against
    the relative coordinates)."""
    _req(u0, "u0"), _req(u1, "u1"), _req(coord, "coord"), _req(d_logits, "d_logits")
    b, q = coord.shape[:2]
    (h0, w0), (h1, w1) = sizes
    if (tuple(coord.shape) != (b, q, 2) or tuple(u0.shape) != (b, h0 * w0, 128) or tuple(u1.shape[1:]) != (h1 * w1, 128) or b % u1.shape[0]
            or tuple(d_logits.shape) != (b, 9, q)):
        raise RuntimeError("liif_mlp_bwd: shape mismatch")
    mk = lambda c: torch.empty((b, c, q), device=coord.device, dtype=torch.float32)  # noqa: E731
    h1_, h2_, h3_, d3, d2 = mk(128), mk(64), mk(64), mk(64), mk(64)
    d1 = du0 = du1 = dwrel = None
    if fuse_first:
        du0 = torch.empty((b, 128, h0, w0), device=coord.device, dtype=torch.float32)
        du1 = torch.empty((u1.shape[0], 128, h1, w1), device=coord.device, dtype=torch.float32)
        dwrel = torch.empty((128, 4), device=coord.device, dtype=torch.float32)
    else:
        d1 = mk(128)
    with _guard(coord.device):
        L.check(L.load().as_liif_mlp_bwd(_p(u0), _p(u1), _p(coord), _p(pack.image), _p(pack_t.image), _p(d_logits), _p(h1_), _p(h2_), _p(h3_),
                                         _p(d3), _p(d2), _p(d1), _p(du0), _p(du1), _p(dwrel), b, u1.shape[0], q, h0, w0, h1, w1, _stream()),
                "liif_mlp_bwd")
    return h1_, h2_, h3_, d3, d2, (du0, du1, dwrel) if fuse_first else d1


def graph_replace_memsets(graph: "torch.cuda.CUDAGraph"):
    """Memset nodes of a captured (keep_graph=True, not yet instantiated) CUDAGraph -> fill kernel nodes (as_graph_replace_memsets).
    Returns (replaced, left)."""
    a, b = C.c_int(0), C.c_int(0)
    L.check(L.load().as_graph_replace_memsets(C.c_void_p(int(graph.raw_cuda_graph())), C.byref(a), C.byref(b)), "graph_replace_memsets")
    return a.value, b.value


class Stamps:
    """Timeline markers of a forward pass (as_stamp): named slots of a device buffer that receive the device wall clock
    (100 MHz) when the issuing stream reaches the marker.  `model.stamps = Stamps(device)` switches the markers of
    models/base.py on (they become kernel nodes of the captured forward); `read()` -> {name: microseconds since the first
    marker}.  Measurement only; the default forward issues none."""
    TICK_US = 0.01  # wall_clock64 on gfx950: constant 100 MHz

    def __init__(self, device, slots: int = 128):
        self.buf = torch.zeros(slots, dtype=torch.int64, device=device)
        self.names = {}
        self.stages = False  # markers after every stage of the feature trunk / cost aggregation (pre-loop diagnosis)
        self.fine = False   # operator-level markers (nn/update.py) on: set by the loop for the iterations it wants resolved
        self.prefix = ""    # prepended to operator-level marker names (the iteration they were issued in)

    def mark(self, name: str, prefixed: bool = False):
        if prefixed:
            name = self.prefix + name
        slot = self.names.setdefault(name, len(self.names))
        if slot >= self.buf.numel():
            raise ValueError("Stamps: out of slots")
        with _guard(self.buf.device):
            L.check(L.load().as_stamp(_p(self.buf), slot, _stream()), "stamp")

    def read(self):
        torch.cuda.synchronize(self.buf.device)
        v = self.buf.cpu().tolist()
        t0 = min(v[s] for s in self.names.values()) if self.names else 0
        return {n: round((v[s] - t0) * self.TICK_US, 2) for n, s in self.names.items()}


_ACTIVE_STAMPS = None


def set_stamps(st: Optional["Stamps"]):
    """Make `st` the process-wide marker sink (None: off).  models/base.py does this for the duration of a forward whose model
    has `stamps` set, so that operator-level code (nn/update.py) can place markers with `ops.mark` without holding the model."""
    global _ACTIVE_STAMPS
    _ACTIVE_STAMPS = st


def mark(name: str):
    if _ACTIVE_STAMPS is not None:
        _ACTIVE_STAMPS.mark(name)


def mark_fine(name: str):
    """Operator-level marker: placed only while the active sink asks for them (`fine`), named with its current prefix."""
    if _ACTIVE_STAMPS is not None and _ACTIVE_STAMPS.fine:
        _ACTIVE_STAMPS.mark(name, prefixed=True)


def split_overflow_count(reset: bool = True) -> int:
    """Split-precision range check: number of waves (since the last reset) in which a kernel met an operand with
    |x| >= 65504 (outside fp16) or NaN.  Such operands are SATURATED to +-65504 — results stay finite but are no longer the
    reference's; switch to `set_precision("fp32")` for such data.  Synchronises the device (diagnostics, tests, bench)."""
    lib, r = L.load(), 1 if reset else 0
    return sum(int(f(r)) for f in (lib.as_liif_split_overflow, lib.as_lookup_split_overflow, lib.as_conv_split_overflow,
                                   lib.as_volumes_split_overflow))
