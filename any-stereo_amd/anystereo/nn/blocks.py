"""Backbone building blocks with the reference's `state_dict` keys
(models/coreContinuous_IGEV/submodule.py:6-252, extractor.py:10-361).  Plain PyTorch modules (MIOpen on the GPU)
for training / autograd; in inference the BatchNorm 3-D convolutions (cost-volume stem and hourglass) run on the
library's direct kernel with BatchNorm folded and LeakyReLU fused (`as_conv3d_k3`).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import grad as G
from .. import ops


# ANYSTEREO_FOLD_POINTWISE=0: BasicConv (2-D BatchNorm, 1x1x1 3-D) and the FeatureAtt gate back on MIOpen + separate norm / activation
_FOLD_POINTWISE = __import__("os").environ.get("ANYSTEREO_FOLD_POINTWISE", "1") != "0"
_CAT_FREE = __import__("os").environ.get("ANYSTEREO_CAT_FREE", "1") != "0"  # 0: materialise the skip concats (A/B)
_NO_SEARCHED_CONV = __import__("os").environ.get("ANYSTEREO_NO_SEARCHED_CONV", "0") == "1"


def fused_ok(x: torch.Tensor, mod: nn.Module) -> bool:
    """Inference fast path: eval mode, CUDA fp32 input, nothing to differentiate."""
    return (not mod.training) and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()


def conv2d_hip_ok(conv: nn.Module) -> bool:
    """Conv2d shapes the implicit-GEMM kernel serves: 1x1 / 3x3 with 'same' padding, no dilation / groups, stride 1 —
    or stride 2 for 3x3 in split precision."""
    if not isinstance(conv, nn.Conv2d):
        return False
    k = conv.kernel_size[0]
    if not (conv.kernel_size in ((1, 1), (3, 3)) and conv.padding == (k // 2, k // 2) and conv.dilation == (1, 1)
            and conv.groups == 1):
        return False
    return conv.stride == (1, 1) or (conv.stride == (2, 2) and k == 3 and ops.get_precision() == "split")


def conv2d_plain(mod: nn.Module, conv: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """conv(x) (no norm, no activation) on the library kernel when it applies, else the module itself (MIOpen)."""
    if conv2d_hip_ok(conv):
        packs = mod.__dict__.setdefault("_hip_packs", {})
        pk = packs.setdefault(id(conv), ops.PackedConv())
        return ops.conv2d([x.contiguous()], pk.get([conv.weight], [conv.bias]), stride=conv.stride[0])
    if (isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and conv.stride[0] == conv.stride[1] and conv.stride[0] > 1):
        # 1x1 with a stride = the 1x1 convolution of the subsampled map
        packs = mod.__dict__.setdefault("_hip_packs", {})
        pk = packs.setdefault(id(conv), ops.PackedConv())
        st = conv.stride[0]
        return ops.conv2d([x[:, :, ::st, ::st].contiguous()], pk.get([conv.weight], [conv.bias]))
    return conv(x)


def _deconv_as_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d(k=4, s=2, p=1) weight [Cin,Cout,4,4] -> the 3x3 convolution weight [4*Cout,Cin,3,3] whose output, read
    through pixel_shuffle(2), is the transposed convolution: out[2y+py, 2x+px] takes, per dimension, exactly two taps of the
    input's 3-neighbourhood — parity 0: offsets (0, -1) with kernel taps (1, 3); parity 1: offsets (+1, 0) with taps (0, 2)."""
    cin, cout = w.shape[:2]
    taps = {0: ((0, 1), (-1, 3)), 1: ((1, 0), (0, 2))}  # parity -> ((input offset, kernel tap), ...)
    w3 = w.new_zeros((cout, 2, 2, cin, 3, 3))
    for py in (0, 1):
        for oy, ky in taps[py]:
            for px in (0, 1):
                for ox, kx in taps[px]:
                    w3[:, py, px, :, oy + 1, ox + 1] = w[:, :, ky, kx].t()
    return w3.reshape(cout * 4, cin, 3, 3)


def deconv2d_k4s2_ok(conv: nn.Module) -> bool:
    return (isinstance(conv, nn.ConvTranspose2d) and conv.kernel_size == (4, 4) and conv.stride == (2, 2) and conv.padding == (1, 1)
            and conv.output_padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and ops.get_precision() == "split")


def deconv2d_k4s2_plain(mod: nn.Module, conv: nn.ConvTranspose2d, x: torch.Tensor) -> torch.Tensor:
    """conv(x) of a 4x4 / stride-2 ConvTranspose2d as ONE 3x3 library convolution to 4*Cout channels + pixel_shuffle (2.25 x
    the FLOPs of the sparse form — these layers sit at 1/8 .. 1/32 resolution — instead of MIOpen's backward-data kernel)."""
    packs = mod.__dict__.setdefault("_hip_packs", {})
    pk = packs.setdefault(id(conv), ops.PackedConv())
    y = ops.conv2d([x.contiguous()], pk.get([conv.weight], [None], transform=_deconv_as_conv3x3))
    return F.pixel_shuffle(y, 2)


def _plain_instance_norm(norm) -> bool:
    return isinstance(norm, nn.InstanceNorm2d) and not norm.affine and not norm.track_running_stats


def conv3d_k3_ok(conv: nn.Module) -> bool:
    return (isinstance(conv, nn.Conv3d) and conv.kernel_size == (3, 3, 3) and conv.padding == (1, 1, 1)
            and conv.stride in ((1, 1, 1), (2, 2, 2)) and conv.dilation == (1, 1, 1) and conv.groups == 1)


def deconv3d_k4s2_ok(conv: nn.Module) -> bool:
    return (isinstance(conv, nn.ConvTranspose3d) and conv.kernel_size == (4, 4, 4) and conv.padding == (1, 1, 1)
            and conv.stride == (2, 2, 2) and conv.dilation == (1, 1, 1) and conv.groups == 1
            and conv.output_padding == (0, 0, 0))


def deconv3d_fused(mod: nn.Module, conv: nn.ConvTranspose3d, bn, x: torch.Tensor, act: int) -> torch.Tensor:
    cache = mod.__dict__.setdefault("_c3d_cache", {})
    fc = cache.setdefault(id(conv), ops.FoldedConv("d3d"))
    w, b = fc.get(conv, bn)
    return ops.deconv3d_k4s2(x.contiguous(), w, b, act)


# Small stride-1 3x3x3 layers of the hourglass (continuous_IGEVstereo.py:22-89) on the MFMA convolution kernel: out[:, d] =
# sum_kd conv2d(x[:, d + kd - 1], W[:, :, kd]) is ONE 2-D convolution with batch = D over the un-materialised channel concat of
# three depth-shifted views of a depth-major copy of x with a zero slice at both ends (ops.conv2d's multi-source input).  The
# direct kernel (as_conv3d_k3, VALU) is latency-bound on these volumes (48 -> 48 at 6x17x30: 83 us for 0.4 GFLOP); the implicit
# GEMM with its split-K takes ~15 us, plus the two layout copies.  ANYSTEREO_CONV3D_MFMA=0 keeps the direct kernel.
_CONV3D_MFMA = __import__("os").environ.get("ANYSTEREO_CONV3D_MFMA", "1") != "0"
_CONV3D_MFMA_MAX_VOXELS = int(__import__("os").environ.get("ANYSTEREO_CONV3D_MFMA_MAX_VOXELS", "40000"))


def conv3d_mfma_ok(conv: nn.Conv3d, x: torch.Tensor) -> bool:
    return (_CONV3D_MFMA and ops.get_precision() == "split" and conv.stride == (1, 1, 1) and x.shape[0] == 1
            and conv.in_channels % 16 == 0 and conv.out_channels >= 16 and x.shape[2] * x.shape[3] * x.shape[4] <= _CONV3D_MFMA_MAX_VOXELS)


def _conv3d_pack2d(mod, conv, bn):
    """PackedConv of the layer as a 2-D convolution over [x(d-1) | x(d) | x(d+1)] channels, BatchNorm folded (fp64 fold, as for
    every other folded layer); rebuilt when the folded weight changes (FoldedConv's key)."""
    cache = mod.__dict__.setdefault("_c3d_cache", {})
    fc = cache.setdefault(("fold2d", id(conv)), ops.FoldedConv())
    w, b = fc.get(conv, bn)  # [Cout, Cin, 3, 3, 3] (module layout), bias | None
    ent = cache.get(("pack2d", id(conv)))
    if ent is None or ent[0] is not w:
        co, ci = w.shape[:2]
        w2 = w.permute(0, 2, 1, 3, 4).reshape(co, 3 * ci, 3, 3).contiguous()  # channel index = kd * Cin + c
        pk = ops.PackedConv()
        ent = (w, pk, w2, b)
        cache[("pack2d", id(conv))] = ent
    _, pk, w2, b = ent
    return pk.get([w2], [b])


def conv3d_mfma(mod: nn.Module, conv: nn.Conv3d, bn, x: torch.Tensor, act: int, gate=None) -> torch.Tensor:
    _, c, d, h, w = x.shape
    xp = x.new_empty((d + 2, c, h, w))
    xp[0].fill_(0.0)      # fill_ is a kernel; zero_() on a contiguous slice is a hipMemsetAsync = a memset NODE under capture
    xp[d + 1].fill_(0.0)
    xp[1:d + 1].copy_(x[0].transpose(0, 1))  # depth-major copy: slice d of the volume is a dense [C, H, W] image
    y = ops.conv2d([xp[0:d], xp[1:d + 1], xp[2:d + 2]], _conv3d_pack2d(mod, conv, bn), act=act)  # [D, Cout, H, W]
    if gate is None:
        return y.transpose(0, 1).unsqueeze(0).contiguous()
    out = y.new_empty((1, y.shape[1], d, h, w))  # the layout copy back and FeatureAtt's gate in one pass
    return torch.mul(y.transpose(0, 1).unsqueeze(0), gate.unsqueeze(2), out=out)


def conv3d_fused(mod: nn.Module, conv: nn.Conv3d, bn, x: torch.Tensor, act: int, gate=None) -> torch.Tensor:
    """gate [B, Cout, Ho, Wo] (FeatureAtt, submodule.py:328-341): the result times the gate, in the producing launch."""
    if conv3d_mfma_ok(conv, x):
        return conv3d_mfma(mod, conv, bn, x.contiguous(), act, gate)
    cache = mod.__dict__.setdefault("_c3d_cache", {})
    fc = cache.setdefault(id(conv), ops.FoldedConv("c3d"))
    w, b = fc.get(conv, bn)
    return ops.conv3d_k3(x.contiguous(), w, b, conv.stride[0], act, gate=None if gate is None else gate.contiguous())


class _SearchedConv(torch.autograd.Function):
    """aten.convolution / convolution_backward with MIOpen's solver SEARCH enabled for these calls only.
    Without a tuning database MIOpen's immediate mode serves the hourglass' 3-D convolutions (17 configurations) with
    naive reference kernels in training — 223 ms of a 359 ms cfg-4 step, 4.6 ms after a search (tools/conv_bwd_times.py) —
    while turning `cudnn.benchmark` on globally makes it search every 2-D configuration of the model too (> 8 min)."""

    @staticmethod
    def forward(ctx, x, w, bias, cfg):
        # Under autocast (the reference's training arithmetic, train_continuous_IGEV.py:206) the convolution runs in the autocast
        # dtype, as nn.Conv3d would: operands are cast HERE (aten.convolution called directly would be cast by autocast's own
        # wrapper, but the tensors saved for backward would keep their dtypes and convolution_backward refuses a mix); the
        # gradients go back in the dtypes the inputs came in.
        ctx.in_dtypes = (x.dtype, w.dtype, None if bias is None else bias.dtype)
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            x, w, bias = x.to(dt), w.to(dt), (None if bias is None else bias.to(dt))
        elif x.dtype != w.dtype:
            x = x.to(w.dtype)
        ctx.save_for_backward(x, w)
        ctx.cfg, ctx.bias_sizes = cfg, None if bias is None else list(bias.shape)
        prev = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = True
        try:
            with torch.autocast("cuda", enabled=False):
                return torch.ops.aten.convolution(x, w, bias, *cfg)
        finally:
            torch.backends.cudnn.benchmark = prev

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        prev = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = True
        try:
            with torch.autocast("cuda", enabled=False):
                gx, gw, gb = torch.ops.aten.convolution_backward(
                    gy.contiguous().to(x.dtype), x, w, ctx.bias_sizes, *ctx.cfg,
                    [ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.bias_sizes is not None and ctx.needs_input_grad[2]])
        finally:
            torch.backends.cudnn.benchmark = prev
        dx, dw, db = ctx.in_dtypes
        return (None if gx is None else gx.to(dx), None if gw is None else gw.to(dw), None if gb is None else gb.to(db), None)


def conv3d_train(conv, x):
    """conv(x) for an nn.Conv3d / nn.ConvTranspose3d; under autograd on the GPU through _SearchedConv."""
    if not (x.is_cuda and x.dim() == 5 and torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)):
        return conv(x)
    if _NO_SEARCHED_CONV:  # diagnostics: MIOpen's immediate mode (naive kernels) instead of the searched solvers
        return conv(x)
    transposed = isinstance(conv, nn.ConvTranspose3d)
    cfg = (list(conv.stride), list(conv.padding), list(conv.dilation), transposed,
           list(conv.output_padding) if transposed else [0, 0, 0], conv.groups)
    return _SearchedConv.apply(x, conv.weight, conv.bias, cfg)


def _conv_nd(is_3d: bool, deconv: bool):
    if is_3d:
        return nn.ConvTranspose3d if deconv else nn.Conv3d
    return nn.ConvTranspose2d if deconv else nn.Conv2d


class _ConvNormAct(nn.Module):
    """conv (no bias) -> norm -> LeakyReLU(0.01); `norm_attr` fixes the state_dict key
    ('bn' for BasicConv, 'IN' for BasicConv_IN; submodule.py:6-33, :76-103)."""

    norm_attr = "bn"

    def __init__(self, cin, cout, deconv=False, is_3d=False, norm=True, relu=True, **kw):
        super().__init__()
        self.relu = relu
        self.use_norm = norm
        self.conv = _conv_nd(is_3d, deconv)(cin, cout, bias=False, **kw)
        setattr(self, self.norm_attr, self._make_norm(cout, is_3d))

    def _make_norm(self, c, is_3d):
        raise NotImplementedError

    def forward_cat(self, xs):
        """self(torch.cat(xs, 1)) without the concatenation where the convolution kernel takes its sources one by one (inference:
        the 1x1x1 blocks behind the hourglass' skip concats, the 3x3 blocks behind the feature decoder's): every source but the
        last needs a multiple of 16 channels (the kernel's K chunk)."""
        xs = list(xs)
        norm = getattr(self, self.norm_attr) if self.use_norm else None
        c = self.conv
        ok = (_FOLD_POINTWISE and _CAT_FREE and len(xs) > 1 and all(fused_ok(t, self) for t in xs) and ops.get_precision() == "split"
              and all(t.shape[1] % 16 == 0 for t in xs[:-1]) and c.groups == 1)
        if ok and xs[0].dim() == 5 and (norm is None or isinstance(norm, nn.BatchNorm3d)) and c.kernel_size == (1, 1, 1) \
                and c.stride == (1, 1, 1) and c.padding == (0, 0, 0) and c.dilation == (1, 1, 1):
            b_, _, d_, h_, w_ = xs[0].shape
            pk = self.__dict__.setdefault("_pk_fold", ops.PackedConv())
            pack = pk.get_folded(c, norm) if norm is not None else pk.get([c.weight], [c.bias])
            y = ops.conv2d([t.contiguous().view(b_, t.shape[1], 1, d_ * h_ * w_) for t in xs], pack,
                           act=L.ACT_LEAKY if self.relu else L.ACT_NONE)
            return y.view(b_, c.out_channels, d_, h_, w_)
        if ok and xs[0].dim() == 4 and _plain_instance_norm(norm) and conv2d_hip_ok(c) and c.stride == (1, 1):
            packs = self.__dict__.setdefault("_hip_packs", {})
            pk = packs.setdefault(id(c), ops.PackedConv())
            y = ops.conv2d([t.contiguous() for t in xs], pk.get([c.weight], [c.bias]))
            return ops.instance_norm_act(y, norm.eps, L.ACT_LEAKY if self.relu else L.ACT_NONE)
        return self(torch.cat(xs, 1))

    def forward(self, x, gate=None):
        """gate (inference, 3-D blocks): FeatureAtt's channel gate [B, Cout, H, W] applied to the block's result — in the
        convolution's own launch where that is the fused 3x3x3 path, as a separate multiply otherwise."""
        if gate is not None:
            norm = getattr(self, self.norm_attr) if self.use_norm else None
            if fused_ok(x, self) and (norm is None or isinstance(norm, nn.BatchNorm3d)) and conv3d_k3_ok(self.conv):
                return conv3d_fused(self, self.conv, norm, x, L.ACT_LEAKY if self.relu else L.ACT_NONE, gate=gate)
            return gate.unsqueeze(2) * self.forward(x)
        norm = getattr(self, self.norm_attr) if self.use_norm else None
        if fused_ok(x, self) and (norm is None or isinstance(norm, nn.BatchNorm3d)):
            if conv3d_k3_ok(self.conv):
                return conv3d_fused(self, self.conv, norm, x, L.ACT_LEAKY if self.relu else L.ACT_NONE)
            if deconv3d_k4s2_ok(self.conv):
                return deconv3d_fused(self, self.conv, norm, x, L.ACT_LEAKY if self.relu else L.ACT_NONE)
        if _FOLD_POINTWISE and fused_ok(x, self) and (norm is None or isinstance(norm, (nn.BatchNorm2d, nn.BatchNorm3d))) and not isinstance(
                self.conv, (nn.ConvTranspose2d, nn.ConvTranspose3d)):
            # frozen BatchNorm folded into the weights, LeakyReLU in the epilogue: one library launch instead of a MIOpen
            # GEMM / convolution + BatchNorm + activation passes
            c, act = self.conv, (L.ACT_LEAKY if self.relu else L.ACT_NONE)
            pk = self.__dict__.setdefault("_pk_fold", ops.PackedConv())
            if x.dim() == 4 and conv2d_hip_ok(c):
                pack = pk.get_folded(c, norm) if norm is not None else pk.get([c.weight], [c.bias])
                return ops.conv2d([x.contiguous()], pack, act=act, stride=c.stride[0])
            if (x.dim() == 5 and c.kernel_size == (1, 1, 1) and c.stride == (1, 1, 1) and c.padding == (0, 0, 0)
                    and c.dilation == (1, 1, 1) and c.groups == 1 and x.shape[2] * x.shape[3] * x.shape[4] < 2 ** 31):
                # 1x1x1: a pointwise map over D*H*W = the 1x1 kernel on the flattened volume
                b_, ci, d_, h_, w_ = x.shape
                pack = pk.get_folded(c, norm) if norm is not None else pk.get([c.weight], [c.bias])
                y = ops.conv2d([x.contiguous().view(b_, ci, 1, d_ * h_ * w_)], pack, act=act)
                return y.view(b_, c.out_channels, d_, h_, w_)
        if fused_ok(x, self) and x.dim() == 4 and _plain_instance_norm(norm):
            # conv (library kernel where it applies, else MIOpen) -> fused InstanceNorm + LeakyReLU
            y = deconv2d_k4s2_plain(self, self.conv, x) if (_FOLD_POINTWISE and deconv2d_k4s2_ok(self.conv)) else conv2d_plain(self, self.conv, x)
            return ops.instance_norm_act(y, norm.eps, L.ACT_LEAKY if self.relu else L.ACT_NONE)
        x = conv3d_train(self.conv, x) if x.dim() == 5 else G.module_conv2d(self, "t", self.conv, x)
        if self.use_norm:
            x = norm(x)
        return F.leaky_relu(x, 0.01) if self.relu else x


class BasicConv(_ConvNormAct):
    norm_attr = "bn"

    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, bn=True, relu=True, **kw):
        super().__init__(in_channels, out_channels, deconv, is_3d, bn, relu, **kw)

    def _make_norm(self, c, is_3d):
        return nn.BatchNorm3d(c) if is_3d else nn.BatchNorm2d(c)


class BasicConv_IN(_ConvNormAct):
    norm_attr = "IN"

    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, IN=True, relu=True, **kw):
        super().__init__(in_channels, out_channels, deconv, is_3d, IN, relu, **kw)

    def _make_norm(self, c, is_3d):
        return nn.InstanceNorm3d(c) if is_3d else nn.InstanceNorm2d(c)


class _Up2x(nn.Module):
    """stride-2 (de)conv, resize-to-skip if shapes differ, concat or add skip, 3x3 conv
    (submodule.py:36-73 Conv2x, :106-144 Conv2x_IN)."""

    block = BasicConv

    def __init__(self, cin, cout, deconv=False, is_3d=False, concat=True, keep_concat=True, norm=True,
                 relu=True, keep_dispc=False):
        super().__init__()
        self.concat = concat
        self.is_3d = is_3d
        if deconv and is_3d and keep_dispc:
            k, s, p = (1, 4, 4), (1, 2, 2), (0, 1, 1)
        else:
            k = (4, 4, 4) if (deconv and is_3d) else (4 if deconv else 3)
            s, p = 2, 1
        self.conv1 = self.block(cin, cout, deconv, is_3d, True, True, kernel_size=k, stride=s, padding=p)
        if concat:
            self.conv2 = self.block(cout * 2, cout * (2 if keep_concat else 1), False, is_3d, norm, relu,
                                    kernel_size=3, stride=1, padding=1)
        else:
            self.conv2 = self.block(cout, cout, False, is_3d, norm, relu, kernel_size=3, stride=1, padding=1)

    def forward(self, x, rem):
        x = self.conv1(x)
        if x.shape != rem.shape:
            x = F.interpolate(x, size=rem.shape[-2:], mode="nearest")
        if self.concat:
            return self.conv2.forward_cat((x, rem))  # the concat is never materialised where the kernel takes two sources
        return self.conv2(x + rem)


class Conv2x(_Up2x):
    block = BasicConv

    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, concat=True, keep_concat=True,
                 bn=True, relu=True, keep_dispc=False):
        super().__init__(in_channels, out_channels, deconv, is_3d, concat, keep_concat, bn, relu, keep_dispc)


class Conv2x_IN(_Up2x):
    block = BasicConv_IN

    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, concat=True, keep_concat=True,
                 IN=True, relu=True, keep_dispc=False):
        super().__init__(in_channels, out_channels, deconv, is_3d, concat, keep_concat, IN, relu, keep_dispc)


class LayerNorm2d(nn.Module):
    """Per-pixel LayerNorm over channels of an NCHW tensor (submodule.py:148-187).
    Plain autograd (the reference's hand-written backward is mathematically the same)."""

    def __init__(self, channels, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.eps = eps

    def forward(self, x):
        mu = x.mean(1, keepdim=True)
        var = (x - mu).pow(2).mean(1, keepdim=True)
        y = (x - mu) / (var + self.eps).sqrt()
        return self.weight.view(1, -1, 1, 1) * y + self.bias.view(1, -1, 1, 1)


class _HighResAgg(nn.Module):
    """PixelUnshuffle(2) -> conv-IN-LeakyReLU -> simplified channel attention -> conv -> norm -> act
    (submodule.py:190-252: HighRes_Aggregation / _LN / _LN_GeLU differ in the head only)."""

    def __init__(self, input_dim, output_dim, head_norm, head_act):
        super().__init__()
        self.embeding = nn.Sequential(
            nn.PixelUnshuffle(2),
            BasicConv_IN(input_dim * 4, output_dim, kernel_size=3, stride=1, padding=1))
        self.sca = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(output_dim, output_dim, 1, bias=True))
        self.head = nn.Sequential(nn.Conv2d(output_dim, output_dim, 3, 1, 1, bias=False), head_norm, head_act)

    def forward(self, x):
        x = self.embeding(x)
        x = x * self.sca(x)
        conv, norm, act = self.head[0], self.head[1], self.head[2]
        code = L.ACT_RELU if isinstance(act, nn.ReLU) else (L.ACT_GELU if isinstance(act, nn.GELU) and act.approximate == "none" else None)
        if fused_ok(x, self) and code is not None:
            if isinstance(norm, LayerNorm2d) and norm.weight.numel() <= 64:
                return ops.layernorm2d_act(conv2d_plain(self, conv, x), norm.weight.detach(), norm.bias.detach(), norm.eps, code)
            if _plain_instance_norm(norm):
                return ops.instance_norm_act(conv2d_plain(self, conv, x), norm.eps, code)
        return self.head(x)


class HighRes_Aggregation(_HighResAgg):
    def __init__(self, input_dim, output_dim):
        super().__init__(input_dim, output_dim, nn.InstanceNorm2d(output_dim), nn.ReLU())


class HighRes_Aggregation_LN(_HighResAgg):
    def __init__(self, input_dim, output_dim):
        super().__init__(input_dim, output_dim, LayerNorm2d(output_dim), nn.ReLU())


class HighRes_Aggregation_LN_GeLU(_HighResAgg):
    def __init__(self, input_dim, output_dim):
        super().__init__(input_dim, output_dim, LayerNorm2d(output_dim), nn.GELU())


def plain_stem(cin, cout, unshuffle: bool):
    """The 'type1' (PixelUnshuffle) and 'IGEV' (stride-2) stems
    (continuous_IGEVstereo.py:105-118, prune_raft_stereo.py:110-134)."""
    if unshuffle:
        return nn.Sequential(nn.PixelUnshuffle(2),
                             BasicConv_IN(cin * 4, cout, kernel_size=3, stride=1, padding=1),
                             nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.InstanceNorm2d(cout), nn.ReLU())
    return nn.Sequential(BasicConv_IN(cin, cout, kernel_size=3, stride=2, padding=1),
                         nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.InstanceNorm2d(cout), nn.ReLU())


class FeatureAtt(nn.Module):
    """Sigmoid channel gate of the cost volume from 2-D features (submodule.py:328-341)."""

    def __init__(self, cv_chan, feat_chan):
        super().__init__()
        self.feat_att = nn.Sequential(BasicConv(feat_chan, feat_chan // 2, kernel_size=1, stride=1, padding=0),
                                      nn.Conv2d(feat_chan // 2, cv_chan, 1))

    def gate_ok(self, feat) -> bool:
        return _FOLD_POINTWISE and fused_ok(feat, self) and conv2d_hip_ok(self.feat_att[1])

    def gate(self, feat):
        """sigmoid(feat_att(feat)) [B, cv_chan, H, W]: depends on the 2-D features only, so the inference schedule computes all
        gates of a pass on a branch of their own beside the cost aggregation (`early`, set per forward by the model)."""
        c = self.feat_att[1]  # 1x1 + bias with the sigmoid in the epilogue
        pk = self.__dict__.setdefault("_pk_gate", ops.PackedConv())
        return ops.conv2d([self.feat_att[0](feat).contiguous()], pk.get([c.weight], [c.bias]), act=L.ACT_SIGMOID)

    early = None  # (gate tensor, event recorded behind it on the branch stream) for the next forward() call, or None

    def take_gate(self, feat):
        """The gate for this forward: the one the model computed early on its branch stream (joined here), or computed now."""
        ent = self.__dict__.pop("early", None)
        if ent is None:
            return self.gate(feat)
        g, ev = ent
        cur = torch.cuda.current_stream(feat.device)
        cur.wait_event(ev)
        g.record_stream(cur)
        return g

    def forward(self, cv, feat):
        if self.gate_ok(feat):
            return self.take_gate(feat).unsqueeze(2) * cv
        return torch.sigmoid(self.feat_att(feat).unsqueeze(2)) * cv

    # inference: the gate rides in the launch that produces the cost volume it multiplies (BasicConv.forward(x, gate=))
    fused_gate = __import__("os").environ.get("ANYSTEREO_FUSED_GATES", "1") != "0"

    def after(self, block, x, feat):
        """self(block(x), feat) with the multiply folded into `block`'s last 3-D convolution where possible."""
        cat = isinstance(x, (tuple, list))  # a channel concat handed over as its parts (the hourglass' skip connections)
        x0 = x[0] if cat else x
        if not (self.fused_gate and self.gate_ok(feat) and fused_ok(x0, self)):
            return self(block(torch.cat(x, 1) if cat else x), feat)
        if isinstance(block, nn.Sequential):
            mods = list(block)
            if cat:
                x = mods[0].forward_cat(x) if isinstance(mods[0], _ConvNormAct) else mods[0](torch.cat(x, 1))
                mods = mods[1:]
            for m in mods[:-1]:
                x = m(x)
            block = mods[-1]
        elif cat:
            x = torch.cat(x, 1)
        g = self.take_gate(feat)  # joined as late as possible: only the last convolution waits for the branch that computes the gates
        if isinstance(block, _ConvNormAct):
            return block(x, gate=g)
        return g.unsqueeze(2) * block(x)


class hourglass(nn.Module):
    """3-D cost aggregation U-Net with feature attention (continuous_IGEVstereo.py:22-89)."""

    def __init__(self, in_channels):
        super().__init__()
        c = in_channels

        def down(ci, co):
            return nn.Sequential(
                BasicConv(ci, co, is_3d=True, bn=True, relu=True, kernel_size=3, padding=1, stride=2, dilation=1),
                BasicConv(co, co, is_3d=True, bn=True, relu=True, kernel_size=3, padding=1, stride=1, dilation=1))

        def up(ci, co, bn=True, relu=True):
            return BasicConv(ci, co, deconv=True, is_3d=True, bn=bn, relu=relu, kernel_size=(4, 4, 4),
                             padding=(1, 1, 1), stride=(2, 2, 2))

        def agg(ci, co):
            return nn.Sequential(
                BasicConv(ci, co, is_3d=True, kernel_size=1, padding=0, stride=1),
                BasicConv(co, co, is_3d=True, kernel_size=3, padding=1, stride=1),
                BasicConv(co, co, is_3d=True, kernel_size=3, padding=1, stride=1))

        self.conv1 = down(c, c * 2)
        self.conv2 = down(c * 2, c * 4)
        self.conv3 = down(c * 4, c * 6)
        self.conv3_up = up(c * 6, c * 4)
        self.conv2_up = up(c * 4, c * 2)
        self.conv1_up = up(c * 2, 8, bn=False, relu=False)
        self.agg_0 = agg(c * 8, c * 4)
        self.agg_1 = agg(c * 4, c * 2)
        self.feature_att_8 = FeatureAtt(c * 2, 64)
        self.feature_att_16 = FeatureAtt(c * 4, 192)
        self.feature_att_32 = FeatureAtt(c * 6, 160)
        self.feature_att_up_16 = FeatureAtt(c * 4, 192)
        self.feature_att_up_8 = FeatureAtt(c * 2, 64)

    def forward(self, x, features):
        c1 = self.feature_att_8.after(self.conv1, x, features[1])
        c2 = self.feature_att_16.after(self.conv2, c1, features[2])
        c3 = self.feature_att_32.after(self.conv3, c2, features[3])
        c2 = self.feature_att_up_16.after(self.agg_0, (self.conv3_up(c3), c2), features[2])
        c1 = self.feature_att_up_8.after(self.agg_1, (self.conv2_up(c2), c1), features[1])
        return self.conv1_up(c1)
