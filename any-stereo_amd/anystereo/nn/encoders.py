"""Feature / context encoders.

Training / autograd / CPU construction run on plain PyTorch modules (MIOpen on the GPU).  In inference
(`eval()`, no grad, CUDA) the BatchNorm residual trunk of the context net takes the fused path: BatchNorm is
folded into the conv weights (`ops.PackedConv.get_folded`), every stride-1 3x3 conv runs on the library's
implicit-GEMM kernel with ReLU and the residual tail `relu(x + relu(.))` fused into its epilogue
(`as_conv2d`, AS_EPI_LINEAR with `h`), and the stride-2 convs keep MIOpen but with folded weights, so no
BatchNorm / ReLU / add pass over the full-resolution 64-channel maps remains.

State-dict keys follow models/coreContinuous_IGEV/extractor.py so reference checkpoints load:
ResidualBlock :10-62, BasicEncoder :126-198, MultiBasicEncoder :200-304, Feature :327-361.
`timm` is not available offline, so the MobileNetV2-100 trunk the reference takes from
`timm.create_model('mobilenetv2_100', features_only=True)` (extractor.py:331) is restated here
with timm's module names (conv_stem / bn1 / blocks[i][j].conv_pw|bn1|conv_dw|bn2|conv_pwl|bn3).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib as L
from .. import grad as G
from .. import ops
from .blocks import BasicConv_IN, Conv2x_IN, conv2d_hip_ok as _conv_hip_ok, conv2d_plain, fused_ok as _fused_ok


def _plain_in(norm) -> bool:
    return isinstance(norm, nn.InstanceNorm2d) and not norm.affine and not norm.track_running_stats


def _norm(kind: str, c: int, groups: int | None = None):
    if kind == "group":
        return nn.GroupNorm(num_groups=groups if groups is not None else c // 8, num_channels=c)
    if kind == "batch":
        return nn.BatchNorm2d(c)
    if kind == "instance":
        return nn.InstanceNorm2d(c)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(kind)


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = _norm(norm_fn, planes, planes // 8)
        self.norm2 = _norm(norm_fn, planes, planes // 8)
        if stride == 1 and in_planes == planes:
            self.downsample = None
        else:
            # norm3 is registered twice (as .norm3 and as .downsample.1), as in the reference
            self.norm3 = _norm(norm_fn, planes, planes // 8)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm3)

        self._pk1, self._pk2, self._pkd = ops.PackedConv(), ops.PackedConv(), ops.PackedConv()
        self._f1, self._fd = ops.FoldedConv(), ops.FoldedConv()

    def forward(self, x):
        if _fused_ok(x, self) and isinstance(self.norm1, nn.BatchNorm2d) and _conv_hip_ok(self.conv2):
            return self._forward_fused(x)
        if _fused_ok(x, self) and _plain_in(self.norm1) and _plain_in(self.norm2):
            return self._forward_fused_in(x)
        # training: own kernels where the shape allows; a frozen BatchNorm (+ ReLU) rides in the convolution (grad.conv_frozen_bn)
        y = G.conv_frozen_bn(self, "t1", self.conv1, self.norm1, x, relu=True)
        y = G.conv_frozen_bn(self, "t2", self.conv2, self.norm2, y, relu=True)
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)

    def _forward_fused_in(self, x):
        """InstanceNorm variant (RAFT feature net): convs on the library kernel where they apply, InstanceNorm + ReLU in
        one fused pass each (`as_instance_norm_act`)."""
        x = x.contiguous()
        y = ops.instance_norm_act(conv2d_plain(self, self.conv1, x), self.norm1.eps, L.ACT_RELU)
        if self.downsample is not None:
            ds_norm = self.downsample[1]
            x = conv2d_plain(self, self.downsample[0], x)
            x = ops.instance_norm_act(x, ds_norm.eps, L.ACT_NONE) if _plain_in(ds_norm) else ds_norm(x)
        # relu(x + relu(IN(conv2 y))) in the normalisation pass
        return ops.instance_norm_act(conv2d_plain(self, self.conv2, y), self.norm2.eps, L.ACT_RELU, residual=x.contiguous())

    def _forward_fused(self, x):
        x = x.contiguous()
        if _conv_hip_ok(self.conv1):
            if self.conv1.stride[0] == 1 and _bs_links():
                # conv1 -> conv2 through a blocked split-fp16 tensor only (ops.BS8): same values, conv2 stages its operands by DMA
                y = ops.BS8.empty(x.shape[0], self.conv1.out_channels, x.shape[2], x.shape[3], x.device)
                ops.conv2d([x], self._pk1.get_folded(self.conv1, self.norm1), act=L.ACT_RELU, out_bs=y, bs_only=True)
            else:
                y = ops.conv2d([x], self._pk1.get_folded(self.conv1, self.norm1), act=L.ACT_RELU, stride=self.conv1.stride[0])
        else:  # stride 2: MIOpen with the folded weights
            w, b = self._f1.get(self.conv1, self.norm1)
            y = nn.functional.conv2d(x, w, b, self.conv1.stride, self.conv1.padding).relu_()
        if self.downsample is not None:
            ds = self.downsample[0]
            if ds.kernel_size == (1, 1) and ds.padding == (0, 0) and ds.groups == 1:
                # 1x1 stride-s shortcut = the same 1x1 convolution on the subsampled map: library kernel instead of a GEMM through
                # MIOpen (90 -> ~35 us for 64 -> 96 at 544x960)
                st = ds.stride[0]
                xs = x if st == 1 else x[:, :, ::st, ::st].contiguous()
                x = ops.conv2d([xs], self._pkd.get_folded(ds, self.downsample[1]))
            else:
                w, b = self._fd.get(ds, self.downsample[1])
                x = nn.functional.conv2d(x, w, b, ds.stride)
        # relu(x + relu(bn2(conv2 y))) in the conv epilogue
        return ops.conv2d([y], self._pk2.get_folded(self.conv2, self.norm2), act=L.ACT_RELU, h=x.contiguous())


def _bs_links() -> bool:
    from .update import _links
    return _links()


def _init_encoder(mod: nn.Module):
    for m in mod.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
            if m.weight is not None:
                nn.init.constant_(m.weight, 1)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


class _Trunk(nn.Module):
    """7x7 stem + three 2-block residual stages shared by both encoders."""

    def __init__(self, norm_fn, downsample):
        super().__init__()
        self.norm_fn = norm_fn
        self.downsample = downsample
        self.norm1 = _norm(norm_fn, 64, 8)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=1 + (downsample > 2), padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = 64
        self.layer1 = self._make_layer(64, 1)
        self.layer2 = self._make_layer(96, 1 + (downsample > 1))
        self.layer3 = self._make_layer(128, 1 + (downsample > 0))

    def _make_layer(self, dim, stride=1):
        seq = nn.Sequential(ResidualBlock(self.in_planes, dim, self.norm_fn, stride),
                            ResidualBlock(dim, dim, self.norm_fn, 1))
        self.in_planes = dim
        return seq

    def _stem_hip_ok(self, x) -> bool:
        c = self.conv1
        return (ops.get_precision() == "split" and x.shape[1] == 3 and tuple(c.weight.shape) == (64, 3, 7, 7) and c.stride == (1, 1)
                and c.padding == (3, 3) and c.dilation == (1, 1) and c.groups == 1)

    def _stem_pack(self):
        if not hasattr(self, "_pk_stem"):
            self._pk_stem = ops.Stem7x7Pack()
        return self._pk_stem

    def trunk(self, x, stage=None):
        stage = stage or (lambda name: None)
        if _fused_ok(x, self) and isinstance(self.norm1, nn.BatchNorm2d):
            if not hasattr(self, "_f_stem"):
                self._f_stem = ops.FoldedConv()
            w, b = self._f_stem.get(self.conv1, self.norm1)
            if self._stem_hip_ok(x):  # 7x7, 3 -> 64, stride 1: one split-precision MFMA launch with bias + ReLU (csrc/stem7x7.hip)
                x = ops.conv7x7_c3(x.contiguous(), self._stem_pack(), w, b, act=L.ACT_RELU)
            else:
                x = nn.functional.conv2d(x, w, b, self.conv1.stride, self.conv1.padding).relu_()
        elif _fused_ok(x, self) and _plain_in(self.norm1):
            if self._stem_hip_ok(x):
                y = ops.conv7x7_c3(x.contiguous(), self._stem_pack(), self.conv1.weight, self.conv1.bias)
            else:
                y = self.conv1(x)
            x = ops.instance_norm_act(y, self.norm1.eps, L.ACT_RELU)
        else:
            x = self.relu1(self.norm1(self.conv1(x)))
        stage("stem")
        x = self.layer1(x)
        stage("layer1")
        x = self.layer2(x)
        stage("layer2")
        x = self.layer3(x)
        stage("layer3")
        return x


class BasicEncoder(_Trunk):
    """RAFT-Stereo feature net: both images batched through the trunk, 1x1 to output_dim."""

    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0, downsample=3):
        super().__init__(norm_fn, downsample)
        self.conv2 = nn.Conv2d(128, output_dim, 1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        _init_encoder(self)

    def forward(self, x, dual_inp=False):
        is_list = isinstance(x, (tuple, list))
        if is_list:
            n = x[0].shape[0]
            x = torch.cat(x, dim=0)
        x = self.conv2(self.trunk(x))
        if self.training and self.dropout is not None:
            x = self.dropout(x)
        return x.split(n, dim=0) if is_list else x


class MultiBasicEncoder(_Trunk):
    """Context net: per scale (1/4, 1/8, 1/16) one head per entry of output_dim
    ([hidden_dims, context_dims]); returns lists of [hidden, context] per scale."""

    def __init__(self, output_dim=[128], norm_fn="batch", dropout=0.0, downsample=3):
        super().__init__(norm_fn, downsample)
        self.layer4 = self._make_layer(128, 2)
        self.layer5 = self._make_layer(128, 2)
        self.outputs04 = nn.ModuleList(
            nn.Sequential(ResidualBlock(128, 128, norm_fn, 1), nn.Conv2d(128, d[2], 3, padding=1)) for d in output_dim)
        self.outputs08 = nn.ModuleList(
            nn.Sequential(ResidualBlock(128, 128, norm_fn, 1), nn.Conv2d(128, d[1], 3, padding=1)) for d in output_dim)
        self.outputs16 = nn.ModuleList(nn.Conv2d(128, d[0], 3, padding=1) for d in output_dim)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        _init_encoder(self)

    on_stage = None  # optional callable(name): timeline markers of the inference schedule (set per forward by the model)

    def forward(self, x, dual_inp=False, num_layers=3):
        stage = self.on_stage or (lambda name: None)
        x = self.trunk(x, stage)
        if dual_inp:
            v = x
            x = x[: x.shape[0] // 2]
        tail = (v,) if dual_inp else ()
        o04 = self._heads(self.outputs04, x)
        stage("heads04")
        if num_layers == 1:
            return (o04,) + tail
        y = self.layer4(x)
        o08 = self._heads(self.outputs08, y)
        stage("heads08")
        if num_layers == 2:
            return (o04, o08) + tail
        z = self.layer5(y)
        o16 = self._heads(self.outputs16, z)
        stage("heads16")
        return (o04, o08, o16) + tail

    # The two heads of a scale (hidden state | context, extractor.py:254-273) apply the same layer shapes to the same input: in
    # inference each layer of the pair is ONE dual launch (as_conv_desc.dual with a residual and dense outputs per convolution)
    # instead of two launches on the pre-loop's side branch — 7 launches per pass instead of 14, twice the blocks per launch on
    # the 1/8 and 1/16 maps.  When the caller applies tanh / relu to the pair right away (continuous_IGEVstereo.py:271-272) the
    # activations ride in the last launch's epilogue (`head_acts`).  Same arithmetic per convolution: results are bit-identical.
    paired_heads = __import__("os").environ.get("ANYSTEREO_PAIRED_HEADS", "1") != "0"
    head_acts = None  # (act of head 0, act of head 1) fused into the heads' last convolution, set by the caller for one forward

    def _heads_plain(self, heads, x):
        outs = [self._head(f, x) for f in heads]
        if self.head_acts is not None:  # the caller asked for the activations: applied here when no epilogue can carry them
            fn = {L.ACT_NONE: (lambda t: t), L.ACT_TANH: torch.tanh, L.ACT_RELU: torch.relu}
            outs = [fn[a](o) for a, o in zip(self.head_acts, outs)]
        return outs

    def _heads(self, heads, x):
        if not (self.paired_heads and len(heads) == 2 and _fused_ok(x, self) and ops.get_precision() == "split" and _bs_links()):
            return self._heads_plain(heads, x)
        fa, fb = heads
        seq = isinstance(fa, nn.Sequential)
        ca, cb = (fa[1], fb[1]) if seq else (fa, fb)
        same = lambda m, n: (m.weight.shape == n.weight.shape and m.kernel_size == n.kernel_size and _conv_hip_ok(m) and _conv_hip_ok(n)
                             and m.stride == (1, 1) and n.stride == (1, 1))
        if isinstance(fa, nn.Sequential) != isinstance(fb, nn.Sequential) or not same(ca, cb):
            return self._heads_plain(heads, x)
        packs = self.__dict__.setdefault("_hip_packs", {})
        pk = lambda c: packs.setdefault(id(c), ops.PackedConv())
        x = x.contiguous()
        b, _, hh, ww = x.shape
        src_a = src_b = x
        if seq:
            ra, rb = fa[0], fb[0]
            if not (isinstance(ra.norm1, nn.BatchNorm2d) and isinstance(rb.norm1, nn.BatchNorm2d) and ra.downsample is None
                    and rb.downsample is None and same(ra.conv1, rb.conv1) and same(ra.conv2, rb.conv2)):
                return self._heads_plain(heads, x)
            c = ra.conv1.out_channels
            ya, yb = ops.BS8.empty(b, c, hh, ww, x.device), ops.BS8.empty(b, c, hh, ww, x.device)
            ops.conv2d([x], ra._pk1.get_folded(ra.conv1, ra.norm1), act=L.ACT_RELU, out_bs=ya, bs_only=True,
                       dual={"src": x, "pack": rb._pk1.get_folded(rb.conv1, rb.norm1), "out_bs": yb})
            za, zb = ops.BS8.empty(b, c, hh, ww, x.device), ops.BS8.empty(b, c, hh, ww, x.device)
            # relu(x + relu(bn2(conv2 y))) per block; the blocks' results only feed the heads' last convolutions: blocked only
            ops.conv2d([ya], ra._pk2.get_folded(ra.conv2, ra.norm2), act=L.ACT_RELU, h=x, out_bs=za, bs_only=True,
                       dual={"src": yb, "pack": rb._pk2.get_folded(rb.conv2, rb.norm2), "h": x, "out_bs": zb})
            src_a, src_b = za, zb
        acts = self.head_acts or (L.ACT_NONE, L.ACT_NONE)
        oa, ob = ops.conv2d([src_a], pk(ca).get([ca.weight], [ca.bias]), act=acts[0],
                            dual={"src": src_b, "pack": pk(cb).get([cb.weight], [cb.bias]), "act": acts[1], "out": True})
        return [oa, ob]


def _plain_conv(mod: nn.Module, conv: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """conv(x) on the implicit-GEMM kernel: the inference fast path, or (training) forward / dgrad / wgrad through grad.Conv2dSame
    where the layer is a stride-1 "same" 1x1 / 3x3 convolution; else the module itself."""
    return conv2d_plain(mod, conv, x) if _fused_ok(x, mod) else G.module_conv2d(mod, f"pc{id(conv)}", conv, x)


def _multi_head(self, f, x):
    if isinstance(f, nn.Sequential):
        return _plain_conv(self, f[1], f[0](x))
    return _plain_conv(self, f, x)


MultiBasicEncoder._head = _multi_head


# ---- MobileNetV2-100 trunk with timm's naming ---------------------------------------------


class _DSConv(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv_dw = nn.Conv2d(cin, cin, 3, stride, 1, groups=cin, bias=False)
        self.bn1 = nn.BatchNorm2d(cin)
        self.conv_pw = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.res = stride == 1 and cin == cout

        self._fdw, self._pk = ops.FoldedConv(), ops.PackedConv()

    def forward(self, x):
        if _fused_ok(x, self):
            x = x.contiguous()
            w, b = self._fdw.get(self.conv_dw, self.bn1)
            y = ops.dwconv3x3(x, w, b, self.conv_dw.stride[0], L.ACT_RELU6)
            return ops.conv2d([y], self._pk.get_folded(self.conv_pw, self.bn2), add=x if self.res else None)
        y = nn.functional.relu6(self.bn1(G.module_dwconv(self.conv_dw, x)))  # training: own forward / dgrad / wgrad kernels
        y = self.bn2(self.conv_pw(y))
        return x + y if self.res else y


class _InvRes(nn.Module):
    def __init__(self, cin, cout, stride, expand=6):
        super().__init__()
        mid = cin * expand
        self.conv_pw = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv_dw = nn.Conv2d(mid, mid, 3, stride, 1, groups=mid, bias=False)
        self.bn2 = nn.BatchNorm2d(mid)
        self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.res = stride == 1 and cin == cout

        self._fdw, self._pk1, self._pk3 = ops.FoldedConv(), ops.PackedConv(), ops.PackedConv()
        self._irpk = ops.IrBlockPack()

    # inference: the whole block as ONE launch with the 6x-expanded tensor in LDS (csrc/irblock.hip) instead of pw / dw / pwl
    # launches with that tensor through HBM; opt-in (ANYSTEREO_FUSED_IR=1) until it beats the three launches on every shape
    fused_ir = {"0": False, "1": True, "auto": "auto"}[__import__("os").environ.get("ANYSTEREO_FUSED_IR", "0")]

    def _ir_pick(self) -> bool:
        """auto: the shapes the one-launch kernel wins stand-alone (tools/kbench_ir.py): the stride-2 blocks on the large maps."""
        if self.fused_ir == "auto":
            return self.conv_dw.stride[0] == 2 and self.conv_pw.in_channels <= 32
        return bool(self.fused_ir)

    def forward(self, x):
        if (_fused_ok(x, self) and self._ir_pick() and ops.get_precision() == "split" and self.conv_pwl.out_channels <= 160
                and self.conv_pw.in_channels <= 256 and self.conv_dw.stride[0] in (1, 2)):
            return ops.ir_block(x.contiguous(), self._irpk.get(self.conv_pw, self.bn1, self.conv_dw, self.bn2, self.conv_pwl, self.bn3),
                                self.conv_dw.stride[0], self.res)
        if _fused_ok(x, self):
            # pw (+bn1, ReLU6) and pwl (+bn3, + skip) on the implicit-GEMM kernel, dw (+bn2, ReLU6) on the direct one
            x = x.contiguous()
            y = ops.conv2d([x], self._pk1.get_folded(self.conv_pw, self.bn1), act=L.ACT_RELU6)
            w, b = self._fdw.get(self.conv_dw, self.bn2)
            y = ops.dwconv3x3(y, w, b, self.conv_dw.stride[0], L.ACT_RELU6)
            return ops.conv2d([y], self._pk3.get_folded(self.conv_pwl, self.bn3), add=x if self.res else None)
        y = nn.functional.relu6(self.bn1(self.conv_pw(x)))
        y = nn.functional.relu6(self.bn2(G.module_dwconv(self.conv_dw, y)))
        y = self.bn3(self.conv_pwl(y))
        return x + y if self.res else y


class MobileNetV2Trunk(nn.Module):
    """conv_stem/bn1/act1/blocks[0..6] of mobilenetv2_100 (what extractor.py:332-342 slices)."""

    CFG = [("ds", 1, 1, 16), ("ir", 2, 2, 24), ("ir", 3, 2, 32), ("ir", 4, 2, 64), ("ir", 3, 1, 96),
           ("ir", 3, 2, 160), ("ir", 1, 1, 320)]

    def __init__(self):
        super().__init__()
        self.conv_stem = nn.Conv2d(3, 32, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(32)
        self.act1 = nn.ReLU6()
        stages, cin = [], 32
        for kind, reps, stride, cout in self.CFG:
            blk = []
            for j in range(reps):
                s = stride if j == 0 else 1
                blk.append(_DSConv(cin, cout, s) if kind == "ds" else _InvRes(cin, cout, s))
                cin = cout
            stages.append(nn.Sequential(*blk))
        self.blocks = nn.Sequential(*stages)


def default_backbone_factory():
    return MobileNetV2Trunk()


class Feature(nn.Module):
    """MobileNetV2 encoder + Conv2x_IN decoder -> [x4 (48ch), x8 (64), x16 (192), x32 (160)]
    (extractor.py:327-361).  `backbone_factory` stands in for timm.create_model."""

    backbone_factory = staticmethod(default_backbone_factory)

    def __init__(self):
        super().__init__()
        model = type(self).backbone_factory()
        chans = [16, 24, 32, 96, 160]
        self.conv_stem, self.bn1, self.act1 = model.conv_stem, model.bn1, model.act1
        b = model.blocks
        self.block0 = nn.Sequential(*b[0:1])
        self.block1 = nn.Sequential(*b[1:2])
        self.block2 = nn.Sequential(*b[2:3])
        self.block3 = nn.Sequential(*b[3:5])
        self.block4 = nn.Sequential(*b[5:6])
        self.deconv32_16 = Conv2x_IN(chans[4], chans[3], deconv=True, concat=True)
        self.deconv16_8 = Conv2x_IN(chans[3] * 2, chans[2], deconv=True, concat=True)
        self.deconv8_4 = Conv2x_IN(chans[2] * 2, chans[1], deconv=True, concat=True)
        self.conv4 = BasicConv_IN(chans[1] * 2, chans[1] * 2, kernel_size=3, stride=1, padding=1)

    def forward(self, x, on_stage=None):
        """on_stage (optional callable(name)): called after each stage of the trunk ("stem", "block0" .. "block4", "deconv32_16",
        "deconv16_8") — the inference schedule records events there (the other pre-loop branch may wait for one of them)."""
        stage = on_stage or (lambda name: None)
        if _fused_ok(x, self) and isinstance(self.bn1, nn.BatchNorm2d):
            if not hasattr(self, "_f_stem"):
                self._f_stem = ops.FoldedConv()
            c = self.conv_stem
            if (c.kernel_size == (3, 3) and c.padding == (1, 1) and c.stride in ((1, 1), (2, 2)) and c.dilation == (1, 1) and c.groups == 1
                    and c.in_channels <= 8 and c.out_channels % 8 == 0 and isinstance(self.act1, nn.ReLU6)):
                # 3 -> 32, stride 2 with the folded BatchNorm and ReLU6: one HBM-bound launch instead of MIOpen + bias + clamp passes
                if not hasattr(self, "_f_stem_few"):
                    self._f_stem_few = ops.FoldedConv("c2d")
                wp, b = self._f_stem_few.get(c, self.bn1)
                x = ops.conv3x3_few(x.contiguous(), wp, b, stride=c.stride[0], act=L.ACT_RELU6)
            else:
                w, b = self._f_stem.get(self.conv_stem, self.bn1)
                x = nn.functional.conv2d(x, w, b, self.conv_stem.stride, self.conv_stem.padding).clamp_(0.0, 6.0)
        else:
            x = self.act1(self.bn1(self.conv_stem(x)))
        stage("stem")
        x2 = self.block0(x)
        stage("block0")
        x4 = self.block1(x2)
        stage("block1")
        x8 = self.block2(x4)
        stage("block2")
        x16 = self.block3(x8)
        stage("block3")
        x32 = self.block4(x16)
        stage("block4")
        x16 = self.deconv32_16(x32, x16)
        stage("deconv32_16")
        x8 = self.deconv16_8(x16, x8)
        stage("deconv16_8")
        x4 = self.conv4(self.deconv8_4(x8, x4))
        return [x4, x8, x16, x32]
