"""LIIF-style continuous implicit upsampler — HIP-backed mirror of the live branches of
models/*/liif.py (MLP :9-25, make_coord :32-45, liif_feat_multiscale_train :108-137,
AffinityFeature :417-446, StructureFeature 'with_v2ISU' :496-499,
liif_out_multi_scale_Training :575-678).

The default configuration (unfold_similarity='with_v2ISU', pos_dim=0) takes the fused kernels; every other option set the
reference can run ('with_ISU', 'with_1_4ISU', 'with_embed_ISU', 'only_ISU', 'only_unfold', no structure feature, Fourier
position encoding fixed or learned, cell decoding, quarter-nearest sampling, three inputs) takes the general latent builder
(csrc/liif_variants.hip), pinned by tests/golden/liif_variants.npz.  Option sets the reference itself cannot run build the same
parameters and fail at forward (see _DEAD).

Data layout: the per-query latent is built CHANNEL-major, latent[B, 228, Q], so that
  * the gather kernel's stores are coalesced along Q,
  * the MLP is a chain of 1x1 convs on the fp32-MFMA implicit-GEMM kernel, and
  * the last layer's output already is the reference's return layout [B, 9, Q].
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import grad as G
from .. import ops
from ..harness.timing import scope


def make_coord(shape, ranges=None, flatten=True):
    """Cell-centre coordinates of a grid (liif.py:32-45); same fp32 evaluation order as the reference."""
    seqs = []
    for i, n in enumerate(shape):
        v0, v1 = (-1, 1) if ranges is None else ranges[i]
        r = (v1 - v0) / (2 * n)
        seqs.append(v0 + r + (2 * r) * torch.arange(n).float())
    ret = torch.stack(torch.meshgrid(*seqs, indexing="ij"), dim=-1)
    return ret.view(-1, ret.shape[-1]) if flatten else ret


class MLP(nn.Module):
    """Linear/ReLU stack with the reference's parameter names (`layers.0.weight`, `layers.2.weight`, ...)."""

    def __init__(self, in_dim, out_dim, hidden_list):
        super().__init__()
        layers, last = [], in_dim
        for hdim in hidden_list:
            layers += [nn.Linear(last, hdim), nn.ReLU()]
            last = hdim
        layers.append(nn.Linear(last, out_dim))
        self.layers = nn.Sequential(*layers)
        self._packs = [ops.PackedConv() for m in self.layers if isinstance(m, nn.Linear)]

    def forward_cm(self, x_cm: torch.Tensor, start: int = 0) -> torch.Tensor:
        """x_cm [B, C, Q] channel-major = input of Linear layer `start` -> [B, out_dim, Q]."""
        b, c, q = x_cm.shape
        x = x_cm.view(b, c, 1, q)
        lin = [m for m in self.layers if isinstance(m, nn.Linear)]
        for i, (m, pk) in enumerate(zip(lin, self._packs)):
            if i < start:
                continue
            act = L.ACT_RELU if i + 1 < len(lin) else L.ACT_NONE
            x = ops.conv2d([x], pk.get([m.weight], [m.bias]), act=act)
        return x.view(b, -1, q)

    def forward(self, x):
        """Reference contract: [..., in_dim] -> [..., out_dim] (liif.py:22-25)."""
        shape = x.shape[:-1]
        flat = x.reshape(1, -1, x.shape[-1]).float()
        y = self.forward_cm(flat.permute(0, 2, 1).contiguous())
        return y.permute(0, 2, 1).reshape(*shape, -1)


# Option sets the REFERENCE cannot run (tests/golden/liif_variants.json records how it fails); they construct — state-dict
# parity — and fail at forward with the reference's failure named, instead of computing something the reference never did.
_DEAD = {
    "dilated": "AffinityFeature with dilation > 1: the reference pads by win_w//2 = 1 whatever the dilation, its Unfold then "
               "yields fewer than H*W windows and the reshape at liif.py:437-438 raises (all '*Dila_*', 'with_1_43*', "
               "'with_3v2ISU' modes)",
    "pos_enconding_new": "pos_enconding_new: PositionEncoder is built with enc_dims = 2 (liif.py:591-592 after :588), its "
                         "frequency table is empty and the projection at liif.py:274 raises",
}


class AffinityFeature(nn.Module):
    """3x3 cosine affinity to the 8 neighbours (liif.py:417-446).  Dilation 1 is the only window the reference can evaluate."""

    def __init__(self, win_h, win_w, dilation, cut):
        super().__init__()
        if (win_h, win_w) != (3, 3):
            raise NotImplementedError("AffinityFeature: only the 3x3 window (lsp_width = lsp_height = 3) is built")
        self.win_w, self.win_h, self.dilation, self.cut = win_w, win_h, dilation, 0
        self._padding = win_w // 2

    def forward(self, feature):
        if self.dilation != 1:
            raise RuntimeError(_DEAD["dilated"])
        x = feature.float().contiguous()
        if G.needs_grad(x):
            return G.StructureFeatureLive.apply(x, False)
        return ops.structure_feature(x)[:, x.shape[1]:]


def convbn(in_planes, out_planes, kernel_size, stride, pad, dilation):
    """conv (no bias) + BatchNorm2d with the reference's layout `0` / `1` (liif.py:27-30)."""
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride,
                                   padding=dilation if dilation > 1 else pad, dilation=dilation, bias=False),
                         nn.BatchNorm2d(out_planes))


class StructureFeature(nn.Module):
    """liif.py:448-572.  Modes the reference can run: 'with_ISU' / 'with_1_4ISU' cat(x, Affi1(x)); 'with_v2ISU' (default)
    cat(x, Affi1(x.detach())); 'with_embed_ISU' sfc_embeding(cat(x, Affi1(x.detach()))); 'only_ISU' Affi1(x).  The dilated
    modes build the reference's sub-modules (same state-dict keys) and raise at forward like the reference does."""

    def __init__(self, affinity_settings, unfold, input_chanels):
        super().__init__()
        self.win_w, self.win_h = affinity_settings["win_w"], affinity_settings["win_h"]
        self.dilation, self.unfold = affinity_settings["dilation"], unfold
        in_c = self.win_w * self.win_h - 1
        d = self.dilation
        self.Affi1 = AffinityFeature(self.win_h, self.win_w, d[0], 0)
        relu_bn = lambda ci, co: nn.Sequential(convbn(ci, co, 1, 1, 0, 1), nn.ReLU(inplace=True))  # noqa: E731
        if "Dila_ISU" in unfold:
            for i in (2, 3, 4):
                setattr(self, f"Affi{i}", AffinityFeature(self.win_h, self.win_w, d[i - 1], 0))
            for i in (1, 2, 3, 4):
                setattr(self, f"sfc_conv{i}", relu_bn(in_c, in_c))
        elif "Dila_3ISU" in unfold:
            self.sfc_embeding = convbn(input_chanels, input_chanels // 4, 1, 1, 0, 1)
            self.Affi2 = AffinityFeature(self.win_h, self.win_w, d[1], 0)
            self.Affi3 = AffinityFeature(self.win_h, self.win_w, d[2], 0)
        elif "Dila_2ISU" in unfold:
            self.sfc_embeding = convbn(input_chanels, input_chanels // 4, 1, 1, 0, 1)
            self.Affi2 = AffinityFeature(self.win_h, self.win_w, d[1], 0)
        elif "with_1_43ISU" in unfold:
            self.Affi2 = AffinityFeature(self.win_h, self.win_w, d[1], 0)
            self.Affi3 = AffinityFeature(self.win_h, self.win_w, d[2], 0)
            for i in (1, 2, 3):
                setattr(self, f"sfc_conv{i}", relu_bn(in_c, in_c // 2))
        elif "with_1_43v2ISU" in unfold or "with_3v2ISU" in unfold:
            self.Affi2 = AffinityFeature(self.win_h, self.win_w, d[1], 0)
            self.Affi3 = AffinityFeature(self.win_h, self.win_w, d[2], 0)
        elif "with_embed_ISU" in unfold:
            self.sfc_embeding = convbn(input_chanels + 8, input_chanels + 8, 1, 1, 0, 1)

    def forward(self, x):
        u = self.unfold
        x = x.float().contiguous()
        grad = G.needs_grad(x)
        if "with_ISU" in u or "with_1_4ISU" in u:
            return G.StructureFeatureLive.apply(x, True) if grad else ops.structure_feature(x)
        if "with_v2ISU" in u:
            return G.StructureFeature.apply(x) if grad else ops.structure_feature(x)
        if "with_embed_ISU" in u:
            conv, bn = self.sfc_embeding[0], self.sfc_embeding[1]
            if grad or bn.training or G.needs_grad(conv.weight):
                return bn(G.module_conv2d(self, "embed", conv, G.StructureFeature.apply(x) if grad else ops.structure_feature(x)))
            if not hasattr(self, "_pk_embed"):
                self._pk_embed = ops.PackedConv()
            return ops.conv2d([ops.structure_feature(x)], self._pk_embed.get_folded(conv, bn))
        if "only_ISU" in u:
            return self.Affi1(x)
        if any(k in u for k in ("Dila_", "with_1_43", "with_3v2ISU")):
            raise RuntimeError(_DEAD["dilated"])
        return x  # liif.py:492-572 falls through to `return x` for any other string


class SpatialEncoding(nn.Module):
    """Fourier features of the relative coordinate (liif.py:339-370): x -> cat(x, sin(x emb^T), cos(x emb^T)) with
    emb = per-axis frequencies 2^linspace(0, sigma, n); a Parameter (`emb`) when require_grad.  On the hot path the encoding is
    evaluated inside the latent builder (ops.liif_latent); `forward` is the reference's tensor-level contract."""

    def __init__(self, in_dim, out_dim, sigma=6, cat_input=True, require_grad=False):
        super().__init__()
        assert out_dim % (2 * in_dim) == 0, "dimension must be dividable"
        # frequency table [in_dim * n, in_dim]: block d holds the n frequencies 2^linspace(0, sigma, n) in column d, zeros
        # elsewhere — every row looks at ONE coordinate (evaluated in fp64 like the reference's numpy table, then fp32)
        n = out_dim // 2 // in_dim
        freqs = torch.pow(torch.tensor(2.0, dtype=torch.float64), torch.linspace(0, sigma, n, dtype=torch.float64))
        table = torch.zeros((in_dim * n, in_dim), dtype=torch.float64)
        for d in range(in_dim):
            table[d * n:(d + 1) * n, d] = freqs
        self.emb = table.to(torch.float32)
        if require_grad:
            self.emb = nn.Parameter(self.emb, requires_grad=True)
        self.in_dim, self.out_dim, self.sigma, self.cat_input, self.require_grad = in_dim, out_dim, sigma, cat_input, require_grad

    def table(self, device):
        if not self.require_grad and self.emb.device != device:
            self.emb = self.emb.to(device)  # the reference moves the plain tensor on first use (liif.py:360-361)
        return self.emb

    def forward(self, x):
        if self.in_dim != 2 or not self.cat_input:
            raise NotImplementedError("SpatialEncoding: only the 2-D, cat_input form the upsampler uses is built")
        shape = x.shape[:-1]
        rel = x.reshape(1, -1, 2).float().contiguous()
        q = rel.shape[1]
        emb = self.table(x.device).contiguous()
        dummy = torch.zeros((1, 1, 1, 1), device=x.device, dtype=torch.float32)
        if G.needs_grad(emb):
            raise NotImplementedError("SpatialEncoding.forward under autograd: use the upsampler (grad.LiifLatent)")
        # the latent builder recomputes rel from coordinates; here rel is GIVEN, so feed it as the coordinate of a 1x1 map whose
        # cell centre is 0 and whose size is 1: rel' = (x - 0) * 1
        lat = torch.empty((1, 1 + 2 + 2 * emb.shape[0], q), device=x.device, dtype=torch.float32)
        ops.liif_latent(dummy, rel, lat, 0, emb=emb)
        return lat[0, 1:].t().reshape(*shape, -1)


class PositionEncoder(nn.Module):
    """liif.py:178-337, as the upsampler constructs it (posenc_type='sinusoid', head=8): `proj` keeps the state-dict keys; the
    reference's forward cannot run in that construction, see _DEAD."""

    def __init__(self, posenc_type=None, complex_transform=False, posenc_scale=6, gauss_scale=1, in_dims=2, enc_dims=256,
                 hidden_dims=32, head=1, gamma=1):
        super().__init__()
        if posenc_type != "sinusoid":
            raise NotImplementedError("PositionEncoder: only the 'sinusoid' construction of the upsampler is mirrored")
        self.posenc_type, self.enc_dims, self.head = posenc_type, enc_dims, head
        self.proj = nn.Linear(enc_dims, head)

    def forward(self, positions, cells=None):
        raise RuntimeError(_DEAD["pos_enconding_new"])


def _cells(coords, scale):
    """decode_cell (liif.py:111-114): ones_like(coords) with both columns set to 2/scale — the reference's own broadcasting
    (scale [B,1] gives one value per batch element; a [B] vector only fits B = 1 or B = Q, as in the reference)."""
    cells = torch.ones_like(coords)
    cells[:, :, 0] = 2 / scale
    cells[:, :, 1] = 2 / scale
    return cells


def _latent_block(feat, coords, unfold9=False, n_samp=1, emb=None, cell=None):
    feat = feat.float().contiguous()
    coords = coords.float().contiguous()
    if G.needs_grad(feat, emb):
        return G.LiifLatent.apply(feat, coords, emb, cell, unfold9, n_samp)
    b, c = feat.shape[:2]
    lat = torch.empty((b, ops.liif_latent_width(c, unfold9, n_samp, 0 if emb is None else emb.shape[0], cell is not None),
                       coords.shape[1]), device=feat.device, dtype=torch.float32)
    ops.liif_latent(feat, coords, lat, 0, unfold9=unfold9, n_samp=n_samp, emb=emb, cell=cell)
    return lat


def liif_feat_multiscale_train(feat, coords, scale=None, local=False, cell=False):
    """(rel_coord [B,Q,2], q_feat [B,Q,C], cells | None) — reference contract of liif.py:108-137."""
    cells = _cells(coords, scale) if cell else None
    c = feat.shape[1]
    lat = _latent_block(feat, coords).permute(0, 2, 1)
    assert not local  # liif.py:136-137
    return lat[..., c:c + 2], lat[..., :c], cells


def liif_feat_multiscale_train_quater(feat, coords, scale=None, local=False, cell=False):
    """Four half-cell shifted nearest samples, rel to their centre (liif.py:140-176): (rel [B,Q,2], q_feat [B,Q,4C], cells)."""
    cells = _cells(coords, scale) if cell else None
    c4 = feat.shape[1] * 4
    lat = _latent_block(feat, coords, n_samp=4).permute(0, 2, 1)
    return lat[..., c4:c4 + 2], lat[..., :c4], cells


class liif_out_multi_scale_Training(nn.Module):
    def __init__(self, pos_dim=24, encoder_dim=256, mlphidden_list=[128, 64, 64], pos_enconding=False,
                 pos_enconding_new=False, local_ensemble=False, decode_cell=False, unfold=False, affinity_settings=None,
                 quater_nearest=None, require_grad=True, number_input=3, chanels=0):
        super().__init__()
        self.local_ensemble, self.decode_cell, self.unfold = local_ensemble, decode_cell, unfold
        self.pos_enconding, self.pos_enconding_new, self.quater_nearest = pos_enconding, pos_enconding_new, quater_nearest
        self.encoder_dim = encoder_dim
        self.pos_dim = 2
        if pos_dim != 0:  # liif.py:587-596
            if pos_enconding:
                self.pos_encoding = SpatialEncoding(2, pos_dim, require_grad=require_grad)
                self.pos_dim = pos_dim + 2
            elif pos_enconding_new:
                self.pos_encoding = PositionEncoder(posenc_type="sinusoid", posenc_scale=10, hidden_dims=self.pos_dim,
                                                    enc_dims=self.pos_dim, head=8)
                self.pos_dim = 8
            else:
                self.pos_encoding = SpatialEncoding(2, pos_dim, require_grad=require_grad)  # built, never applied
        self.outputdim = 9
        in_c = affinity_settings["win_h"] * affinity_settings["win_w"] - 1
        imnet_in_dim = encoder_dim
        self._single_sf = False
        if unfold is not None:  # liif.py:599-635
            if unfold is False or not isinstance(unfold, str):
                raise TypeError("unfold_similarity must be a mode string or None (liif.py:600 tests `in self.unfold`)")
            self._single_sf = any(k in unfold for k in ("with_1_4ISU", "with_1_43ISU", "with_1_43v2ISU"))
            if self._single_sf:
                self.to_sf_l2 = StructureFeature(affinity_settings, unfold, input_chanels=None)
            else:
                self.to_sf_l2 = nn.ModuleList(StructureFeature(affinity_settings, unfold, input_chanels=c) for c in chanels)
            table = (("only_unfold", lambda d: d * 9), ("with_1_4ISU", lambda d: d + in_c), ("with_1_43ISU", lambda d: d + (in_c // 2) * 3),
                     ("with_1_43v2ISU", lambda d: d + in_c * 3), ("with_3v2ISU", lambda d: d + in_c * 3 * number_input),
                     ("with_ISU", lambda d: d + in_c * number_input), ("with_v2ISU", lambda d: d + in_c * number_input),
                     ("with_embed_ISU", lambda d: d + in_c * number_input), ("only_ISU", lambda d: in_c * number_input),
                     ("with_Dila_ISU", lambda d: d + in_c * 4 * number_input), ("only_Dila_ISU", lambda d: in_c * 4 * number_input),
                     ("with_Dila_3ISU", lambda d: d + in_c * 3 * number_input), ("only_Dila_3ISU", lambda d: in_c * 3 * number_input),
                     ("with_Dila_2ISU", lambda d: d + in_c * 2 * number_input), ("only_Dila_2ISU", lambda d: in_c * 2 * number_input))
            for key, fn in table:
                if key in unfold:
                    imnet_in_dim = fn(imnet_in_dim)
                    break
            else:
                assert False, f"unknown unfold_similarity {unfold!r}"  # liif.py:634-635
        if quater_nearest is not None:
            self.outputdim = 4
            if "both" in quater_nearest:
                imnet_in_dim = imnet_in_dim * 4
        imnet_in_dim = imnet_in_dim + self.pos_dim * number_input
        if decode_cell:
            imnet_in_dim = imnet_in_dim + 2 * number_input
        self.imnet = MLP(imnet_in_dim, self.outputdim, hidden_list=mlphidden_list)
        # the fused / low-resolution-first-layer fast paths serve the default option set; everything else takes the general
        # latent builder (same function, csrc/liif_variants.hip)
        self._default_variant = (unfold == "with_v2ISU" and not pos_enconding and not pos_enconding_new and not decode_cell
                                 and not local_ensemble and quater_nearest is None)

    # ---- general path: any option set the reference can run ---------------------------------------------------------------
    def _structure(self, feats):
        """The per-source feature maps after the `unfold_similarity` stage (liif.py:652-660)."""
        if self.unfold is None or "only_unfold" in self.unfold:
            return [f.float().contiguous() for f in feats]
        if self._single_sf:
            return [self.to_sf_l2(f) if i == 0 else f.float().contiguous() for i, f in enumerate(feats)]
        return [sf(f) for sf, f in zip(self.to_sf_l2, feats)]

    def _mask_logits_general(self, feats, coord, scale):
        if self.local_ensemble:
            raise AssertionError("local_ensemble: liif_feat_multiscale_train ends in `assert False` for it (liif.py:136-137)")
        if self.pos_enconding_new and hasattr(self, "pos_encoding") and isinstance(self.pos_encoding, PositionEncoder):
            raise RuntimeError(_DEAD["pos_enconding_new"])
        with scope("structure_feature"):
            sfs = self._structure(feats)
        unfold9 = self.unfold is not None and "only_unfold" in self.unfold
        n_samp = 4 if (self.quater_nearest is not None and "both" in self.quater_nearest) else 1
        emb = self.pos_encoding.table(coord.device).contiguous() if (self.pos_enconding and hasattr(self, "pos_encoding")) else None
        cell = _cells(coord, scale).contiguous() if self.decode_cell else None
        widths = [ops.liif_latent_width(s.shape[1], unfold9, n_samp, 0 if emb is None else emb.shape[0], cell is not None) for s in sfs]
        if sum(widths) != self.imnet.layers[0].weight.shape[1]:
            raise RuntimeError(f"liif: inputs hold {sum(widths)} latent channels but the MLP expects {self.imnet.layers[0].weight.shape[1]}")
        lin = [m for m in self.imnet.layers if isinstance(m, nn.Linear)]
        if G.needs_grad(*sfs, emb, *self.imnet.parameters()):
            x = torch.cat([G.LiifLatent.apply(s, coord, emb, cell, unfold9, n_samp) for s in sfs], dim=1)
            for i, m in enumerate(lin):
                x = G.pointwise_linear(self, id(m), x, m, i + 1 < len(lin))
            return x
        b, q = coord.shape[:2]
        out = torch.empty((b, self.outputdim, q), device=coord.device, dtype=torch.float32)
        # slabs keep the [B,ctot,Q] latent below ~1 GB whatever the option set (only_unfold: 1876 channels)
        qmax = max(1 << 12, min(self.query_chunk, (1 << 28) // max(1, b * sum(widths))))
        for q0 in range(0, q, qmax):
            q1 = min(q, q0 + qmax)
            cs = coord if (q0 == 0 and q1 == q) else coord[:, q0:q1].contiguous()
            cl = cell if (cell is None or (q0 == 0 and q1 == q)) else cell[:, q0:q1].contiguous()
            latent = torch.empty((b, sum(widths), q1 - q0), device=coord.device, dtype=torch.float32)
            off = 0
            with scope("liif_gather"):
                for s, wd in zip(sfs, widths):
                    ops.liif_latent(s, cs, latent, off, unfold9=unfold9, n_samp=n_samp, emb=emb, cell=cl)
                    off += wd
            with scope("liif_mlp"):
                res = self.imnet.forward_cm(latent)
            if q0 == 0 and q1 == q:
                return res
            out[:, :, q0:q1] = res
        return out

    def forward(self, feats, coord, scale=None):
        """feats: list of [B,C_i,H_i,W_i]; coord [B,Q,2] (row, col) -> mask logits [B,9,Q] (liif.py:644-678)."""
        coord = coord.float().contiguous()
        b, q = coord.shape[:2]
        if not self._default_variant:
            return self._mask_logits_general(feats, coord, scale)
        if G.needs_grad(*feats, *self.imnet.parameters()):
            return self._mask_logits_train(feats, coord)
        with scope("structure_feature"):
            sfs = [sf(f) for sf, f in zip(self.to_sf_l2, feats)]
        ctot = sum(s.shape[1] + 2 for s in sfs)
        lin = [m for m in self.imnet.layers if isinstance(m, nn.Linear)]
        pre = self._first_layer_lowres(sfs, lin[0]) if (self.fused_first_layer and len(sfs) <= 2 and len(lin) > 1) else None
        # queries are processed in slabs of <= 2^20 so the [B,228,Q] latent stays below 1 GB (Middlebury-F has
        # 5.7 M queries = 5.2 GB if materialised at once, liif.py:675) and inside the kernels' 32-bit offsets
        qmax = self.query_chunk
        if q <= qmax:
            return self._mask_logits(sfs, coord, ctot, pre)
        out = torch.empty((b, self.outputdim, q), device=coord.device, dtype=torch.float32)
        for q0 in range(0, q, qmax):
            q1 = min(q, q0 + qmax)
            out[:, :, q0:q1] = self._mask_logits(sfs, coord[:, q0:q1].contiguous(), ctot, pre)
        return out

    query_chunk = 1 << 20
    # The first Linear layer commutes with the nearest gather: its feature blocks are applied once per LOW-resolution
    # pixel (two 1x1 convs) and the per-query kernel only gathers + adds (csrc/liif.hip, as_liif_gather_mlp1).
    fused_first_layer = True

    def _first_layer_lowres(self, sfs, lin0):
        if not hasattr(self, "_pk_u"):
            self._pk_u = [ops.PackedConv() for _ in range(2)]
        us, rel_cols, off = [], [], 0
        with scope("liif_mlp_lowres"):
            for s, pk in zip(sfs, self._pk_u):
                c = s.shape[1]
                us.append(ops.conv2d([s], pk.get([lin0.weight], [None], transform=lambda w, o=off, c=c: w[:, o:o + c])))
                rel_cols.append(off + c)
                off += c + 2
            # slices + cat only: index tensors would need a host->device copy, which a graph capture forbids
            wrel = torch.cat([lin0.weight.detach()[:, o:o + 2] for o in rel_cols], dim=1).float().contiguous()
        return us, wrel, None if lin0.bias is None else lin0.bias.detach().float().contiguous()

    # ---- fused inference pipeline (csrc/liif_fused.hip) -----------------------------------------------------------
    fused_tail = __import__("os").environ.get("ANYSTEREO_FUSED_LIIF", "1") != "0"

    parallel_inputs = __import__("os").environ.get("ANYSTEREO_LIIF_PARALLEL", "1") != "0"

    def _side_stream(self, device):
        streams = self.__dict__.setdefault("_streams", {})
        if device not in streams:
            streams[device] = torch.cuda.Stream(device=device)
        return streams[device]

    def fused_ok(self, feats_parts, coord) -> bool:
        """The one-kernel tail exists for the default configuration: <= 2 inputs, MLP 128-64-64-9, split-precision mode,
        inference.  Everything else takes the staged path (same function)."""
        lin = [m for m in self.imnet.layers if isinstance(m, nn.Linear)]
        return (self.fused_tail and self._default_variant and coord.is_cuda and not torch.is_grad_enabled() and len(feats_parts) in (1, 2)
                and [tuple(m.weight.shape) for m in lin[1:]] == [(64, 128), (64, 64), (9, 64)] and lin[0].weight.shape[0] == 128
                and ops.get_precision() == "split" and all(len(ps) <= 2 for ps in feats_parts)
                and all(p.shape[1] % 16 == 0 for ps in feats_parts for p in ps))

    fused_train_mlp = __import__("os").environ.get("ANYSTEREO_LIIF_TRAIN_FUSED", "1") != "0"
    early_static = __import__("os").environ.get("ANYSTEREO_LIIF_EARLY_STATIC", "1") != "0"
    # Opt-in (ANYSTEREO_LIIF_DIRECT=1), measured and NOT kept as the default: the second input (stem_2x, 32 + 8 channels at 1/2
    # resolution) handed to the tail as RAW channels-last rows, its first-layer product taken per query there
    # (ops.liif_tail(direct1=...)).  It removes that input's 128-channel first-layer table (66.8 MB at 960x540, written once and
    # gathered ~2.3 x; the raw rows are 25 MB) — and is SLOWER: the whole upsampler 180 vs 158 us at cfg 2, 405 vs 339 us at
    # cfg 3, 1276 vs 1064 us at cfg 5 (hipGraph replays, tools/kbench_liif.py): the tail is bound by its per-tile instruction
    # stream (36 more MFMAs + three operand splits per tile, 20 spilled registers at 8 waves around one 90 KB LDS image), not by
    # the bytes it gathers.  Same results (tests/test_hip_parity.py::test_liif_tail_direct_second_input).
    direct_second_input = __import__("os").environ.get("ANYSTEREO_LIIF_DIRECT", "0") == "1"

    def _direct_slot(self, slot: int, c: int) -> bool:
        return self.direct_second_input and slot == 1 and c <= 48

    def _input_rows(self, slot, parts, aff, pk):
        """Rows of LIIF input `slot` for the tail: its first-layer product at low resolution, or (direct slot) its raw channels."""
        if self._direct_slot(slot, sum(p_.shape[1] for p_ in parts) + 8):
            with scope("liif_rows"):
                return ops.liif_rows_cl(parts + [aff])
        with scope("liif_mlp_lowres"):
            return ops.liif_lowres_cl(parts + [aff], pk)

    def precompute_static(self, feats_parts, slot, stream, coord=None):
        """Affinity + first MLP layer at low resolution of input `slot`, whose maps do not change during the GRU loop (stem_2x):
        issued on `stream` BEFORE the loop, so the post-loop upsampler only has the hidden-state input's chain in front of the
        tail kernel.  feats_parts as in `upsample_fused` (the other inputs are read for their channel counts only).  The result
        is picked up by the next `upsample_fused` call that sees the same tensors; `clear_static()` drops it."""
        self.__dict__.pop("_early_static", None)
        # only when the fused tail will pick the rows up (fused_ok's conditions that do not depend on the coordinates)
        if not (self.early_static and self.fused_tail and self._default_variant and 0 < slot < len(feats_parts) <= 2
                and not torch.is_grad_enabled() and ops.get_precision() == "split"
                and all(p_.is_cuda and p_.shape[1] % 16 == 0 for ps in feats_parts for p_ in ps)):
            return
        lin = [m for m in self.imnet.layers if isinstance(m, nn.Linear)]
        w1 = lin[0].weight
        if not hasattr(self, "_lowres_packs"):
            self._lowres_packs = [ops.LiifLowresPack() for _ in range(2)]
        o_ = sum(sum(p_.shape[1] for p_ in ps) + 8 + 2 for ps in feats_parts[:slot])
        src = list(feats_parts[slot])
        parts = [p_.float().contiguous() for p_ in src]
        c = sum(p_.shape[1] for p_ in parts) + 8
        pk = self._lowres_packs[slot]
        pk.get(w1, o_, c)  # built on the current stream, before the branch
        main = torch.cuda.current_stream(parts[0].device)
        stream.wait_stream(main)
        with torch.cuda.stream(stream):
            with scope("structure_feature"):
                aff = ops.liif_affinity(parts)
            u = self._input_rows(slot, parts, aff, pk.get(w1, o_, c))
            # the tail's query-order hint (ops.liif_query_rows) depends on the coordinates alone: found here, off the tail's path
            row_len = None
            if (coord is not None and ops.PATCH_ORDER and coord.is_cuda and coord.dtype == torch.float32 and coord.is_contiguous()
                    and coord.dim() == 3):
                row_len = ops.liif_query_rows(coord)
            done = torch.cuda.Event()
            done.record(stream)
        for t_ in parts:
            t_.record_stream(stream)
        self.__dict__["_early_static"] = {"slot": slot, "src": src, "parts": parts, "u": u, "done": done, "pack": pk.get(w1, o_, c),
                                          "row_len": row_len, "coord": coord if row_len is not None else None}

    def clear_static(self):
        """Drop the early rows; JOINS their branch into the current stream (a forked branch that nobody waited for would leave a
        hipGraph capture with unjoined work — e.g. when the upsampler took the staged path after all)."""
        early = self.__dict__.pop("_early_static", None)
        if early is not None:
            torch.cuda.current_stream(early["u"].device).wait_event(early["done"])

    def upsample_fused(self, feats_parts, coord, disp, scale_vec, want_logits=False):
        """feats_parts: per LIIF input the list of NCHW tensors whose channel concat is that input (e.g. [[stem_4x, net0],
        [stem_2x]]: the concat of continuous_IGEVstereo.py:195 is never materialised).  coord [B,Q,2] is clamped IN PLACE
        (submodule.py:366).  -> disp_up [B,1,Q] (and the mask logits [B,9,Q], `forward`'s return value, when asked)."""
        lin = [m for m in self.imnet.layers if isinstance(m, nn.Linear)]
        w1 = lin[0].weight
        if not hasattr(self, "_lowres_packs"):
            self._lowres_packs = [ops.LiifLowresPack() for _ in range(2)]
        us, sizes, rel_cols, off = [], [], [], 0
        # the per-input chains (affinity -> first layer at low resolution) are independent, small and latency-bound: the
        # second input's chain runs on a side stream (a parallel branch of the captured graph) and joins before the tail
        main = torch.cuda.current_stream(coord.device)
        side = self._side_stream(coord.device) if (self.parallel_inputs and len(feats_parts) > 1) else None
        # everything the chains read is made ready on `main` first (input conversions, weight packs: a pack is built by a kernel
        # on the current stream the first time it is asked for) ...
        prep, o_ = [], 0
        for parts, lpk in zip(feats_parts, self._lowres_packs):
            parts = [p_.float().contiguous() for p_ in parts]
            c = sum(p_.shape[1] for p_ in parts) + 8
            prep.append((parts, c, lpk.get(w1, o_, c)))
            o_ += c + 2
        # ... and the branch point is HERE, before the first chain is issued (a side.wait_stream(main) issued after it made the
        # second chain wait for the first: the two ran back to back, 168 us instead of ~125)
        fork = main.record_event() if side is not None else None
        early = self.__dict__.get("_early_static")  # the loop-invariant input's rows, computed beside the GRU loop
        for i, (parts, c, pk) in enumerate(prep):
            if (early is not None and i == early["slot"] and len(parts) == len(early["parts"])
                    and all(a is b_ for a, b_ in zip(early["src"], feats_parts[i])) and early["pack"] is pk):
                main.wait_event(early["done"])
                early["joined"] = True
                early["u"].record_stream(main)
                us.append(early["u"])
                sizes.append(tuple(parts[0].shape[2:]))
                rel_cols.append(off + c)
                off += c + 2
                continue
            on_side = side is not None and i == 1
            if on_side:
                side.wait_event(fork)
            with torch.cuda.stream(side if on_side else main):
                with scope("structure_feature"):
                    aff = ops.liif_affinity(parts)
                us.append(self._input_rows(i, parts, aff, pk))
            if on_side:
                for t_ in parts:
                    t_.record_stream(side)
            sizes.append(tuple(parts[0].shape[2:]))
            rel_cols.append(off + c)
            off += c + 2
        if side is not None:
            main.wait_stream(side)
            us[1].record_stream(main)
        if off != w1.shape[1]:
            raise RuntimeError(f"liif: inputs hold {off} latent channels but the MLP expects {w1.shape[1]}")
        if not hasattr(self, "_tail_pack"):
            self._tail_pack = ops.LiifTailPack()
        pack = self._tail_pack.get(lin, rel_cols)
        direct1 = prep[1][2] if (len(prep) > 1 and self._direct_slot(1, prep[1][1])) else None
        row_len = None
        if early is not None and early.get("joined") and early.get("coord") is coord and early.get("row_len") is not None:
            row_len = early["row_len"]  # its branch was joined above (main.wait_event(early["done"]))
            row_len.record_stream(main)
        elif ops.PATCH_ORDER:
            row_len = ops.liif_query_rows(coord)
        with scope("liif_tail"):
            return ops.liif_tail(us[0], us[1] if len(us) > 1 else None, sizes, coord, pack, disp, scale_vec,
                                 clamp_inplace=True, want_logits=want_logits, direct1=direct1, row_len=row_len)

    def _mask_logits_train(self, feats, coord):
        """Differentiable form (liif.py:652-678).  With <= 2 sources the first Linear layer is applied at LOW resolution
        (two library 1x1 convs under autograd) and the per-query stage is the fused HIP gather + add + ReLU with its HIP
        scatter-add backward (grad.LiifGatherMlp1); otherwise the latent [B,228,Q] is gathered per source.  The remaining
        layers run forward and dgrad as 1x1 convs on the implicit-GEMM kernel (grad.PointwiseLinear).
        The training loop upsamples EVERY iteration's disparity (train_continuous_IGEV.py:219 needs all predictions); of the
        feature maps only the first (stem_4x | hidden state) changes between iterations, so the structure feature and the
        low-resolution first layer of the others (stem_2x) are computed once per forward and reused — same values, the
        gradients of the reuses are summed by autograd."""
        lin = [m for m in self.imnet.layers if isinstance(m, nn.Linear)]
        lowres = self.fused_first_layer and len(feats) <= 2 and len(lin) > 1
        cache = self.__dict__.get("_train_static")  # installed per forward by the model's GRU loop; None = no reuse
        w1, off, sfs, us, rel_cols, rel_idx = lin[0].weight, 0, [], [], [], []
        # the one-kernel MLP (ANYSTEREO_LIIF_TRAIN_FUSED=0: layer by layer) serves the default two-input option set in split mode
        fused = (self.fused_train_mlp and lowres and len(feats) == 2 and coord.is_cuda and ops.get_precision() == "split"
                 and lin[0].weight.shape[0] == 128 and [tuple(m.weight.shape) for m in lin[1:]] == [(64, 128), (64, 64), (9, 64)]
                 and coord.shape[0] * 128 * coord.shape[1] * 4 < 0x7FFFFFF0)
        for i, (sf, f) in enumerate(zip(self.to_sf_l2, feats)):
            ent = cache.get(i) if (cache is not None and i > 0) else None
            if ent is not None and ent[0] is f and ent[1] == (f._version, w1._version, torch.is_grad_enabled()):
                s, u = ent[2], ent[3]
            else:
                s = sf(f)
                u = self._lowres_first_layer(i, s, w1[:, off:off + s.shape[1], None, None]) if lowres else None
                if cache is not None and i > 0:
                    cache[i] = (f, (f._version, w1._version, torch.is_grad_enabled()), s, u)
            nb = coord.shape[0]
            if s.shape[0] != nb:
                # an input shared by the n batched evaluations of one forward (models/base.py::_upsample_batched hands stem_2x
                # over ONCE, batch B, beside n*B hidden states): its structure feature and low-resolution first layer are computed
                # at batch B and the rows repeated — autograd sums the n uses' gradients in repeat's backward (the fused MLP
                # reads element b % B instead of a repeated copy)
                if nb % s.shape[0]:
                    raise RuntimeError(f"liif: input {i} has batch {s.shape[0]}, the queries {nb}")
                k = nb // s.shape[0]
                if not (fused and i == 1):
                    s = s.repeat(k, 1, 1, 1) if not lowres else s
                    u = u.repeat(k, 1, 1, 1) if u is not None else None
            sfs.append(s)
            us.append(u)
            rel_cols.append(w1[:, off + s.shape[1]:off + s.shape[1] + 2])
            rel_idx.append(off + s.shape[1])
            off += s.shape[1] + 2
        if fused:
            # gather + first-layer finish + the three remaining layers as ONE forward kernel; its backward recomputes the
            # activations per query tile (grad.LiifMlpTail) — no [B,128|64,Q] tensor is written by the forward or saved
            toks, stashes = [], []
            for m in lin[1:]:
                w_, b_, st = G.anchored(self, id(m), "linear", (m.weight,), (m.bias,))
                toks += [w_, b_]
                stashes.append(st)
            if not hasattr(self, "_tail_pack"):
                self._tail_pack = ops.LiifTailPack()
            if not hasattr(self, "_tail_pack_t"):
                self._tail_pack_t = ops.LiifMlpBwdPack()
            return G.LiifMlpTail.apply(us[0].contiguous(), us[1].contiguous(), coord, torch.cat(rel_cols, dim=1).contiguous(), lin[0].bias,
                                       *toks, self._tail_pack.get(lin, rel_idx), self._tail_pack_t, stashes)
        if lowres:
            x = G.LiifGatherMlp1.apply(us[0].contiguous(), us[1].contiguous() if len(us) > 1 else None, coord,
                                       torch.cat(rel_cols, dim=1).contiguous(), lin[0].bias)
            lin = lin[1:]
        else:
            x = torch.cat([G.LiifGather.apply(s, coord) for s in sfs], dim=1)
        for i, m in enumerate(lin):
            x = G.pointwise_linear(self, id(m), x, m, i + 1 < len(lin))
        return x

    def _lowres_first_layer(self, i, s, w):
        """The first Linear layer's feature block of input i at low resolution under autograd: a 1x1 convolution on this library's
        kernels (forward, dgrad, wgrad — grad.Conv2dSame; the weight is a column block of the layer's matrix, autograd carries
        its gradient back into it), else the library's."""
        if not s.is_cuda:
            raise RuntimeError(f"anystereo liif first layer: expected a CUDA tensor (the hot path has no CPU fallback), got {s.device}")
        if s.shape[1] >= 16:  # fp16 / bf16 inputs (autocast training) are cast for the kernel: fp32 result, as the MLP expects
            return G.conv2d_same(self, f"u{i}", s if s.dtype == torch.float32 else s.float(), w.float().contiguous(), None)
        return F.conv2d(s.float(), w.float())  # < 16 input channels (non-default option sets): below the MFMA kernel's K granularity

    def _mask_logits(self, sfs, coord, ctot, pre=None):
        b, q = coord.shape[:2]
        if pre is not None:
            us, wrel, b1 = pre
            with scope("liif_gather"):
                h1 = ops.liif_gather_mlp1(us[0], us[1] if len(us) > 1 else None, coord, wrel, b1)
            with scope("liif_mlp"):
                return self.imnet.forward_cm(h1, start=1)
        latent = torch.empty((b, ctot, q), device=coord.device, dtype=torch.float32)
        off = 0
        with scope("liif_gather"):
            for s in sfs:
                ops.liif_gather(s, coord, latent, off)
                off += s.shape[1] + 2
        with scope("liif_mlp"):
            return self.imnet.forward_cm(latent)
