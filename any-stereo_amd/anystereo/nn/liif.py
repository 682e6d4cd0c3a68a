"""LIIF-style continuous implicit upsampler — HIP-backed mirror of the live branches of
models/*/liif.py (MLP :9-25, make_coord :32-45, liif_feat_multiscale_train :108-137,
AffinityFeature :417-446, StructureFeature 'with_v2ISU' :496-499,
liif_out_multi_scale_Training :575-678).

Only the default-configuration branch is implemented (unfold_similarity='with_v2ISU', pos_dim=0,
no positional encoding / cell decode / local ensemble / quarter-nearest); any other option raises
at construction (SURVEY.md §2 row 5 lists them as out of scope).

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


class AffinityFeature(nn.Module):
    """3x3 cosine affinity (liif.py:417-446); dilation 1 only (the 'with_v2ISU' branch uses Affi1)."""

    def __init__(self, win_h, win_w, dilation, cut):
        super().__init__()
        if (win_h, win_w, dilation) != (3, 3, 1):
            raise NotImplementedError("AffinityFeature: only the 3x3, dilation-1 window of the default config is built")
        self.win_w, self.win_h, self.dilation, self.cut = win_w, win_h, dilation, 0

    def forward(self, feature):
        x = feature.float().contiguous()
        return ops.structure_feature(x)[:, x.shape[1]:]


class StructureFeature(nn.Module):
    """cat(x, Affi1(x.detach())) — the 'with_v2ISU' branch (liif.py:496-499)."""

    def __init__(self, affinity_settings, unfold, input_chanels):
        super().__init__()
        if unfold != "with_v2ISU":
            raise NotImplementedError(f"StructureFeature: unfold_similarity={unfold!r} is not built (default: 'with_v2ISU')")
        self.win_w, self.win_h = affinity_settings["win_w"], affinity_settings["win_h"]
        self.dilation, self.unfold = affinity_settings["dilation"], unfold
        self.Affi1 = AffinityFeature(self.win_h, self.win_w, self.dilation[0], 0)

    def forward(self, x):
        x = x.float().contiguous()
        return G.StructureFeature.apply(x) if G.needs_grad(x) else ops.structure_feature(x)


def liif_feat_multiscale_train(feat, coords, scale=None, local=False, cell=False):
    """(rel_coord [B,Q,2], q_feat [B,Q,C], None) — reference contract of liif.py:108-137."""
    if local or cell:
        raise NotImplementedError("local ensemble / cell decoding are not built (off in the default config)")
    feat = feat.float().contiguous()
    coords = coords.float().contiguous()
    b, c = feat.shape[:2]
    q = coords.shape[1]
    if G.needs_grad(feat):
        lat = G.LiifGather.apply(feat, coords)
    else:
        lat = torch.empty((b, c + 2, q), device=feat.device, dtype=torch.float32)
        ops.liif_gather(feat, coords, lat, 0)
    lat = lat.permute(0, 2, 1)
    return lat[..., c:], lat[..., :c], None


class liif_out_multi_scale_Training(nn.Module):
    def __init__(self, pos_dim=24, encoder_dim=256, mlphidden_list=[128, 64, 64], pos_enconding=False,
                 pos_enconding_new=False, local_ensemble=False, decode_cell=False, unfold=False, affinity_settings=None,
                 quater_nearest=None, require_grad=True, number_input=3, chanels=0):
        super().__init__()
        unsupported = {"pos_dim": pos_dim != 0, "pos_enconding": pos_enconding, "pos_enconding_new": pos_enconding_new,
                       "local_ensemble": local_ensemble, "decode_cell": decode_cell,
                       "quater_nearest": quater_nearest is not None, "unfold": unfold != "with_v2ISU"}
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"liif_out_multi_scale_Training: non-default options {bad} are not built")
        self.local_ensemble, self.decode_cell, self.unfold = local_ensemble, decode_cell, unfold
        self.pos_enconding, self.pos_enconding_new, self.quater_nearest = pos_enconding, pos_enconding_new, quater_nearest
        self.encoder_dim = encoder_dim
        self.pos_dim = 2
        self.outputdim = 9
        in_c = affinity_settings["win_h"] * affinity_settings["win_w"] - 1
        self.to_sf_l2 = nn.ModuleList(StructureFeature(affinity_settings, unfold, input_chanels=c) for c in chanels)
        imnet_in_dim = encoder_dim + in_c * number_input + self.pos_dim * number_input
        self.imnet = MLP(imnet_in_dim, self.outputdim, hidden_list=mlphidden_list)

    def forward(self, feats, coord, scale=None):
        """feats: list of [B,C_i,H_i,W_i]; coord [B,Q,2] (row, col) -> mask logits [B,9,Q] (liif.py:644-678)."""
        coord = coord.float().contiguous()
        b, q = coord.shape[:2]
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
        return (self.fused_tail and coord.is_cuda and not torch.is_grad_enabled() and len(feats_parts) in (1, 2)
                and [tuple(m.weight.shape) for m in lin[1:]] == [(64, 128), (64, 64), (9, 64)] and lin[0].weight.shape[0] == 128
                and ops.get_precision() == "split" and all(len(ps) <= 2 for ps in feats_parts)
                and all(p.shape[1] % 16 == 0 for ps in feats_parts for p in ps))

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
        for i, (parts, c, pk) in enumerate(prep):
            on_side = side is not None and i == 1
            if on_side:
                side.wait_event(fork)
            with torch.cuda.stream(side if on_side else main):
                with scope("structure_feature"):
                    aff = ops.liif_affinity(parts)
                with scope("liif_mlp_lowres"):
                    us.append(ops.liif_lowres_cl(parts + [aff], pk))
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
        with scope("liif_tail"):
            return ops.liif_tail(us[0], us[1] if len(us) > 1 else None, sizes, coord, pack, disp, scale_vec,
                                 clamp_inplace=True, want_logits=want_logits)

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
        w1, off, sfs, us, rel_cols = lin[0].weight, 0, [], [], []
        for i, (sf, f) in enumerate(zip(self.to_sf_l2, feats)):
            ent = cache.get(i) if (cache is not None and i > 0) else None
            if ent is not None and ent[0] is f and ent[1] == (f._version, w1._version, torch.is_grad_enabled()):
                s, u = ent[2], ent[3]
            else:
                s = sf(f)
                u = F.conv2d(s, w1[:, off:off + s.shape[1], None, None]) if lowres else None
                if cache is not None and i > 0:
                    cache[i] = (f, (f._version, w1._version, torch.is_grad_enabled()), s, u)
            sfs.append(s)
            us.append(u)
            rel_cols.append(w1[:, off + s.shape[1]:off + s.shape[1] + 2])
            off += s.shape[1] + 2
        if lowres:
            x = G.LiifGatherMlp1.apply(us[0].contiguous(), us[1].contiguous() if len(us) > 1 else None, coord,
                                       torch.cat(rel_cols, dim=1).contiguous(), lin[0].bias)
            lin = lin[1:]
        else:
            x = torch.cat([G.LiifGather.apply(s, coord) for s in sfs], dim=1)
        for i, m in enumerate(lin):
            x = G.pointwise_linear(self, id(m), x, m, i + 1 < len(lin))
        return x

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
