"""TEST INFRASTRUCTURE (see oracle/__init__.py) — operator-level CPU oracle.

Every function restates one reference operator of SURVEY.md §8(a) with explicit
index arithmetic (gathers / slices) instead of the reference's ``grid_sample`` /
``unfold`` formulation, so that it is an independent second statement of the same
mathematics.  All citations are ``/root/reference``-relative ``file:line``.

Pinned by tests/test_oracle_golden.py against vectors captured from the imported
reference (tests/golden/make_golden.py).  Works in any float dtype (fp32 for
parity, fp64 for tight checks of the HIP kernels).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# a1 / a2  correlation volume + pyramids
# --------------------------------------------------------------------------------------


def all_pairs_corr(f1: torch.Tensor, f2: torch.Tensor) -> torch.Tensor:
    """corr[b,y,x1,x2] = sum_c f1[b,c,y,x1] * f2[b,c,y,x2]  (no 1/sqrt(C) scaling).

    models/coreContinuous_IGEV/geometry.py:63-72, models/corePrune_RAFT/geometry.py:46-55.
    Returned as [B,h,w1,w2] (the reference's [B,h,w1,1,w2] without the unit axis).
    """
    return torch.einsum("bcyx,bcyz->byxz", f1, f2)


def pool_pairs(v: torch.Tensor) -> torch.Tensor:
    """Mean of adjacent pairs along the last axis, floor(n/2) outputs, a trailing odd
    element is dropped: F.avg_pool2d(x,[1,2],stride=[1,2]) (geometry.py:24,28)."""
    n = v.shape[-1] // 2
    return (v[..., 0:2 * n:2] + v[..., 1:2 * n:2]) / 2


def corr_pyramid(corr: torch.Tensor, num_levels: int):
    """geometry.py:27-29 / corePrune_RAFT/geometry.py:17-19; levels pooled along x2."""
    out = [corr]
    for _ in range(num_levels - 1):
        out.append(pool_pairs(out[-1]))
    return out


def geo_pyramid(gev: torch.Tensor, num_levels: int):
    """gev [B,G,D,h,w] -> list of [B,h,w,G,D>>i], pooled along D (geometry.py:17-25)."""
    v = gev.permute(0, 3, 4, 1, 2).contiguous()
    out = [v]
    for _ in range(num_levels - 1):
        out.append(pool_pairs(out[-1]))
    return out


# --------------------------------------------------------------------------------------
# a3  pyramid lookup
# --------------------------------------------------------------------------------------


def lerp_zero_pad(vol: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Linear interpolation of vol[..., W] at real positions x[..., K]; taps outside
    [0, W-1] contribute zero.  This is what grid_sample(bilinear, zeros,
    align_corners=True) computes for an H==1 image (utils.py:59-73)."""
    w = vol.shape[-1]
    x0 = torch.floor(x)
    t = x - x0
    i0 = x0.long()
    i1 = i0 + 1
    ok0 = (i0 >= 0) & (i0 < w)
    ok1 = (i1 >= 0) & (i1 < w)
    v0 = torch.gather(vol, -1, i0.clamp(0, w - 1)) * ok0.to(vol.dtype)
    v1 = torch.gather(vol, -1, i1.clamp(0, w - 1)) * ok1.to(vol.dtype)
    return (1 - t) * v0 + t * v1


def geo_corr_lookup(geo_pyr, corr_pyr, disp: torch.Tensor, radius: int) -> torch.Tensor:
    """Combined_Geo_Encoding_Volume.__call__ (geometry.py:34-60) and, with geo_pyr
    empty/None, CorrBlock1D.__call__ (corePrune_RAFT/geometry.py:24-43).

    geo_pyr[i] [B,h,w,G,D_i], corr_pyr[i] [B,h,w,W2_i], disp [B,1,h,w].
    Output [B, L*(2r+1)*(G+1), h, w]; per level the channels are
    [geo c*(2r+1)+k for c<G][corr k], levels concatenated (geometry.py:48,55,57-60).
    """
    b, _, h, w = disp.shape
    dt = disp.dtype
    taps = torch.arange(-radius, radius + 1, dtype=dt, device=disp.device)  # dx = linspace(-r,r,2r+1)
    xs = torch.arange(w, dtype=dt, device=disp.device).view(1, 1, w)  # coords (continuous_IGEVstereo.py:280)
    d = disp[:, 0]  # [B,h,w]
    outs = []
    for i, corr in enumerate(corr_pyr):
        s = float(2 ** i)
        if geo_pyr:
            geo = geo_pyr[i]  # [B,h,w,G,Di]
            g = geo.shape[3]
            xg = (d / s).unsqueeze(-1) + taps  # [B,h,w,K]   geometry.py:43
            xg = xg.unsqueeze(3).expand(b, h, w, g, taps.numel())
            outs.append(lerp_zero_pad(geo, xg).reshape(b, h, w, -1))
        xc = (xs / s - d / s).unsqueeze(-1) + taps  # geometry.py:52
        outs.append(lerp_zero_pad(corr, xc))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------
# a18  corr_sampler (sampler/sampler_kernel.cu) — python emulation of the kernel loops
# --------------------------------------------------------------------------------------


def corr_sampler_forward(volume: torch.Tensor, coords: torch.Tensor, radius: int) -> torch.Tensor:
    """sampler_forward_kernel, sampler/sampler_kernel.cu:19-60 (vectorised over pixels).
    volume [N,H1,W1,W2]; coords [N,2,H1,W1] (channel 0 = x); -> [N,2r+1,H1,W1]."""
    n, h1, w1, w2 = volume.shape
    x0 = coords[:, 0]
    fl = torch.floor(x0)
    dx = (x0 - fl).to(volume.dtype)
    rd = 2 * radius + 1
    out = torch.zeros(n, rd, h1, w1, dtype=volume.dtype, device=volume.device)
    for i in range(rd + 1):
        x1 = fl.long() - radius + i
        ok = (x1 >= 0) & (x1 < w2)
        s = torch.gather(volume, -1, x1.clamp(0, w2 - 1).unsqueeze(-1))[..., 0] * ok.to(volume.dtype)
        if i > 0:
            out[:, i - 1] += s * dx
        if i < rd:
            out[:, i] += s * (1 - dx)
    return out


def corr_sampler_backward(volume: torch.Tensor, coords: torch.Tensor, corr_grad: torch.Tensor,
                          radius: int) -> torch.Tensor:
    """sampler_backward_kernel, sampler/sampler_kernel.cu:63-104."""
    n, h1, w1, w2 = volume.shape
    x0 = coords[:, 0]
    fl = torch.floor(x0)
    dx = (x0 - fl).to(volume.dtype)
    rd = 2 * radius + 1
    grad = torch.zeros_like(volume)
    for i in range(rd + 1):
        x1 = fl.long() - radius + i
        ok = ((x1 >= 0) & (x1 < w2)).to(volume.dtype)
        g = torch.zeros_like(dx)
        if i > 0:
            g = g + corr_grad[:, i - 1] * dx
        if i < rd:
            g = g + corr_grad[:, i] * (1 - dx)
        grad.scatter_add_(-1, x1.clamp(0, w2 - 1).unsqueeze(-1), (g * ok).unsqueeze(-1))
    return grad


# --------------------------------------------------------------------------------------
# a4 / a5  group-wise correlation volume, disparity regression
# --------------------------------------------------------------------------------------


def gwc_volume(fl: torch.Tensor, fr: torch.Tensor, maxdisp: int, groups: int) -> torch.Tensor:
    """vol[b,g,d,y,x] = mean_{c in group g} fl[b,c,y,x]*fr[b,c,y,x-d] for x>=d else 0.
    models/coreContinuous_IGEV/submodule.py:253-271."""
    b, c, h, w = fl.shape
    cg = c // groups
    vol = fl.new_zeros(b, groups, maxdisp, h, w)
    for d in range(min(maxdisp, w)):
        prod = fl[..., d:] * fr[..., : w - d]
        vol[:, :, d, :, d:] = prod.view(b, groups, cg, h, w - d).mean(dim=2)
    return vol


def disparity_regression(prob: torch.Tensor, maxdisp: int) -> torch.Tensor:
    """sum_d d * prob[b,d,y,x] -> [B,1,h,w]   (submodule.py:321-325)."""
    d = torch.arange(maxdisp, dtype=prob.dtype, device=prob.device).view(1, maxdisp, 1, 1)
    return (prob * d).sum(dim=1, keepdim=True)


# --------------------------------------------------------------------------------------
# a6 - a10  update block
# --------------------------------------------------------------------------------------


def _conv(m, x, pad):
    return F.conv2d(x, m.weight.to(x.dtype), None if m.bias is None else m.bias.to(x.dtype), padding=pad)


def motion_encoder(enc, disp: torch.Tensor, corr: torch.Tensor) -> torch.Tensor:
    """BasicMotionEncoder.forward, update.py:84-92."""
    cor = F.relu(_conv(enc.convc1, corr, 0))
    cor = F.relu(_conv(enc.convc2, cor, 1))
    dsp = F.relu(_conv(enc.convd1, disp, 3))
    dsp = F.relu(_conv(enc.convd2, dsp, 1))
    out = F.relu(_conv(enc.conv, torch.cat([cor, dsp], dim=1), 1))
    return torch.cat([out, disp], dim=1)


def conv_gru(gru, h, cz, cr, cq, *xs):
    """ConvGRU.forward, update.py:33-41."""
    x = torch.cat(xs, dim=1)
    hx = torch.cat([h, x], dim=1)
    z = torch.sigmoid(_conv(gru.convz, hx, 1) + cz)
    r = torch.sigmoid(_conv(gru.convr, hx, 1) + cr)
    q = torch.tanh(_conv(gru.convq, torch.cat([r * h, x], dim=1), 1) + cq)
    return (1 - z) * h + z * q


def disp_head(head, x):
    """DispHead.forward, update.py:23-24."""
    return _conv(head.conv2, F.relu(_conv(head.conv1, x, 1)), 1)


def pool2x(x):
    """update.py:94-95: 3x3 mean, stride 2, zero pad 1, divisor always 9."""
    b, c, h, w = x.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    xp = F.pad(x, (1, 1, 1, 1))
    acc = torch.zeros(b, c, ho, wo, dtype=x.dtype, device=x.device)
    for dy in range(3):
        for dx in range(3):
            acc = acc + xp[:, :, dy:dy + 2 * ho - 1:2, dx:dx + 2 * wo - 1:2]
    return acc / 9


def interp_to(x, ho: int, wo: int):
    """update.py:100-102: bilinear, align_corners=True resize to (ho, wo)."""
    b, c, h, w = x.shape
    dt = x.dtype

    def axis(n_in, n_out):
        if n_out > 1:
            pos = torch.arange(n_out, dtype=dt, device=x.device) * ((n_in - 1) / (n_out - 1))
        else:
            pos = torch.zeros(1, dtype=dt, device=x.device)
        i0 = pos.floor().long().clamp(0, n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        return i0, i1, pos - i0.to(dt)

    y0, y1, ty = axis(h, ho)
    x0, x1, tx = axis(w, wo)
    top = x[:, :, y0][:, :, :, x0] * (1 - tx) + x[:, :, y0][:, :, :, x1] * tx
    bot = x[:, :, y1][:, :, :, x0] * (1 - tx) + x[:, :, y1][:, :, :, x1] * tx
    return top * (1 - ty).view(1, 1, -1, 1) + bot * ty.view(1, 1, -1, 1)


def update_block(ub, net, inp, corr=None, disp=None, iter04=True, iter08=True, iter16=True, update=True):
    """BasicMultiUpdateBlock.forward, update.py:116-136 (n_gru_layers from ub.args)."""
    net = list(net)
    n = ub.args.n_gru_layers
    if iter16:
        net[2] = conv_gru(ub.gru16, net[2], *inp[2], pool2x(net[1]))
    if iter08:
        if n > 2:
            net[1] = conv_gru(ub.gru08, net[1], *inp[1], pool2x(net[0]),
                              interp_to(net[2], net[1].shape[2], net[1].shape[3]))
        else:
            net[1] = conv_gru(ub.gru08, net[1], *inp[1], pool2x(net[0]))
    if iter04:
        mf = motion_encoder(ub.encoder, disp, corr)
        if n > 1:
            net[0] = conv_gru(ub.gru04, net[0], *inp[0], mf,
                              interp_to(net[1], net[0].shape[2], net[0].shape[3]))
        else:
            net[0] = conv_gru(ub.gru04, net[0], *inp[0], mf)
    if not update:
        return net
    return net, disp_head(ub.disp_head, net[0])


# --------------------------------------------------------------------------------------
# a12 - a17  LIIF implicit upsampler
# --------------------------------------------------------------------------------------

_OFFS8 = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


def affinity(feature: torch.Tensor, dilation: int = 1) -> torch.Tensor:
    """AffinityFeature.forward, liif.py:432-446: cosine affinity between each pixel and its
    8 neighbours (row-major, centre skipped), zero outside the image, clamped at >= 0."""
    b, c, h, w = feature.shape
    fn = feature / feature.norm(dim=1, keepdim=True).clamp_min(1e-12)  # F.normalize(dim=1)
    p = dilation
    fp = F.pad(fn, (p, p, p, p))
    outs = []
    for oy, ox in _OFFS8:
        nb = fp[:, :, p + oy * dilation:p + oy * dilation + h, p + ox * dilation:p + ox * dilation + w]
        outs.append((nb * fn).sum(dim=1))
    return torch.stack(outs, dim=1).clamp_min(0)


def structure_feature_v2isu(x: torch.Tensor) -> torch.Tensor:
    """StructureFeature.forward 'with_v2ISU' branch, liif.py:496-499."""
    return torch.cat([x, affinity(x.detach(), 1)], dim=1)


def nearest_index(coord: torch.Tensor, n: int) -> torch.Tensor:
    """grid_sample(mode='nearest', align_corners=False) source index along an axis of size n
    for a normalised coordinate (already clamped): nearbyint(((c+1)*n-1)/2), half-to-even."""
    return torch.round(((coord + 1) * n - 1) / 2).long()


def liif_query(feat: torch.Tensor, coords: torch.Tensor):
    """liif_feat_multiscale_train, liif.py:108-137.
    feat [B,C,H',W']; coords [B,Q,2] as (row, col) in [-1,1].
    Returns rel [B,Q,2] (unclamped coord - cell centre, scaled by (H',W')) and q_feat [B,Q,C]."""
    b, c, lh, lw = feat.shape
    cc = coords.clamp(-1 + 1e-6, 1 - 1e-6)
    iy = nearest_index(cc[..., 0], lh)
    ix = nearest_index(cc[..., 1], lw)
    oky = (iy >= 0) & (iy < lh)
    okx = (ix >= 0) & (ix < lw)
    ok = (oky & okx).to(feat.dtype)
    iyc, ixc = iy.clamp(0, lh - 1), ix.clamp(0, lw - 1)
    flat = feat.reshape(b, c, lh * lw)
    idx = (iyc * lw + ixc).unsqueeze(1).expand(b, c, -1)
    q_feat = (torch.gather(flat, 2, idx) * ok.unsqueeze(1)).permute(0, 2, 1)
    # make_coord cell centres, liif.py:32-45: v0 + r + 2r*i with r = 1/n
    dt = feat.dtype
    cy = (-1 + 1.0 / lh + (2.0 / lh) * iyc.to(dt)) * ok
    cx = (-1 + 1.0 / lw + (2.0 / lw) * ixc.to(dt)) * ok
    rel = torch.stack([(coords[..., 0] - cy) * lh, (coords[..., 1] - cx) * lw], dim=-1)
    return rel, q_feat


def mlp(imnet, x: torch.Tensor) -> torch.Tensor:
    """MLP.forward, liif.py:22-25 (Linear/ReLU stack)."""
    lin = [m for m in imnet.layers if isinstance(m, torch.nn.Linear)]
    for i, m in enumerate(lin):
        x = F.linear(x, m.weight.to(x.dtype), m.bias.to(x.dtype))
        if i + 1 < len(lin):
            x = F.relu(x)
    return x


def liif_up_mask(liif, feats, coord: torch.Tensor) -> torch.Tensor:
    """liif_out_multi_scale_Training.forward, liif.py:644-678 (default-config branch:
    unfold='with_v2ISU', no pos-encoding, no cell decode, no local ensemble)."""
    latent = []
    for f in feats:
        f = structure_feature_v2isu(f)
        rel, q = liif_query(f, coord)
        latent.append(torch.cat([q, rel], dim=-1))
    lat = torch.cat(latent, dim=-1)
    b, q, _ = lat.shape
    out = mlp(liif.imnet, lat.reshape(b * q, -1)).view(b, q, -1)
    return out.permute(0, 2, 1)


def convex_upsample(disp_low: torch.Tensor, mask: torch.Tensor, hr_coord: torch.Tensor) -> torch.Tensor:
    """context_upsample_multiscale_train, submodule.py:357-372.
    disp_low [B,1,h,w] (already scaled), mask [B,9,Q] (softmaxed), hr_coord [B,Q,2] -> [B,Q].
    NB the reference clamps hr_coord IN PLACE (submodule.py:366); callers that care pass a clone."""
    b, _, h, w = disp_low.shape
    cc = hr_coord.clamp(-1 + 1e-6, 1 - 1e-6)
    iy = nearest_index(cc[..., 0], h)
    ix = nearest_index(cc[..., 1], w)
    dp = F.pad(disp_low[:, 0], (1, 1, 1, 1))  # zero border == F.unfold(.,3,1,1)
    flat = dp.reshape(b, -1)
    out = torch.zeros(b, hr_coord.shape[1], dtype=disp_low.dtype, device=disp_low.device)
    for k in range(9):
        ky, kx = k // 3, k % 3
        idx = (iy + ky) * (w + 2) + (ix + kx)
        out = out + torch.gather(flat, 1, idx) * mask[:, k]
    return out


def upsample_disp(model, disp, hidden, stem_4x, stem_2x, hr_coord, scale):
    """upsample_disp, continuous_IGEVstereo.py:192-237 / prune_raft_stereo.py:200-242
    (multi_training branch, no disparity_norm, quater_nearest None, stem_1x None)."""
    x = torch.cat((stem_4x, hidden), 1) if stem_4x is not None else hidden
    d = disp * 4.0 * scale.view(-1, 1, 1, 1)
    feats = [x, stem_2x] if stem_2x is not None else [x]
    m = F.softmax(liif_up_mask(model.liif_up, feats, hr_coord), dim=1)
    return convex_upsample(d, m, hr_coord).unsqueeze(1)


# --------------------------------------------------------------------------------------
# §8(f) rows used by the harness tests
# --------------------------------------------------------------------------------------


# ---- §8 f4: the off-by-default options of the implicit upsampler -------------------------------------------------------
def structure_feature_mode(x: torch.Tensor, mode, embed=None) -> torch.Tensor:
    """StructureFeature.forward for the modes the reference can run (liif.py:492-535): 'with_ISU' / 'with_1_4ISU' cat(x, aff(x)),
    'with_v2ISU' cat(x, aff(x.detach())), 'with_embed_ISU' convbn(cat(x, aff(x.detach()))) (eval BatchNorm), 'only_ISU' aff(x)."""
    if "with_ISU" in mode or "with_1_4ISU" in mode:
        return torch.cat([x, affinity(x, 1)], dim=1)
    if "with_v2ISU" in mode:
        return structure_feature_v2isu(x)
    if "with_embed_ISU" in mode:
        conv, bn = embed[0], embed[1]
        y = F.conv2d(structure_feature_v2isu(x), conv.weight.to(x.dtype))
        s = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).to(x.dtype)
        return (y - bn.running_mean.to(x.dtype).view(1, -1, 1, 1)) * s.view(1, -1, 1, 1) + bn.bias.to(x.dtype).view(1, -1, 1, 1)
    if "only_ISU" in mode:
        return affinity(x, 1)
    return x


def unfold3x3(feat: torch.Tensor) -> torch.Tensor:
    """F.unfold(feat, 3, padding=1).view(B, 9C, H, W) (liif.py:655): channel c*9 + ky*3 + kx, zero padded."""
    b, c, h, w = feat.shape
    fp = F.pad(feat, (1, 1, 1, 1))
    return torch.stack([fp[:, :, ky:ky + h, kx:kx + w] for ky in range(3) for kx in range(3)], dim=2).reshape(b, c * 9, h, w)


def _gather_nearest(feat, cc):
    b, c, lh, lw = feat.shape
    iy = nearest_index(cc[..., 0], lh).clamp(0, lh - 1)
    ix = nearest_index(cc[..., 1], lw).clamp(0, lw - 1)
    idx = (iy * lw + ix).unsqueeze(1).expand(b, c, -1)
    q_feat = torch.gather(feat.reshape(b, c, lh * lw), 2, idx).permute(0, 2, 1)
    dt = feat.dtype
    cy = -1 + 1.0 / lh + (2.0 / lh) * iy.to(dt)
    cx = -1 + 1.0 / lw + (2.0 / lw) * ix.to(dt)
    return q_feat, torch.stack([cy, cx], dim=-1)


def liif_query_quater(feat: torch.Tensor, coords: torch.Tensor):
    """liif_feat_multiscale_train_quater, liif.py:140-176: four nearest samples at coords + (vx/H', vy/W') + 1e-6 for
    (vx, vy) in (-1,-1), (-1,1), (1,-1), (1,1), features concatenated; rel to the mean of the first and last cell centres."""
    lh, lw = feat.shape[-2:]
    qs, cs = [], []
    for vx in (-1, 1):
        for vy in (-1, 1):
            sh = torch.tensor([vx * (2 / lh / 2) + 1e-6, vy * (2 / lw / 2) + 1e-6], dtype=coords.dtype, device=coords.device)
            q, c = _gather_nearest(feat, (coords + sh).clamp(-1 + 1e-6, 1 - 1e-6))
            qs.append(q)
            cs.append(c)
    centre = (cs[0] + cs[3]) / 2
    rel = (coords - centre) * torch.tensor([lh, lw], dtype=coords.dtype, device=coords.device)
    return rel, torch.cat(qs, dim=-1)


def spatial_encoding(rel: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """SpatialEncoding.forward, liif.py:359-367 (cat_input)."""
    y = rel[..., 0:1] * emb[:, 0].to(rel.dtype) + rel[..., 1:2] * emb[:, 1].to(rel.dtype)
    return torch.cat([rel, torch.sin(y), torch.cos(y)], dim=-1)


def liif_up_mask_general(liif, feats, coord: torch.Tensor, scale) -> torch.Tensor:
    """liif_out_multi_scale_Training.forward, liif.py:644-678, with every option the reference can run."""
    u = liif.unfold
    latent = []
    for i, f in enumerate(feats):
        if u is not None:
            if "only_unfold" in u:
                f = unfold3x3(f)
            elif any(k in u for k in ("with_1_4ISU", "with_1_43ISU", "with_1_43v2ISU")):
                if i == 0:
                    f = structure_feature_mode(f, u)
            else:
                f = structure_feature_mode(f, u, getattr(liif.to_sf_l2[i], "sfc_embeding", None))
        if liif.quater_nearest is not None and "both" in liif.quater_nearest:
            rel, q = liif_query_quater(f, coord)
        else:
            rel, q = liif_query(f, coord)
        if liif.pos_enconding:
            rel = spatial_encoding(rel, liif.pos_encoding.emb.to(rel.device))
        parts = [q, rel]
        if liif.decode_cell:
            cells = torch.ones_like(coord)
            cells[:, :, 0] = 2 / scale
            cells[:, :, 1] = 2 / scale
            parts.append(cells)
        latent.append(torch.cat(parts, dim=-1))
    lat = torch.cat(latent, dim=-1)
    b, q, _ = lat.shape
    return mlp(liif.imnet, lat.reshape(b * q, -1)).view(b, q, -1).permute(0, 2, 1)


def convex_upsample_quater(disp_low: torch.Tensor, mask: torch.Tensor, hr_coord: torch.Tensor) -> torch.Tensor:
    """context_upsample_multiscale_train_quaterp, submodule.py:375-399: disp_low [B,1,h,w], mask [B,4,Q] -> [B,Q]."""
    b, _, h, w = disp_low.shape
    out = 0
    k = 0
    for vx in (-1, 1):
        for vy in (-1, 1):
            sh = torch.tensor([vx * (2 / h / 2) + 1e-6, vy * (2 / w / 2) + 1e-6], dtype=hr_coord.dtype, device=hr_coord.device)
            d, _ = _gather_nearest(disp_low, (hr_coord + sh).clamp(-1 + 1e-6, 1 - 1e-6))
            out = out + d[..., 0] * mask[:, k]
            k += 1
    return out


def make_coord(shape):
    """Cell-centre coordinates in [-1,1] for a grid of `shape` -> [*shape, 2] (liif.py:32-45)."""
    seqs = []
    for n in shape:
        r = 1.0 / n
        seqs.append(-1 + r + (2 * r) * torch.arange(n).float())
    return torch.stack(torch.meshgrid(*seqs, indexing="ij"), dim=-1)


def epe(d_est, d_gt, mask):
    """EPE_metric, metrics_utils/metrics.py:84-90 (per image masked mean L1, then mean over images)."""
    vals = [(d_est[i][mask[i]] - d_gt[i][mask[i]]).abs().mean() for i in range(d_est.shape[0])]
    return torch.stack(vals).mean()
