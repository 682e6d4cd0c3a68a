"""Differentiable forms of the HBM-bound hot-path operators (training, SURVEY.md §8 cfg 4).

Each `torch.autograd.Function` below pairs the HIP forward of `anystereo.ops` with a HIP backward
(csrc/backward.hip, csrc/lookup.hip) — the transposes autograd derives for the reference's einsum /
avg_pool2d / grid_sample / unfold / softmax call sites.  The only library arithmetic is the pair of plain
batched GEMMs of the all-pairs correlation's backward (rocBLAS through torch.matmul).  The nn modules switch
to these when gradients are required; under `torch.no_grad()` they call `ops` directly.
"""
from __future__ import annotations

import os
import weakref

import torch

from . import _lib as L
from . import ops
from .ops import _guard, _p, _req, _stream


def needs_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


def _c(t):
    return t.float().contiguous()


# ---- a1/a2: all-pairs correlation + pooled pyramid (geometry.py:63-72, :27-29) ---------------------------------
class CorrBuildPyramid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2, num_levels):
        ctx.save_for_backward(f1, f2)
        ctx.num_levels = num_levels
        return tuple(ops.corr_build_pyramid(f1, f2, num_levels))

    @staticmethod
    def backward(ctx, *d_levels):
        f1, f2 = ctx.saved_tensors
        b, c, h, w1 = f1.shape
        w2 = f2.shape[3]
        lv = [torch.zeros((b, h, w1, w2 >> i), device=f1.device, dtype=torch.float32) if g is None else _c(g)
              for i, g in enumerate(d_levels)]
        d0 = torch.empty((b, h, w1, w2), device=f1.device, dtype=torch.float32)
        pp, keep = L.ptr_array([t.data_ptr() for t in lv])
        with _guard(f1.device):
            L.check(L.load().as_corr_pyramid_bwd(pp, _p(d0), b * h * w1, w2, len(lv), _stream()), "corr_pyramid_bwd")
        # corr[b,y,i,j] = sum_c f1[b,c,y,i] f2[b,c,y,j]:  d f1[b,:,y,:] = f2[b,:,y,:] @ d0[b,y]^T,  d f2[b,:,y,:] = f1[b,:,y,:] @ d0[b,y]
        f1r, f2r = f1.permute(0, 2, 1, 3), f2.permute(0, 2, 1, 3)  # [B,H,C,W]
        df1 = torch.matmul(f2r, d0.transpose(2, 3)).permute(0, 2, 1, 3).contiguous() if ctx.needs_input_grad[0] else None
        df2 = torch.matmul(f1r, d0).permute(0, 2, 1, 3).contiguous() if ctx.needs_input_grad[1] else None
        return df1, df2, None


# ---- a2: geometry-encoding pyramid (geometry.py:17-25) ---------------------------------------------------------
class GeoPyramid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gev, num_levels):
        ctx.shape = tuple(gev.shape)
        return tuple(ops.geo_pyramid(gev, num_levels))

    @staticmethod
    def backward(ctx, *d_levels):
        b, g, d, h, w = ctx.shape
        dev = next(t for t in d_levels if t is not None).device
        lv = [torch.zeros((b, h, w, d >> i, g), device=dev, dtype=torch.float32) if t is None else _c(t)
              for i, t in enumerate(d_levels)]
        out = torch.empty(ctx.shape, device=dev, dtype=torch.float32)
        pp, keep = L.ptr_array([t.data_ptr() for t in lv])
        with _guard(dev):
            L.check(L.load().as_geo_pyramid_bwd(pp, _p(out), b, g, d, h, w, len(lv), _stream()), "geo_pyramid_bwd")
        return out, None


# ---- a3: lookup (gradients to the volumes only; disp arrives detached, continuous_IGEVstereo.py:285) -----------
class LookupAnchor(torch.autograd.Function):
    """Identity on the pyramid levels that all lookups of one forward hang off: the lookups' backward passes add their windows to
    ONE zero-filled gradient per level (`holder`), and this node — reached after the last of them — hands those to the levels,
    instead of `iters` zero-filled full-size volumes that autograd sums pairwise."""

    @staticmethod
    def forward(ctx, holder, *levels):
        ctx.holder = holder
        ctx.set_materialize_grads(False)
        return tuple(t.view_as(t) for t in levels)

    @staticmethod
    def backward(ctx, *g):
        acc, ctx.holder["acc"] = ctx.holder.get("acc"), None
        if acc is None:
            return (None, *g)
        out = [*acc[0], *acc[1]]
        return (None, *[o if gi is None else o + gi for o, gi in zip(out, g)])


class Lookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, radius, n_geo, holder, *levels):
        geo, corr = list(levels[:n_geo]), list(levels[n_geo:])
        ctx.radius, ctx.n_geo, ctx.holder = radius, n_geo, holder
        ctx.geo_shapes = [tuple(t.shape) for t in geo]
        ctx.corr_shapes = [tuple(t.shape) for t in corr]
        ctx.save_for_backward(disp)
        return ops.geo_corr_lookup(geo, corr, disp, radius)

    @staticmethod
    def backward(ctx, d_out):
        (disp,) = ctx.saved_tensors
        if ctx.holder is not None:  # levels came through a LookupAnchor: accumulate, it returns the sums
            ctx.holder["acc"] = ops.geo_corr_lookup_backward(disp, _c(d_out), ctx.geo_shapes, ctx.corr_shapes, ctx.radius,
                                                             into=ctx.holder.get("acc"))
            return (None,) * (4 + len(ctx.geo_shapes) + len(ctx.corr_shapes))
        d_geo, d_corr = ops.geo_corr_lookup_backward(disp, _c(d_out), ctx.geo_shapes, ctx.corr_shapes, ctx.radius)
        return (None, None, None, None, *d_geo, *d_corr)


# ---- a4: group-wise correlation volume (submodule.py:253-271) ---------------------------------------------------
class GwcVolume(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fl, fr, maxdisp, groups):
        ctx.save_for_backward(fl, fr)
        ctx.maxdisp, ctx.groups = maxdisp, groups
        return ops.gwc_volume(fl, fr, maxdisp, groups)

    @staticmethod
    def backward(ctx, d_vol):
        fl, fr = ctx.saved_tensors
        b, c, h, w = fl.shape
        d_vol = _c(d_vol)
        dfl, dfr = torch.empty_like(fl), torch.empty_like(fr)
        with _guard(fl.device):
            L.check(L.load().as_gwc_volume_bwd(_p(fl), _p(fr), _p(d_vol), _p(dfl), _p(dfr), b, c, h, w, ctx.maxdisp, ctx.groups,
                                               _stream()), "gwc_volume_bwd")
        return dfl, dfr, None, None


# ---- a5: (softmax +) disparity regression (continuous_IGEVstereo.py:267-268, submodule.py:321-325) --------------
class DisparityRegression(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, apply_softmax):
        ctx.save_for_backward(cost)
        ctx.apply_softmax = bool(apply_softmax)
        return ops.disparity_regression(cost, ctx.apply_softmax)

    @staticmethod
    def backward(ctx, d_out):
        (cost,) = ctx.saved_tensors
        b, d, h, w = cost.shape
        d_out = _c(d_out)
        d_cost = torch.empty_like(cost)
        with _guard(cost.device):
            L.check(L.load().as_disparity_regression_bwd(_p(cost), _p(d_out), _p(d_cost), b, d, h, w, 1 if ctx.apply_softmax else 0,
                                                         _stream()), "disparity_regression_bwd")
        return d_cost, None


# ---- a12/a13: cat(x, affinity(x.detach())) (liif.py:496-499) ----------------------------------------------------
class StructureFeature(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.c = x.shape[1]
        return ops.structure_feature(x)

    @staticmethod
    def backward(ctx, d_out):
        return d_out[:, :ctx.c].contiguous()  # the affinity is computed on x.detach()


# ---- a14: nearest gather + relative coordinates (liif.py:108-137) -----------------------------------------------
class LiifGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, coord):
        b, c, h, w = feat.shape
        q = coord.shape[1]
        lat = torch.empty((b, c + 2, q), device=feat.device, dtype=torch.float32)
        ops.liif_gather(feat, coord, lat, 0)
        # not save_for_backward: the caller clamps hr_coord IN PLACE afterwards, every iteration (submodule.py:366), which
        # would trip autograd's version check; the backward clamps again itself and clamping is idempotent
        ctx.coord = coord
        ctx.shape = (b, c, h, w)
        return lat

    @staticmethod
    def backward(ctx, d_lat):
        coord = ctx.coord
        b, c, h, w = ctx.shape
        return ops.liif_scatter_add(_c(d_lat), coord, c, h, w), None


# ---- a14 + first Linear/ReLU of a15 through the low-resolution identity (csrc/liif.hip, as_liif_gather_mlp1) -----
class LiifGatherMlp1(torch.autograd.Function):
    """h1[b,:,q] = relu(u0[b,:,n0(q)] + u1[b,:,n1(q)] + wrel·rel(q) + bias).  Backward: the ReLU mask, two scatter-adds
    (HIP) and two small reductions over the queries for wrel / bias."""

    @staticmethod
    def forward(ctx, u0, u1, coord, wrel, bias):
        h1 = ops.liif_gather_mlp1(u0, u1, coord, wrel, bias)
        ctx.save_for_backward(h1)
        ctx.coord = coord  # see LiifGather
        ctx.s0 = tuple(u0.shape)
        ctx.s1 = None if u1 is None else tuple(u1.shape)
        ctx.has_bias = bias is not None
        return h1

    @staticmethod
    def backward(ctx, d_h1):
        (h1,) = ctx.saved_tensors
        coord = ctx.coord
        d_pre = torch.ops.aten.threshold_backward(_c(d_h1), h1, 0.0)  # ReLU backward, one launch
        b, c, h0, w0 = ctx.s0
        d_u0 = ops.liif_scatter_add(d_pre, coord, c, h0, w0) if ctx.needs_input_grad[0] else None
        d_u1 = None
        sizes = [(h0, w0)]
        if ctx.s1 is not None:
            sizes.append(ctx.s1[2:])
            if ctx.needs_input_grad[1]:
                d_u1 = ops.liif_scatter_add(d_pre, coord, c, ctx.s1[2], ctx.s1[3])
        d_wrel = d_bias = None
        if ctx.needs_input_grad[3]:
            rel, _ = ops.liif_rel_key(coord, sizes)
            d_wrel = torch.matmul(d_pre, rel.transpose(1, 2)).sum(0)
        if ctx.has_bias and ctx.needs_input_grad[4]:
            d_bias = d_pre.sum((0, 2))
        return d_u0, d_u1, None, d_wrel, d_bias


# ---- a14 + a15 as ONE forward kernel and ONE data-gradient kernel (csrc/liif_fused.hip, as_liif_mlp_fwd / _bwd) ------------
_LIIF_FUSE_FIRST = os.environ.get("ANYSTEREO_LIIF_TRAIN_FUSE_FIRST", "1") != "0"


class LiifMlpTail(torch.autograd.Function):
    """logits[b,:,q] = MLP(relu(u0[b,:,n0(q)] + u1[b % B1,:,n1(q)] + wrel.rel(q) + b1)) for the default 128-64-64-9 MLP
    (liif.py:9-25, :644-678).  The forward keeps NO per-query activation; the backward kernel recomputes them per 32-query tile
    with the forward's instruction sequence, runs the data-gradient chain on the matrix cores and writes the operands of the
    weight gradients (stashed for the step's batched launches, or reduced here without deferral) and of the first layer's
    scatter-add.  u0 [B,128,H0,W0], u1 [B1,128,H1,W1] (NCHW, as the low-resolution first layer produces them)."""

    @staticmethod
    def forward(ctx, u0, u1, coord, wrel, b1, w2, b2, w3, b3, w4, b4, pack, pack_t, stashes):
        cl = lambda u: u.detach().permute(0, 2, 3, 1).reshape(u.shape[0], -1, u.shape[1]).contiguous()  # noqa: E731
        u0c, u1c = cl(u0), cl(u1)
        sizes = [tuple(u0.shape[2:]), tuple(u1.shape[2:])]
        logits = ops.liif_mlp_fwd(u0c, u1c, sizes, coord, pack)
        ctx.save_for_backward(u0c, u1c, w2, w3, w4)
        ctx.coord = coord  # see LiifGather
        ctx.sizes, ctx.pack, ctx.pack_t, ctx.stashes = sizes, pack, pack_t, stashes
        ctx.bias = (b1 is not None, b2 is not None, b3 is not None, b4 is not None)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        u0c, u1c, w2, w3, w4 = ctx.saved_tensors
        coord, sizes = ctx.coord, ctx.sizes
        d_logits = _c(d_logits)
        need = ctx.needs_input_grad
        nb, b1n = u0c.shape[0], u1c.shape[0]
        (h0, w0), (h1s, w1s) = sizes
        d_wrel = d_b1 = None
        if _LIIF_FUSE_FIRST and need[0] and need[1] and not ops.get_deterministic():
            # the first layer's consumers inside the kernel: d1 is scattered into both maps (the shared second input's n evaluations add
            # into its B1 elements directly) and reduced against the relative coordinates; it never reaches memory
            h1, h2, h3, d3, d2, (d_u0, d_u1, d_wrel) = ops.liif_mlp_bwd(u0c, u1c, sizes, coord, ctx.pack, ctx.pack_t.get(w2, w3, w4), d_logits,
                                                                         fuse_first=True)
            d1 = None
            if not need[3]:
                d_wrel = None
        else:
            h1, h2, h3, d3, d2, d1 = ops.liif_mlp_bwd(u0c, u1c, sizes, coord, ctx.pack, ctx.pack_t.get(w2, w3, w4), d_logits)
            d_u0 = ops.liif_scatter_add(d1, coord, 128, h0, w0) if need[0] else None
            d_u1 = None
            if need[1]:
                d_u1 = ops.liif_scatter_add(d1, coord, 128, h1s, w1s)
                if b1n != nb:  # every use of the shared input: evaluation e = (i, b) read element b
                    d_u1 = d_u1.view(nb // b1n, b1n, 128, h1s, w1s).sum(0)
            if need[3]:
                rel, _ = ops.liif_rel_key(coord, sizes)
                d_wrel = torch.matmul(d1, rel.transpose(1, 2)).sum(0)
        if ctx.bias[0] and need[4]:
            # every query lands in exactly one pixel of source 0 (the coordinates are clamped into the map), so the sum over the
            # queries is the sum over that map's pixels: a 92 MB reduction instead of another pass over the 1.5 GB of d1
            d_b1 = d_u0.sum((0, 2, 3)) if d_u0 is not None else d1.sum((0, 2))
        grads = [None] * 6  # w2, b2, w3, b3, w4, b4
        for li, (x, d) in enumerate(((h1, d2), (h2, d3), (h3, d_logits))):
            want_w, want_b = need[5 + 2 * li], ctx.bias[1 + li] and need[6 + 2 * li]
            st = None if ctx.stashes is None else ctx.stashes[li]
            if st is not None:
                if want_w or want_b:
                    st.xs.append(x), st.ds.append(d)
            elif want_w or want_b:
                grads[2 * li], grads[2 * li + 1] = _wgrad_linear(d, x, want_w, want_b)
        return (d_u0, d_u1, None, d_wrel, d_b1, *grads, None, None, None)


# ---- weight gradients of layers that run once per GRU iteration: one batched reduction per step ----------------------
# A layer of the update block / the LIIF MLP is applied `iters` times per training step (train_continuous_IGEV.py:214-239), so
# autograd would run `iters` small wgrad reductions per weight and add them up one by one.  Instead every call's backward only
# stashes its (input, output gradient) pair, and the WeightAnchor node — an identity the weight passes through once per step,
# which autograd therefore reaches after ALL of the weight's uses — reduces the whole stack in one launch and hands the weight
# its gradient once (one AccumulateGrad, hence one DDP hook, per parameter and step).
_DEFER = os.environ.get("ANYSTEREO_DEFER_WGRAD", "1") != "0"


# The deferred-gradient anchors (WeightAnchor stashes, ContextAnchor holders) are cached on their module for the duration of ONE
# training forward: `begin_forward()` (called at the top of every model forward) opens a new epoch and every anchor of an older epoch is
# rebuilt on first use.  So a forward whose backward never ran (a probe pass, an exception, a validation pass with gradients enabled)
# cannot hand its autograd nodes, its stashed activations or its context tensor to the next forward; the autograd nodes of a forward
# hold their own stash / holder objects, so two forwards followed by their two backward() calls stay independent.
_EPOCH = 0
_HOLDERS = weakref.WeakSet()  # modules that cache anchors of the current forward


def begin_forward() -> int:
    """New forward: drop every cached anchor of the previous one NOW (not lazily at a layer's first use) — with it die the
    autograd nodes it kept alive, in particular the parameters' AccumulateGrad nodes, which are bound to the stream of the
    forward that created them (a hipGraph capture after eager warm-up steps would otherwise meet default-stream nodes)."""
    global _EPOCH
    _EPOCH += 1
    for mod in list(_HOLDERS):
        mod.__dict__.pop("_wgrad_anchors", None)
        mod.__dict__.pop("_ctx_anchor", None)
    _HOLDERS.clear()
    return _EPOCH


class _Stash:
    __slots__ = ("key", "kind", "xs", "ds", "done", "w_tok", "b_tok", "epoch")

    def __init__(self, key, kind):
        self.key, self.kind, self.xs, self.ds, self.done, self.epoch = key, kind, [], [], False, _EPOCH


def _stack(ts):
    return ts[0] if len(ts) == 1 else torch.cat(ts, 0)


# "hip": as_conv2d_wgrad (bf16 hi/lo split MFMA, ~2^-16 relative per product) for every 1x1 / 3x3 layer (tools/kbench_wgrad.py at
# the cfg-4 shapes: 1.9-3.2x the library's fp32 wgrad, 64 -> 64 and 1/16-resolution layers included); "library": MIOpen only
_WGRAD = os.environ.get("ANYSTEREO_WGRAD", "hip")


def _hip_wgrad(weight) -> bool:
    return _WGRAD != "library" and weight.dim() == 4 and weight.shape[2] in (1, 3)


def _wgrad_conv(d, x, weight, bias_sizes, want_w, want_b):
    k = weight.shape[2]
    if want_w and x.is_cuda and _hip_wgrad(weight):
        d_w, d_b = ops.conv2d_wgrad(_c(x), _c(d), k, want_bias=want_b)
        return d_w, d_b
    _, d_w, d_b = torch.ops.aten.convolution_backward(d, x, weight, bias_sizes, [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1,
                                                      [False, want_w, want_b])
    return d_w, d_b


def _wgrad_linear(d, x, want_w, want_b):
    return (torch.bmm(d, x.transpose(1, 2)).sum(0) if want_w else None), (d.sum((0, 2)) if want_b else None)


class WeightAnchor(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stash, weight, bias):
        ctx.stash, ctx.bias_sizes = stash, None if bias is None else list(bias.shape)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(weight)
        return weight.view_as(weight), None if bias is None else bias.view_as(bias)

    @staticmethod
    def backward(ctx, g_w, g_b):
        st = ctx.stash
        (weight,) = ctx.saved_tensors
        st.done = True
        xs, ds, st.xs, st.ds = st.xs, st.ds, [], []
        want_w, want_b = ctx.needs_input_grad[1], ctx.bias_sizes is not None and ctx.needs_input_grad[2]
        groups = {}
        for x, d in zip(xs, ds):  # calls of one layer share a shape; group anyway
            groups.setdefault((tuple(x.shape), tuple(d.shape)), ([], []))
            groups[(tuple(x.shape), tuple(d.shape))][0].append(x)
            groups[(tuple(x.shape), tuple(d.shape))][1].append(d)
        for gx, gd in groups.values():
            if st.kind == "conv" and _hip_wgrad(weight) and want_w and gx[0].is_cuda:
                # all iterations of the step in one launch, straight from the per-iteration tensors (no stacking copy)
                d_w, d_b = ops.conv2d_wgrad([_c(t) for t in gx], [_c(t) for t in gd], weight.shape[2], want_bias=want_b)
            elif st.kind == "conv7" and want_w and gx[0].is_cuda:
                d_w, d_b = ops.conv7x7_c1_wgrad([_c(t) for t in gx], [_c(t) for t in gd], want_bias=want_b)
            elif st.kind == "linear" and _WGRAD != "library" and want_w and gx[0].is_cuda:
                # a Linear layer over [B,C,Q] activations = a 1x1 convolution over a one-row image of Q pixels
                d_w, d_b = ops.conv2d_wgrad([_c(t).unsqueeze(2) for t in gx], [_c(t).unsqueeze(2) for t in gd], 1, want_bias=want_b)
                d_w = d_w.view(weight.shape)
            else:
                x, d = _stack(gx), _stack(gd)
                d_w, d_b = _wgrad_conv(d, x, weight, ctx.bias_sizes, want_w, want_b) if st.kind == "conv" else _wgrad_linear(d, x, want_w, want_b)
            if want_w:
                g_w = d_w if g_w is None else g_w + d_w
            if want_b:
                g_b = d_b if g_b is None else g_b + d_b
        return None, g_w, g_b


def anchored(mod, name, kind, weights, biases):
    """(weight, bias, stash) of layer `name` of `mod` for this training forward: `weights` / `biases` are tuples of parameters
    (concatenated along dim 0 when there are several: convz|convr).  stash None = gradients are not deferred."""
    def build():
        w = weights[0] if len(weights) == 1 else torch.cat(list(weights))
        b = None if biases[0] is None else (biases[0] if len(biases) == 1 else torch.cat(list(biases)))
        return w, b
    if not (_DEFER and torch.is_grad_enabled() and any(p.requires_grad for p in weights)):
        return (*build(), None)
    slot = mod.__dict__.setdefault("_wgrad_anchors", {})
    _HOLDERS.add(mod)
    key = tuple((id(p), p._version) for p in (*weights, *biases) if p is not None)
    st = slot.get(name)
    if st is None or st.done or st.key != key or st.epoch != _EPOCH:
        st = slot[name] = _Stash(key, kind)
        st.w_tok, st.b_tok = WeightAnchor.apply(st, *build())
    return st.w_tok, st.b_tok, st


# ---- remaining Linear/ReLU layers of a15 over channel-major activations (liif.py:9-25) ---------------------------
class PointwiseLinear(torch.autograd.Function):
    """y[b,:,q] = act(W x[b,:,q] + bias), x [B,C,Q].  Forward and dgrad run on the implicit-GEMM conv kernel (1x1, the
    dgrad with W^T); wgrad and the bias gradient are library reductions over the queries."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, pack_f, pack_b, stash=None):
        b, c, q = x.shape
        y = ops.conv2d_plain(x.view(b, c, 1, q), pack_f.get([weight], [bias]), L.ACT_RELU if relu else L.ACT_NONE).view(b, -1, q)
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.pack_b, ctx.has_bias, ctx.stash = pack_b, bias is not None, stash
        return y

    @staticmethod
    def backward(ctx, d_y):
        x, weight, y = ctx.saved_tensors
        b, c, q = x.shape
        d = _c(d_y if y is None else torch.ops.aten.threshold_backward(d_y, y, 0.0))  # ReLU backward in one launch (was compare + multiply)
        d_x = d_w = d_b = None
        if ctx.needs_input_grad[0]:
            if weight.shape[0] >= 16:
                pk = ctx.pack_b.get_dgrad(weight)
                d_x = ops.conv2d_plain(d.view(b, -1, 1, q), pk).view(b, c, q)
            else:  # a handful of output channels (the 9 mask logits): not worth a K-padded MFMA launch
                d_x = torch.matmul(weight.t(), d)
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.stash is not None:
            if want_w or want_b:
                ctx.stash.xs.append(x), ctx.stash.ds.append(d)
        else:
            d_w, d_b = _wgrad_linear(d, x, want_w, want_b)
        return d_x, d_w, d_b, None, None, None, None


# ---- a6/a7/a9: stride-1 "same" convolutions of the update block (update.py:16-92) -------------------------------
class Conv2dSame(torch.autograd.Function):
    """y = act(conv(x, W) + bias), 1x1 or 3x3, stride 1, zero "same" padding.  Forward and dgrad run on the implicit-GEMM
    kernel (dgrad = the same kernel on W transposed and flipped); wgrad / bias gradient on the library (MIOpen)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, pack_f, pack_b, stash=None):
        y = ops.conv2d_plain(x, pack_f.get([weight], [bias]), L.ACT_RELU if relu else L.ACT_NONE)
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.pack_b, ctx.bias_sizes, ctx.stash = pack_b, None if bias is None else list(bias.shape), stash
        return y

    @staticmethod
    def backward(ctx, d_y):
        x, weight, y = ctx.saved_tensors
        d = _c(d_y if y is None else torch.ops.aten.threshold_backward(d_y, y, 0.0))  # ReLU backward in one launch (was compare + multiply)
        k = weight.shape[2]
        d_x = d_w = d_b = None
        if ctx.needs_input_grad[0]:
            pk = ctx.pack_b.get_dgrad(weight)
            d_x = ops.conv2d_plain(d, pk)
        want_w, want_b = ctx.needs_input_grad[1], ctx.bias_sizes is not None and ctx.needs_input_grad[2]
        if want_w or want_b:
            if ctx.stash is not None:
                ctx.stash.xs.append(x), ctx.stash.ds.append(d)
            else:
                d_w, d_b = _wgrad_conv(d, x, weight, ctx.bias_sizes, want_w, want_b)
        return d_x, d_w, d_b, None, None, None, None


# ---- a6: the 7x7 convolution of the one-channel disparity map + ReLU (update.py:81,87) ------------------------------------
class Conv7x7C1Relu(torch.autograd.Function):
    """y = relu(conv7x7(disp [B,1,H,W], W [Cout,1,7,7], padding 3) + bias).  The loop detaches the disparity before every
    iteration (continuous_IGEVstereo.py:285), so the backward is the weight / bias gradient only (as_conv7x7_c1_wgrad_multi, all
    iterations of a step in one launch through the layer's stash); a caller that does want d_disp gets it from the library."""

    @staticmethod
    def forward(ctx, x, weight, bias, stash=None):
        y = ops.conv7x7_c1_relu(x, weight, bias)
        ctx.save_for_backward(x, weight, y)
        ctx.has_bias, ctx.stash = bias is not None, stash
        return y

    @staticmethod
    def backward(ctx, d_y):
        x, weight, y = ctx.saved_tensors
        d = _c(torch.ops.aten.threshold_backward(d_y, y, 0.0))
        d_x = d_w = d_b = None
        if ctx.needs_input_grad[0]:
            d_x = torch.ops.aten.convolution_backward(d, x, weight, None, [1, 1], [3, 3], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if want_w or want_b:
            if ctx.stash is not None:
                ctx.stash.xs.append(x), ctx.stash.ds.append(d)
            else:
                d_w, d_b = ops.conv7x7_c1_wgrad(x, d, want_bias=want_b)
                d_w = d_w if want_w else None
        return d_x, d_w, d_b, None


def conv7x7_c1_relu(mod, name, x, conv):
    """relu(conv(x)) of the 7x7, one-input-channel nn.Conv2d `conv` under autograd on this library's kernels (Cout <= 64), else the
    module + F.relu."""
    if (x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (7, 7) and conv.padding == (3, 3) and conv.stride == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels == 1 and conv.out_channels <= 64
            and conv.padding_mode == "zeros" and os.environ.get("ANYSTEREO_TRAIN_CONV7", "1") != "0"):
        w, b, stash = anchored(mod, name, "conv7", (conv.weight,), (conv.bias,))
        return Conv7x7C1Relu.apply(_c(x), w, b, stash)
    return torch.nn.functional.relu(conv(x))


# ---- a7: ConvGRU gate math as two fused stages (update.py:36-41) ---------------------------------------------------
class ContextAnchor(torch.autograd.Function):
    """Identity on the GRU context tensor (cz | cr | cq of one level, continuous_IGEVstereo.py:273) that every iteration's gate
    stages hang off: their backward passes add the context-window gradients into ONE accumulator (`holder`), handed to the
    context once — instead of 3 x iters sliced gradients that autograd sums one by one."""

    @staticmethod
    def forward(ctx, holder, base):
        ctx.holder = holder
        ctx.set_materialize_grads(False)
        return base.view_as(base)

    @staticmethod
    def backward(ctx, g):
        acc, ctx.holder["acc"] = ctx.holder.get("acc"), None
        ctx.holder["done"] = True
        if acc is None:
            return None, g
        return None, acc if g is None else acc + g


def context_anchor(mod, base):
    """(anchored view of `base`, holder) for this forward, cached on `mod` per context tensor; (base, None) without deferral."""
    if not (_DEFER and torch.is_grad_enabled() and base.requires_grad and base.is_cuda):
        return base, None
    ent = mod.__dict__.get("_ctx_anchor")
    if ent is None or ent[0] is not base or ent[1] != base._version or ent[3].get("done") or ent[3].get("epoch") != _EPOCH:
        holder = {"epoch": _EPOCH}
        _HOLDERS.add(mod)
        ent = mod.__dict__["_ctx_anchor"] = (base, base._version, ContextAnchor.apply(holder, base), holder)
    return ent[2], ent[3]


def _ctx_acc(holder, base):
    if holder.get("acc") is None:
        holder["acc"] = torch.zeros_like(base)
    return holder["acc"]


class GruGatesZR(torch.autograd.Function):
    """(z, r*h) from the convz‖convr output; cz, cr are passed for the autograd graph, the kernel reads them in place from
    the context tensor they are views of (`base`, channel offset `coff`).  With `holder` (grad.context_anchor) `base` is the
    anchored context and its gradient is accumulated by the backward kernel instead of returned through cz / cr."""

    @staticmethod
    def forward(ctx, lin, cz, cr, h, base, coff, holder=None):
        b, c, hh, ww = h.shape
        z, r, rh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        with _guard(h.device):
            L.check(L.load().as_gru_gates_zr(_p(lin), _p(base), base.shape[1], coff, _p(h), _p(z), _p(r), _p(rh), b, c, hh, ww,
                                             _stream()), "gru_gates_zr")
        ctx.save_for_backward(z, r, h)
        ctx.holder, ctx.coff, ctx.base_like = holder, coff, (base if holder is not None else None)
        return z, rh

    @staticmethod
    def backward(ctx, d_z, d_rh):
        z, r, h = ctx.saved_tensors
        b, c, hh, ww = h.shape
        d_z = None if d_z is None else _c(d_z)
        d_rh = None if d_rh is None else _c(d_rh)
        d_lin = torch.empty((b, 2 * c, hh, ww), device=h.device, dtype=torch.float32)
        d_h = torch.empty_like(h)
        with _guard(h.device):
            if ctx.holder is not None:
                acc = _ctx_acc(ctx.holder, ctx.base_like)
                L.check(L.load().as_gru_gates_zr_bwd_ctx(_p(d_z), _p(d_rh), _p(z), _p(r), _p(h), _p(d_lin), _p(d_h), _p(acc), acc.shape[1],
                                                         ctx.coff, b, c, hh, ww, _stream()), "gru_gates_zr_bwd")
                return d_lin, None, None, d_h, None, None, None
            L.check(L.load().as_gru_gates_zr_bwd(_p(d_z), _p(d_rh), _p(z), _p(r), _p(h), _p(d_lin), _p(d_h), b, c, hh, ww, _stream()),
                    "gru_gates_zr_bwd")
        return d_lin, d_lin[:, :c], d_lin[:, c:], d_h, None, None, None


class GruGatesQ(torch.autograd.Function):
    """h' = (1 - z) h + z tanh(lin + cq)."""

    @staticmethod
    def forward(ctx, lin, cq, z, h, base, coff, holder=None):
        b, c, hh, ww = h.shape
        out, t = torch.empty_like(h), torch.empty_like(h)
        with _guard(h.device):
            L.check(L.load().as_gru_gates_q(_p(lin), _p(base), base.shape[1], coff, _p(z), _p(h), _p(out), _p(t), b, c, hh, ww,
                                            _stream()), "gru_gates_q")
        ctx.save_for_backward(z, t, h)
        ctx.holder, ctx.coff, ctx.base_like = holder, coff, (base if holder is not None else None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        z, t, h = ctx.saved_tensors
        b, c, hh, ww = h.shape
        d_lin, d_z, d_h = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        with _guard(h.device):
            if ctx.holder is not None:
                acc = _ctx_acc(ctx.holder, ctx.base_like)
                L.check(L.load().as_gru_gates_q_bwd_ctx(_p(_c(d_out)), _p(z), _p(t), _p(h), _p(d_lin), _p(d_z), _p(d_h), _p(acc), acc.shape[1],
                                                        ctx.coff, b, c, hh, ww, _stream()), "gru_gates_q_bwd")
                return d_lin, None, d_z, d_h, None, None, None
            L.check(L.load().as_gru_gates_q_bwd(_p(_c(d_out)), _p(z), _p(t), _p(h), _p(d_lin), _p(d_z), _p(d_h), b, c, hh, ww, _stream()),
                    "gru_gates_q_bwd")
        return d_lin, d_lin, d_z, d_h, None, None, None


def conv2d_same(mod, name, x, weight, bias, relu=False):
    """Conv2dSame with the forward / dgrad weight packs cached on `mod` under `name`.  `weight` / `bias`: a parameter, or a tuple
    of parameters that are concatenated along the output channels (convz | convr as one convolution)."""
    packs = mod.__dict__.setdefault("_train_packs", {})
    pf, pb = packs.setdefault(name, (ops.PackedConv(), ops.PackedConv()))
    ws = weight if isinstance(weight, tuple) else (weight,)
    bs = bias if isinstance(bias, tuple) else (bias,)
    w, b, stash = anchored(mod, name, "conv", ws, bs)
    return Conv2dSame.apply(_c(x), w, b, relu, pf, pb, stash)


_TRAIN_BACKBONE = os.environ.get("ANYSTEREO_TRAIN_BACKBONE_CONVS", "1") != "0"


def module_conv2d(mod, name, conv, x):
    """`conv(x)` of an nn.Conv2d under autograd.  Stride-1 "same" 1x1 / 3x3 convolutions (no dilation, no groups) of CUDA fp32
    tensors run forward, dgrad and wgrad on this library's kernels (Conv2dSame); anything else is the module itself."""
    if (_TRAIN_BACKBONE and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and isinstance(conv, torch.nn.Conv2d)
            and conv.kernel_size in ((1, 1), (3, 3)) and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding == (conv.kernel_size[0] // 2,) * 2 and conv.padding_mode == "zeros" and x.shape[1] >= 16
            and (x.requires_grad or conv.weight.requires_grad)):
        return conv2d_same(mod, name, x, conv.weight, conv.bias)
    return conv(x)


class DwConv3x3(torch.autograd.Function):
    """Depthwise 3x3 convolution (padding 1, stride 1|2, no bias) of MobileNetV2's conv_dw layers: forward, data gradient and weight
    gradient on this library's kernels — MIOpen serves fp32 convolutions with groups == channels with its naive reference solvers
    only (72 calls, 1.65 ms per cfg-4 step)."""

    @staticmethod
    def forward(ctx, x, weight, stride):
        x = _c(x)
        ctx.save_for_backward(x, weight)
        ctx.stride = stride
        return ops.dwconv3x3(x, weight.detach().contiguous(), None, stride)

    @staticmethod
    def backward(ctx, d_out):
        x, weight = ctx.saved_tensors
        d_out = _c(d_out)
        d_x = d_w = None
        if ctx.needs_input_grad[0]:
            d_x = ops.dwconv3x3_backward_data(d_out, weight.detach().contiguous(), x.shape[2], x.shape[3], ctx.stride)
        if ctx.needs_input_grad[1]:
            d_w = ops.dwconv3x3_wgrad(x, d_out, ctx.stride)
        return d_x, d_w, None


_TRAIN_DW = os.environ.get("ANYSTEREO_TRAIN_DWCONV", "1") != "0"


def module_dwconv(conv, x):
    """`conv(x)` of a depthwise nn.Conv2d under autograd: 3x3 / padding 1 / stride 1|2 / no bias / groups == channels on CUDA fp32
    runs on DwConv3x3; anything else is the module itself (ANYSTEREO_TRAIN_DWCONV=0: always the module)."""
    c = x.shape[1]
    if (_TRAIN_DW and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and isinstance(conv, torch.nn.Conv2d)
            and conv.groups == c and conv.in_channels == c and conv.out_channels == c and conv.kernel_size == (3, 3)
            and conv.padding == (1, 1) and conv.stride in ((1, 1), (2, 2)) and conv.dilation == (1, 1) and conv.bias is None
            and conv.padding_mode == "zeros" and x.shape[0] * c <= 65535 and (x.requires_grad or conv.weight.requires_grad)):
        return DwConv3x3.apply(x, conv.weight, conv.stride[0])
    return conv(x)


_TRAIN_FOLD_BN = os.environ.get("ANYSTEREO_TRAIN_FOLD_BN", "1") != "0"
_FOLD_MIN_ELEMS = 1 << 21  # output elements from which two BatchNorm passes cost more than the fold's ~11 weight-sized launches


def conv_frozen_bn(mod, name, conv, bn, x, relu=False):
    """act(bn(conv(x))) under autograd for a FROZEN BatchNorm2d — eval mode inside the training step (`freeze_bn`,
    train_continuous_IGEV.py:189), so its statistics are constants: y = conv(x; W * s) + (beta - mean * s [+ bias * s]),
    s = gamma / sqrt(var + eps), with W * s and the bias built by differentiable operations on the WEIGHT-sized tensors.  The
    convolution then runs forward, dgrad and wgrad on this library's kernels with bias and ReLU in the epilogue and the bias gradient
    in the weight-gradient launch: the BatchNorm forward pass, its backward pass and the ReLU passes over the activation tensor
    (0.6 ms per step for the context network's five full-resolution layers alone) disappear.  Same function as bn(conv(x)); the
    gradients of gamma / beta / W follow from the chain rule through the fold.  Small maps keep the plain form."""
    cout = conv.out_channels
    big = x.shape[0] * cout * x.shape[2] * x.shape[3] >= _FOLD_MIN_ELEMS
    if not (_TRAIN_FOLD_BN and big and isinstance(bn, torch.nn.BatchNorm2d) and not bn.training and bn.affine and bn.track_running_stats
            and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and isinstance(conv, torch.nn.Conv2d)
            and conv.kernel_size in ((1, 1), (3, 3)) and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding == (conv.kernel_size[0] // 2,) * 2 and conv.padding_mode == "zeros" and x.shape[1] >= 16
            and _TRAIN_BACKBONE):
        y = bn(module_conv2d(mod, name, conv, x))
        return torch.relu(y) if relu else y
    # 1 / sqrt(var + eps) and mean / sqrt(var + eps): constants while the layer is frozen (cached per buffer version)
    cache = mod.__dict__.setdefault("_frozen_bn_consts", {})
    key = (bn.running_mean.data_ptr(), bn.running_mean._version, bn.running_var.data_ptr(), bn.running_var._version, bn.eps)
    ent = cache.get(name)
    if ent is None or ent[0] != key:
        with torch.no_grad():
            rs = torch.rsqrt(bn.running_var.float() + bn.eps)
            ent = (key, rs, bn.running_mean.float() * rs)
        cache[name] = ent
    _, rs, mrs = ent
    s_ = bn.weight * rs
    w = conv.weight * s_.view(-1, 1, 1, 1)
    b = torch.addcmul(bn.bias, bn.weight, mrs, value=-1.0)
    if conv.bias is not None:
        b = torch.addcmul(b, conv.bias, s_)
    return conv2d_same(mod, name, x, w, b, relu=relu)


def pointwise_linear(mod, key, x, lin, relu):
    """PointwiseLinear of the nn.Linear `lin` with packs and the step's weight anchor cached on `mod` under `key`."""
    packs = mod.__dict__.setdefault("_train_packs", {})
    pf, pb = packs.setdefault(key, (ops.PackedConv(), ops.PackedConv()))
    w, b, stash = anchored(mod, key, "linear", (lin.weight,), (lin.bias,))
    return PointwiseLinear.apply(x.contiguous(), w, b, relu, pf, pb, stash)


# ---- a8: the update block's resamplers (update.py:94-102) -----------------------------------------------------------
class Pool2x(torch.autograd.Function):
    """avg_pool2d(x, 3, stride=2, padding=1); backward = as_pool2x_bwd (gather form)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return ops.pool2x(_c(x))

    @staticmethod
    def backward(ctx, d_out):
        b, c, h, w = ctx.shape
        d_x = torch.empty(ctx.shape, device=d_out.device, dtype=torch.float32)
        with _guard(d_out.device):
            L.check(L.load().as_pool2x_bwd(_p(_c(d_out)), _p(d_x), b, c, h, w, _stream()), "pool2x_bwd")
        return d_x


class InterpBilinear(torch.autograd.Function):
    """F.interpolate(x, (ho, wo), mode="bilinear", align_corners=True); backward = as_interp_bilinear_ac_bwd (gather form)."""

    @staticmethod
    def forward(ctx, x, ho, wo):
        ctx.shape, ctx.dest = tuple(x.shape), (int(ho), int(wo))
        return ops.interp(_c(x), int(ho), int(wo))

    @staticmethod
    def backward(ctx, d_out):
        b, c, h, w = ctx.shape
        d_x = torch.empty(ctx.shape, device=d_out.device, dtype=torch.float32)
        with _guard(d_out.device):
            L.check(L.load().as_interp_bilinear_ac_bwd(_p(_c(d_out)), _p(d_x), b, c, h, w, ctx.dest[0], ctx.dest[1], _stream()),
                    "interp_bilinear_ac_bwd")
        return d_x, None, None


# ---- a16/a17: (softmax +) convex 3x3 upsampling at the queries (submodule.py:357-372) ---------------------------
# ---- §8 f4: off-by-default upsampler options (csrc/liif_variants.hip) ---------------------------------------------------
class StructureFeatureLive(torch.autograd.Function):
    """cat(x, affinity(x)) / affinity(x) with the affinity of the LIVE map ('with_ISU', 'with_1_4ISU', 'only_ISU',
    liif.py:493-495,501-503,534-535): the gradient reaches x through F.normalize and the eight dot products."""

    @staticmethod
    def forward(ctx, x, with_x):
        out = ops.structure_feature(x)
        if not with_x:
            out = out[:, x.shape[1]:].contiguous()
        ctx.save_for_backward(x, out)
        ctx.with_x = bool(with_x)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, out = ctx.saved_tensors
        return ops.affinity_backward(x, out, _c(d_out), ctx.with_x), None


class LiifLatent(torch.autograd.Function):
    """One source's block of the MLP input (ops.liif_latent).  Backward: the feature channels are scattered back into the
    map (HIP); a learned frequency table gets d_emb = sum_q (d_sin cos - d_cos sin) rel from the block itself (it holds
    rel, sin and cos), a [n,Q] x [Q,2] reduction."""

    @staticmethod
    def forward(ctx, feat, coord, emb, cell, unfold9, n_samp):
        b, c, h, w = feat.shape
        q = coord.shape[1]
        n_enc = 0 if emb is None else emb.shape[0]
        lat = torch.empty((b, ops.liif_latent_width(c, unfold9, n_samp, n_enc, cell is not None), q), device=feat.device,
                          dtype=torch.float32)
        ops.liif_latent(feat, coord, lat, 0, unfold9=unfold9, n_samp=n_samp, emb=emb, cell=cell)
        ctx.coord = coord  # see LiifGather
        ctx.shape, ctx.opts, ctx.n_enc = (b, c, h, w), (bool(unfold9), int(n_samp)), n_enc
        ctx.emb_grad = emb is not None and emb.requires_grad
        if ctx.emb_grad:
            ctx.save_for_backward(lat)
        return lat

    @staticmethod
    def backward(ctx, d_lat):
        b, c, h, w = ctx.shape
        unfold9, n_samp = ctx.opts
        d_lat = _c(d_lat)
        d_feat = ops.liif_latent_backward(d_lat, ctx.coord, 0, c, h, w, unfold9=unfold9, n_samp=n_samp) if ctx.needs_input_grad[0] else None
        d_emb = None
        if ctx.emb_grad:
            (lat,) = ctx.saved_tensors
            o, n = (9 * c if unfold9 else c) * n_samp, ctx.n_enc
            rel, sn, cs = lat[:, o:o + 2], lat[:, o + 2:o + 2 + n], lat[:, o + 2 + n:o + 2 + 2 * n]
            dy = d_lat[:, o + 2:o + 2 + n] * cs - d_lat[:, o + 2 + n:o + 2 + 2 * n] * sn  # [B,n,Q]
            d_emb = torch.einsum("bnq,bkq->nk", dy, rel)
        return d_feat, None, d_emb, None, None, None


class ConvexUpsampleQuater(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, mask, coord, scale, mask_is_logits):
        ctx.save_for_backward(disp, mask, scale)
        ctx.coord = coord
        ctx.logits = bool(mask_is_logits)
        return ops.convex_upsample_quater(disp, mask, coord, scale=scale, mask_is_logits=ctx.logits)

    @staticmethod
    def backward(ctx, d_out):
        disp, mask, scale = ctx.saved_tensors
        coord = ctx.coord
        b, _, h, w = disp.shape
        q = coord.shape[1]
        d_out = _c(d_out)
        d_mask = torch.empty_like(mask)
        d_disp = torch.empty_like(disp) if ctx.needs_input_grad[0] else None
        if d_disp is not None:
            ops.warn_nondeterministic("ConvexUpsampleQuater.backward (quarter-nearest convex upsampling)")
        with _guard(disp.device):
            L.check(L.load().as_convex_upsample_quater_bwd(_p(disp), _p(scale), _p(mask), _p(coord), _p(d_out), _p(d_mask), _p(d_disp),
                                                           b, h, w, q, 1 if ctx.logits else 0, _stream()), "convex_upsample_quater_bwd")
        return d_disp, d_mask, None, None, None


class ConvexUpsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, mask, coord, scale, mask_is_logits):
        ctx.save_for_backward(disp, mask, scale)
        ctx.coord = coord  # see LiifGather
        ctx.logits = bool(mask_is_logits)
        return ops.convex_upsample(disp, mask, coord, scale=scale, mask_is_logits=ctx.logits)

    @staticmethod
    def backward(ctx, d_out):
        disp, mask, scale = ctx.saved_tensors
        coord = ctx.coord
        b, _, h, w = disp.shape
        q = coord.shape[1]
        d_out = _c(d_out)
        d_mask = torch.empty_like(mask)
        det = ops.get_deterministic() and ctx.needs_input_grad[0]
        d_disp = torch.empty_like(disp) if (ctx.needs_input_grad[0] and not det) else None
        with _guard(disp.device):
            L.check(L.load().as_convex_upsample_bwd(_p(disp), _p(scale), _p(mask), _p(coord), _p(d_out), _p(d_mask), _p(d_disp),
                                                    b, h, w, q, 1 if ctx.logits else 0, _stream()), "convex_upsample_bwd")
        if det:
            # d_disp without atomics: the nine per-query contributions g * l_k * (4 * scale) summed per source pixel in a fixed order
            # (ops.liif_scatter_add's deterministic form), then the nine tap planes shifted onto the pixels they belong to
            probs = torch.softmax(mask, dim=1) if ctx.logits else mask
            mul = (4.0 * scale.reshape(-1, 1, 1).float()) if scale is not None else 1.0
            taps = ops.liif_scatter_add((probs * d_out * mul).contiguous(), coord, 9, h, w)  # [B,9,h,w]: tap k of the queries AT each pixel
            pad = torch.nn.functional.pad(taps, (1, 1, 1, 1))
            d_disp = torch.zeros_like(disp)
            for k in range(9):
                dy, dx = k // 3 - 1, k % 3 - 1  # a query at (iy, ix) adds tap k to pixel (iy + dy, ix + dx)
                d_disp = d_disp + pad[:, k:k + 1, 1 - dy:1 - dy + h, 1 - dx:1 - dx + w]
        return d_disp, d_mask, None, None, None

