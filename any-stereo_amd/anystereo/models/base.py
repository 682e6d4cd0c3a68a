"""Shared skeleton of the two Any-Stereo models (continuous_IGEVstereo.py:91-305,
prune_raft_stereo.py:92-297): context net -> [volume build] -> iters x (lookup -> multi-level
ConvGRU -> disp += delta) -> LIIF convex upsampling at arbitrary query coordinates.

The CNN backbones run on PyTorch-ROCm; every operator of SURVEY.md §8(a) is dispatched to
libanystereo_hip.so through `anystereo.ops`.  The small `_hot_*` hooks exist so the test oracle
(/oracle, CPU) can substitute its own restatement of the same operators on the same module tree;
the product never imports the oracle.
"""
from __future__ import annotations

import argparse
import os

import torch
import torch.nn as nn

from .. import grad as G
from .. import ops
from ..harness.timing import scope
from ..nn import functional as AF
from ..nn.liif import liif_out_multi_scale_Training
from ..nn.update import BasicMultiUpdateBlock


def default_args(model: str = "continuous_IGEVStereo", **over) -> argparse.Namespace:
    """Training-script defaults that fix every shape (SURVEY.md Appendix C;
    train_continuous_IGEV.py:284-369, train_continuous_Raft.py:298-385) with `multi_training` on
    (the only functional arbitrary-scale branch)."""
    igev = "IGEV" in model
    d = dict(
        model=model, hidden_dims=[128, 128, 128], n_gru_layers=3, n_downsample=2, corr_radius=4,
        corr_levels=2 if igev else 4, slow_fast_gru=False, max_disp=192 if igev else 700, agg_type="type5",
        unfold_similarity="with_v2ISU", lsp_width=3, lsp_height=3, lsp_dilation=[1, 2, 4, 8],
        mlphidden_list=[128, 64, 64], pos_dim=0, pos_enconding=False, pos_enconding_new=False, local_ensemble=False,
        decode_cell=False, unfold=False, quater_nearest=None, require_grad=False, Raw_Mask_dim=32,
        disparity_norm=False, disparity_norm2=False, multi_training=True, multi_input_training=False,
        mixed_precision=False, train_iters=16, valid_iters=32, corr_implementation="reg", shared_backbone=False)
    d.update(over)
    return argparse.Namespace(**d)


class ContinuousStereoBase(nn.Module):
    geo_channels = 0  # 8 for IGEV (geometry encoding volume), 0 for RAFT

    # ---- construction helpers -----------------------------------------------------------------
    def _check_args(self, args):
        if not (args.multi_training or args.multi_input_training):
            raise NotImplementedError(
                "only the multi_training (arbitrary-scale LIIF) branch is built; the reference's fixed-scale branch "
                "is dimensionally inconsistent with the default dims (SURVEY.md §0 item 8)")

    def _make_update_block(self, args):
        return BasicMultiUpdateBlock(args, hidden_dims=args.hidden_dims, geo_channels=self.geo_channels)

    def _make_liif(self, args, indim, chanels):
        aff = {"win_w": args.lsp_width, "win_h": args.lsp_height, "dilation": args.lsp_dilation}
        return liif_out_multi_scale_Training(
            encoder_dim=indim, mlphidden_list=args.mlphidden_list, pos_dim=args.pos_dim, pos_enconding=args.pos_enconding,
            pos_enconding_new=args.pos_enconding_new, local_ensemble=args.local_ensemble, decode_cell=args.decode_cell,
            unfold=args.unfold_similarity, affinity_settings=aff, quater_nearest=args.quater_nearest,
            require_grad=args.require_grad, number_input=len(chanels), chanels=chanels)

    def freeze_bn(self):
        bns = self.__dict__.get("_bn2d_modules")
        if bns is None:  # the module tree is fixed after construction; the training step calls this every step
            bns = self.__dict__["_bn2d_modules"] = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        for m in bns:
            m.eval()

    # ---- timeline markers (measurement) ---------------------------------------------------------
    stamps = None  # an ops.Stamps: markers at the phase boundaries of a pass become nodes of the captured forward (bench.py)

    def _mark(self, name: str):
        st = self.__dict__.get("stamps")
        if st is not None:
            st.mark(name)

    stamp_iters = (15, 16)  # GRU iterations whose inner stages are marked too (two consecutive ones: a full period)

    def _pixel_grid(self, b, h, w, device):
        """The reference's `coords` argument of the lookup (x index of every pixel, continuous_IGEVstereo.py:279): a constant of
        the shape — built once per (shape, device) OUTSIDE a capture (the forward's eager warm-up passes) and reused, so the
        replayed graph carries no arange / cast / repeat launches between the cost aggregation and the loop.  The HIP lookup
        regenerates the grid itself (`_as_pixel_grid`); callers that substitute their own lookup read the tensor."""
        cache = self.__dict__.setdefault("_pixel_grids", {})
        key = (b, h, w, str(device))
        g = cache.get(key) if os.environ.get("ANYSTEREO_GRID_CACHE", "1") != "0" else None
        if g is None:
            g = torch.arange(w, device=device).float().reshape(1, 1, w, 1).repeat(b, h, 1, 1)
            g._as_pixel_grid = True  # the kernels regenerate this grid: mark it so the lookup need not compare it
            if not (g.is_cuda and torch.cuda.is_current_stream_capturing()):  # a capture's allocations belong to its graph
                if len(cache) > 8:
                    cache.clear()
                cache[key] = g
        return g

    # ---- hot-path hooks (HIP) ------------------------------------------------------------------
    def _hot_update(self, net_list, inp_list, corr, disp, **flags):
        return self.update_block(net_list, inp_list, corr, disp, **flags)

    def _hot_upsample(self, disp, x, stem_2x, hr_coord, scale_vec):
        feats = [x, stem_2x] if stem_2x is not None else [x]
        disp = disp.float().contiguous()
        if G.needs_grad(disp, x, *self.liif_up.parameters()):
            return self._hot_upsample_train(disp, feats, hr_coord, scale_vec)
        logits = self.liif_up(feats, hr_coord, scale_vec)  # [B,9,Q]
        hr_coord.clamp_(-1 + 1e-6, 1 - 1e-6)  # side effect of context_upsample_multiscale_train (submodule.py:366)
        with scope("convex_upsample"):
            return ops.convex_upsample(disp, logits, hr_coord, scale=scale_vec, mask_is_logits=True)

    def _upsample_general(self, disp, feats, hr, scale_vec, scale):
        """upsample_disp for the option sets outside the default (continuous_IGEVstereo.py:192-237): three inputs (agg_type
        'type2'), the non-default `liif_up` options, quarter-nearest convex sum (submodule.py:375-399), disparity_norm(2)."""
        a = self.args
        w = disp.shape[-1]
        d = disp.float()
        norm = getattr(a, "disparity_norm", False)
        norm2 = getattr(a, "disparity_norm2", False) and self.geo_channels > 0  # prune_raft_stereo.py has no disparity_norm2
        if norm:
            d, sv = d / w, None
        elif norm2:
            d, sv = d / w * 1024, None
        else:
            sv = scale_vec  # disp * 4 * scale is applied inside the convex kernel
        d = d.contiguous()
        logits = self.liif_up(feats, hr, scale if torch.is_tensor(scale) else scale_vec.view(-1, 1)).contiguous()
        train = G.needs_grad(d, logits)
        with scope("convex_upsample"):
            if a.quater_nearest is not None:
                up = (G.ConvexUpsampleQuater.apply(d, logits, hr, sv, True) if train
                      else ops.convex_upsample_quater(d, logits, hr, scale=sv, mask_is_logits=True))
            else:
                hr.clamp_(-1 + 1e-6, 1 - 1e-6)  # side effect of context_upsample_multiscale_train (submodule.py:366)
                up = (G.ConvexUpsample.apply(d, logits, hr, sv, True) if train
                      else ops.convex_upsample(d, logits, hr, scale=sv, mask_is_logits=True))
        if norm:
            up = up * torch.round(w * 4.0 * scale_vec.view(-1, 1, 1))
        elif norm2:
            up = up / 1024 * torch.round(w * 4.0 * scale_vec.view(-1, 1, 1))
        return up

    # Training: the queries are random samples of the HR grid (stereo_datasets.py:190-193), ~16 per 1/4-res pixel.  The whole
    # per-query stage (gather, MLP, softmax, convex combination) is order-independent, so it runs on the queries SORTED by
    # source pixel — the gathers read coalesced and the scatter-add backward pre-sums each pixel's run inside a wave instead
    # of issuing one contended atomic per element (csrc/backward.hip) — and the result is put back in the caller's order.
    # The permutation is computed once per forward (`_iterate` clears it): hr_coord is the same tensor in every iteration.
    sort_queries = True

    def _query_order(self, hr_coord, sizes):
        cache = self.__dict__.setdefault("_qorder", {})
        k = (hr_coord.data_ptr(), tuple(hr_coord.shape), tuple(sizes))
        if k not in cache:
            _, key = ops.liif_rel_key(hr_coord, sizes, want_rel=False, want_key=True)
            perm = torch.argsort(key, dim=1)
            inv = torch.empty_like(perm)
            inv.scatter_(1, perm, torch.arange(perm.shape[1], device=perm.device).expand_as(perm))
            cache.clear()
            cache[k] = (perm, inv)
        return cache[k]

    def _hot_upsample_train(self, disp, feats, hr_coord, scale_vec):
        sizes = [tuple(f.shape[2:]) for f in feats]
        order = self._query_order(hr_coord, sizes) if (self.sort_queries and len(feats) <= 2) else None
        hr = hr_coord if order is None else torch.gather(hr_coord, 1, order[0].unsqueeze(-1).expand(-1, -1, 2))
        logits = self.liif_up(feats, hr, scale_vec)
        hr_coord.clamp_(-1 + 1e-6, 1 - 1e-6)  # reference side effect (submodule.py:366)
        if order is not None:
            hr = hr.clamp(-1 + 1e-6, 1 - 1e-6)
        with scope("convex_upsample"):
            out = G.ConvexUpsample.apply(disp, logits.contiguous(), hr, scale_vec, True)
        return out if order is None else torch.gather(out, 2, order[1].unsqueeze(1))

    # ---- whole-forward hipGraph ---------------------------------------------------------------
    # A 32-iteration forward is ~2000 short launches; replaying it as ONE captured graph removes the host
    # launch path (PyTorch dispatch + ctypes) from the critical path.  Opt-in (`enable_graph(True)`), inference
    # only; inputs are copied into static buffers, the result is a clone.
    # A replay runs no Python: it reads the packed / BatchNorm-folded weight buffers that existed at capture time.  So a
    # graph is keyed on (input shapes, iters, device, matrix-core mode, WEIGHTS FINGERPRINT): an optimizer step,
    # load_state_dict(), .to() / .half() or set_precision() between two calls selects (captures) another graph instead of
    # replaying stale weights; train() / load_state_dict() / _apply() drop every graph.  Not seen by the fingerprint:
    # writes through `.data` (they do not bump the version counter) — call `enable_graph(True)` again after such edits.
    # The cache is an LRU of `graph_cache_size` entries (each holds a private memory pool with a whole forward), so
    # datasets with per-image sizes (ETH3D, Middlebury) do not grow memory without bound.
    graph_cache_size = 2

    def enable_graph(self, flag: bool = True):
        self._use_graph = bool(flag)
        self._graphs = {}

    def _weights_fingerprint(self):
        ver = ptr = n = 0
        for t in list(self.parameters()) + list(self.buffers()):
            ver += t._version
            ptr ^= t.data_ptr() + 0x9E3779B97F4A7C15 * n & 0xFFFFFFFFFFFFFFFF
            n += 1
        return (n, ver, ptr)

    def _drop_graphs(self):
        if self.__dict__.get("_graphs"):
            self.__dict__["_graphs"] = {}

    def train(self, mode: bool = True):
        self._drop_graphs()
        return super().train(mode)

    def load_state_dict(self, *a, **k):
        self._drop_graphs()
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._drop_graphs()
        self.__dict__.pop("_pixel_grids", None)
        return super()._apply(fn, *a, **k)

    def _reduced_precision(self, image1) -> bool:
        """`args.mixed_precision` in inference: the reference wraps the feature nets and the update block in autocast (fp16
        operands, continuous_IGEVstereo.py:245,287).  Here it selects the one-MFMA fp16-operand mode of the HIP convolution
        kernels (fp32 storage and accumulation stay; ops.fast_fp16) for the whole forward instead of torch.autocast — a
        second arithmetic mode with its own tolerance, never the parity mode.  Training keeps torch.autocast + fp32 kernels."""
        return (bool(getattr(self.args, "mixed_precision", False)) and image1.is_cuda and not torch.is_grad_enabled()
                and not self.training and ops.get_precision() == "split")

    def forward(self, image1, image2, iters=12, flow_init=None, test_mode=False, hr_coord=None, scale=1.0, output_raw=None):
        """Reference signature (continuous_IGEVstereo.py:239, prune_raft_stereo.py:246)."""
        if self._reduced_precision(image1) and not ops.get_fast_fp16():
            with ops.fast_fp16(True):
                return self.forward(image1, image2, iters, flow_init, test_mode, hr_coord, scale, output_raw)
        if (getattr(self, "_use_graph", False) and test_mode and not torch.is_grad_enabled() and image1.is_cuda
                and torch.is_tensor(scale) and not output_raw and not self.training):
            return self._forward_graphed(image1, image2, iters, hr_coord, scale)
        return self._forward_impl(image1, image2, iters=iters, flow_init=flow_init, test_mode=test_mode,
                                  hr_coord=hr_coord, scale=scale, output_raw=output_raw)

    def _forward_impl_marked(self, *a, **k):
        ops.set_stamps(self.__dict__.get("stamps"))
        try:
            return self._forward_impl(*a, **k)
        finally:
            ops.set_stamps(None)

    def _forward_graphed(self, image1, image2, iters, hr_coord, scale):
        key = (tuple(image1.shape), tuple(hr_coord.shape), tuple(scale.shape), int(iters), image1.device.index,
               ops.get_precision(), ops.get_fast_fp16() or bool(getattr(self.args, "mixed_precision", False)),
               self._weights_fingerprint(), id(self.__dict__.get("stamps")))
        graphs = self.__dict__.setdefault("_graphs", {})
        ent = graphs.pop(key, None)
        if ent is None:
            while len(graphs) >= max(1, self.graph_cache_size):  # least recently used first (dict order = recency)
                graphs.pop(next(iter(graphs)))
            st = [t.detach().clone() for t in (image1, image2, hr_coord, scale)]
            side = torch.cuda.Stream(device=image1.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up: MIOpen solver search, weight packing, allocator
                for _ in range(2):
                    st[2].copy_(hr_coord)
                    self._forward_impl_marked(st[0], st[1], iters=iters, test_mode=True, hr_coord=st[2], scale=st[3])
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(image1.device)
            keep = os.environ.get("ANYSTEREO_INFER_GRAPH_KEEP", "1") != "0"  # 0 (diagnostics): instantiate at capture end, no node rewrite
            g = torch.cuda.CUDAGraph(keep_graph=True) if keep else torch.cuda.CUDAGraph()
            st[2].copy_(hr_coord)
            with torch.cuda.graph(g):
                out = self._forward_impl_marked(st[0], st[1], iters=iters, test_mode=True, hr_coord=st[2], scale=st[3])
            # memset nodes (a library zero-filling through hipMemsetAsync) are not reliably ordered inside long graphs on this ROCm
            # stack (csrc/graph.hip): rewritten as fill kernel nodes before instantiation.  This library issues none
            # itself; the count is kept for inspection.
            if keep:
                self.__dict__["_graph_memsets"] = ops.graph_replace_memsets(g)
                g.instantiate()
            ent = (g, st, out)
        graphs[key] = ent  # (re-)insert as most recently used
        g, st, out = ent
        st[0].copy_(image1)
        st[1].copy_(image2)
        st[2].copy_(hr_coord)
        st[3].copy_(scale)
        g.replay()
        return out.clone()

    # ---- reference API -------------------------------------------------------------------------
    def upsample_disp(self, disp, hidden_layer, stem_4x, stem_2x, stem_1x, hr_coord=None, scale=1):
        """[B,1,h,w] disparity at 1/4 res -> [B,1,Q] at the query coordinates
        (continuous_IGEVstereo.py:192-237, prune_raft_stereo.py:200-242)."""
        b = disp.shape[0]
        if torch.is_tensor(scale):
            scale_vec = scale.reshape(-1).float().to(disp.device)
            if scale_vec.numel() == 1 and b > 1:
                scale_vec = scale_vec.expand(b)
            scale_vec = scale_vec.contiguous()
        else:
            scale_vec = torch.full((b,), float(scale), device=disp.device, dtype=torch.float32)
        hr = hr_coord if (hr_coord.dtype == torch.float32 and hr_coord.is_contiguous()) else hr_coord.float().contiguous()
        a = self.args
        if (stem_1x is not None or not self.liif_up._default_variant or getattr(a, "disparity_norm", False)
                or (getattr(a, "disparity_norm2", False) and self.geo_channels > 0)):
            x = torch.cat((stem_4x.float(), hidden_layer.float()), 1) if stem_4x is not None else hidden_layer.float()
            feats = [x.contiguous(), stem_2x] if stem_1x is None else [stem_1x, stem_2x, x.contiguous()]  # :207-210
            return self._upsample_general(disp, [f for f in feats if f is not None], hr, scale_vec, scale)
        parts = [[stem_4x, hidden_layer] if stem_4x is not None else [hidden_layer]] + ([[stem_2x]] if stem_2x is not None else [])
        if (type(self)._hot_upsample is ContinuousStereoBase._hot_upsample
                and not G.needs_grad(disp, hidden_layer, *self.liif_up.parameters()) and self.liif_up.fused_ok(parts, hr)):
            # inference: affinity -> first MLP layer at low resolution -> ONE per-query kernel (gather, MLP, softmax, convex
            # combination); cat(stem_4x, hidden) (:195), the latent, the hidden layers and the mask never reach HBM
            return self.liif_up.upsample_fused(parts, hr, disp.float().contiguous(), scale_vec)
        x = torch.cat((stem_4x.float(), hidden_layer.float()), 1) if stem_4x is not None else hidden_layer.float()
        return self._hot_upsample(disp, x.contiguous(), None if stem_2x is None else stem_2x.float().contiguous(), hr, scale_vec)

    # Inference schedule of the GRU loop (same operators, same operands, same results as the loop below).
    # In the reference order, iteration i+1 starts with lookup(disp_i) -> motion encoder, but only gru04 consumes the
    # motion features: gru16 and gru08 of iteration i+1 depend on the hidden states alone.  So the chain
    #     disp_head(i) -> disp += delta -> lookup(i+1) -> motion encoder(i+1)          (side stream)
    # runs concurrently with
    #     gru16(i+1) -> gru08(i+1)                                                    (main stream)
    # — both chains are made of kernels that leave most of the 256 CUs idle at 1/8 and 1/16 resolution — and the
    # two streams meet only at gru04.  Fork/join is by events, so the whole thing is capturable as one hipGraph.
    # (Measured and not kept: a third stream for the encoder's disparity branch — 0.70 vs 0.60 ms per iteration under
    # graph replay; and a stream that waits on a stream which waited on it crashes capture on this stack.)
    pipelined_loop = os.environ.get("ANYSTEREO_PIPELINED_LOOP", "1") != "0"
    # Measured and not kept (off by default): issuing the h-part of gru04's gate conv at the top of the iteration
    # (ConvGRU.pre_zr) to fill the CUs the small kernels leave idle — 0.66 vs 0.60 ms per iteration on the same box.
    split_gate_conv = os.environ.get("ANYSTEREO_SPLIT_GATE_CONV", "0") != "0"

    # gru16 of iteration i+1 needs only the 1/8 and 1/16 states of iteration i (gru08(i)'s output), so it can run beside
    # gru04(i) on a third stream instead of in front of gru08(i+1) on the main one (shorter 1/8-1/16 chain between two
    # gru04 launches).  ANYSTEREO_EARLY_GRU16=0 keeps the in-order schedule.
    early_gru16 = os.environ.get("ANYSTEREO_EARLY_GRU16", "1") != "0"
    # ... and with it the bilinear resize of its result to 1/8 resolution (gru08's third source), instead of in front of gru08
    early_interp16 = os.environ.get("ANYSTEREO_EARLY_INTERP16", "1") != "0"

    def _iterate_pipelined(self, lookup_fn, net, inp, disp, coords, iters):
        from ..nn.update import interp, pool2x
        ub = self.update_block
        dev = disp.device
        main = torch.cuda.current_stream(dev)
        # serial_streams (measurement only): the same kernels in the same order on ONE stream, so that an event pair around a
        # launch times that kernel alone (bench.py's per-kernel table)
        serial = bool(getattr(self, "serial_streams", False))
        side = main if serial else ub._side_stream(dev)
        side.wait_stream(main)
        fused = ub.encoder.fused_lookup_ok(lookup_fn)  # lookup -> convc1 as one kernel (no [B,162,h,w] tensor)
        enc = (lambda d: ub.encoder.forward_fused_lookup(d, lookup_fn)) if fused else (lambda d: ub.encoder(d, lookup_fn(d, coords)))
        with torch.cuda.stream(side):
            mf = enc(disp)
        front = fused and ub.encoder.fused_front and ub.disp_head.taps_ok(net[0])
        early = self.early_gru16 and iters > 1
        s16 = (main if serial else ub._side_stream(dev, 1)) if early else None
        net2_next = up16_next = None
        for itr in range(iters):
            sink = ops._ACTIVE_STAMPS
            if sink is not None:
                sink.fine, sink.prefix = itr in self.stamp_iters, f"it{itr}."
            mk = ops.mark_fine
            mk("main_begin")
            pre = ub.gru04.pre_zr(net[0], *(inp[0])) if self.split_gate_conv else None
            if net2_next is None:
                net[2] = ub.gru16(net[2], *(inp[2]), pool2x(net[1]))
            else:  # computed on the third stream during the previous iteration's gru04
                main.wait_stream(s16)
                net[2] = net2_next
                net[2].record_stream(main)
                net2_next = None
            if up16_next is None:
                up16 = interp(net[2], net[1])
            else:  # resized on the third stream right behind the gru16 that produced it: off the 1/8-resolution chain
                up16, up16_next = up16_next, None
                (up16.t if isinstance(up16, ops.BS8) else up16).record_stream(main)
            net[1] = ub.gru08(net[1], *(inp[1]), pool2x(net[0]), up16)
            mk("gru08_end")
            if early and itr + 1 < iters:
                s16.wait_stream(main)  # net[1], net[2] of this iteration are final
                with torch.cuda.stream(s16):
                    net2_next = ub.gru16(net[2], *(inp[2]), pool2x(net[1]))
                    if self.early_interp16:
                        up16_next = interp(net2_next, net[1])  # only net[1]'s size is used
                net[1].record_stream(s16)
                net[2].record_stream(s16)
            up = interp(net[1], net[0])
            mk("interp08_end")
            main.wait_stream(side)  # motion features (and the disparity they were computed from) are ready
            mf.record_stream(main)
            mk("gru04_begin")
            net[0] = ub.gru04(net[0], *(inp[0]), mf, up, pre_zr=pre)
            mk("gru04_end")
            net[0].record_stream(side)
            twin = getattr(net[0], "_as_bs", None)  # blocked twin of the hidden state: read by the head on the side stream
            if twin is not None:
                twin[0].record_stream(side)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if front and itr + 1 < iters:
                    # head conv1 -> ONE launch: disp += delta, lookup + convc1, 7x7 conv -> the encoder's remaining two convs
                    mf, disp = ub.encoder.forward_front(ub.disp_head.taps(net[0]), ub.disp_head, disp, lookup_fn)
                else:
                    disp = ub.disp_head(net[0], addend=disp)
                    mk("head_end")
                    if itr + 1 < iters:
                        mf = enc(disp)
                    mk("encoder_end")
        main.wait_stream(side)
        disp.record_stream(main)
        if ops._ACTIVE_STAMPS is not None:
            ops._ACTIVE_STAMPS.fine, ops._ACTIVE_STAMPS.prefix = False, ""
        return disp

    batched_train_upsample = os.environ.get("ANYSTEREO_BATCHED_TRAIN_LIIF", "1") != "0"

    def _upsample_batched(self, pend, stem_4x, stem_2x, stem_1x, hr_coord, scale):
        """upsample_disp of all iterations' (disparity, hidden state) pairs as batched calls over (iteration, sample), in groups
        that keep the per-query activations [n*B,128,Q] under 2^31 bytes -> the list of [B,1,Q] predictions, one per iteration."""
        b = pend[0][0].shape[0]
        q = hr_coord.shape[1]
        group = max(1, min(len(pend), int((1 << 31) * 0.7) // max(1, b * 128 * q * 4)))
        sc = scale.reshape(-1, 1).float()
        if sc.shape[0] == 1 and b > 1:
            sc = sc.expand(b, 1)
        preds = []
        base_order = None
        for g0 in range(0, len(pend), group):
            part = pend[g0:g0 + group]
            n = len(part)
            rep = (lambda t: None if t is None else (t if n == 1 else t.repeat(n, *([1] * (t.dim() - 1)))))
            hc = rep(hr_coord.detach()).contiguous()
            if self.sort_queries and stem_1x is None and n > 1:
                # the query order depends on the coordinates only: sort the B samples once and repeat the permutation, instead of
                # an argsort over n*B rows (the cache entry _hot_upsample_train looks up is keyed on the repeated tensor)
                sizes = [tuple(part[0][1].shape[2:])] + ([tuple(stem_2x.shape[2:])] if stem_2x is not None else [])
                if base_order is None:
                    base_order = self._query_order(hr_coord.detach(), sizes)
                perm, inv = base_order
                self.__dict__["_qorder"] = {(hc.data_ptr(), tuple(hc.shape), tuple(sizes)): (perm.repeat(n, 1), inv.repeat(n, 1))}
            # stem_2x (the upsampler's second input) goes in ONCE, at batch B: nn/liif.py computes its rows at that batch and
            # repeats them (16 x less work than a repeated input for the loop-invariant branch); the option sets outside the
            # default take every input repeated
            s2 = stem_2x if (stem_1x is None and self.liif_up._default_variant and self.liif_up.fused_first_layer) else rep(stem_2x)
            up = self.upsample_disp(torch.cat([d for d, _ in part], 0), torch.cat([h for _, h in part], 0), rep(stem_4x), s2,
                                    rep(stem_1x), hr_coord=hc, scale=rep(sc).contiguous())
            preds.extend(up.split(b, 0))
        hr_coord.clamp_(-1 + 1e-6, 1 - 1e-6)  # the reference's side effect on the caller's tensor (submodule.py:366)
        return list(preds)

    def _iterate(self, lookup_fn, net_list, inp_list, disp, coords, iters, test_mode, stem_4x, stem_2x, hr_coord, scale, stem_1x=None):
        """The GRU loop shared by both models (continuous_IGEVstereo.py:284-301, prune_raft_stereo.py:276-291)."""
        a = self.args
        disp_preds = []
        disp_up = None
        ub = self.update_block
        self.__dict__.pop("_qorder", None)
        if getattr(self, "liif_up", None) is not None:
            # reuse of the iteration-invariant upsampler branch (stem_2x) across this forward's iterations; a fresh cache per
            # forward, so no entry outlives the autograd graph it belongs to
            self.liif_up.__dict__["_train_static"] = {}
        if (test_mode and iters > 0 and a.n_gru_layers == 3 and not a.slow_fast_gru and disp.is_cuda
                and not torch.is_grad_enabled() and getattr(ub, "parallel_encoder", False)
                and type(self)._hot_update is ContinuousStereoBase._hot_update and self.pipelined_loop):
            liif = getattr(self, "liif_up", None)
            if liif is not None and stem_2x is not None and stem_1x is None and hasattr(liif, "precompute_static"):
                # the stem_2x input of the upsampler does not change during the loop: its affinity + low-resolution first layer
                # run on a branch of their own beside the loop instead of in front of the tail kernel after it
                liif.precompute_static([[stem_4x, net_list[0]] if stem_4x is not None else [net_list[0]], [stem_2x]], 1,
                                       ub._side_stream(disp.device, 2), coord=hr_coord)
            self._mark("loop_begin")
            disp = self._iterate_pipelined(lookup_fn, net_list, inp_list, disp, coords, iters)
            self._mark("loop_end")
            disp_up = self.upsample_disp(disp, net_list[0], stem_4x, stem_2x, stem_1x, hr_coord=hr_coord, scale=scale)
            self._mark("pass_end")
            if liif is not None and hasattr(liif, "clear_static"):
                liif.clear_static()
            return disp, disp_up, [disp_up]
        # Training: the upsampler of every iteration (train_continuous_IGEV.py:219 needs all predictions) does not feed back
        # into the loop, so its `iters` evaluations are issued AFTER the loop as batched calls over (iteration, sample) — same
        # values per query (the per-query stage does not look at its batch neighbours), the weight gradients summed over one
        # big call instead of `iters` small ones — and ~70 launches per iteration leave the host-bound step's launch count.
        batch_up = (self.batched_train_upsample and not test_mode and iters > 1 and disp.is_cuda and torch.is_grad_enabled()
                    and type(self)._hot_upsample is ContinuousStereoBase._hot_upsample and type(self)._hot_update is ContinuousStereoBase._hot_update
                    and torch.is_tensor(scale) and hr_coord is not None)
        pend = []
        for itr in range(iters):
            disp = disp.detach()
            geo_feat = lookup_fn(disp, coords)
            if a.n_gru_layers == 3 and a.slow_fast_gru:
                net_list = self._hot_update(net_list, inp_list, None, None, iter16=True, iter08=False, iter04=False, update=False)
            if a.n_gru_layers >= 2 and a.slow_fast_gru:
                net_list = self._hot_update(net_list, inp_list, None, None, iter16=a.n_gru_layers == 3, iter08=True,
                                            iter04=False, update=False)
            net_list, delta = self._hot_update(net_list, inp_list, geo_feat, disp, iter16=a.n_gru_layers == 3,
                                               iter08=a.n_gru_layers >= 2)
            disp = disp + delta
            if test_mode and itr < iters - 1:
                continue
            if batch_up:
                pend.append((disp, net_list[0]))
                continue
            disp_up = self.upsample_disp(disp, net_list[0], stem_4x, stem_2x, stem_1x, hr_coord=hr_coord, scale=scale)
            disp_preds.append(disp_up)
        if pend:
            disp_preds = self._upsample_batched(pend, stem_4x, stem_2x, stem_1x, hr_coord, scale)
            disp_up = disp_preds[-1]
        if getattr(self, "liif_up", None) is not None:
            self.liif_up.__dict__.pop("_train_static", None)
        return disp, disp_up, disp_preds
