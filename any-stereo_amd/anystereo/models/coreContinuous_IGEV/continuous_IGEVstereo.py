"""continuous_IGEVStereo — same constructor, forward() signature, return values and state_dict keys
as models/coreContinuous_IGEV/continuous_IGEVstereo.py:91-305, with the hot path on HIP.

    model = continuous_IGEVStereo(args)
    disp_up = model(image1, image2, iters=32, hr_coord=coord[None], scale=torch.tensor([[s]]), test_mode=True)
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ... import grad as G
from ...nn import blocks as B
from ...nn import functional as AF
from ...nn.encoders import Feature, MultiBasicEncoder, _plain_conv
from ...nn.geometry import Combined_Geo_Encoding_Volume
from ..base import ContinuousStereoBase

hourglass = B.hourglass


def _stems(agg_type):
    if "type1" in agg_type:
        return B.plain_stem(3, 32, True), B.plain_stem(32, 48, True)
    if "type3" in agg_type:
        return B.HighRes_Aggregation(3, 32), B.HighRes_Aggregation(32, 48)
    if "type4" in agg_type:
        return B.HighRes_Aggregation_LN(3, 32), B.HighRes_Aggregation_LN(32, 48)
    if "type5" in agg_type:
        return B.HighRes_Aggregation_LN_GeLU(3, 32), B.HighRes_Aggregation_LN_GeLU(32, 48)
    raise AssertionError(f"unsupported agg_type {agg_type!r}")


class continuous_IGEVStereo(ContinuousStereoBase):
    geo_channels = 8
    geo_block = Combined_Geo_Encoding_Volume

    def __init__(self, args):
        super().__init__()
        self._check_args(args)
        self.args = args
        self.name_buff = []
        context_dims = args.hidden_dims
        self.max_disp = args.max_disp
        self.multi_training = args.multi_training
        self.multi_input_training = args.multi_input_training
        self.cnet = MultiBasicEncoder(output_dim=[args.hidden_dims, context_dims], norm_fn="batch", downsample=args.n_downsample)
        self.update_block = self._make_update_block(args)
        self.context_zqr_convs = nn.ModuleList(
            nn.Conv2d(context_dims[i], args.hidden_dims[i] * 3, 3, padding=1) for i in range(args.n_gru_layers))
        self.agg_type = args.agg_type
        self.feature = Feature()
        indim = 48 + 32
        if "type2" in args.agg_type and not any(t in args.agg_type for t in ("type1", "type3", "type4", "type5")):
            # three upsampler inputs: a full-resolution stem in front of the two pixel-unshuffle stems (:137-158)
            self.stem_1 = nn.Sequential(B.BasicConv_IN(3, 8, kernel_size=3, stride=1, padding=1),
                                        nn.Conv2d(8, 8, 3, 1, 1, bias=False), nn.InstanceNorm2d(8), nn.ReLU())
            self.stem_2, self.stem_4 = B.plain_stem(8, 32, True), B.plain_stem(32, 48, True)
            indim = 48 + 32 + 8
            chanels = [8, 32, 48 + args.hidden_dims[2]]
        else:
            self.stem_2, self.stem_4 = _stems(args.agg_type)
            chanels = [48 + args.hidden_dims[2], 32]
        self.conv = B.BasicConv_IN(96, 96, kernel_size=3, padding=1, stride=1)
        self.desc = nn.Conv2d(96, 96, kernel_size=1, padding=0, stride=1)
        self.liif_up = self._make_liif(args, indim + args.hidden_dims[2], chanels)
        self.corr_stem = B.BasicConv(8, 8, is_3d=True, kernel_size=3, stride=1, padding=1)
        self.corr_feature_att = B.FeatureAtt(8, 96)
        self.cost_agg = B.hourglass(8)
        self.classifier = nn.Conv3d(8, 1, 3, 1, 1, bias=False)

    # hot-path hooks -----------------------------------------------------------------------------
    def _hot_gwc(self, match_left, match_right):
        return AF.build_gwc_volume(match_left, match_right, self.args.max_disp // 4, 8)

    def _hot_init_disp(self, cost):
        return AF.softmax_disparity_regression(cost)

    def _hot_lookup_fn(self, match_left, match_right, gev):
        return self.geo_block(match_left.float(), match_right.float(), gev.float(), radius=self.args.corr_radius,
                              num_levels=self.args.corr_levels)

    parallel_context = os.environ.get("ANYSTEREO_PARALLEL_CONTEXT", "1") != "0"
    parallel_stems = os.environ.get("ANYSTEREO_PARALLEL_STEMS", "1") != "0"
    # which branch of the forked pre-loop is ISSUED first (the feature trunk on the main stream, or the stems + context network on
    # the side stream): same kernels, same dependencies, bit-identical results — only the node order of the captured graph
    trunk_first = os.environ.get("ANYSTEREO_TRUNK_FIRST", "0") != "0"
    # (with trunk_first) the context network — the branch's heavy full-resolution convolutions — starts only when the feature trunk
    # is done: the trunk's ~60 short dependent launches then run beside the light stems only, the context network beside the
    # cost aggregation.  Measured: see DESIGN.md §0 (round 6)
    context_after_trunk = os.environ.get("ANYSTEREO_CONTEXT_AFTER_TRUNK", "0") != "0"
    # ... or when the trunk has passed one of its stages (nn/encoders.py::Feature.forward: "block0" .. "deconv16_8"): "" = no wait
    context_after_stage = os.environ.get("ANYSTEREO_CONTEXT_AFTER_STAGE", "")

    # The six feature-attention gates of the cost aggregation (submodule.py:328-341 via continuous_IGEVstereo.py:60-88) depend on
    # the 2-D features only: in inference they are computed on a branch stream right behind the feature trunk, in the order the
    # cost aggregation consumes them, instead of as 12 small launches inside its serial chain (each FeatureAtt waits for its own).
    early_gates = os.environ.get("ANYSTEREO_EARLY_GATES", "1") != "0"

    def _early_gates(self, features_left, image1):
        if not (self.early_gates and image1.is_cuda and B.fused_ok(features_left[0], self)):
            return
        h = self.cost_agg
        order = [(self.corr_feature_att, 0), (h.feature_att_8, 1), (h.feature_att_16, 2), (h.feature_att_32, 3),
                 (h.feature_att_up_16, 2), (h.feature_att_up_8, 1)]
        if not all(isinstance(m, B.FeatureAtt) and m.gate_ok(features_left[i]) for m, i in order):
            return
        main = torch.cuda.current_stream(image1.device)
        br = self.update_block._side_stream(image1.device, 2)
        br.wait_stream(main)
        with torch.cuda.stream(br):
            for m, i in order:
                g = m.gate(features_left[i])
                ev = torch.cuda.Event()
                ev.record(br)
                m.early = (g, ev)
        for f in features_left:
            f.record_stream(br)

    def _stems_fwd(self, image):
        """(stem_1x | None, stem_2x, stem_4x) of one image batch (continuous_IGEVstereo.py:247-256)."""
        s1 = self.stem_1(image) if hasattr(self, "stem_1") else None
        s2 = self.stem_2(image if s1 is None else s1)
        return s1, s2, self.stem_4(s2)

    def _context(self, image1):
        """Hidden-state initialisation and the per-level context terms (continuous_IGEVstereo.py:270-273)."""
        from ... import _lib as L
        st = self.__dict__.get("stamps")
        self.cnet.on_stage = (lambda name: self._mark("cnet_" + name)) if (st is not None and st.stages) else None
        fuse = (B.fused_ok(image1, self.cnet) and getattr(self.cnet, "paired_heads", False)
                and all(len(o) == 2 for o in (self.cnet.outputs04, self.cnet.outputs08, self.cnet.outputs16)))
        if fuse:  # tanh (:271) and relu (:272) in the epilogue of the heads' last (paired) convolution
            self.cnet.head_acts = (L.ACT_TANH, L.ACT_RELU)
            try:
                cnet_list = self.cnet(image1, num_layers=self.args.n_gru_layers)
            finally:
                self.cnet.head_acts = None
            net_list = [x[0] for x in cnet_list]
            inp_list = [x[1] for x in cnet_list]
        else:
            cnet_list = self.cnet(image1, num_layers=self.args.n_gru_layers)
            net_list = [torch.tanh(x[0]) for x in cnet_list]
            inp_list = [torch.relu(x[1]) for x in cnet_list]
        return net_list, [_plain_conv(self, conv, i) for i, conv in zip(inp_list, self.context_zqr_convs)]

    def _forward_impl(self, image1, image2, iters=12, flow_init=None, test_mode=False, hr_coord=None, scale=1.0, output_raw=None):
        """Estimate disparity between a pair of frames (images are 0..255 float)."""
        G.begin_forward()  # deferred-gradient anchors are scoped to this forward (grad.py)
        a = self.args
        self._mark("pass_begin")
        fast = B.fused_ok(image1, self) and image2.dtype == image1.dtype and image2.shape == image1.shape
        if fast:
            # inference: left and right image as ONE batch through the (per-sample) feature net, stems and descriptor head — same
            # arithmetic per sample, half the launches, twice the blocks per launch.  The pair is concatenated FIRST and normalised
            # once, in place ((x / 255) * 2 - 1 per element, the reference's operations, :242-243): 4 launches instead of 7 in front
            # of both pre-loop branches
            n = image1.shape[0]
            both = torch.cat((image1, image2), 0).div_(255.0).mul_(2).sub_(1.0)
            image1, image2 = both[:n], both[n:]
        else:
            image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
            image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        with torch.autocast("cuda", enabled=bool(a.mixed_precision) and image1.is_cuda and not self._reduced_precision(image1)):
            side = None
            if fast and self.parallel_context and image1.is_cuda:
                # the stems and the context network do not depend on the feature trunk, and most of the trunk / cost-aggregation
                # kernels at 1/8 .. 1/32 resolution leave CUs idle: run them on the second stream (captured as a parallel
                # branch); the main stream picks the stems up before the descriptor head and the context before the GRU loop
                main = torch.cuda.current_stream(image1.device)
                side = self.update_block._side_stream(image1.device)
                if self.trunk_first:
                    # node order of the captured graph = issue order, and a replay feeds its queues in node order: the trunk's ~100
                    # short kernels go first, the second branch forks off the point BEFORE them (an event, not wait_stream)
                    forked = torch.cuda.Event()
                    forked.record(main)
                    self._mark("trunk_begin")
                    stage_ev = {}

                    def on_stage(name, want=self.context_after_stage):
                        if name == want:
                            stage_ev[name] = torch.cuda.Event()
                            stage_ev[name].record(main)
                            self._mark("trunk_" + name)
                    feats = self.feature(both, on_stage=on_stage if self.context_after_stage else None)
                    self._mark("trunk_end")
                    trunk_done = stage_ev.get(self.context_after_stage)
                    if trunk_done is None:
                        trunk_done = torch.cuda.Event()
                        trunk_done.record(main)
                    side.wait_event(forked)
                else:
                    side.wait_stream(main)
                with torch.cuda.stream(side):
                    if self.parallel_stems:
                        stem_1b, stem_2b, stem_4b = self._stems_fwd(both)
                        self._mark("stems_end")
                        stems_done = torch.cuda.Event()
                        stems_done.record(side)
                    if self.trunk_first and (self.context_after_trunk or self.context_after_stage):
                        side.wait_event(trunk_done)
                    net_list, ctx_list = self._context(image1)
                    self._mark("context_end")
            if fast:
                if not (side is not None and self.trunk_first):
                    self._mark("trunk_begin")
                    st = self.__dict__.get("stamps")
                    feats = self.feature(both, on_stage=(lambda name: self._mark("trunk_" + name)) if (st is not None and st.stages) else None)
                    self._mark("trunk_end")
                if side is not None and self.parallel_stems:
                    main.wait_event(stems_done)
                    stem_2b.record_stream(main)
                    stem_4b.record_stream(main)
                    if stem_1b is not None:
                        stem_1b.record_stream(main)
                else:
                    stem_1b, stem_2b, stem_4b = self._stems_fwd(both)
                feats[0] = torch.cat((feats[0], stem_4b), 1)
                match = _plain_conv(self, self.desc, self.conv(feats[0]))
                features_left = [f[:n] for f in feats]
                stem_2x, stem_4x = stem_2b[:n], stem_4b[:n]
                stem_1x = None if stem_1b is None else stem_1b[:n]
                match_left, match_right = match[:n], match[n:]
            else:
                features_left = self.feature(image1)
                features_right = self.feature(image2)
                stem_1x, stem_2x, stem_4x = self._stems_fwd(image1)
                _, _, stem_4y = self._stems_fwd(image2)
                features_left[0] = torch.cat((features_left[0], stem_4x), 1)
                features_right[0] = torch.cat((features_right[0], stem_4y), 1)
                match_left = _plain_conv(self, self.desc, self.conv(features_left[0]))
                match_right = _plain_conv(self, self.desc, self.conv(features_right[0]))
            self._early_gates(features_left, image1)
            gwc_volume = self._hot_gwc(match_left, match_right)
            gwc_volume = self.corr_feature_att.after(self.corr_stem, gwc_volume, features_left[0])
            geo_encoding_volume = self.cost_agg(gwc_volume, features_left)
            if B.fused_ok(geo_encoding_volume, self) and B.conv3d_k3_ok(self.classifier):
                cost = B.conv3d_fused(self, self.classifier, None, geo_encoding_volume, 0)
            else:
                cost = B.conv3d_train(self.classifier, geo_encoding_volume)
            init_disp = self._hot_init_disp(cost.squeeze(1))
            self._mark("cost_agg_end")
            del gwc_volume
            if side is None:
                net_list, ctx_list = self._context(image1)
            else:
                main.wait_stream(side)
                for t in net_list + ctx_list:
                    t.record_stream(main)
        net_list = [n.float() for n in net_list]
        # cz, cr, cq stay VIEWS of one [B,3*hidden,h,w] tensor: the GRU kernels index it in place
        inp_list = [list(c.float().split(split_size=c.shape[1] // 3, dim=1)) for c in ctx_list]

        geo_fn = self._hot_lookup_fn(match_left, match_right, geo_encoding_volume)
        b, c, h, w = match_left.shape
        coords = self._pixel_grid(b, h, w, match_left.device)
        disp, disp_up, disp_preds = self._iterate(geo_fn, net_list, inp_list, init_disp.float(), coords, iters, test_mode,
                                                  stem_4x, stem_2x, hr_coord, scale, stem_1x=stem_1x)
        if test_mode:
            return disp_up
        return init_disp.squeeze(1), disp_preds
