"""continuous_RaftStereo — same constructor, forward() signature, return values and state_dict keys as
models/corePrune_RAFT/prune_raft_stereo.py:92-297, with the hot path on HIP (no geometry encoding
volume: correlation pyramid of 4 levels, zero initial disparity)."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ... import grad as G
from ...nn import blocks as B
from ...nn.encoders import BasicEncoder, MultiBasicEncoder
from ...nn.geometry import CorrBlock1D
from ..base import ContinuousStereoBase


def _stems(agg_type):
    if "IGEV" in agg_type:
        return B.plain_stem(3, 32, False), B.plain_stem(32, 48, False)
    if "type1" in agg_type:
        return B.plain_stem(3, 32, True), B.plain_stem(32, 48, True)
    if "type3" in agg_type:
        return B.HighRes_Aggregation(3, 32), B.HighRes_Aggregation(32, 48)
    if "type4" in agg_type:
        return B.HighRes_Aggregation_LN(3, 32), B.HighRes_Aggregation_LN(32, 48)
    if "type5" in agg_type:
        return B.HighRes_Aggregation_LN_GeLU(3, 32), B.HighRes_Aggregation_LN_GeLU(32, 48)
    return None, None


class continuous_RaftStereo(ContinuousStereoBase):
    geo_channels = 0
    corr_block = CorrBlock1D

    def __init__(self, args):
        super().__init__()
        self._check_args(args)
        self.args = args
        self.multi_training = args.multi_training
        self.multi_input_training = args.multi_input_training
        self.agg_type = args.agg_type
        context_dims = args.hidden_dims
        self.cnet = MultiBasicEncoder(output_dim=[args.hidden_dims, context_dims], norm_fn="batch", downsample=args.n_downsample)
        self.update_block = self._make_update_block(args)
        self.context_zqr_convs = nn.ModuleList(
            nn.Conv2d(context_dims[i], args.hidden_dims[i] * 3, 3, padding=1) for i in range(args.n_gru_layers))
        self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", downsample=args.n_downsample)
        s2, s4 = _stems(args.agg_type)
        if s2 is not None:
            self.stem_2, self.stem_4 = s2, s4
            indim, chanels = 48 + 32, [48 + args.hidden_dims[2], 32]
            if "IGEV" in args.agg_type:
                # the reference leaves `chanels` undefined on this branch (prune_raft_stereo.py:110-121)
                chanels = [48 + args.hidden_dims[2], 32]
        elif "type2" in args.agg_type:
            # three upsampler inputs: a full-resolution stem in front of the two pixel-unshuffle stems (:156-177)
            self.stem_1 = nn.Sequential(B.BasicConv_IN(3, 8, kernel_size=3, stride=1, padding=1),
                                        nn.Conv2d(8, 8, 3, 1, 1, bias=False), nn.InstanceNorm2d(8), nn.ReLU())
            self.stem_2, self.stem_4 = B.plain_stem(8, 32, True), B.plain_stem(32, 48, True)
            indim, chanels = 48 + 32 + 8, [8, 32, 48 + args.hidden_dims[2]]
        else:
            indim, chanels = 0, [args.hidden_dims[2]]
        self._has_stems = hasattr(self, "stem_2")
        self.liif_up = self._make_liif(args, indim + args.hidden_dims[2], chanels)

    def _hot_lookup_fn(self, match_left, match_right):
        return self.corr_block(match_left.float(), match_right.float(), radius=self.args.corr_radius,
                               num_levels=self.args.corr_levels)

    parallel_context = os.environ.get("ANYSTEREO_PARALLEL_CONTEXT", "1") != "0"

    def _context(self, image1):
        a = self.args
        cnet_list = self.cnet(image1, num_layers=a.n_gru_layers)
        net_list = [torch.tanh(x[0]) for x in cnet_list]
        inp_list = [torch.relu(x[1]) for x in cnet_list]
        ctx_list = [G.module_conv2d(self, f"ctx{k}", conv, i) for k, (i, conv) in enumerate(zip(inp_list, self.context_zqr_convs))]
        stem_1x = stem_2x = stem_4x = None
        if self._has_stems:
            stem_1x = self.stem_1(image1) if hasattr(self, "stem_1") else None
            stem_2x = self.stem_2(image1 if stem_1x is None else stem_1x)
            stem_4x = self.stem_4(stem_2x)
        self.__dict__["_stem_1x"] = stem_1x
        return net_list, ctx_list, stem_2x, stem_4x

    def _forward_impl(self, image1, image2, iters=12, flow_init=None, test_mode=False, hr_coord=None, scale=1.0, output_raw=False):
        G.begin_forward()  # deferred-gradient anchors are scoped to this forward (grad.py)
        a = self.args
        image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
        image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        with torch.autocast("cuda", enabled=bool(a.mixed_precision) and image1.is_cuda and not self._reduced_precision(image1)):
            side = None
            if B.fused_ok(image1, self) and self.parallel_context and image1.is_cuda:
                # context network (+ stems) on the second stream while the feature network runs (see continuous_IGEVStereo)
                main = torch.cuda.current_stream(image1.device)
                side = self.update_block._side_stream(image1.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    net_list, ctx_list, stem_2x, stem_4x = self._context(image1)
            match_left, match_right = self.fnet([image1, image2])
            if side is None:
                net_list, ctx_list, stem_2x, stem_4x = self._context(image1)
            else:
                main.wait_stream(side)
                for t in net_list + ctx_list + [t for t in (stem_2x, stem_4x, self.__dict__.get("_stem_1x")) if t is not None]:
                    t.record_stream(main)
        net_list = [n.float() for n in net_list]
        inp_list = [list(c.float().split(split_size=c.shape[1] // 3, dim=1)) for c in ctx_list]

        corr_fn = self._hot_lookup_fn(match_left, match_right)
        b, c, h, w = match_left.shape
        coords = self._pixel_grid(b, h, w, match_left.device)
        disp0 = match_left.new_zeros((b, 1, h, w), dtype=torch.float32)
        disp, disp_up, disp_preds = self._iterate(corr_fn, net_list, inp_list, disp0, coords, iters, test_mode,
                                                  stem_4x, stem_2x, hr_coord, scale, stem_1x=self.__dict__.pop("_stem_1x", None))
        if test_mode:
            return (disp, disp_up) if output_raw else disp_up
        return disp_preds
