"""Correlation / geometry-encoding volumes and their per-iteration lookup — HIP-backed mirror of
models/coreContinuous_IGEV/geometry.py:6-72 and models/corePrune_RAFT/geometry.py:6-55.

Same constructor and call signatures as the reference classes:
    fn = Combined_Geo_Encoding_Volume(fmap1, fmap2, geo_volume, num_levels=2, radius=4)
    feat = fn(disp [B,1,h,w], coords [B,h,w,1])      -> [B, L*9*(G+1), h, w] float32 contiguous
Differences that do not change results: the all-pairs product and every pooled level are produced
by one kernel (as_corr_build_pyramid); the geometry volume is stored [B,h,w,D,G] instead of
[B*h*w,G,1,D]; `coords` is accepted for signature parity but must be the pixel-column grid the
reference always passes (continuous_IGEVstereo.py:280) — it is regenerated in-kernel.
"""
from __future__ import annotations

import weakref

import torch

from .. import grad as G
from .. import ops
from ..harness.timing import scope


_needs_grad = G.needs_grad


class Combined_Geo_Encoding_Volume:
    def __init__(self, init_fmap1, init_fmap2, geo_volume, num_levels=2, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        f1 = init_fmap1.float().contiguous()
        f2 = init_fmap2.float().contiguous()
        with scope("corr_build"):
            if _needs_grad(f1, f2):
                self.init_corr_pyramid = list(G.CorrBuildPyramid.apply(f1, f2, num_levels))
            else:
                self.init_corr_pyramid = ops.corr_build_pyramid(f1, f2, num_levels)
        self.geo_volume_pyramid = []
        if geo_volume is not None:
            gev = geo_volume.float().contiguous()
            with scope("geo_pyramid"):
                if _needs_grad(gev):
                    self.geo_volume_pyramid = list(G.GeoPyramid.apply(gev, num_levels))
                else:
                    self.geo_volume_pyramid = ops.geo_pyramid(gev, num_levels)

    def _check_coords(self, coords, disp):
        """`coords` must be the pixel-column grid arange(w) the reference always passes (continuous_IGEVstereo.py:280,
        prune_raft_stereo.py:272): the kernels regenerate it.  Grids built by this package's models carry a mark; any other
        tensor is compared once per (tensor, version) — anything else raises instead of giving silently different numbers."""
        b, _, h, w = disp.shape
        if tuple(coords.shape) != (b, h, w, 1):
            raise RuntimeError(f"lookup: coords must be [B,h,w,1] = {(b, h, w, 1)}, got {tuple(coords.shape)}")
        if getattr(coords, "_as_pixel_grid", False):
            return
        key = (coords.data_ptr(), coords._version, tuple(coords.shape))
        ok = getattr(self, "_coords_ok", None)
        if ok is not None and ok[0] == key and ok[1]() is coords:  # the verified tensor itself, not a recycled address
            return
        if coords.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("lookup: an unverified `coords` tensor inside a stream capture — call once eagerly first")
        want = torch.arange(w, device=coords.device, dtype=coords.dtype).view(1, 1, w, 1)
        if not torch.equal(coords, want.expand(b, h, w, 1)):
            raise RuntimeError("lookup: only coords == arange(w) per row (the grid of continuous_IGEVstereo.py:280) is supported; "
                               "the lookup position is x - disp with x the pixel column")
        self._coords_ok = (key, weakref.ref(coords))

    def __call__(self, disp, coords=None):
        disp = disp.float().contiguous()
        if coords is not None:
            self._check_coords(coords, disp)
        levels = list(self.geo_volume_pyramid) + list(self.init_corr_pyramid)
        if _needs_grad(*levels):
            if getattr(self, "_anchored", None) is None:  # once per forward: every iteration's lookup hangs off these views
                self._holder = {}
                self._anchored = G.LookupAnchor.apply(self._holder, *levels) if G._DEFER else tuple(levels)
            return G.Lookup.apply(disp, self.radius, len(self.geo_volume_pyramid), self._holder if G._DEFER else None, *self._anchored)
        with scope("lookup"):
            return ops.geo_corr_lookup(self.geo_volume_pyramid, self.init_corr_pyramid, disp, self.radius)

    def fused_convc1_ok(self) -> bool:
        """lookup -> convc1 in one kernel: inference, split precision, radius 4 with (G, L) = (8, 2) or (0, 4)."""
        levels = list(self.geo_volume_pyramid) + list(self.init_corr_pyramid)
        return (not _needs_grad(*levels) and ops.get_precision() == "split"
                and ops.lookup_convc1_supported(self.geo_volume_pyramid, self.init_corr_pyramid, self.radius))

    def lookup_convc1(self, disp, pack, out_bs=None, want_f32=False):
        """relu(convc1(self(disp))) without the [B,162,h,w] tensor (geometry.py:34-60 + update.py:84-85)."""
        with scope("lookup_convc1"):
            return ops.lookup_convc1(self.geo_volume_pyramid, self.init_corr_pyramid, disp.float().contiguous(), self.radius, pack,
                                     out_bs=out_bs, want_f32=want_f32)

    def loop_front(self, taps, head_bias, disp_old, pack, w7, b7, copy_out, copy_coff):
        """disp += delta (from the head's tap planes), lookup + convc1 and the encoder's 7x7 conv in one launch (ops.loop_front)."""
        with scope("loop_front"):
            return ops.loop_front(self.geo_volume_pyramid, self.init_corr_pyramid, taps, head_bias, disp_old.float().contiguous(),
                                  self.radius, pack, w7, b7, copy_out=copy_out, copy_coff=copy_coff)

    @staticmethod
    def corr(fmap1, fmap2):
        """All-pairs correlation only -> [B,h,w1,1,w2] (geometry.py:63-72)."""
        lv = ops.corr_build_pyramid(fmap1.float().contiguous(), fmap2.float().contiguous(), 1)[0]
        b, h, w1, w2 = lv.shape
        return lv.view(b, h, w1, 1, w2)


class CorrBlock1D(Combined_Geo_Encoding_Volume):
    """RAFT-Stereo style 1-D correlation pyramid (corePrune_RAFT/geometry.py:6-55)."""

    def __init__(self, init_fmap1, init_fmap2, num_levels=2, radius=4, mask_invalid=False):
        # mask_invalid is a no-op in the reference as well (`corr == torch.tril(corr)` discards its result, :53-54)
        super().__init__(init_fmap1, init_fmap2, None, num_levels=num_levels, radius=radius)

    @staticmethod
    def corr(fmap1, fmap2, mask_invalid=False):
        return Combined_Geo_Encoding_Volume.corr(fmap1, fmap2)
