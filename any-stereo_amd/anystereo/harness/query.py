"""§8(f1) query-grid + padding harness: what defines `hr_coord` / `scale` for arbitrary-scale
evaluation (evaluation.py:67-89 pad_for_multi_train, evaluation_validate.py:92-106
pad_for_multi_train_Fixed, models/*/utils/utils.py:7-26 InputPadder, liif.py:32-45 make_coord).

`InputPadder.get_pad_num()` is called but never defined in the reference; from its
use (`coord[p[0]:H-p[1], p[2]:W-p[3]]`) it returns [top, bottom, left, right]."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from ..nn.liif import make_coord


class InputPadder:
    """Replicate-pad to a multiple of `divis_by` ('sintel' mode splits the padding on both sides)."""

    def __init__(self, dims, mode="sintel", divis_by=8):
        self.ht, self.wd = dims[-2:]
        pad_ht = (((self.ht // divis_by) + 1) * divis_by - self.ht) % divis_by
        pad_wd = (((self.wd // divis_by) + 1) * divis_by - self.wd) % divis_by
        if mode == "sintel":
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        assert all(x.ndim == 4 for x in inputs)
        return [F.pad(x, self._pad, mode="replicate") for x in inputs]

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        return x[..., self._pad[2]:ht - self._pad[3], self._pad[0]:wd - self._pad[1]]

    def get_pad_num(self):
        return [self._pad[2], self._pad[3], self._pad[0], self._pad[1]]


def pad_for_multi_train(image1, image2, scale_test: float, divis_by: int = 32):
    """Down-scale by `scale_test` (bicubic), pad, and build the query coordinates of the WANTED
    full-resolution output inside the padded low-res frame.  Returns
    (image1_pad, image2_pad, hr_coord [H*W,2], scaled pad_num)  — evaluation.py:67-89."""
    assert scale_test > 0.99
    h_want, w_want = image1.shape[-2:]
    h_lr = int(math.ceil(h_want / float(scale_test)))
    w_lr = int(math.ceil(w_want / float(scale_test)))
    if scale_test > 1:
        image1 = F.interpolate(image1, (h_lr, w_lr), mode="bicubic", align_corners=False)
        image2 = F.interpolate(image2, (h_lr, w_lr), mode="bicubic", align_corners=False)
    padder = InputPadder(image1.shape, divis_by=divis_by)
    image1_pad, image2_pad = padder.pad(image1, image2)
    h_hr = int(image1_pad.shape[2] * scale_test)
    w_hr = int(image1_pad.shape[3] * scale_test)
    coord = make_coord([h_hr, w_hr], flatten=False)
    p = [int(i * scale_test) for i in padder.get_pad_num()]
    coord = coord[p[0]:h_hr - p[1], p[2]:w_hr - p[3], :]
    if coord.shape[0] != h_want or coord.shape[1] != w_want:
        coord = F.interpolate(coord.permute(2, 0, 1).unsqueeze(0), (h_want, w_want), mode="bilinear").squeeze(0).permute(1, 2, 0)
    return image1_pad, image2_pad, coord.contiguous().view(h_want * w_want, -1), p


def pad_for_multi_train_fixed(image1, image2, scale: int, divis_by: int = 16):
    """Fixed integer up-scaling of the given low-res pair (evaluation_validate.py:92-106)."""
    h_want, w_want = image1.shape[-2] * scale, image1.shape[-1] * scale
    padder = InputPadder(image1.shape, divis_by=divis_by)
    image1_pad, image2_pad = padder.pad(image1, image2)
    h_hr, w_hr = image1_pad.shape[2] * scale, image1_pad.shape[3] * scale
    coord = make_coord([h_hr, w_hr], flatten=False)
    p = [round(i * scale) for i in padder.get_pad_num()]
    coord = coord[p[0]:h_hr - p[1], p[2]:w_hr - p[3], :]
    assert coord.shape[0] == h_want and coord.shape[1] == w_want
    return image1_pad, image2_pad, coord.contiguous().view(h_want * w_want, -1), p
