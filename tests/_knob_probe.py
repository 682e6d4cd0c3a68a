"""Child process of test_hip_parity.py::test_schedule_knobs_keep_results: the loop's launches (and two training-shaped ones) at the
cfg-2 sizes on deterministic operands, results saved to argv[1] (.npz).  The environment of the process selects the kernel
schedule (csrc/conv.hip / lookup.hip read their AS_* knobs once, at the first launch)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
from anystereo import _lib as Lb  # noqa: E402
from anystereo import ops  # noqa: E402
from anystereo.harness.synthetic import det_uniform  # noqa: E402

dev = "cuda:0"
b, h, w = 1, 136, 240


class _Res(dict):
    def __setitem__(self, k, v):  # tensors, or tuples of tensors / None (an op's several results)
        if isinstance(v, (tuple, list)):
            for i, t in enumerate(v):
                if torch.is_tensor(t):
                    dict.__setitem__(self, f"{k}.{i}", t)
        elif torch.is_tensor(v):
            dict.__setitem__(self, k, v)


res = _Res()


def to_bs(x):
    bb, cc, hh, ww = x.shape
    hi = x.half()
    lo = ((x - hi.float()) * 2048.0).half()
    return ops.BS8(torch.stack([hi, lo], 1).view(bb, 2, cc // 8, 8, hh, ww).permute(0, 1, 2, 4, 5, 3).contiguous(), cc)


def pack(cout, cin, k, seed, s=0.03):
    return ops.PackedConv().get([det_uniform((cout, cin, k, k), seed, -s, s).to(dev)], [det_uniform((cout,), seed + 1, -0.1, 0.1).to(dev)])


pzr, pq = pack(256, 384, 3, 30, 0.02), pack(128, 384, 3, 32, 0.02)
for name, div in (("04", 1), ("08", 2), ("16", 4)):
    hh, ww = h // div, w // div
    xs32 = [det_uniform((b, 128, hh, ww), 40 + i).to(dev) for i in range(3)]
    xs = [to_bs(t) for t in xs32]
    cx = det_uniform((b, 384, hh, ww), 50).to(dev)
    z_ = det_uniform((b, 128, hh, ww), 51, 0.0, 1.0).to(dev)
    o1, o2 = ops.BS8.empty(b, 128, hh, ww, dev), ops.BS8.empty(b, 128, hh, ww, dev)
    ops.conv2d(xs, pzr, add=cx, add_coff=0, epilogue=Lb.EPI_GRU_ZR, h=xs32[0], out_bs=o1, bs_only=True)
    res["zr" + name] = o1.float()
    res["q" + name] = ops.conv2d(xs, pq, add=cx, add_coff=256, epilogue=Lb.EPI_GRU_Q, h=xs32[0], z=z_, out_bs=o2)
    res["q_bs" + name] = o2.float()
    res["pool" + name] = ops.pool2x_bs(xs32[0]).float()
    if div > 1:
        res["interp" + name] = ops.interp_bs(xs32[1], hh * 2, ww * 2).float()
# motion encoder and disparity head, blocked links
x64 = [to_bs(det_uniform((b, 64, h, w), 90 + i).to(dev)) for i in range(2)]
cd = ops.BS8.empty(b, 128, h, w, dev)
ops.conv2d([x64[0]], pack(64, 64, 3, 92, 0.05), act=Lb.ACT_RELU, out_bs=cd, out_bs_coff=0, bs_only=True,
           dual={"src": x64[1], "pack": pack(64, 64, 3, 94, 0.05), "out_coff": 64, "out_bs_coff": 64})
res["enc_c2d2"] = cd.float()
mf = ops.BS8.empty(b, 128, h, w, dev)
ops.conv2d([cd], pack(127, 128, 3, 96, 0.04), act=Lb.ACT_RELU, out_bs=mf, out_bs_coff=0, bs_only=True)
res["enc_conv"] = mf.float()[:, :127]
x128 = to_bs(det_uniform((b, 128, h, w), 10).to(dev))
res["head_taps"] = ops.conv2d([x128], pack(256, 128, 3, 98, 0.04), act=Lb.ACT_RELU, epilogue=Lb.EPI_RELU_TAPS,
                              tap_w=det_uniform((256, 9), 100, -0.05, 0.05).to(dev))
# fp32 sources: a context-network layer and two training-shaped layers (batch 4, 40x80)
xf = det_uniform((1, 64, 2 * h, 2 * w), 61).to(dev)
res["cnet_64"] = ops.conv2d([xf], pack(64, 64, 3, 62, 0.05), act=Lb.ACT_RELU)
xt = det_uniform((4, 128, 40, 80), 63).to(dev)
res["train_128_256"] = ops.conv2d([xt], pack(256, 128, 3, 64, 0.04))
res["train_128_128_1x1"] = ops.conv2d([xt], pack(128, 128, 1, 66, 0.08), act=Lb.ACT_RELU)
# the lookup fused with convc1
f1, f2 = det_uniform((b, 96, h, w), 1).to(dev), det_uniform((b, 96, h, w), 2).to(dev)
gev = det_uniform((b, 8, 48, h, w), 3).to(dev)
disp = det_uniform((b, 1, h, w), 4, 0.0, 40.0).to(dev)
corr, geo = ops.corr_build_pyramid(f1, f2, 2), ops.geo_pyramid(gev, 2)
plc1 = ops.LookupConvPack().get((det_uniform((64, 162, 1, 1), 77) * (3.0 / 162) ** 0.5).to(dev), det_uniform((64,), 78, -0.1, 0.1).to(dev))
cor = ops.BS8.empty(b, 64, h, w, dev)
ops.lookup_convc1(geo, corr, disp, 4, plc1, out_bs=cor)
res["lookup_convc1"] = cor.float()
torch.cuda.synchronize()
np.savez(sys.argv[1], **{k: v.detach().float().cpu().numpy() for k, v in res.items() if torch.is_tensor(v)})
print("saved", len(res))
