"""Every implicit-GEMM convolution launch of one eager inference forward (cfg 2, 1 GRU iteration) with its shape, stand-alone time
(HIP events around the launch) and achieved TFLOP/s — where the one-shot part of the pass spends its matrix time.
    python tools/preloop_convs.py"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo import ops  # noqa: E402
from anystereo.harness.query import pad_for_multi_train  # noqa: E402
from anystereo.harness.synthetic import fill_module_deterministic, synthetic_pair  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402

dev = "cuda:0"
model = __models__["continuous_IGEVStereo"](default_args("continuous_IGEVStereo")).eval()
fill_module_deterministic(model, base_seed=1)
model = model.to(dev)
model.serial_streams = True
img1, img2 = synthetic_pair(1, 540, 960, shift=8, seed=1234)
i1, i2, coord, _ = pad_for_multi_train(img1, img2, 1.0, divis_by=32)
i1, i2, coord = i1.to(dev), i2.to(dev), coord.unsqueeze(0).to(dev)
sc = torch.tensor([[1.0]], device=dev)
recs = []
real = ops.conv2d


def timed(srcs, pack, *a, **k):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = real(srcs, pack, *a, **k)
    e.record()
    b, _, h, w = srcs[0].shape
    stride = k.get("stride", 1)
    ho, wo = (h, w) if stride == 1 else ((h - 1) // 2 + 1, (w - 1) // 2 + 1)
    mult = 2 if k.get("dual") is not None else 1
    recs.append((s, e, pack.cin, pack.cout, pack.ks, stride, b, ho, wo, mult, k.get("epilogue", 0), all(isinstance(t, ops.BS8) for t in srcs)))
    return out


with torch.no_grad():
    for _ in range(2):
        model(i1, i2, iters=1, test_mode=True, hr_coord=coord.clone(), scale=sc)
    torch.cuda.synchronize()
    ops.conv2d = timed
    import anystereo.nn.blocks as B_  # modules hold `ops` by reference: patching the attribute is enough
    model(i1, i2, iters=1, test_mode=True, hr_coord=coord.clone(), scale=sc)
    torch.cuda.synchronize()
    ops.conv2d = real
tot_us = tot_gf = 0.0
rows = []
for s, e, cin, cout, ks, stride, b, ho, wo, mult, epi, bs in recs:
    us = s.elapsed_time(e) * 1e3
    gf = 2.0 * b * ho * wo * cin * cout * ks * ks * mult / 1e9
    rows.append((us, gf, cin, cout, ks, stride, b, ho, wo, mult, epi, bs))
    tot_us += us
    tot_gf += gf
print(f"{len(rows)} conv launches, {tot_us / 1e3:.2f} ms (host-inclusive event spans, eager), {tot_gf:.1f} GFLOP, {tot_gf / tot_us * 1e3:.0f} TFLOP/s overall" if tot_us else "none")
agg = {}
for r in rows:
    k = r[2:]
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += r[0]
    a[2] += r[1]
print(f"{'n':>3s} {'us each':>8s} {'GFLOP':>7s} {'TF/s':>6s}  Cin->Cout k s  B x H x W  x{'':2s} epi blocked")
for k, (n, us, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    cin, cout, ks, stride, b, ho, wo, mult, epi, bs = k
    print(f"{n:3d} {us / n:8.1f} {gf / n:7.2f} {gf / us * 1e3:6.0f}  {cin:4d}->{cout:<4d} {ks} {stride}  {b}x{ho}x{wo} x{mult} epi{epi} {'bs' if bs else 'fp32'}   total {us:7.1f} us")
