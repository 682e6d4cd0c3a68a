"""Which convolutions of the cfg-4 training step still run on MIOpen (run on the GPU box): torch.profiler with shapes, the
library convolution ops grouped by (op, input shape, weight shape) with calls and device time per step.

    python tools/train_lib_convs.py [--steps 2]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo.harness.synthetic import fill_module_deterministic  # noqa: E402
from anystereo.harness.train import Trainer, synthetic_train_batch  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    dev = "cuda:0"
    args = default_args("continuous_IGEVStereo")
    m = __models__["continuous_IGEVStereo"](args)
    fill_module_deterministic(m, base_seed=1)
    tr = Trainer(m.to(dev), train_iters=16, max_disp=args.max_disp, graph=False)
    batch = synthetic_train_batch(4, 160, 320, seed=0, device=dev)
    for _ in range(3):
        tr.step(batch)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(a.steps):
            tr.step(batch)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    names = ("aten::miopen_convolution", "aten::convolution_backward", "aten::miopen_depthwise_convolution", "aten::miopen_convolution_transpose",
             "aten::miopen_batch_norm", "aten::native_batch_norm_backward", "aten::miopen_batch_norm_backward", "aten::native_batch_norm",
             "aten::threshold_backward", "aten::cat", "aten::add_", "aten::add", "aten::copy_")
    for e in prof.key_averages(group_by_input_shape=True):
        if e.key in names:
            shp = [tuple(s) for s in (e.input_shapes or [])[:3] if s]
            k = (e.key, str(shp[:2]) if "conv" in e.key else str(shp[:1]))
            agg[k][0] += e.count
            agg[k][1] += getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0)
    tot = collections.Counter()
    for (op, shp), (n, t) in agg.items():
        tot[op] += t
    print("device ms per step by op:", {k: round(v / a.steps / 1e3, 2) for k, v in tot.most_common()})
    for (op, shp), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{t / a.steps / 1e3:7.3f} ms  n={n // a.steps:4d}  {op:36s} {shp}")


if __name__ == "__main__":
    main()
