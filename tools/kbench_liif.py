"""Stand-alone timing of the fused LIIF pipeline at a BASELINE configuration's shapes (default cfg 2: 1/4-res map 136x240,
518 400 queries): affinity, low-resolution first layer, per-query tail.  For rocprofv3 --kernel-trace / --pmc runs.
    python tools/kbench_liif.py [cfg2|cfg3|cfg5] [reps]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch  # noqa: E402

from anystereo import ops  # noqa: E402
from anystereo.harness import workloads as WL  # noqa: E402
from anystereo.harness.synthetic import det_uniform  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
wl = WL.WORKLOADS[cfg]
model, args = WL.build_model(wl, device=dev)
i1, i2, coord, sc = WL.build_inputs(wl, device=dev)
h, w = i1.shape[-2] // 4, i1.shape[-1] // 4
stem4 = det_uniform((1, 48, h, w), 1).to(dev)
net0 = det_uniform((1, 128, h, w), 2).to(dev)
stem2 = det_uniform((1, 32, 2 * h, 2 * w), 3).to(dev)
disp = det_uniform((1, 1, h, w), 4, 0.0, 60.0).to(dev)
sv = sc.reshape(-1).float().contiguous()
up = model.liif_up
with torch.no_grad():
    for _ in range(3):
        out = up.upsample_fused([[stem4, net0], [stem2]], coord.clone(), disp, sv)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c2 = coord.clone()
    s.record()
    for _ in range(reps):
        out = up.upsample_fused([[stem4, net0], [stem2]], c2, disp, sv)
    e.record()
    torch.cuda.synchronize()
if len(sys.argv) > 3 and sys.argv[3] == "graph":  # the pipeline captured once and replayed: GPU time without the host launch path
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            up.upsample_fused([[stem4, net0], [stem2]], c2, disp, sv)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                out = up.upsample_fused([[stem4, net0], [stem2]], c2, disp, sv)
        g.replay()
        torch.cuda.synchronize()
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
    print(f"{cfg}: graph replay {s.elapsed_time(e) / reps * 1e3:.1f} us per upsample_fused call (direct_second_input={up.direct_second_input})")
print(f"{cfg}: Q={coord.shape[1]} fused LIIF {s.elapsed_time(e) / reps * 1e3:.1f} us per call (host-paced eager launches), "
      f"finite={bool(torch.isfinite(out).all())} overflow_waves={ops.split_overflow_count()}")
