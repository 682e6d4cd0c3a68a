"""Host-side cost of one whole-forward hipGraph replay (cfg 2): is hipGraphLaunch asynchronous on this stack, and how much host
time sits between two passes?   python tools/graph_launch_cost.py"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo.harness import workloads as WL  # noqa: E402

wl = WL.WORKLOADS["cfg2"]
model, args = WL.build_model(wl, device="cuda:0")
model.enable_graph(True)
i1, i2, coord, sc = (t.to("cuda:0") for t in WL.build_inputs(wl))
with torch.no_grad():
    for _ in range(3):
        out = model(i1, i2, iters=wl.iters, test_mode=True, hr_coord=coord.clone(), scale=sc)
    torch.cuda.synchronize()
    ent = next(iter(model._graphs.values()))
    g = ent[0]
    for label, fn in (("g.replay() alone", lambda: g.replay()),
                      ("model(...) (fingerprint + 4 input copies + replay + clone)",
                       lambda: model(i1, i2, iters=wl.iters, test_mode=True, hr_coord=coord, scale=sc)),
                      ("weights fingerprint alone", lambda: model._weights_fingerprint())):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # 5 back to back: does the host run ahead of the GPU?
        t3 = time.perf_counter()
        for _ in range(5):
            fn()
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        print(f"{label}: host returns after {1e3 * (t1 - t0):.3f} ms, device done after {1e3 * (t2 - t0):.3f} ms; "
              f"5 calls: host {1e3 * (t4 - t3):.3f} ms, device done {1e3 * (t5 - t3):.3f} ms ({1e3 * (t5 - t3) / 5:.3f} per call)")
