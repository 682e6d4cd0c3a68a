"""How many memset nodes the captured inference graphs hold, per BASELINE config (run on the GPU box):
    python tools/infer_graph_memsets.py
Prints (replaced, left) of as_graph_replace_memsets for cfg1 (RAFT), cfg2, cfg3 and checks graph == eager on each."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo.harness.workloads import WORKLOADS, build_inputs, build_model  # noqa: E402

for name in ("cfg1", "cfg2", "cfg3"):
    wl = WORKLOADS[name]
    model = build_model(wl, device="cuda:0")[0].eval()
    inp = build_inputs(wl, device="cuda:0")
    with torch.no_grad():
        eager = model(*inp[:2], iters=wl.iters, test_mode=True, hr_coord=inp[2].clone(), scale=inp[3])
        model.enable_graph(True)
        outs = [model(*inp[:2], iters=wl.iters, test_mode=True, hr_coord=inp[2].clone(), scale=inp[3]) for _ in range(3)]
    d = max((o - eager).abs().max().item() for o in outs)
    print(f"{name}: memset nodes (replaced, left) = {model.__dict__.get('_graph_memsets')}; max |graph - eager| over 3 replays = {d:.3e}", flush=True)
    del model
    torch.cuda.empty_cache()
