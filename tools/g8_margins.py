"""Per-tensor deviation of the product's G8 gradients from the reference fixture next to the reference's own 128-perturbation spread
(tests/golden/train_*_sens.npz): the data behind the per-tensor limits of tests/test_hip_parity.py::_g8_limits.
    python tools/g8_margins.py out.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "any-stereo_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_hip_parity as T  # noqa: E402

out = {}
for name in ("igev", "raft"):
    z = np.load(os.path.join(ROOT, "tests", "golden", f"train_{name}.npz"))
    s = np.load(os.path.join(ROOT, "tests", "golden", f"train_{name}_sens.npz"))
    names = [str(n) for n in z["names"]]
    dev = {str(n): float(d) for n, d in zip(s["names"], s["norm_dev"])}
    for mode in ("split", "fp32"):
        loss, preds, grads = T._g8_run(name, mode)
        norms = np.array([float(grads[n].double().norm()) for n in names])
        zero = z["norms"] < T.G8_ZERO_REF * z["norms"].max()
        rel = np.where(zero, 0.0, np.abs(norms - z["norms"]) / (z["norms"] + 1e-6 * z["norms"].max()))
        rows = [{"name": n, "rel": float(r), "spread": dev[n]} for n, r in zip(names, rel)]
        full = []
        for i, n in enumerate(str(x) for x in z["full_names"]):
            want = torch.from_numpy(z[f"g{i}"])
            e = ((grads[n].cpu() - want).abs().max() / want.abs().max()).item()
            full.append({"name": n, "elem": e, "spread": float(s["full_dev"][i])})
        out[f"{name}_{mode}"] = {"norms": rows, "full": full}
        small = [r for r in rows if r["spread"] < 1e-4]
        print(name, mode, "tensors with spread < 1e-4:", len(small), "max rel", max(r["rel"] for r in small),
              "exceeding own spread:", sum(r["rel"] > r["spread"] for r in small),
              "exceeding max(spread, 1e-4):", sum(r["rel"] > max(r["spread"], 1e-4) for r in small),
              "exceeding max(spread, 3e-4):", sum(r["rel"] > max(r["spread"], 3e-4) for r in small))
json.dump(out, open(sys.argv[1], "w"))
