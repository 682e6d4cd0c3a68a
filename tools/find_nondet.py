"""Locate run-to-run differences: records every module output of an eager forward and compares with a second run."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo.harness.query import pad_for_multi_train  # noqa: E402
from anystereo.harness.synthetic import fill_module_deterministic, synthetic_pair  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402

dev = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 4
model = __models__["continuous_IGEVStereo"](default_args("continuous_IGEVStereo")).eval()
fill_module_deterministic(model, base_seed=1)
model = model.to(dev)
img1, img2 = synthetic_pair(1, 540, 960, shift=8, seed=1234)
i1, i2, coord, _ = pad_for_multi_train(img1, img2, 1.0, divis_by=32)
i1, i2, coord = i1.to(dev), i2.to(dev), coord.unsqueeze(0).to(dev)
sc = torch.tensor([[1.0]], device=dev)
log = []


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in flat(x)]
    return []


def hook(name):
    def f(m, inp, out):
        log.append((name, [t.detach().clone() for t in flat(out)]))
    return f


for name, m in model.named_modules():
    if name:
        m.register_forward_hook(hook(name))


def run():
    log.clear()
    with torch.no_grad():
        out = model(i1, i2, iters=iters, test_mode=True, hr_coord=coord.clone(), scale=sc)
    torch.cuda.synchronize()
    return out.clone(), list(log)


ref_out, ref_log = run()
for t in range(trials):
    out, lg = run()
    same = torch.equal(out, ref_out)
    first = None
    for (n1, a), (n2, b) in zip(ref_log, lg):
        assert n1 == n2
        for x, y in zip(a, b):
            if not torch.equal(x, y):
                first = (n1, tuple(x.shape), (x - y).abs().max().item(), int((x != y).sum().item()))
                break
        if first:
            break
    print(f"trial {t}: final equal={same} first differing module: {first}", flush=True)
