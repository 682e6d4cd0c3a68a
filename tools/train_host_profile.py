"""Where the HOST time of the cfg-4 training step goes (run on the GPU box): torch.profiler CPU-side totals of a few steps.

    python tools/train_host_profile.py [--steps 3] [--top 40]
Prints wall ms per step, then ops by self CPU time (per step) and the Python-level cProfile view of the same steps.
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "any-stereo_amd"))
import torch  # noqa: E402

from anystereo.harness.synthetic import fill_module_deterministic  # noqa: E402
from anystereo.harness.train import Trainer, synthetic_train_batch  # noqa: E402
from anystereo.models import __models__, default_args  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--top", type=int, default=40)
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
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step(batch)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"plain: {t_all / a.steps * 1e3:.1f} ms per step wall, host done issuing after {t_issue / a.steps * 1e3:.1f} ms per step")
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        for _ in range(a.steps):
            tr.step(batch)
        torch.cuda.synchronize()
    ev = prof.key_averages()
    rows = sorted(ev, key=lambda e: -e.self_cpu_time_total)
    tot = sum(e.self_cpu_time_total for e in ev)
    print(f"torch.profiler (CPU): {tot / a.steps / 1e3:.1f} ms of op self time per step (all threads), {sum(e.count for e in ev) // a.steps} op calls per step")
    for e in rows[:a.top]:
        print(f"  {e.self_cpu_time_total / a.steps / 1e3:7.2f} ms  n={e.count // a.steps:5d}  {e.key[:90]}")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        tr.step(batch)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime")
    print("cProfile (main thread only; the backward pass runs on autograd's thread), by own time:")
    st.print_stats(a.top)


if __name__ == "__main__":
    main()
