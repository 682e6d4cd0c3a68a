import os
import re
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "any-stereo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- run order of the GPU suite -------------------------------------------------------------------------------------------
# The driver runs `pytest -x -m gpu`: one failure stops the run, so the order decides how much evidence a failure leaves behind.
# Cheapest and most local first: kernel-level forward parity -> kernel-level backward / weight-gradient / fused-kernel parity ->
# the torch.ops boundary -> the off-by-default upsampler variants -> whole-model and golden training steps -> the BASELINE
# configurations at full size -> the tests that run bench.py as a child process.  (CPU tests keep their file order.)
_WHOLE_MODEL = re.compile(r"test_(whole_model|training_step_vs_reference|training_step_reduced_precision|training_step_is_bit_repeatable|training_batched_upsampler|"
                          r"trainer_graphed_step|training_fused_liif_mlp_equals_layered|reduced_precision_mode|update_block_links_on_off)")
_BACKWARD = re.compile(r"backward|wgrad|dgrad|deferred|training_fused_gates|mlp_tail_function|accumulates_over_iterations")
_BENCH_CMD = re.compile(r"bench_command")


def gpu_tier(nodeid: str) -> int:
    fname, _, test = nodeid.partition("::")
    fname = os.path.basename(fname)
    if _BENCH_CMD.search(test):
        return 7
    if fname == "test_hip_fullsize.py":
        return 6
    if fname == "test_torch_ops.py":
        return 3
    if fname == "test_liif_variants.py":
        return 4
    if fname == "test_hip_parity.py":
        if _WHOLE_MODEL.search(test):
            return 5
        return 2 if _BACKWARD.search(test) else 1
    return 1


_TIER_NAMES = {1: "kernel forward parity", 2: "kernel backward / wgrad parity", 3: "torch.ops boundary", 4: "upsampler variants",
               5: "whole model + golden training step", 6: "full-size configurations", 7: "bench.py command lines"}


def pytest_collection_modifyitems(session, config, items):
    gpu = [it for it in items if it.get_closest_marker("gpu") is not None]
    if not gpu:
        return
    order = {id(it): i for i, it in enumerate(items)}
    gpu_sorted = sorted(gpu, key=lambda it: (gpu_tier(it.nodeid), order[id(it)]))
    it_gpu = iter(gpu_sorted)
    items[:] = [next(it_gpu) if it.get_closest_marker("gpu") is not None else it for it in items]
    config._anystereo_gpu_order = [it.nodeid for it in gpu_sorted]


def pytest_report_collectionfinish(config, start_path, items):
    sel = [it.nodeid for it in items if it.get_closest_marker("gpu") is not None]
    if not sel:
        return []
    lines, at = ["GPU suite run order (tests/conftest.py):"], 0
    for tier in sorted(_TIER_NAMES):
        n = sum(1 for nid in sel if gpu_tier(nid) == tier)
        if n:
            lines.append(f"  tier {tier} ({_TIER_NAMES[tier]}): tests {at + 1}-{at + n}")
            at += n
    return lines


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    import torch

    def load(name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        return {k: torch.from_numpy(np.ascontiguousarray(z[k])) for k in z.files}

    return load
