"""Headline benchmark (BASELINE.json): stereo pairs/s and ms per GRU iteration of
coreContinuous_IGEV inference on a 960x540 SceneFlow-shape synthetic pair, 32 iterations, fp32.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; every rank runs the same per-GPU workload on its own pair (independent
units, no data-path collective: "replicas", weak scaling).  A step = one full forward pass
(backbones on PyTorch-ROCm/MIOpen, hot path on libanystereo_hip.so) with inputs resident in HBM.
Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      dominant kernel (by time) measured with HIP events on its launch stream in the timed steps
  rooflines     the same figure for every hot kernel class (north_star quotes build+lookup vs HBM)
  cpu_baseline  the CPU oracle (oracle/model.py) timed on this host's cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "any-stereo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA dense peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: fp16/bf16 MFMA dense peak (~2.5 PF)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--iters", type=int, default=32)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--pairs-per-gpu", type=int, default=1,
                    help="stereo pairs per forward on each GPU (1 = the reference's evaluation protocol; >1 = throughput mode)")
    ap.add_argument("--no-batched", action="store_true", help="skip the extra 4-pairs-per-forward throughput measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run the GRU loop eagerly instead of as a captured hipGraph")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer = the headline benchmark (default); train = cfg 4: DDP training steps at 160x320, 16 GRU iterations")
    ap.add_argument("--batch-per-gpu", type=int, default=4, help="train mode: samples per rank (global batch 32 = 4 x 8)")
    ap.add_argument("--train-iters", type=int, default=16)
    return ap.parse_args()


def algorithmic(B, h, w, Q, iters, C=96, L=2, G=8, D=48, r=4):
    """Algorithmic bytes / flops per launch."""
    P = B * h * w
    build_b = 4 * (2 * B * C * h * w + sum(P * (w >> i) for i in range(L)))
    geo_b = 4 * (B * G * D * h * w + sum(P * G * (D >> i) for i in range(L)))
    lookup_b = 4 * P * (L * (G + 1) * (2 * r + 2) + 1 + L * (G + 1) * (2 * r + 1))
    return {
        "corr_build": {"bound": "hbm", "bytes": build_b, "flops": 2 * C * P * w},
        "geo_pyramid": {"bound": "hbm", "bytes": geo_b},
        "lookup": {"bound": "hbm", "bytes": lookup_b},
        "gwc_volume": {"bound": "hbm", "bytes": 4 * (2 * B * C * h * w + B * G * D * h * w)},
        # 3x3 convs of gru04: zr = (3*128 -> 256), q = (3*128 -> 128)
        "gru04_zr_conv": {"bound": "mfma", "flops": 2 * P * 384 * 9 * 256},
        "gru04_q_conv": {"bound": "mfma", "flops": 2 * P * 384 * 9 * 128},
        "disp_head_conv1": {"bound": "mfma", "flops": 2 * P * 128 * 9 * 256},
        # layers 2..4 at query resolution (the first Linear layer runs at low resolution: liif_mlp_lowres)
        "liif_mlp": {"bound": "mfma", "flops": 2 * Q * B * (128 * 64 + 64 * 64 + 64 * 9)},
    }


def train_main(a):
    """SURVEY.md §8d cfg 4: IGEV training, 4 samples per GPU at 160x320 network input, 51 200 HR queries per sample, 16 GRU
    iterations with the LIIF upsampler every iteration, AdamW + OneCycleLR; one process per GPU, DDP over RCCL.  A step =
    zero_grad + forward + loss + backward (+ bucketed gradient all-reduce) + clip + optimizer + scheduler step."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = world > 1 or "RANK" in os.environ  # under torchrun even one rank goes through RCCL + DDP
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        td.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from anystereo import _lib
    from anystereo.harness.synthetic import fill_module_deterministic
    from anystereo.harness.train import Trainer, synthetic_train_batch
    from anystereo.models import __models__, default_args
    _lib.load()
    args = default_args("continuous_IGEVStereo")
    model = __models__["continuous_IGEVStereo"](args)
    fill_module_deterministic(model, base_seed=1)
    model = model.to(dev)
    tr = Trainer(model, train_iters=a.train_iters, max_disp=args.max_disp, force_ddp=dist)
    h, w = (160, 320) if (a.height, a.width) == (540, 960) else (a.height, a.width)
    batch = synthetic_train_batch(a.batch_per_gpu, h, w, seed=rank, device=dev)
    losses = []
    for _ in range(a.warmup):
        losses.append(float(tr.step(batch)[0]))
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        torch.cuda._sleep(2000)  # ~1 us `spin_kernel`: step marker for tools/profile_train.sh (separates MIOpen's search in warm-up)
        loss, _ = tr.step(batch)
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses.append(float(loss))
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        nparam = sum(p.numel() for p in model.parameters())
        print(json.dumps({
            "metric": "train_samples_per_s", "value": round(world * a.batch_per_gpu * a.steps / dt, 3), "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg4 continuous_IGEVStereo training {h}x{w}, {a.train_iters} GRU iters, LIIF every iter, "
                                   f"Q={batch[2].shape[1]} queries/sample, AdamW+OneCycleLR, clip 1.0",
                       "global_batch": world * a.batch_per_gpu, "parallelism": f"ddp x{world} (RCCL all-reduce of {nparam} fp32 grads)"},
            "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)],
            "backward": "HIP kernels for the volume/lookup/gwc/LIIF/convex-upsample transposes and the update-block/MLP dgrad; wgrad and backbone convs on MIOpen/rocBLAS",
            "roofline": None, "cpu_baseline": None}))
    if dist:
        td.barrier()
        td.destroy_process_group()


def main():
    a = parse()
    if a.mode == "train":
        return train_main(a)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = world > 1
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        td.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from anystereo import _lib
    from anystereo.harness import timing
    from anystereo.harness.query import pad_for_multi_train
    from anystereo.harness.synthetic import fill_module_deterministic, synthetic_pair
    from anystereo.models import __models__, default_args

    _lib.load()
    from anystereo import ops as _ops
    precision = _ops.get_precision()
    torch.backends.cudnn.benchmark = bool(int(os.environ.get("ANYSTEREO_MIOPEN_BENCHMARK", "0")))
    args = default_args("continuous_IGEVStereo")
    model = __models__["continuous_IGEVStereo"](args).eval()
    fill_module_deterministic(model, base_seed=1)  # random-init weights of the named architecture (no checkpoints offline)
    model = model.to(dev)

    img1, img2 = synthetic_pair(1, a.height, a.width, shift=8, seed=1234 + rank)
    i1, i2, coord, _ = pad_for_multi_train(img1, img2, a.scale, divis_by=32)
    nb = max(1, a.pairs_per_gpu)
    i1, i2 = i1.to(dev).repeat(nb, 1, 1, 1), i2.to(dev).repeat(nb, 1, 1, 1)
    coord = coord.unsqueeze(0).to(dev).repeat(nb, 1, 1)
    scale = torch.tensor([[a.scale]] * nb, device=dev)
    Q = coord.shape[1]
    hp, wp = i1.shape[-2:]
    use_graph = not a.no_graph
    if hasattr(model, "enable_graph"):
        model.enable_graph(use_graph)

    def step(iters=a.iters):
        with torch.no_grad():
            return model(i1, i2, iters=iters, test_mode=True, hr_coord=coord, scale=scale)

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    graphed = use_graph and hasattr(model, "enable_graph")
    if not graphed:
        timing.enable(True)       # HIP events around every hot-kernel launch, on the launch stream
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if graphed:
        # the timed steps replay one captured hipGraph (no per-kernel events inside a replay): take the
        # per-kernel device times from instrumented EAGER passes of the identical workload right after
        model.enable_graph(False)
        par = model.update_block.parallel_encoder
        model.update_block.parallel_encoder = False   # isolated kernel durations: no co-running stream
        step()
        timing.enable(True)
        for _ in range(2):
            # a ~60 ms device-side delay first: the host then enqueues the pass AHEAD of the GPU, so an event pair measures the
            # kernel between them and not the Python / ctypes time between recording the start event and the launch (which
            # inflated the multi-source conv launches by 10-20 % against rocprofv3's durations of the same run)
            torch.cuda._sleep(120_000_000)
            step()
        kstats = timing.collect()
        timing.enable(False)
        model.update_block.parallel_encoder = par
        model.enable_graph(True)
        ksteps = 2
    else:
        kstats = timing.collect()
        timing.enable(False)
        ksteps = a.steps
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out).all()

    # ms per GRU iteration = (t32 - t8) / 24 on the same inputs, outside the timed region
    def timed(iters, reps=3):
        step(iters)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            step(iters)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps
    lo = max(1, a.iters // 4)
    ms_iter = (timed(a.iters) - timed(lo)) / (a.iters - lo) * 1e3 if a.iters > lo else None

    # throughput mode (reported next to the headline, never as `value`): 4 pairs per forward fill the 1/8- and
    # 1/16-resolution kernels that leave most CUs idle at one pair; every rank measures, rank 0 reports the job total
    batched = None
    if nb == 1 and not a.no_batched:
        nbb = 4
        bi1, bi2 = i1.repeat(nbb, 1, 1, 1), i2.repeat(nbb, 1, 1, 1)
        bcoord, bscale = coord.repeat(nbb, 1, 1), scale.repeat(nbb, 1)
        with torch.no_grad():
            for _ in range(2):
                model(bi1, bi2, iters=a.iters, test_mode=True, hr_coord=bcoord, scale=bscale)
            torch.cuda.synchronize()
            if dist:
                td.barrier()
            tb = time.perf_counter()
            for _ in range(3):
                model(bi1, bi2, iters=a.iters, test_mode=True, hr_coord=bcoord, scale=bscale)
            torch.cuda.synchronize()
            if dist:
                td.barrier()
            dtb = time.perf_counter() - tb
        if dist:
            tt = torch.tensor([dtb], device=dev, dtype=torch.float64)
            td.all_reduce(tt, op=td.ReduceOp.MAX)
            dtb = float(tt.item())
        batched = {"pairs_per_gpu": nbb, "value": round(world * nbb * 3 / dtb, 4), "unit": "pairs/s",
                   "ms_per_step": round(dtb / 3 * 1e3, 3), "steps": 3}
        del bi1, bi2, bcoord, bscale

    if rank == 0:
        alg = algorithmic(nb, hp // 4, wp // 4, Q, a.iters)
        # the short HBM-bound kernels (10-60 us) are re-timed as 20 back-to-back launches (one hipGraph) between ONE
        # event pair: a start/stop pair around a single launch adds ~3 us of its own (rocprofv3 durations confirm)
        micro = micro_time_small_kernels(dev, hp // 4, wp // 4)
        for k, v in micro.items():
            kstats[k] = v
        traffic = load_pmc_traffic()
        rooflines = {}
        for name, st in kstats.items():
            if name not in alg or st["count"] == 0:
                continue
            avg_s = st["total_ms"] / st["count"] * 1e-3
            e = alg[name]
            if e["bound"] == "hbm":
                ach = e["bytes"] / avg_s / 1e9
                rooflines[name] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic.get(name),
                                   "avg_us": round(avg_s * 1e6, 2), "launches": st["count"], "total_ms": round(st["total_ms"], 3)}
            else:
                ach = e["flops"] / avg_s / 1e12
                # split precision spends 3 fp16 MFMA products per algorithmic product: the attainable peak for
                # ALGORITHMIC flops is the fp16 dense peak / 3; in fp32 mode it is the fp32-input MFMA peak
                peak = MFMA_F16_PEAK_TFLOPS / 3.0 if precision == "split" else MFMA_F32_PEAK_TFLOPS
                rooflines[name] = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                                   "frac": round(ach / peak, 4), "traffic": traffic.get(name),
                                   "avg_us": round(avg_s * 1e6, 2), "launches": st["count"], "total_ms": round(st["total_ms"], 3)}
        dominant = max(rooflines, key=lambda k: rooflines[k]["total_ms"]) if rooflines else None
        # north_star's own bar: >= 40 % of the HBM roofline on volume build + lookup (a1-a3) — each kernel and the pair together
        ns = None
        if all(k in rooflines for k in ("corr_build", "geo_pyramid", "lookup")):
            ks = ("corr_build", "geo_pyramid", "lookup")
            t_us = sum(rooflines[k]["avg_us"] for k in ks)
            byt = sum(rooflines[k]["achieved"] * rooflines[k]["avg_us"] for k in ks)  # GB/s * us = kB
            ns = {"build_plus_lookup_hbm_frac": round(byt / t_us / HBM_PEAK_GBS, 4), "target": 0.40,
                  "per_kernel": {k: rooflines[k]["frac"] for k in ks},
                  "note": "one build (all-pairs pyramid + geometry pyramid) + one lookup, algorithmic bytes / measured time / 8 TB/s"}
        cpu = None
        if not a.no_cpu_baseline and world == 1:  # reported at N=1 only (rank 0), bounded sample
            cpu = cpu_baseline(args, model, img1, img2, a)
        line = {
            "metric": "stereo pairs/sec (coreContinuous_IGEV inference, 32-iter GRU, 960x540)",
            "value": round(world * nb * a.steps / dt, 4), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if precision == "fp32" else "f32 (3xf16 split-precision MFMA, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"coreContinuous_IGEV inference, {a.width}x{a.height} SceneFlow-shape synthetic pair "
                                   f"(padded {wp}x{hp}), {a.iters} GRU iters, scale {a.scale}, Q={Q} queries, {nb} pair(s) per GPU, "
                                   f"random-init weights", "pairs_per_gpu": nb, "parallelism": f"replicas x{world}",
                       "gru_loop": "hipGraph" if use_graph and hasattr(model, "enable_graph") else "eager"},
            "ms_per_gru_iter": None if ms_iter is None else round(ms_iter, 4),
            "roofline": dict(rooflines[dominant], kernel=dominant) if dominant else None,
            "roofline_source": ("HIP events on the launch stream around each hot-kernel launch, " +
                                ("2 eager passes of the same workload right after the timed hipGraph replays" if graphed
                                 else "inside the timed steps")),
            "rooflines": rooflines,
            "north_star_roofline": ns,
            "kernel_times_us": {k: {"avg": round(v["total_ms"] / max(v["count"], 1) * 1e3, 2), "n": v["count"] // ksteps}
                                for k, v in sorted(kstats.items(), key=lambda kv: -kv[1]["total_ms"])},
            "throughput_mode": batched,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist:
        td.barrier()
        td.destroy_process_group()


def micro_time_small_kernels(dev, h, w, reps=20):
    """corr_build and lookup on operands of the workload's shapes (C=96, L=2, G=8, D=48): `reps` launches captured
    into one hipGraph (the host launch path, ~20 us per call through Python + ctypes, is longer than these kernels)
    and replayed between one HIP event pair on the launch stream."""
    from anystereo import ops
    from anystereo.harness.synthetic import det_uniform
    f1 = det_uniform((1, 96, h, w), 1).to(dev)
    f2 = det_uniform((1, 96, h, w), 2).to(dev)
    gev = det_uniform((1, 8, 48, h, w), 3).to(dev)
    disp = det_uniform((1, 1, h, w), 4, 0.0, 40.0).to(dev)
    corr = ops.corr_build_pyramid(f1, f2, 2)
    geo = ops.geo_pyramid(gev, 2)
    out = {}
    for name, fn in (("corr_build", lambda: ops.corr_build_pyramid(f1, f2, 2)),
                     ("lookup", lambda: ops.geo_corr_lookup(geo, corr, disp, 4)),
                     ("gwc_volume", lambda: ops.gwc_volume(f1, f2, 48, 8)),
                     ("geo_pyramid", lambda: ops.geo_pyramid(gev, 2))):
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        out[name] = {"count": reps, "total_ms": s.elapsed_time(e)}
        del g
    return out


def load_pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command (profiles/rNN_pmc_traffic.json, produced by tools/profile_round.sh; gfx950 FETCH_SIZE correction applied)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return {}
    d = json.load(open(files[-1]))
    return {k: (None if v.get("hbm_bytes") is None else int(v["hbm_bytes"])) for k, v in d.items() if not k.startswith("_")}


def cpu_baseline(args, model, img1, img2, a):
    """The CPU oracle (same weights) timed on this host: 1 pair of the same workload; if the machine is
    slow the GRU iteration count of the sample is reduced and the 32-iteration figure extrapolated."""
    from anystereo.harness.query import pad_for_multi_train
    from oracle.model import OracleIGEV
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    torch.set_num_threads(threads)
    ref = OracleIGEV(args).eval()
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    i1, i2, coord, _ = pad_for_multi_train(img1, img2, a.scale, divis_by=32)
    coord = coord.unsqueeze(0)
    sc = torch.tensor([[a.scale]])

    def run(iters):
        t = time.perf_counter()
        with torch.no_grad():
            ref(i1, i2, iters=iters, test_mode=True, hr_coord=coord.clone(), scale=sc)
        return time.perf_counter() - t
    run(1)  # warm-up (thread pools, allocator)
    t2, t6 = run(2), run(6)
    per_iter = max((t6 - t2) / 4, 1e-6)
    est_full = t2 + per_iter * (a.iters - 2)
    if est_full <= 45.0:
        t_full = run(a.iters)
        sample = f"1 pair, full workload ({a.iters} iters), fp32, {threads} threads"
    else:
        t_full = est_full
        sample = (f"1 pair at 2 and 6 GRU iters ({t2:.1f}s, {t6:.1f}s), extrapolated to {a.iters} iters, fp32, "
                  f"{threads} threads")
    return {"value": round(1.0 / t_full, 5), "unit": "pairs/s", "cores": threads, "kind": "port", "sample": sample,
            "s_per_pair": round(t_full, 3), "ms_per_gru_iter": round(per_iter * 1e3, 2)}


if __name__ == "__main__":
    main()
