"""Headline benchmark (BASELINE.json): stereo pairs/s and ms per GRU iteration of coreContinuous_IGEV inference on a
960x540 SceneFlow-shape synthetic pair, 32 iterations (cfg 2), fp32 storage.

    python bench.py [--gpus N --steps K --warmup W]            N > 1: this process spawns N ranks itself (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     (the driver's form)

One process per GPU over RCCL; every rank runs the same per-GPU workload on its own pair (independent units, no
data-path collective: "replicas", weak scaling).  A step = one full forward pass (backbones on PyTorch-ROCm/MIOpen, hot path
on libanystereo_hip.so) with inputs resident in HBM.  `value` is the median of three timed blocks of --steps steps (`value_spread`).
With a process group (N > 1, or any N under torch.distributed.run) the same ranks then run the cfg-4 TRAINING leg (`train_leg`:
graphed steps, one flat RCCL all-reduce per step) and `train_mode` carries its N-rank figures; one rank without a launcher runs that
leg as a child process.  Rank 0 prints ONE JSON line (stdout carries nothing else).  Extra objects on that line:
  pass_phases     pre-loop / loop / upsampler wall time of a replayed pass from marker kernels inside the graph (no profiler)
  roofline        dominant kernel (by time), HIP events on its launch stream
  rooflines       the same figure for every hot kernel class; HBM-bound ones warm (working set in the Infinity Cache),
                  cold (operand sets rotated, > 256 MB) and co-scheduled inside the two-stream GRU loop
  parity          EPE of this run's output against the CPU oracle's output on the same input, split and fp32 mode
  fp32_mode       the same workload in exact-fp32 MFMA mode (the same-precision figure next to the split-precision headline)
  reduced_precision_mode  the same workload with fp16 operands / one MFMA per product (the reference's autocast path)
  other_configs   cfg 3 (KITTI x2.0), cfg 5 (Middlebury-F x1.5, 48 iterations) and cfg 1 (corePrune_RAFT, with its EPE vs the oracle)
                  pairs/s, N = 1 only
  train_mode      cfg 4: samples/s, ms per step, exchange_ms, ranks_seen, scaling (N ranks) | the 1-rank child's line + its
                  reduced-precision leg (the reference's autocast arithmetic)
  cpu_baseline    the CPU oracle (oracle/model.py) timed on this host as BASELINE.md §4 states: threads = usable physical cores, 1 warm-up
                  + 3 runs, median, per-stage split; cfg 2 (and cfg 1 under `also`)
`--mode train` = cfg 4 (DDP training step).  Without a visible GPU the launcher protocol alone runs (gloo, "dry_run").
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "any-stereo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

RANK_CPUS = None  # this rank's CPU set (pin_rank)
torch = None  # imported by main() in the worker ranks only: the spawning parent never touches torch or the GPU

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA dense peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: fp16/bf16 MFMA dense peak (~2.5 PF)
# What the chip sustains on this instruction mix with RANDOM operands (it lowers its clock under MFMA load; MI355X_MICROARCH.md "DVFS
# give-back"): tools/experiments/mfma_shape2.hip — the conv consumer's exact work per wave (64 x 64 tile, 3 MFMAs per product, every
# operand re-read from LDS), software-pipelined to 98.5 % issue efficiency, four waves per CU on 256 CUs: 1.51 PFLOP/s fp16 =
# 503 TFLOP/s of algorithmic (split) flops at an in-kernel clock of 1.52 GHz; the less tightly issued round-4 form of the same loop
# gives the same 2.39 us per 16-channel chunk at 1.70 GHz (profiles/r05_mfma_shape_microbench.txt).  Reported beside `frac`, which stays
# against the nominal dense peak.
MFMA_F16_RANDOM_DATA_TFLOPS = 1510.0
INFINITY_CACHE_MB = 256
TIMED_BLOCKS = 3  # `value` is the median of this many timed blocks of `--steps` steps each


_REAL_STDOUT = None  # the process's stdout as it was at start; fd 1 itself is pointed at stderr while the ranks work


def claim_stdout():
    """Rank 0 prints ONE JSON line on stdout — and nothing else may: RCCL writes a version banner to fd 1 when its first
    communicator comes up, MIOpen / hipBLASLt may log there too.  fd 1 is therefore re-pointed at stderr for the life of the
    process (native writes and stray prints end up in the log) and the line goes out through a duplicate of the original fd."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        try:
            sys.stdout.flush()
            fd = os.dup(1)
            os.dup2(2, 1)
            _REAL_STDOUT = fd
        except OSError:  # stdout / stderr closed by the launcher: print through sys.stdout as before
            _REAL_STDOUT = None


# Rehearsal of the N-rank protocol on a ONE-GPU box (ANYSTEREO_BENCH_ONE_GPU_REHEARSAL=1): every rank uses device 0 and the
# process group is gloo (RCCL refuses two ranks on one device).  Everything else — the rank protocol, the Trainer, the flat
# exchange, the agreement rounds, the watchdog — is the code an N-GPU node runs.  The line says so; its numbers are NOT a result.
REHEARSAL = os.environ.get("ANYSTEREO_BENCH_ONE_GPU_REHEARSAL", "0") == "1"


def emit(line: dict):
    if REHEARSAL:
        line = dict(line, rehearsal="all ranks on GPU 0 over gloo (ANYSTEREO_BENCH_ONE_GPU_REHEARSAL=1): protocol check, not a measurement")
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg5"],
                    help="workload of `value` (harness/workloads.py); the default is the configuration BASELINE.json's metric is quoted on")
    ap.add_argument("--height", type=int, default=None, help="custom workload: wanted output height (with --width)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--scale", type=float, default=None)
    ap.add_argument("--pairs-per-gpu", type=int, default=1,
                    help="stereo pairs per forward on each GPU (1 = the reference's evaluation protocol; >1 = throughput mode)")
    ap.add_argument("--no-batched", action="store_true", help="skip the extra 4-pairs-per-forward throughput measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip fp32_mode / other_configs / cold-cache timing (profiling runs)")
    ap.add_argument("--no-graph", action="store_true", help="run the GRU loop eagerly instead of as a captured hipGraph")
    ap.add_argument("--serial-loop", action="store_true",
                    help="profiling runs: every kernel of the GRU loop on ONE stream (same kernels, same order), so that a rocprofv3 "
                         "average is the kernel's own duration, as the HIP-event averages in kernel_times_us are; implies --no-graph")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer = the headline benchmark (default); train = cfg 4: DDP training steps at 160x320, 16 GRU iterations")
    ap.add_argument("--batch-per-gpu", type=int, default=4, help="train mode: samples per rank (global batch 32 = 4 x 8)")
    ap.add_argument("--train-iters", type=int, default=16)
    ap.add_argument("--train-quick", action="store_true",
                    help="train mode: the timed steps and the dominant kernel's roofline only (no eager comparison, no CPU baseline): "
                         "what the default inference run starts as a child process for its `train_mode` object")
    ap.add_argument("--no-train-mode", action="store_true", help="infer mode: skip the `train_mode` object (cfg 4, 1 rank, child process)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun -> N child ranks, one per GPU; the parent touches no GPU
# ------------------------------------------------------------------------------------------------------------------


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a) -> int:
    """Spawn `a.gpus` fresh processes running this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what
    torch.distributed.run would export).  Rank 0 inherits stdout (the ONE JSON line); the other ranks' stdout goes to
    stderr.  Exit code = the worst child's."""
    port = _free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def pin_rank(local: int, local_world: int):
    """Give this rank its own share of the host BEFORE torch (OpenMP pools, the autograd engine's threads) or HIP starts a
    thread: the allowed CPUs (cgroup / taskset aware) are split into `local_world` contiguous core sets, the process is bound to
    set `local` (os.sched_setaffinity: every later thread inherits it) and the OpenMP / MKL pools are sized to it.  The
    training step is host-bound: eight ranks that each start a pool as wide as the machine and migrate across
    sockets contend for the same cores.  ANYSTEREO_PIN=0 leaves placement to the OS.  Returns the CPU list (for the bench line).
    No process is re-executed: this runs in the child before its first GPU call."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:  # not Linux
        return None
    if os.environ.get("ANYSTEREO_PIN", "1") == "0" or local_world <= 1:
        return cpus
    n = len(cpus) // local_world
    if n < 1:
        return cpus  # fewer cores than ranks: nothing sensible to pin to
    mine = cpus[local * n:(local + 1) * n]
    os.sched_setaffinity(0, mine)
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[k] = str(len(mine))
    return mine


def _dist_init(backend_gpu: bool, local: int):
    import torch.distributed as td
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend_gpu and not REHEARSAL:
        td.init_process_group("nccl", device_id=torch.device("cuda", local))   # "nccl" IS RCCL on ROCm
    else:
        td.init_process_group("gloo")
    return td


def dry_main(a, rank, world):
    """No GPU visible: rehearse the N-rank protocol only (rendezvous, barrier, timed steps, MAX over ranks, gather of the
    per-rank values) over gloo.  Nothing of the hot path runs and no throughput is claimed (`value` is null)."""
    td = _dist_init(False, 0) if (world > 1 or "RANK" in os.environ) else None
    x = torch.ones(64, 64)
    for _ in range(a.warmup):
        x = x @ x * 1e-3
    if td:
        td.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        x = x @ x * 1e-3
        time.sleep(0.002)
    if td:
        td.barrier()
    dt = time.perf_counter() - t0
    per_rank = [dt]
    cpu_sets = [RANK_CPUS]
    if td:
        cpu_sets = [None] * world
        td.all_gather_object(cpu_sets, RANK_CPUS)
        t = torch.tensor([dt], dtype=torch.float64)
        gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        td.all_gather(gathered, t)
        per_rank = [float(g.item()) for g in gathered]
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
    # the training leg of an N-rank run, same code path as on the GPU (train_leg), on the stand-in module over gloo
    train_mode = None
    if td and not a.no_train_mode:
        guard = _LegWatchdog(rank, float(os.environ.get("ANYSTEREO_TRAIN_LEG_TIMEOUT", "240")))
        try:
            train_mode = train_leg(a, rank, world, 0, td, None, dry=True)
        except Exception as ex:
            train_mode = {"error": repr(ex)[:300]} if rank == 0 else None
        guard.done()
    if rank == 0:
        sys.stderr.write("bench.py: no GPU visible - dry run of the launch protocol over gloo; the hot path has no CPU fallback\n")
        emit(({"metric": "stereo pairs/sec (coreContinuous_IGEV inference, 32-iter GRU, 960x540)", "value": None,
                          "train_mode": train_mode,
                          "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "launch protocol only (no GPU visible)", "parallelism": f"replicas x{world}",
                                     "backend": "gloo"},
                          "per_rank_step_s": [round(v / a.steps, 5) for v in per_rank], "per_rank_cpus": cpu_sets,
                          "torch_threads": torch.get_num_threads(), "roofline": None, "cpu_baseline": None}))
    if td:
        td.barrier()
        td.destroy_process_group()


def dry_train_main(a, rank, world):
    """No GPU visible, --mode train: rehearse the N-rank TRAINING protocol over gloo — per-rank batch shard, the Trainer's flat
    gradient exchange (rank 0's weights broadcast before the first step, one all-reduce of the concatenated gradients between
    backward and the update, collective overflow decisions), barrier-bracketed timed steps, MAX over ranks, the same steps with
    the exchange off (`exposed_allreduce_ms`) — on a small stand-in module with the reference's forward signature.  Nothing of the
    hot path runs and no throughput is claimed (`value` is null): what is checked is that the line the 8-GPU driver run prints
    has its fields and that the ranks end a step with identical parameters."""
    td = _dist_init(False, 0) if world > 1 else None
    from anystereo.harness.train import Trainer

    class DryStereo(torch.nn.Module):  # (image1, image2, iters=, hr_coord=, scale=) -> list of [B,1,Q] predictions
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(6, 8, 3, padding=1)
            self.head = torch.nn.Linear(10, 1)

        def freeze_bn(self):
            pass

        def forward(self, image1, image2, iters=2, hr_coord=None, scale=None, **_):
            f = torch.relu(self.conv(torch.cat([image1, image2], 1) / 255.0)).mean((2, 3))  # [B,8]
            x = torch.cat([f.unsqueeze(1).expand(-1, hr_coord.shape[1], -1), hr_coord], -1)  # [B,Q,10]
            d = self.head(x).transpose(1, 2)
            return [d * (i + 1) for i in range(iters)]

    torch.manual_seed(100 + rank)  # every rank builds DIFFERENT weights: the Trainer must make them rank 0's
    model = DryStereo()
    tr = Trainer(model, train_iters=2, max_disp=192, ddp_impl="flat", num_steps=100)
    g = torch.Generator().manual_seed(7 + rank)
    bsz, q = a.batch_per_gpu, 64
    batch = (torch.rand(bsz, 3, 16, 32, generator=g) * 255, torch.rand(bsz, 3, 16, 32, generator=g) * 255,
             torch.rand(bsz, q, 2, generator=g) * 2 - 1, torch.rand(bsz, 1, q, generator=g) * 60 + 0.5, torch.ones(bsz, 1))
    for _ in range(max(1, a.warmup)):
        tr.step(batch)

    def timed(n, sync_grads=True):
        if td:
            td.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            loss, _ = tr.step(batch, sync_grads=sync_grads)
        if td:
            td.barrier()
        return time.perf_counter() - t0, float(loss)

    dt, loss = timed(a.steps)
    flat = torch.cat([p.detach().reshape(-1).double() for p in model.parameters()])
    check = torch.stack([flat.sum(), flat.abs().sum()])
    per_rank, sums, cpu_sets, overlap = [dt], [check.tolist()], [RANK_CPUS], None
    if td:
        t = torch.tensor([dt], dtype=torch.float64)
        gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        td.all_gather(gathered, t)
        per_rank = [float(v.item()) for v in gathered]
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
        gs = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        td.all_gather(gs, check)
        sums = [v.tolist() for v in gs]
        cpu_sets = [None] * world
        td.all_gather_object(cpu_sets, RANK_CPUS)
        dt_ns, _ = timed(max(2, a.steps), sync_grads=False)  # last: the ranks' parameters diverge from here on
        t2 = torch.tensor([dt_ns], dtype=torch.float64)
        td.all_reduce(t2, op=td.ReduceOp.MAX)
        ms_sync, ms_nosync = dt / a.steps * 1e3, float(t2.item()) / max(2, a.steps) * 1e3
        overlap = {"ms_per_step_with_allreduce": round(ms_sync, 3), "ms_per_step_no_sync": round(ms_nosync, 3),
                   "exposed_allreduce_ms": round(ms_sync - ms_nosync, 3), "gradient_bytes": 4 * flat.numel(), "ddp": tr.ddp_mode}
    if rank == 0:
        sys.stderr.write("bench.py: no GPU visible - dry run of the training launch protocol over gloo; the hot path has no CPU fallback\n")
        emit(({"metric": "train_samples_per_s", "value": None, "unit": "samples/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": TRAIN_DTYPE, "data": "synthetic", "dry_run": True,
                          "config": {"workload": "training launch protocol only (no GPU visible; stand-in module)",
                                     "global_batch": world * bsz, "parallelism": f"ddp x{world} (flat all-reduce over gloo)"},
                          "per_rank_samples_per_s": [round(bsz * a.steps / v, 3) for v in per_rank],
                          "samples_per_s_protocol": round(world * bsz * a.steps / dt, 3),
                          "allreduce_overlap": overlap, "per_rank_cpus": cpu_sets, "torch_threads": torch.get_num_threads(),
                          "parameters_equal_across_ranks": all(v == sums[0] for v in sums), "loss_last": loss,
                          "trainer": {"graph": bool(tr.use_graph), "gradient_exchange": tr.ddp_mode}, "roofline": None, "cpu_baseline": None}))
    if td:
        td.barrier()
        td.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
# algorithmic work
# ------------------------------------------------------------------------------------------------------------------


class GpuTelemetry:
    """Socket power and shader clock of the rank's GPU, sampled from sysfs (hwmon of the amdgpu device: `power1_average` /
    `power1_input` in microwatts, `freq1_input` in Hz) by a host thread WHILE the timed loop runs — so that "power-limited
    clock" in DESIGN.md is a measurement of the run that produced `value`.  Reads are plain file reads (no tool is spawned,
    nothing touches the GPU); a box that does not expose the files yields nulls."""

    def __init__(self, dev_index: int, period_s: float = 0.01):
        import glob
        self.period, self.samples, self._stop, self._thread = period_s, [], False, None
        self.power_file = self.clock_file = self.cap_file = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(dev_index)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            want = None
        cards = []
        for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
            if "-" in os.path.basename(c):
                continue
            try:
                addr = os.path.basename(os.path.realpath(os.path.join(c, "device")))
            except OSError:
                continue
            cards.append((c, addr))
        pick = [c for c, addr in cards if want and addr.lower().startswith(want)] or ([cards[0][0]] if len(cards) == 1 else [])
        if not pick:
            return
        for hw in glob.glob(os.path.join(pick[0], "device", "hwmon", "hwmon*")):
            for f in ("power1_average", "power1_input"):
                if self.power_file is None and os.path.exists(os.path.join(hw, f)):
                    self.power_file = os.path.join(hw, f)
            if os.path.exists(os.path.join(hw, "freq1_input")):
                self.clock_file = os.path.join(hw, "freq1_input")
            if os.path.exists(os.path.join(hw, "power1_cap")):
                self.cap_file = os.path.join(hw, "power1_cap")

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _loop(self):
        while not self._stop:
            pw = self._read(self.power_file) if self.power_file else None
            ck = self._read(self.clock_file) if self.clock_file else None
            self.samples.append((pw, ck))
            time.sleep(self.period)

    def start(self):
        if self.power_file or self.clock_file:
            import threading
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()
        return self

    def stop(self) -> dict:
        self._stop = True
        if self._thread is not None:
            self._thread.join(1.0)
        pw = [p * 1e-6 for p, _ in self.samples if p is not None]
        ck = [c * 1e-6 for _, c in self.samples if c is not None]
        cap = self._read(self.cap_file) if self.cap_file else None
        return {"gpu_power_w": round(sum(pw) / len(pw), 1) if pw else None, "gpu_power_w_max": round(max(pw), 1) if pw else None,
                "gpu_power_cap_w": round(cap * 1e-6, 1) if cap else None,
                "gpu_sclk_mhz": round(sum(ck) / len(ck), 1) if ck else None, "gpu_sclk_mhz_min": round(min(ck), 1) if ck else None,
                "telemetry_samples": len(self.samples),
                "telemetry_source": "sysfs hwmon (power1_average, freq1_input), host thread sampling every %.0f ms during the timed loop" % (self.period * 1e3)
                                    if (pw or ck) else "unavailable on this box (no readable amdgpu hwmon files)"}


def algorithmic(B, h, w, Q, C=96, L=2, G=8, D=48, r=4):
    """Algorithmic bytes / flops per launch."""
    P = B * h * w
    build_b = 4 * (2 * B * C * h * w + sum(P * (w >> i) for i in range(L)))
    geo_b = 4 * (B * G * D * h * w + sum(P * G * (D >> i) for i in range(L)))
    lookup_b = 4 * P * (L * (G + 1) * (2 * r + 2) + 1 + L * (G + 1) * (2 * r + 1))
    # LIIF: compulsory bytes 4*[B(184+160)hw + 2Q + Bhw + Q]; flops Q*84096 (the product applies the first layer at low
    # resolution, so it executes fewer — the ALGORITHMIC figure is the reference's)
    liif_b = 4 * (B * (184 + 160) * h * w + 2 * Q * B + B * h * w + Q * B)
    return {
        "corr_build": {"bound": "hbm", "bytes": build_b, "flops": 2 * C * P * w},
        "geo_pyramid": {"bound": "hbm", "bytes": geo_b},
        "lookup": {"bound": "hbm", "bytes": lookup_b},
        "lookup_convc1": {"bound": "hbm", "bytes": 4 * P * (L * (G + 1) * (2 * r + 2) + 1) + 4 * P * 64},
        "gwc_volume": {"bound": "hbm", "bytes": 4 * (2 * B * C * h * w + B * G * D * h * w)},
        # 3x3 convs of gru04: zr = (3*128 -> 256), q = (3*128 -> 128)
        "gru04_zr_conv": {"bound": "mfma", "flops": 2 * P * 384 * 9 * 256},
        "gru04_q_conv": {"bound": "mfma", "flops": 2 * P * 384 * 9 * 128},
        "gru08_zr_conv": {"bound": "mfma", "flops": 2 * (P // 4) * 384 * 9 * 256},
        "gru16_zr_conv": {"bound": "mfma", "flops": 2 * (P // 16) * 256 * 9 * 256},
        "disp_head_conv1": {"bound": "mfma", "flops": 2 * P * 128 * 9 * 256},
        # layers 2..4 at query resolution (the first Linear layer runs at low resolution: liif_mlp_lowres)
        "liif_mlp": {"bound": "mfma", "flops": 2 * Q * B * (128 * 64 + 64 * 64 + 64 * 9)},
        # a15 sits on the MFMA roofline in SURVEY.md §8(d): ALGORITHMIC flops = Q * 84 096 (the reference's 228->128->64->64->9 MLP
        # per query); the kernel EXECUTES layers 2-4 only (the first layer commutes with the nearest gather and runs once per
        # low-resolution pixel: liif_mlp_lowres) — both figures are reported, `frac` uses the algorithmic one
        "liif_tail": {"bound": "mfma", "flops": Q * B * 84096, "flops_executed": 2 * Q * B * (128 * 64 + 64 * 64 + 64 * 9), "bytes": liif_b},
    }


def src_hash() -> str:
    """Content hash of the kernel sources: ties a committed profile to the code it was taken on (the GPU box has no .git)."""
    hsh = hashlib.sha256()
    d = os.path.join(ROOT, "any-stereo_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            hsh.update(open(os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


def load_pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
    (profiles/rNN_pmc_traffic.json, produced by tools/profile_round.sh + tools/summarize_pmc.py with the guide's gfx950
    FETCH_SIZE correction).  NOT measured in this run: the line says so under `traffic_source`."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return {}, None
    d = json.load(open(files[-1]))
    meta = d.get("_meta", {})
    now = src_hash()
    source = {"file": os.path.relpath(files[-1], ROOT), "measured_in_this_run": False,
              "profile_src_hash": meta.get("src_hash"), "current_src_hash": now,
              "kernel_sources_unchanged_since_profile": meta.get("src_hash") == now,
              "collected": meta.get("collected"), "method": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE "
              "doubled for kernels whose reads are 16 B/lane (MI355X_MICROARCH.md §HBM)"}
    return {k: (None if v.get("hbm_bytes") is None else int(v["hbm_bytes"])) for k, v in d.items() if not k.startswith("_")}, source


# ------------------------------------------------------------------------------------------------------------------
# cfg 4: training
# ------------------------------------------------------------------------------------------------------------------


# arithmetic of the training step, spelled out: storage, accumulation, loss, norms, optimizer in fp32; the
# matrix-core products of forward and data gradient are the 3 x f16 split (22 significand bits), those of the batched weight
# gradient a 2-term bf16 split (hi.hi + hi.lo + lo.hi, ~16 significand bits, fp32's exponent range)
TRAIN_DTYPE = "fp32 storage and accumulation; forward + dgrad 3xf16 split MFMA (22 bits); wgrad 2-term bf16 split MFMA (~16 bits)"


def train_mode_child(a, steps=5, warmup=4, timeout=420):
    """cfg 4 inside the DEFAULT run's line: `bench.py --mode train --train-quick` at 1 rank as a CHILD process
    (its own HIP context; a crash or a hang there cannot take the headline line with it), 3 eager warm-up steps + the capture +
    `steps` graphed steps.  Returns the child's line reduced to what the driver should see, or {"error": ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "train", "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup),
           "--train-quick", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired:
        return {"error": f"child timed out after {timeout} s"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"child exit {r.returncode}: {r.stderr[-300:]}"}
    d = json.loads(lines[-1])
    import math
    return {"workload": d["config"]["workload"], "global_batch": d["config"]["global_batch"], "n_gpus": d["n_gpus"],
            "metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
            "warmup": d["warmup"], "dtype": d["dtype"], "loss_first_last": d["loss_first_last"],
            "loss_finite": all(math.isfinite(v) for v in d["loss_first_last"]), "trainer": d["trainer"], "loss_scale": d["loss_scale"],
            "split_overflow_events": d["split_overflow"]["events"], "roofline": d["roofline"],
            "grad_bytes": d.get("grad_bytes"), "exchange_ms": d.get("exchange_ms"), "exchange_how": d.get("exchange_how"),
            "reduced_precision": d.get("reduced_precision"),
            "how": "child process `bench.py --mode train --train-quick`, inputs resident, graphed gradient half + eager clip/AdamW",
            "child_wall_s": round(time.perf_counter() - t0, 1)}


class _DryStereo:
    """Stand-in module of the no-GPU rehearsals: the reference's forward signature on a few parameters."""

    @staticmethod
    def build():
        class DryStereo(torch.nn.Module):  # (image1, image2, iters=, hr_coord=, scale=) -> list of [B,1,Q] predictions
            def __init__(self):
                super().__init__()
                self.conv = torch.nn.Conv2d(6, 8, 3, padding=1)
                self.head = torch.nn.Linear(10, 1)

            def freeze_bn(self):
                pass

            def forward(self, image1, image2, iters=2, hr_coord=None, scale=None, **_):
                f = torch.relu(self.conv(torch.cat([image1, image2], 1) / 255.0)).mean((2, 3))  # [B,8]
                x = torch.cat([f.unsqueeze(1).expand(-1, hr_coord.shape[1], -1), hr_coord], -1)  # [B,Q,10]
                d = self.head(x).transpose(1, 2)
                return [d * (i + 1) for i in range(iters)]
        return DryStereo()


class _LegWatchdog:
    """Bounds the in-process training leg of an N-rank run.  `stash` (rank 0) is the line as far as it is known; if `done()` has
    not been called `seconds` after construction, rank 0 prints the stashed line with train_mode = {"error": "timed out"} and
    every rank ends its process (os._exit: a peer may be stuck inside a collective that will never complete)."""

    def __init__(self, rank, seconds):
        import threading
        self.rank, self.stash, self._done = rank, None, False
        self._t = threading.Timer(seconds, self._fire)
        self._t.daemon = True
        self._t.start()
        self.seconds = seconds

    def rearm(self, stash):
        self.stash = stash

    def done(self):
        self._done = True
        self._t.cancel()

    def _fire(self):
        if self._done:
            return
        sys.stderr.write(f"bench.py: rank {self.rank}: the training leg did not finish within {self.seconds:.0f} s - leaving without it\n")
        if self.rank == 0 and self.stash is not None:
            line = dict(self.stash)
            if getattr(self, "key", "train_mode") == "train_mode":
                line["train_mode"] = {"error": f"training leg timed out after {self.seconds:.0f} s"}
            emit(line)
        os._exit(0 if (self.rank != 0 or self.stash is not None) else 3)


TRAIN_LEG_KEYS = ("n_gpus", "global_batch", "value", "unit", "per_rank_samples_per_s", "ms_per_step", "exchange_ms", "ranks_seen",
                  "one_rank_ms_per_step", "scaling", "steps", "warmup", "grad_bytes", "trainer", "loss_first_last", "loss_finite")


def train_leg(a, rank, world, local, td, dev, dry=False):
    """The N-rank TRAINING leg of `bench.py --gpus N` (north_star: 1 -> 8-GPU training throughput scaling; the reference shards its
    batch with nn.DataParallel, train_continuous_IGEV.py:184,214-239).  Runs on the SAME ranks and process group as the
    inference leg, after it (no new process, nothing re-executed): cfg 4 — 4 samples per rank at 160x320, 16 GRU iterations,
    51 200 queries per sample — through `Trainer`: the gradient half of the step replayed as one captured hipGraph, ONE RCCL
    all-reduce of the flat fp32 gradient vector, clip + AdamW.  Timed like the headline: barrier + synchronize on both sides of
    K steps, MAX over ranks.  Then, reported only: the all-reduce alone at N ranks (HIP events), and K steps of every rank with
    the exchange off (`one_rank_ms_per_step`; the ranks' weights diverge from there on, so it comes last) ->
    scaling = N * mean(one-rank step) / N-rank step, an in-job estimate (the driver computes its own from its N=1 run).
    dry=True (no GPU): the same protocol and the same keys over gloo on a stand-in module; no throughput is claimed.
    Returns the `train_mode` object on rank 0, None elsewhere.  A failure on one rank is agreed on by all ranks (MIN all-reduce
    of an ok flag between the phases) so nobody is left waiting in a collective."""
    import math
    from anystereo.harness.train import Trainer, synthetic_train_batch
    t_leg = time.perf_counter()
    steps = max(2, min(a.steps, 20))
    warm = max(4, min(a.warmup, 6))  # 3 eager steps (solver search, packs, optimizer state) + the capture
    bsz = a.batch_per_gpu
    cdev = torch.device("cpu") if dry else dev

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def agree(ok: bool) -> bool:
        if td is None:
            return ok
        f = torch.tensor([1 if ok else 0], device=cdev, dtype=torch.int32)
        td.all_reduce(f, op=td.ReduceOp.MIN)
        return bool(f.item())

    err = None
    tr = batch = None
    try:
        if dry:
            torch.manual_seed(100 + rank)  # every rank builds DIFFERENT weights: the Trainer must make them rank 0's
            model = _DryStereo.build()
            tr = Trainer(model, train_iters=2, max_disp=192, ddp_impl="flat", num_steps=100, force_ddp=td is not None)
            g = torch.Generator().manual_seed(7 + rank)
            q = 64
            batch = (torch.rand(bsz, 3, 16, 32, generator=g) * 255, torch.rand(bsz, 3, 16, 32, generator=g) * 255,
                     torch.rand(bsz, q, 2, generator=g) * 2 - 1, torch.rand(bsz, 1, q, generator=g) * 60 + 0.5, torch.ones(bsz, 1))
            what = "training launch protocol only (no GPU visible; stand-in module)"
        else:
            from anystereo.harness.synthetic import fill_module_deterministic
            from anystereo.models import __models__, default_args
            targs = default_args("continuous_IGEVStereo")
            model = __models__["continuous_IGEVStereo"](targs)
            fill_module_deterministic(model, base_seed=1)
            model = model.to(dev)
            tr = Trainer(model, train_iters=a.train_iters, max_disp=targs.max_disp, force_ddp=td is not None)
            batch = synthetic_train_batch(bsz, 160, 320, seed=rank, device=dev)
            what = (f"cfg4 continuous_IGEVStereo training 160x320, {a.train_iters} GRU iters, LIIF every iter, "
                    f"Q={batch[2].shape[1]} queries/sample, AdamW+OneCycleLR, clip 1.0")
    except Exception as ex:
        err = "setup: " + repr(ex)[:300]
    if not agree(err is None):
        return {"error": err or "another rank failed during setup"} if rank == 0 else None

    losses = []

    def timed(n, sync_grads=True):
        sync()
        if td is not None:
            td.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            loss, _ = tr.step(batch, sync_grads=sync_grads)
        sync()
        own = time.perf_counter() - t0
        if td is not None:
            td.barrier()
        sync()
        return time.perf_counter() - t0, own, float(loss)

    def gather(v):
        if td is None:
            return [v]
        t = torch.tensor([v], device=cdev, dtype=torch.float64)
        outs = [torch.zeros_like(t) for _ in range(world)]
        td.all_gather(outs, t)
        return [float(o.item()) for o in outs]

    try:
        for _ in range(warm):
            losses.append(float(tr.step(batch)[0]))
    except Exception as ex:
        err = "warm-up: " + repr(ex)[:300]
    if not agree(err is None):
        return {"error": err or "another rank failed during the warm-up steps"} if rank == 0 else None
    try:
        dt, own, loss = timed(steps)
        losses.append(loss)
    except Exception as ex:
        err = "timed steps: " + repr(ex)[:300]
    if not agree(err is None):
        return {"error": err or "another rank failed during the timed steps"} if rank == 0 else None
    dt = max(gather(dt))
    per_rank_own = gather(own)
    nparam = sum(p.numel() for p in tr.model.parameters() if p.requires_grad)
    # the collective alone, at N ranks: what the flat exchange issues once per step
    exchange_ms = None
    exchange_how = "no process group (one rank without a launcher): nothing is exchanged"
    if td is not None:
        try:
            flat = torch.zeros(nparam, device=cdev, dtype=torch.float32)
            for _ in range(3):
                td.all_reduce(flat)
            sync()
            reps = 20
            if dry:
                t0 = time.perf_counter()
                for _ in range(reps):
                    td.all_reduce(flat)
                ms = (time.perf_counter() - t0) / reps * 1e3
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    td.all_reduce(flat)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
            exchange_ms = round(max(gather(ms)), 4)
            exchange_how = (f"{reps} all-reduces of the {4 * nparam} B flat fp32 gradient vector over {td.get_backend()} at {world} rank(s), "
                            + ("host clock" if dry else "HIP events on the collective's stream") + ", MAX over ranks")
            del flat
        except Exception as ex:
            exchange_how = "unavailable: " + repr(ex)[:200]
        if not agree(True):
            return {"error": "a rank failed in the exchange probe"} if rank == 0 else None
    # identity of the ranks' devices (a SCALE record must show N distinct GPUs)
    seen = {"rank": rank, "local_rank": local, "host": socket.gethostname()}
    if not dry:
        pr = torch.cuda.get_device_properties(dev)
        seen.update(device_index=dev.index, name=pr.name,
                    pci="%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
                    uuid=str(getattr(pr, "uuid", "")) or None)
    ranks_seen = [seen]
    if td is not None:
        ranks_seen = [None] * world
        td.all_gather_object(ranks_seen, seen)
    # every rank alone: the same steps with the exchange off (LAST: the ranks' parameters diverge from here on)
    one_rank = None
    try:
        _, own_ns, _ = timed(steps, sync_grads=False)
    except Exception as ex:
        err = "no-exchange steps: " + repr(ex)[:300]
    if agree(err is None):
        one_rank = [round(v / steps * 1e3, 3) for v in gather(own_ns)]
    if rank != 0:
        return None
    ms_step = dt / steps * 1e3
    res = {"workload": what, "n_gpus": world, "global_batch": world * bsz, "metric": "train_samples_per_s",
           "value": None if dry else round(world * bsz * steps / dt, 3), "unit": "samples/s",
           "samples_per_s_protocol": round(world * bsz * steps / dt, 3) if dry else None,
           "per_rank_samples_per_s": [round(bsz * steps / v, 3) for v in per_rank_own],
           "ms_per_step": round(ms_step, 3), "steps": steps, "warmup": warm,
           "grad_bytes": 4 * nparam, "exchange_ms": exchange_ms, "exchange_how": exchange_how,
           "ranks_seen": ranks_seen,
           "distinct_devices": len({(r_.get("host"), r_.get("pci") or r_.get("device_index") or r_.get("rank")) for r_ in ranks_seen}),
           "one_rank_ms_per_step": one_rank,
           "scaling": None if not one_rank else round(world * (sum(one_rank) / len(one_rank)) / ms_step, 3),
           "scaling_how": "N x mean(one_rank_ms_per_step: every rank's own steps with the exchange off, all ranks running at once) / "
                          "ms_per_step at N ranks; the driver's SCALE record divides by its own N=1 run instead",
           "dtype": TRAIN_DTYPE, "dry_run": bool(dry),
           "trainer": {"graph": bool(tr.use_graph), "graph_scope": tr.graph_scope if tr.use_graph else None,
                       "gradient_exchange": tr.ddp_mode},
           "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)],
           "loss_finite": all(math.isfinite(v) for v in losses),
           "loss_scale": tr.loss_scale, "split_overflow_events": list(tr.overflow_events),
           "how": "same ranks and process group as the inference leg, after it; inputs resident; graphed gradient half + one flat "
                  "all-reduce + eager clip/AdamW; barrier + synchronize around the K steps, MAX over ranks",
           "leg_wall_s": round(time.perf_counter() - t_leg, 1)}
    if err:
        res["error_no_exchange_steps"] = err
    return res


def _exchange_probe(a, rank, dist, dev, nbytes):
    """What the flat gradient exchange costs through RCCL on this box: 20 all-reduces of the flat fp32 vector, HIP events on the
    collective's stream.  With a single rank that is the collective's launch + the in-place pass over the vector (no link traffic),
    the floor under the N-rank figure (ring: + 2*(N-1)/N*bytes per link)."""
    exchange = {"grad_bytes": nbytes, "exchange_ms": None, "how": None}
    if not ((dist or rank == 0) and not a.no_extras):  # every rank of a group takes part in its collective
        return exchange
    try:
        import torch.distributed as tdd
        own_pg = not tdd.is_initialized()
        if own_pg:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            tdd.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        flat = torch.zeros(nbytes // 4, device=dev, dtype=torch.float32)
        for _ in range(3):
            tdd.all_reduce(flat)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            tdd.all_reduce(flat)
        e1.record()
        torch.cuda.synchronize()
        exchange["exchange_ms"] = round(e0.elapsed_time(e1) / 20, 4)
        exchange["how"] = (f"20 all-reduces of the {nbytes} B flat fp32 gradient vector through RCCL, {tdd.get_world_size()} rank(s), "
                           "HIP events on the collective's stream; what the Trainer's flat exchange issues once per step")
        del flat
        if own_pg:
            tdd.destroy_process_group()
    except Exception as ex:
        exchange["how"] = "unavailable: " + repr(ex)[:200]
    return exchange


def train_main(a, rank, world, local):
    """SURVEY.md §8d cfg 4: IGEV training, 4 samples per GPU at 160x320 network input, 51 200 HR queries per sample, 16 GRU
    iterations with the LIIF upsampler every iteration, AdamW + OneCycleLR; one process per GPU, DDP over RCCL.  A step =
    zero_grad + forward + loss + backward (+ gradient all-reduce) + clip + optimizer + scheduler step, as the Trainer issues it by
    default: the gradient half replayed as one captured hipGraph, then the (flat) gradient exchange, clip and AdamW eager."""
    dist = world > 1 or "RANK" in os.environ  # under a launcher even one rank goes through RCCL + DDP
    td = _dist_init(True, local) if dist else None
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
    h, w = (160, 320) if a.height is None else (a.height, a.width)
    batch = synthetic_train_batch(a.batch_per_gpu, h, w, seed=rank, device=dev)
    losses = []
    for _ in range(max(1, a.warmup)):
        losses.append(float(tr.step(batch)[0]))

    def timed(n, sync_grads=True):
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            torch.cuda._sleep(2000)  # ~1 us `spin_kernel`: step marker for tools/profile_train.sh
            loss, _ = tr.step(batch, sync_grads=sync_grads)
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, loss

    dt, loss = timed(a.steps)
    losses.append(float(loss))
    per_rank = [dt]
    overlap = None
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        td.all_gather(gathered, t)
        per_rank = [float(g.item()) for g in gathered]
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
        # what the gradient all-reduce costs on the critical path: the same steps with DDP's reducer off (no_sync: every
        # rank steps on its local gradient) — outside the timed region, reported only
        n2 = max(2, min(a.steps, 5))
        dt_ns, _ = timed(n2, sync_grads=False)
        t2 = torch.tensor([dt_ns], device=dev, dtype=torch.float64)
        td.all_reduce(t2, op=td.ReduceOp.MAX)
        ms_sync, ms_nosync = dt / a.steps * 1e3, float(t2.item()) / n2 * 1e3
        nbytes = 4 * sum(p.numel() for p in model.parameters() if p.requires_grad)
        overlap = {"ms_per_step_with_allreduce": round(ms_sync, 2), "ms_per_step_no_sync": round(ms_nosync, 2),
                   "exposed_allreduce_ms": round(ms_sync - ms_nosync, 2), "gradient_bytes": nbytes,
                   "ddp": tr.ddp_mode}
    roof = cpu = eager = reduced = None
    if rank == 0 and world == 1 and not a.no_extras:
        try:
            reduced = train_reduced_precision(a, args, batch, dev, losses)
        except Exception as ex:  # never lose the headline line to the side measurement
            reduced = {"error": repr(ex)[:300]}
    if rank == 0 and world == 1 and a.train_quick:
        roof = train_roofline(a, batch, dev)
    elif rank == 0 and world == 1 and not a.no_extras:
        try:
            eager = train_eager_step(a, args, batch, dev)
        except Exception as ex:  # never lose the headline line to the side measurement
            eager = {"error": repr(ex)[:300]}
        roof = train_roofline(a, batch, dev)
        if not a.no_cpu_baseline:
            cpu = train_cpu_baseline(a, args, model, batch)
    line = None
    if rank == 0:
        nparam = sum(p.numel() for p in model.parameters())
        line = ({
            "metric": "train_samples_per_s", "value": round(world * a.batch_per_gpu * a.steps / dt, 3), "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": TRAIN_DTYPE, "data": "synthetic",
            "config": {"workload": f"cfg4 continuous_IGEVStereo training {h}x{w}, {a.train_iters} GRU iters, LIIF every iter, "
                                   f"Q={batch[2].shape[1]} queries/sample, AdamW+OneCycleLR, clip 1.0",
                       "global_batch": world * a.batch_per_gpu, "parallelism": f"ddp x{world} (RCCL all-reduce of {nparam} fp32 grads)"},
            "per_rank_samples_per_s": [round(a.batch_per_gpu * a.steps / v, 3) for v in per_rank],
            "host": {"cpus_per_rank": len(RANK_CPUS or []), "pinned": world > 1 and os.environ.get("ANYSTEREO_PIN", "1") != "0",
                     "torch_threads": torch.get_num_threads()},
            "allreduce_overlap": overlap,
            "grad_bytes": 4 * sum(p.numel() for p in model.parameters() if p.requires_grad), "exchange_ms": None, "exchange_how": None,
            "trainer": {"graph": bool(tr.use_graph), "graph_scope": tr.graph_scope if tr.use_graph else None,
                        "gradient_exchange": tr.ddp_mode},
            "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)],
            "backward": "HIP kernels for the volume/lookup/gwc/LIIF/convex-upsample/pool/interp/gate transposes, forward + dgrad + wgrad of "
                        "every stride-1 1x1 / 3x3 convolution (update block, MLP Linear layers, backbone layers with >= 16 input "
                        "channels; one batched wgrad launch per layer and step); strided / 3-channel / 3-D backbone convs, BatchNorm "
                        "and elementwise glue on MIOpen / ATen",
            "loss_scale": tr.loss_scale,
            "split_overflow": {"policy": tr.overflow_policy, "check_every": tr.overflow_check_every,
                               "saturated_waves_at_end": tr._poll_overflow() if tr._on_gpu() else None,
                               "events": tr.overflow_events, "skipped_steps": tr.skipped_steps},
            "eager_step": eager, "reduced_precision": reduced,
            "library": _lib.library_info(), "roofline": roof, "cpu_baseline": cpu})
    # The exchange probe comes LAST and is bounded: with one rank and no launcher it creates a 1-rank RCCL group of its own; if
    # that initialisation (or a collective) hangs on some box, rank 0 still prints the training line it already has
    nbytes = 4 * sum(p.numel() for p in model.parameters() if p.requires_grad)
    guard = _LegWatchdog(rank, float(os.environ.get("ANYSTEREO_EXCHANGE_PROBE_TIMEOUT", "90")))
    guard.rearm(None if line is None else dict(line, exchange_how="unavailable: the probe did not return in time"))
    guard.key = "exchange_probe"
    exchange = _exchange_probe(a, rank, dist, dev, nbytes)
    guard.done()
    if rank == 0:
        line.update(exchange_ms=exchange["exchange_ms"], exchange_how=exchange["how"])
        emit(line)
    if dist:
        td.barrier()
        td.destroy_process_group()


def train_reduced_precision(a, args, batch, dev, split_losses):
    """cfg 4 in the reference's OWN training arithmetic (it always trains under autocast + GradScaler,
    train_continuous_IGEV.py:206,288: fp16 operands, fp32 accumulation): the same graphed step with the forward and data-gradient
    convolutions in the one-MFMA mode (fp16 operands = the hi parts only, fp32 accumulate and storage; ops.fast_fp16), weight
    gradients unchanged (bf16 hi + lo).  A second, labelled figure with its own tolerance (tests/test_hip_parity.py::
    test_training_step_reduced_precision_vs_reference) — never the headline, whose split mode is WIDER than the reference's."""
    from anystereo import ops
    from anystereo.harness.synthetic import fill_module_deterministic
    from anystereo.harness.train import Trainer
    from anystereo.models import __models__
    m = __models__["continuous_IGEVStereo"](args)
    fill_module_deterministic(m, base_seed=1)
    losses = []
    with ops.fast_fp16(True):
        tr = Trainer(m.to(dev), train_iters=a.train_iters, max_disp=args.max_disp)
        for _ in range(max(4, a.warmup)):  # 3 eager steps + the capture, all inside the mode: the graph's launches carry it
            losses.append(float(tr.step(batch)[0]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss, _ = tr.step(batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        losses.append(float(loss))
    n = min(len(split_losses), len(losses)) - 1  # the warm-up steps both runs took from the same weights on the same batch
    rel = [abs(x - y) / max(abs(y), 1e-12) for x, y in zip(losses[:n], split_losses[:n])]
    return {"value": round(a.batch_per_gpu * a.steps / dt, 3), "unit": "samples/s", "ms_per_step": round(dt / a.steps * 1e3, 2), "steps": a.steps,
            "dtype": "fp16 operands (1 MFMA per product) in forward + dgrad convolutions, fp32 accumulate and storage; wgrad 2-term bf16 split",
            "graphed": bool(tr.use_graph), "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)],
            "max_rel_loss_diff_vs_split_over_warmup_steps": max(rel) if rel else None,
            "note": "the reference's autocast arithmetic (train_continuous_IGEV.py:206,288); labelled second figure, never `value`"}


def train_eager_step(a, args, batch, dev, n_cmp=6):
    """The EAGER step (Trainer(graph=False)) next to the headline, which is the Trainer's default — the gradient half of the step
    replayed as one captured hipGraph (harness/train.py): a fresh default trainer's loss trajectory (3 eager warm-up steps + the
    capture + n_cmp replays) is checked against a fresh eager trainer started from the same weights on the same batch, then
    `steps` eager steps are timed.  The eager step is host-bound: its time follows the box's host and its load."""
    from anystereo.harness.synthetic import fill_module_deterministic
    from anystereo.harness.train import Trainer
    from anystereo.models import __models__

    def fresh(graph):
        m = __models__["continuous_IGEVStereo"](args)
        fill_module_deterministic(m, base_seed=1)
        return Trainer(m.to(dev), train_iters=a.train_iters, max_disp=args.max_disp, graph=graph)

    def run(tr, n):
        out = []
        for _ in range(n):
            out.append(tr.step(tuple(t.clone() for t in batch))[0])
        torch.cuda.synchronize()
        return [float(v) for v in out]

    n = 3 + n_cmp
    tr = fresh(None)
    got = run(tr, n)
    graphed, memsets = bool(tr.use_graph and tr._graph is not None), getattr(tr, "graph_memsets", None)
    del tr
    torch.cuda.empty_cache()
    eager = fresh(False)
    want = run(eager, n)
    rel = [abs(g - w) / max(abs(w), 1e-12) for g, w in zip(got, want)]
    ok = all(r == r and r < 5e-3 for r in rel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eager.step(batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"how": "Trainer(graph=False) / ANYSTEREO_TRAIN_GRAPH=0: every launch issued by the host (host-bound)",
            "ms_per_step": round(dt / a.steps * 1e3, 2), "samples_per_s": round(a.batch_per_gpu * a.steps / dt, 3), "steps": a.steps,
            "default_step_is_graphed": graphed, "memset_nodes_replaced_left": memsets,
            "loss_trajectory_default_matches_eager": ok, "max_rel_loss_diff_over_replays": max(rel[3:]) if len(rel) > 3 else None,
            "losses_eager": [round(v, 4) for v in want], "losses_default": [round(v, 4) for v in got]}


def train_roofline(a, batch, dev):
    """The largest single launch of a training step: the weight gradient of the 1/4-resolution GRU's z|r convolution over all
    `train_iters` iterations (as_conv2d_wgrad, 3 bf16 MFMAs per product), timed here with HIP events on its own stream."""
    from anystereo import ops
    b, h4, w4 = batch[0].shape[0], batch[0].shape[2] // 4, batch[0].shape[3] // 4
    n, cin, cout = a.train_iters * b, 384, 256
    x = torch.randn(n, cin, h4, w4, device=dev)
    dy = torch.randn(n, cout, h4, w4, device=dev) * 1e-4
    for _ in range(2):
        ops.conv2d_wgrad(x, dy, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.conv2d_wgrad(x, dy, 3)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    flop = 2.0 * n * h4 * w4 * cin * 9 * cout
    peak = 2500.0 / 3.0
    ach = flop / (us * 1e-6) / 1e12
    return {"kernel": "wgrad_kernel<3,1> + wgrad_finish_kernel (gru04 convz|convr, all iterations of a step in one launch)",
            "bound": "mfma", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 3),
            "peak_note": "2.5 PF dense bf16 / 3 MFMAs per product (bf16 hi/lo operand split)", "us_per_launch": round(us, 1),
            "algorithmic_gflop_per_launch": round(flop / 1e9, 1), "traffic": None, "timing": "HIP events around 5 launches"}


def train_cpu_baseline(a, args, model, batch):
    """One full optimisation step (forward + backward + clip + AdamW) of the CPU oracle model on ONE sample of the batch, same
    weights — the oracle's operators are explicit-index torch code, so autograd differentiates them as it does the reference."""
    try:
        from anystereo.harness.metrics import fetch_optimizer, train_step
        from oracle.model import OracleIGEV
        threads = max(1, min(os.cpu_count() or 1, 64))
        torch.set_num_threads(threads)
        ref = OracleIGEV(args)
        ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
        ref.train()
        ref.freeze_bn()
        opt, sched = fetch_optimizer(2e-4, 1e-5, 1000, ref.parameters())
        one = tuple(t[:1].cpu().contiguous() for t in batch)
        t0 = time.perf_counter()
        loss, _ = train_step(ref, opt, sched, None, one, a.train_iters, max_disp=args.max_disp)
        dt = time.perf_counter() - t0
        return {"value": round(1.0 / dt, 4), "unit": "samples/s", "cores": threads, "kind": "port", "s_per_sample": round(dt, 2),
                "sample": f"one full step (forward + backward + clip + AdamW) on 1 of the {batch[0].shape[0]} samples, "
                          f"{a.train_iters} GRU iters, Q={batch[2].shape[1]}, fp32, {threads} threads, no warm-up",
                "loss": round(float(loss), 4)}
    except Exception as e:  # the baseline is a report, never a reason to lose the line
        return {"value": None, "unit": "samples/s", "kind": "port", "error": f"{type(e).__name__}: {e}"[:200]}


# ------------------------------------------------------------------------------------------------------------------
# inference
# ------------------------------------------------------------------------------------------------------------------


class Runner:
    """One workload resident on one GPU: model, inputs, step()."""

    def __init__(self, wl, dev, seed, pairs=1, graph=True, model=None, args=None):
        from anystereo.harness import workloads as WL
        self.wl, self.dev = wl, dev
        if model is None:
            model, args = WL.build_model(wl, device=dev)
        self.model, self.args = model, args
        self.cpu_inputs = WL.build_inputs(wl, seed=seed)
        self.i1, self.i2, self.coord, self.scale = WL.build_inputs(wl, seed=seed, pairs=pairs, device=dev)
        self.pairs = pairs
        self.Q = self.coord.shape[1]
        self.hp, self.wp = self.i1.shape[-2:]
        self.graph = graph and hasattr(model, "enable_graph")
        if hasattr(model, "enable_graph"):
            model.enable_graph(self.graph)

    def step(self, iters=None):
        with torch.no_grad():
            return self.model(self.i1, self.i2, iters=iters or self.wl.iters, test_mode=True, hr_coord=self.coord, scale=self.scale)

    def time_steps(self, n, iters=None):
        self.step(iters)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            out = self.step(iters)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n, out


def kernel_stats(run, passes=2, co_scheduled=False):
    """Per-kernel device times: HIP events on the launch stream around every hot-kernel launch in instrumented EAGER passes
    of the identical workload (a graph replay has no per-kernel events).  co_scheduled=False: the loop's kernels are issued in
    the same order on ONE stream, so each duration is the kernel alone; True: the loop runs on its three streams as it does in
    the timed steps (a kernel's events then include what the other streams' kernels take from it)."""
    from anystereo.harness import timing
    model = run.model
    graphed = run.graph
    if graphed:
        model.enable_graph(False)
    model.serial_streams = not co_scheduled   # same kernels, same order; one stream -> no co-running kernel in an event pair
    liif_par = getattr(model.liif_up, "parallel_inputs", False)
    model.liif_up.parallel_inputs = liif_par and co_scheduled
    run.step()
    timing.enable(True)
    for _ in range(passes):
        # a ~60 ms device-side delay first: the host then enqueues the pass AHEAD of the GPU, so an event pair measures the
        # kernel between them and not the Python / ctypes time between recording the start event and the launch
        torch.cuda._sleep(120_000_000)
        run.step()
    ks = timing.collect()
    timing.enable(False)
    model.serial_streams = False
    model.liif_up.parallel_inputs = liif_par
    if graphed:
        model.enable_graph(True)
    return ks, passes


def micro_time_small_kernels(dev, h, w, reps=20, cold=False):
    """The short HBM-bound kernels (10-60 us) on operands of the workload's shapes (C=96, L=2, G=8, D=48): `reps` launches
    captured into one hipGraph (the host launch path, ~20 us per call through Python + ctypes, is longer than these kernels)
    and replayed between one HIP event pair on the launch stream.
    cold=False: every launch on the SAME operands (120-250 MB: they stay in the 256 MB Infinity Cache).
    cold=True:  the launches rotate over enough operand sets that > 2 x 256 MB pass between two uses of the same line, so
    every launch streams from HBM."""
    from anystereo import ops
    from anystereo.harness.synthetic import det_uniform

    def make_set(k):
        f1 = det_uniform((1, 96, h, w), 1 + 10 * k).to(dev)
        f2 = det_uniform((1, 96, h, w), 2 + 10 * k).to(dev)
        gev = det_uniform((1, 8, 48, h, w), 3 + 10 * k).to(dev)
        disp = det_uniform((1, 1, h, w), 4 + 10 * k, 0.0, 40.0).to(dev)
        return {"f1": f1, "f2": f2, "gev": gev, "disp": disp, "corr": ops.corr_build_pyramid(f1, f2, 2), "geo": ops.geo_pyramid(gev, 2)}

    # convc1 (1x1, 162 -> 64, update.py:78) for the fused lookup + convc1 kernel the GRU loop launches (models/base.py)
    wc1 = (det_uniform((64, 162, 1, 1), 77) * (3.0 / 162) ** 0.5).to(dev)
    bc1 = (det_uniform((64,), 78) * 0.1).to(dev)
    pack = ops.LookupConvPack().get(wc1, bc1)
    P = h * w
    set_mb = {"corr_build": 4 * (2 * 96 * P + P * w * 1.5) / 1e6, "lookup": 4 * (P * 48 * 8 * 1.5 + P * w * 1.5 + 163 * P) / 1e6,
              "lookup_convc1": 4 * (P * 48 * 8 * 1.5 + P * w * 1.5 + 65 * P) / 1e6,
              "gwc_volume": 4 * (2 * 96 * P + 8 * 48 * P) / 1e6, "geo_pyramid": 4 * (8 * 48 * P * 2.5) / 1e6}
    nsets = 1
    if cold:
        nsets = max(3, int(2.2 * INFINITY_CACHE_MB / min(set_mb.values())) + 1)
    sets = [make_set(k) for k in range(nsets)]
    for st in sets:
        st["cor"] = ops.BS8.empty(1, 64, h, w, dev)
    fns = {"corr_build": lambda s: ops.corr_build_pyramid(s["f1"], s["f2"], 2),
           "lookup": lambda s: ops.geo_corr_lookup(s["geo"], s["corr"], s["disp"], 4),
           "lookup_convc1": lambda s: ops.lookup_convc1(s["geo"], s["corr"], s["disp"], 4, pack, out_bs=s["cor"]),
           "gwc_volume": lambda s: ops.gwc_volume(s["f1"], s["f2"], 48, 8),
           "geo_pyramid": lambda s: ops.geo_pyramid(s["gev"], 2)}
    out = {}
    for name, fn in fns.items():
        n = reps if not cold else max(reps, nsets * 2)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for k in range(3):
                fn(sets[k % nsets])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(n):
                fn(sets[k % nsets])
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        out[name] = {"count": n, "total_ms": s.elapsed_time(e), "operand_sets": nsets,
                     "bytes_between_reuse_mb": round(set_mb[name] * nsets, 1)}
        del g
    return out


def roofline_table(kstats, alg, precision, traffic):
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
            if precision == "split":
                rooflines[name]["frac_of_measured_mfma_rate"] = round(ach / (MFMA_F16_RANDOM_DATA_TFLOPS / 3.0), 4)
            if "flops_executed" in e:  # a kernel that executes fewer flops than the reference's statement of the operator
                rooflines[name]["algorithmic_gflop"] = round(e["flops"] / 1e9, 2)
                rooflines[name]["executed_gflop"] = round(e["flops_executed"] / 1e9, 2)
                rooflines[name]["frac_executed"] = round(e["flops_executed"] / avg_s / 1e12 / peak, 4)
            if "bytes" in e:
                rooflines[name]["compulsory_bytes"] = e["bytes"]
    return rooflines


def infer_main(a, rank, world, local):
    dist = world > 1 or "RANK" in os.environ  # under a launcher even one rank joins a process group (RCCL): the training leg uses it
    td = _dist_init(True, local) if dist else None
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from anystereo import _lib
    from anystereo import ops as _ops
    from anystereo.harness import workloads as WL
    _lib.load()
    precision = _ops.get_precision()
    torch.backends.cudnn.benchmark = bool(int(os.environ.get("ANYSTEREO_MIOPEN_BENCHMARK", "0")))
    wl = WL.WORKLOADS[a.config]
    if a.height is not None and a.width is not None:
        wl = WL.custom(a.height, a.width, a.scale or 1.0, a.iters or 32)
    elif a.iters is not None or a.scale is not None:
        wl = WL.Workload(wl.name, wl.model, wl.height, wl.width, a.scale or wl.scale, a.iters or wl.iters, wl.protocol,
                         wl.divis_by, wl.what)
    nb = max(1, a.pairs_per_gpu)
    run = Runner(wl, dev, seed=1234 + rank, pairs=nb, graph=not (a.no_graph or a.serial_loop))
    if a.serial_loop:
        run.model.serial_streams = True
    model = run.model

    for _ in range(a.warmup):
        out = run.step()
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    from anystereo.harness import timing
    if not run.graph:
        timing.enable(True)       # HIP events around every hot-kernel launch, on the launch stream
    # `value` = the MEDIAN of `TIMED_BLOCKS` timed blocks; each block times exactly `--steps` steps between barrier +
    # synchronize on both sides and is reduced with MAX over the ranks.  Box-to-box and run-to-run spread of this workload is a
    # few per cent (profiles/r05_bench_lease_spread.txt): one block is one sample of it, the line carries all of them.
    block_dt, block_tel, block_own = [], [], []
    for _ in range(TIMED_BLOCKS):
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        torch.cuda.synchronize()
        telemetry = GpuTelemetry(local).start()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = run.step()
        torch.cuda.synchronize()
        d_own = time.perf_counter() - t0   # this rank's own K steps (before it waits for the others)
        if dist:
            td.barrier()
        torch.cuda.synchronize()
        d_blk = time.perf_counter() - t0
        block_tel.append(telemetry.stop())
        if dist:
            t = torch.tensor([d_blk], device=dev, dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            d_blk = float(t.item())
        block_dt.append(d_blk)
        block_own.append(d_own)
    order = sorted(range(TIMED_BLOCKS), key=lambda i: block_dt[i])
    mid = order[TIMED_BLOCKS // 2]      # the same index on every rank: block_dt is the all-reduced figure
    dt, telemetry = block_dt[mid], block_tel[mid]
    out_split = out.float().cpu() if precision == "split" else None
    out_this = out.float().cpu()
    per_rank = [block_own[mid]]
    if dist:
        t = torch.tensor([block_own[mid]], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        td.all_gather(gathered, t)
        per_rank = [float(g.item()) for g in gathered]
    assert torch.isfinite(out).all()
    if run.graph:
        kstats, ksteps = kernel_stats(run, passes=2)
    else:
        kstats, ksteps = timing.collect(), a.steps
        timing.enable(False)

    # ms per GRU iteration = (t32 - t8) / 24 on the same inputs, outside the timed region
    lo = max(1, wl.iters // 4)
    ms_iter = (run.time_steps(3)[0] - run.time_steps(3, lo)[0]) / (wl.iters - lo) * 1e3 if wl.iters > lo else None

    # throughput mode (reported next to the headline, never as `value`): 4 pairs per forward fill the 1/8- and
    # 1/16-resolution kernels that leave most CUs idle at one pair; every rank measures, rank 0 reports the job total
    batched = None
    if nb == 1 and not a.no_batched:
        nbb = 4
        rb = Runner(wl, dev, seed=1234 + rank, pairs=nbb, graph=run.graph, model=model, args=run.args)
        for _ in range(2):
            rb.step()
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        tb = time.perf_counter()
        for _ in range(3):
            rb.step()
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
        del rb

    # The N-rank TRAINING leg (cfg 4) on the same ranks and process group, right after the inference leg: what lets a SCALE record
    # (bench.py --gpus 1,2,4,8) answer north_star's training-scaling criterion.  One rank without a launcher keeps the child
    # process of the default line (below).  A watchdog bounds the leg: if it has not returned in time, rank 0 prints the line it
    # already has with train_mode = {"error": ...} and every rank leaves (nobody waits in a collective for a peer that died).
    train_mode_n = None
    leg_guard = None
    if dist and not a.no_train_mode and wl.name == "cfg2":
        leg_guard = _LegWatchdog(rank, float(os.environ.get("ANYSTEREO_TRAIN_LEG_TIMEOUT", "240")))
        if rank == 0:  # the contract's fields of the headline, known before the leg starts: what the watchdog prints if it must
            leg_guard.rearm({
                "metric": "stereo pairs/sec (coreContinuous_IGEV inference, 32-iter GRU, 960x540)",
                "value": round(world * nb * a.steps / dt, 4), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32" if precision == "fp32" else "f32 (3xf16 split-precision MFMA, fp32 accumulate)", "data": "synthetic",
                "config": {"workload": f"{wl.what} (padded {run.wp}x{run.hp}), {wl.iters} GRU iters, scale {wl.scale}, Q={run.Q} queries, "
                                       f"{nb} pair(s) per GPU, random-init weights", "name": wl.name, "pairs_per_gpu": nb,
                           "parallelism": f"replicas x{world}", "gru_loop": "hipGraph" if run.graph else "eager"},
                "per_rank_pairs_per_s": [round(nb * a.steps / v, 4) for v in per_rank],
                "ms_per_gru_iter": None if ms_iter is None else round(ms_iter, 4),
                "roofline": None, "cpu_baseline": None, "truncated": "printed by the training leg's watchdog"})
        try:
            train_mode_n = train_leg(a, rank, world, local, td, dev)
        except Exception as ex:  # a rank-local failure outside the agreed phases
            train_mode_n = {"error": repr(ex)[:300]} if rank == 0 else None
        leg_guard.done()
        torch.cuda.empty_cache()

    if rank == 0:
        h4, w4 = run.hp // 4, run.wp // 4
        alg = algorithmic(nb, h4, w4, run.Q)
        traffic, traffic_source = load_pmc_traffic()
        # the short HBM-bound kernels are re-timed as back-to-back launches of one hipGraph between ONE event pair: a
        # start/stop pair around a single launch adds ~3 us of its own (rocprofv3 durations confirm)
        micro = micro_time_small_kernels(dev, h4, w4)
        serial_loop_lookup = kstats.get("lookup_convc1")  # HIP events around the loop's own launches (one stream)
        for k, v in micro.items():
            kstats[k] = v
        rooflines = roofline_table(kstats, alg, precision, traffic)
        extras = not a.no_extras and world == 1
        if extras:
            cold = roofline_table(micro_time_small_kernels(dev, h4, w4, cold=True), alg, precision, {})
            co, _ = kernel_stats(run, passes=1, co_scheduled=True)
            co = roofline_table({k: v for k, v in co.items() if k in ("lookup", "lookup_convc1")}, alg, precision, {})
            for k, r in rooflines.items():
                if k in cold:
                    r["cache_state"] = "warm: 20 launches on one operand set (fits the 256 MB Infinity Cache)"
                    r["frac_cold"], r["avg_us_cold"] = cold[k]["frac"], cold[k]["avg_us"]
                if k in co:
                    r["frac_in_loop"], r["avg_us_in_loop"] = co[k]["frac"], co[k]["avg_us"]
        dominant = max(rooflines, key=lambda k: rooflines[k]["total_ms"]) if rooflines else None
        # north_star's own bar: >= 40 % of the HBM roofline on volume build + lookup (a1-a3) — each kernel and the group together.
        # The lookup in that figure is the kernel the timed GRU loop LAUNCHES: lookup_convc1 (lookup fused with the encoder's
        # first conv, models/base.py::_iterate_pipelined), isolated-warm, cold, and with its in-loop (co-scheduled) duration;
        # the stand-alone `lookup` kernel (the reference-contract entry, not launched by the loop) is a separate line.
        ns = None
        if all(k in rooflines for k in ("corr_build", "geo_pyramid")) and ("lookup_convc1" in rooflines or "lookup" in rooflines):
            lk = "lookup_convc1" if "lookup_convc1" in rooflines else "lookup"
            ks = ("corr_build", "geo_pyramid", lk)
            if serial_loop_lookup and serial_loop_lookup.get("count") and lk == "lookup_convc1":
                rooflines[lk]["avg_us_loop_serial"] = round(serial_loop_lookup["total_ms"] / serial_loop_lookup["count"] * 1e3, 2)

            def pooled(fk, uk, names=ks, over=None):
                over = over or {}
                t_us = sum(over.get(k, (None, rooflines[k][uk]))[1] for k in names)
                byt = sum(rooflines[k]["frac"] * HBM_PEAK_GBS * rooflines[k]["avg_us"] for k in names)  # algorithmic bytes x 1e-3
                return round(byt / t_us / HBM_PEAK_GBS, 4)
            ns = {"build_plus_lookup_hbm_frac": pooled("frac", "avg_us"), "target": 0.40, "lookup_kernel": lk,
                  "per_kernel": {k: rooflines[k]["frac"] for k in ks},
                  "note": "one build (all-pairs pyramid + geometry pyramid) + one lookup AS THE GRU LOOP LAUNCHES IT (lookup fused "
                          "with convc1), algorithmic bytes / measured time / 8 TB/s; warm = operands resident in the Infinity Cache, "
                          "cold = streamed from HBM, in_loop = the lookup's duration while co-scheduled with the other streams' kernels"}
            if all("frac_cold" in rooflines[k] for k in ks):
                ns["build_plus_lookup_hbm_frac_cold"] = pooled("frac_cold", "avg_us_cold")
                ns["per_kernel_cold"] = {k: rooflines[k]["frac_cold"] for k in ks}
            if "avg_us_in_loop" in rooflines[lk]:
                ns["build_plus_lookup_hbm_frac_in_loop"] = pooled("frac", "avg_us", over={lk: (None, rooflines[lk]["avg_us_in_loop"])})
                ns["lookup_in_loop"] = {"avg_us": rooflines[lk]["avg_us_in_loop"], "frac": rooflines[lk]["frac_in_loop"]}
            # as a pair actually runs: ONE build (both pyramids) and `iters` lookups — algorithmic bytes of all of them over the sum
            # of their durations (lookup: in-loop where measured, else isolated)
            def abytes(k):
                return rooflines[k]["frac"] * HBM_PEAK_GBS * rooflines[k]["avg_us"]
            lk_us_loop = rooflines[lk].get("avg_us_in_loop", rooflines[lk]["avg_us"])
            t_build = rooflines["corr_build"]["avg_us"] + rooflines["geo_pyramid"]["avg_us"]
            b_build = abytes("corr_build") + abytes("geo_pyramid")
            ns["pair_weighted"] = {
                "iters": wl.iters,
                "hbm_frac_lookup_in_loop": round((b_build + wl.iters * abytes(lk)) / (t_build + wl.iters * lk_us_loop) / HBM_PEAK_GBS, 4),
                "hbm_frac_lookup_isolated": round((b_build + wl.iters * abytes(lk)) / (t_build + wl.iters * rooflines[lk]["avg_us"]) / HBM_PEAK_GBS, 4),
                "note": "(build + geo + iters x lookup algorithmic bytes) / (their durations) / 8 TB/s: the figure weighted as one stereo pair "
                        "runs (1 build : iters lookups); the lookup dominates it"}
            if "lookup" in rooflines and lk != "lookup":
                ns["standalone_lookup_kernel"] = {"frac": rooflines["lookup"]["frac"], "frac_cold": rooflines["lookup"].get("frac_cold"),
                                                  "avg_us": rooflines["lookup"]["avg_us"],
                                                  "note": "reference-contract entry ([B,162,h,w] result); not launched by the timed loop"}
        cpu = None
        parity = None
        oracle_out = None
        if not a.no_cpu_baseline and world == 1:  # reported at N=1 only (rank 0), bounded sample
            cpu, oracle_out = cpu_baseline(run)
            if extras and wl.name == "cfg2":
                try:
                    cpu["also"] = {"cfg1": cpu_baseline_cfg1()}
                except Exception as ex:  # never lose the headline line to the side measurement
                    cpu["also"] = {"cfg1": {"error": repr(ex)}}
        fp32_mode = None
        if extras and precision == "split":
            fp32_mode, out_fp32 = fp32_mode_run(run, alg, traffic)
            parity = {"metric": "EPE = mean |disparity - CPU oracle| over all queries of the timed workload, px",
                      "tolerance": 1e-3, "queries": run.Q * nb, "iters": wl.iters,
                      "epe_split_vs_fp32": float((out_split - out_fp32).abs().mean())}
            if oracle_out is not None:
                parity["epe_vs_oracle"] = {"split": float((out_split - oracle_out).abs().mean()),
                                           "fp32": float((out_fp32 - oracle_out).abs().mean())}
                parity["max_abs_vs_oracle"] = {"split": float((out_split - oracle_out).abs().max()),
                                               "fp32": float((out_fp32 - oracle_out).abs().max())}
                parity["oracle_mean_abs_disparity"] = float(oracle_out.abs().mean())
        elif oracle_out is not None:
            parity = {"metric": "EPE = mean |disparity - CPU oracle| over all queries of the timed workload, px",
                      "tolerance": 1e-3, "queries": run.Q * nb, "iters": wl.iters,
                      "epe_vs_oracle": {precision: float((out_this - oracle_out).abs().mean())}}
        reduced = None
        if extras and precision == "split":
            try:
                reduced = reduced_precision_run(run, alg, oracle_out, out_split)
            except Exception as ex:
                reduced = {"error": repr(ex)}
        others = None
        if extras and wl.name == "cfg2":
            others = {}
            for name in ("cfg3", "cfg5"):
                try:
                    r2 = Runner(WL.WORKLOADS[name], dev, seed=1234, graph=run.graph, model=model, args=run.args)
                    t_step, o2 = r2.time_steps(3)
                    lo2 = max(1, r2.wl.iters // 4)
                    t_lo, _ = r2.time_steps(2, lo2)
                    others[name] = {"workload": f"{r2.wl.what}, {r2.wl.iters} GRU iters, padded {r2.wp}x{r2.hp}, Q={r2.Q}",
                                    "value": round(1.0 / t_step, 3), "unit": "pairs/s", "ms_per_step": round(t_step * 1e3, 2),
                                    "ms_per_gru_iter": round((t_step - t_lo) / (r2.wl.iters - lo2) * 1e3, 4), "steps": 3,
                                    "finite": bool(torch.isfinite(o2).all())}
                    # the upsampler's kernels at this configuration's query count (HIP events, kernels one at a time)
                    ks2, _ = kernel_stats(r2, passes=1)
                    others[name]["liif_us"] = {k: round(v["total_ms"] / max(v["count"], 1) * 1e3, 1) for k, v in ks2.items()
                                               if k in ("liif_tail", "liif_mlp_lowres", "structure_feature", "convex_upsample", "liif_mlp")}
                    if ks2.get("liif_tail", {}).get("count"):  # on the MFMA roofline, as SURVEY.md §8(d) places a15
                        a2 = algorithmic(1, r2.hp // 4, r2.wp // 4, r2.Q)["liif_tail"]
                        t_tail = ks2["liif_tail"]["total_ms"] / ks2["liif_tail"]["count"] * 1e-3
                        pk = (MFMA_F16_PEAK_TFLOPS / 3.0 if precision == "split" else MFMA_F32_PEAK_TFLOPS) * 1e12
                        others[name]["liif_tail_mfma_frac"] = round(a2["flops"] / t_tail / pk, 4)
                        others[name]["liif_tail_mfma_frac_executed"] = round(a2["flops_executed"] / t_tail / pk, 4)
                    del r2, o2
                    model.enable_graph(run.graph)  # drop that shape's graph pool
                except Exception as ex:
                    others[name] = {"error": repr(ex)}
            # cfg 1: the other model family (corePrune_RAFT, 256x512, 8 iterations) — its own weights, checked against its oracle
            try:
                wl1 = WL.WORKLOADS["cfg1"]
                m1, a1 = WL.build_model(wl1, device=dev)
                m1.enable_graph(run.graph)
                r1 = Runner(wl1, dev, seed=1234, graph=run.graph, model=m1, args=a1)
                t_step, o1 = r1.time_steps(5)
                rec = {"workload": f"{wl1.what}, {wl1.iters} GRU iters, padded {r1.wp}x{r1.hp}, Q={r1.Q}", "value": round(1.0 / t_step, 3),
                       "unit": "pairs/s", "ms_per_step": round(t_step * 1e3, 2), "steps": 5, "finite": bool(torch.isfinite(o1).all())}
                if not a.no_cpu_baseline:
                    from oracle.model import OracleRAFT
                    ref = OracleRAFT(a1).eval()
                    ref.load_state_dict({k: v.cpu() for k, v in m1.state_dict().items()})
                    i1, i2, coord, sc = r1.cpu_inputs
                    with torch.no_grad():
                        oo = ref(i1, i2, iters=wl1.iters, test_mode=True, hr_coord=coord.clone(), scale=sc)
                    rec["epe_vs_oracle"] = float((o1.float().cpu() - oo.float()).abs().mean())
                others["cfg1"] = rec
                del r1, m1, o1
            except Exception as ex:
                others["cfg1"] = {"error": repr(ex)}
        pass_phases = None
        if extras and run.graph:
            # where a replayed pass spends its time, from marker kernels inside the graph (no profiler: rocprofv3's interception
            # serialises the pre-loop's two branches and reports 5.6 ms for what takes 3.9 ms, profiles/r06_pass_phases_preloop_ab.txt)
            try:
                from anystereo.harness.phases import phases
                ph = phases(model, (run.i1, run.i2, run.coord, run.scale), wl.iters, reps=5)
                pass_phases = {"pre_loop_wall_ms": round(ph["pre_loop_us"] / 1e3, 3), "loop_ms": round(ph["loop_us"] / 1e3, 3),
                               "post_loop_ms": round(ph["post_loop_us"] / 1e3, 3), "ms_per_gru_iter_in_graph": round(ph["us_per_iter"] / 1e3, 4),
                               "pass_ms": round(ph["pass_us"] / 1e3, 3), "markers_us": ph["markers_us"], "how": ph["how"]}
            except Exception as ex:
                pass_phases = {"error": repr(ex)[:300]}
        train_mode = train_mode_n
        if train_mode is None and extras and not dist and wl.name == "cfg2" and not a.no_train_mode:
            # this process is idle on the GPU now: the child has the chip to itself
            torch.cuda.synchronize()
            try:
                train_mode = train_mode_child(a)
            except Exception as ex:
                train_mode = {"error": repr(ex)[:300]}
        line = {
            "metric": "stereo pairs/sec (coreContinuous_IGEV inference, 32-iter GRU, 960x540)",
            "value": round(world * nb * a.steps / dt, 4), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "timed_blocks": TIMED_BLOCKS,
            "value_spread": {"how": f"`value` / `ms_per_step` = the median of {TIMED_BLOCKS} timed blocks of {a.steps} steps each (barrier + synchronize "
                                    "on both sides of every block, MAX over ranks per block)",
                             "pairs_per_s": [round(world * nb * a.steps / v, 4) for v in block_dt],
                             "min": round(world * nb * a.steps / max(block_dt), 4), "max": round(world * nb * a.steps / min(block_dt), 4),
                             "median": round(world * nb * a.steps / dt, 4),
                             "telemetry_per_block": block_tel},
            "dtype": "f32" if precision == "fp32" else "f32 (3xf16 split-precision MFMA, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"{wl.what} (padded {run.wp}x{run.hp}), {wl.iters} GRU iters, scale {wl.scale}, Q={run.Q} queries, "
                                   f"{nb} pair(s) per GPU, random-init weights", "name": wl.name, "pairs_per_gpu": nb,
                       "parallelism": f"replicas x{world}", "gru_loop": "hipGraph" if run.graph else "eager"},
            "per_rank_pairs_per_s": [round(nb * a.steps / v, 4) for v in per_rank],
            "host": {"cpus_per_rank": len(RANK_CPUS or []), "pinned": world > 1 and os.environ.get("ANYSTEREO_PIN", "1") != "0",
                     "torch_threads": torch.get_num_threads(), **telemetry},
            "library": _lib.library_info(),
            "ms_per_gru_iter": None if ms_iter is None else round(ms_iter, 4),
            "pass_phases": pass_phases,
            "roofline": dict(rooflines[dominant], kernel=dominant) if dominant else None,
            "roofline_source": ("HIP events on the launch stream around each hot-kernel launch, " +
                                ("2 eager passes of the same workload right after the timed hipGraph replays" if run.graph
                                 else "inside the timed steps")),
            "traffic_source": traffic_source,
            "rooflines": rooflines,
            "north_star_roofline": ns,
            "kernel_times_us": {k: {"avg": round(v["total_ms"] / max(v["count"], 1) * 1e3, 2), "n": max(1, v["count"] // ksteps)}
                                for k, v in sorted(kstats.items(), key=lambda kv: -kv[1]["total_ms"])},
            "parity": parity,
            "fp32_mode": fp32_mode,
            "reduced_precision_mode": reduced,
            "other_configs": others,
            "throughput_mode": batched,
            "train_mode": train_mode,
            "cpu_baseline": cpu,
        }
        emit(line)
    if dist:
        td.barrier()
        td.destroy_process_group()


def fp32_mode_run(run, alg, traffic):
    """The timed workload once more in exact-fp32 MFMA mode (`v_mfma_f32_32x32x2_f32`, peak 157.3 TFLOP/s): the
    same-precision figure, driver-visible, next to the split-precision headline."""
    from anystereo import ops
    ops.set_precision("fp32")
    try:
        t_step, out = run.time_steps(3)
        ks, _ = kernel_stats(run, passes=1)
        roofs = roofline_table(ks, alg, "fp32", {})
        keep = {k: roofs[k] for k in ("gru04_zr_conv", "gru04_q_conv", "disp_head_conv1") if k in roofs}
        dom = max(keep, key=lambda k: keep[k]["total_ms"]) if keep else None
        res = {"value": round(run.pairs / t_step, 4), "unit": "pairs/s", "ms_per_step": round(t_step * 1e3, 3), "steps": 3,
               "dtype": "f32 (exact fp32-input MFMA)", "roofline": dict(keep[dom], kernel=dom) if dom else None, "rooflines": keep}
        return res, out.float().cpu()
    finally:
        ops.set_precision("split")
        run.model.enable_graph(run.graph)


def reduced_precision_run(run, alg, oracle_out, out_split):
    """The timed workload in the one-MFMA mode (fp16 operands, fp32 accumulate: the reference's `mixed_precision` /
    autocast path, continuous_IGEVstereo.py:287) — a second arithmetic mode with its own tolerance, never `value`."""
    from anystereo import ops
    with ops.fast_fp16(True):
        try:
            t_step, out = run.time_steps(5)
            ks, _ = kernel_stats(run, passes=1)
        finally:
            pass
    run.model.enable_graph(run.graph)
    out = out.float().cpu()
    res = {"value": round(run.pairs / t_step, 4), "unit": "pairs/s", "ms_per_step": round(t_step * 1e3, 3), "steps": 5,
           "dtype": "f16 operands (1 MFMA per product), f32 accumulate and storage", "tolerance_px": 5e-2,
           "epe_vs_split_mode": float((out - out_split).abs().mean())}
    if oracle_out is not None:
        res["epe_vs_oracle"] = float((out - oracle_out).abs().mean())
    if "gru04_zr_conv" in ks and ks["gru04_zr_conv"]["count"]:
        st = ks["gru04_zr_conv"]
        avg_s = st["total_ms"] / st["count"] * 1e-3
        ach = alg["gru04_zr_conv"]["flops"] / avg_s / 1e12
        res["roofline"] = {"kernel": "gru04_zr_conv", "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F16_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(ach / MFMA_F16_PEAK_TFLOPS, 4), "avg_us": round(avg_s * 1e6, 2)}
    return res


def host_cpu_info() -> dict:
    """CPU model and core counts of this host: physical cores = distinct (physical id, core id) pairs of /proc/cpuinfo (the
    BASELINE.md §4 thread count), next to what this process may actually use (affinity mask, cgroup v2 cpu.max quota)."""
    model, pairs, logical = None, set(), 0
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                k, _, v = ln.partition(":")
                k, v = k.strip(), v.strip()
                if k == "processor":
                    logical += 1
                    phys = core = None
                elif k == "model name" and model is None:
                    model = v
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = round(float(q) / float(per), 2)
    except (OSError, ValueError):
        pass
    physical = len(pairs) or None
    usable = min(v for v in (physical or logical or 1, affinity, None if quota is None else max(1, int(quota))) if v)
    return {"model": model, "logical_cpus": logical or (os.cpu_count() or 1), "physical_cores": physical, "affinity_cpus": affinity,
            "cgroup_cpu_quota": quota, "threads_used": max(1, usable)}


def _staged_oracle(cls):
    """The CPU oracle with a wall-clock timer around each hot-path stage (build, lookup, update_block, LIIF):
    the hooks oracle/model.py substitutes are the instrumentation points; everything else of a pass is `backbone_other`."""
    class Staged(cls):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.__dict__["stage_s"] = {}

        def _t(self, name, fn, *a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            st = self.__dict__["stage_s"]
            st[name] = st.get(name, 0.0) + time.perf_counter() - t
            return r

        def _hot_gwc(self, *a, **k):
            return self._t("build", super()._hot_gwc, *a, **k)

        def _hot_init_disp(self, *a, **k):
            return self._t("build", super()._hot_init_disp, *a, **k)

        def _hot_lookup_fn(self, *a, **k):
            fn = self._t("build", super()._hot_lookup_fn, *a, **k)
            return lambda *la, **lk: self._t("lookup", fn, *la, **lk)

        def _hot_update(self, *a, **k):
            return self._t("update_block", super()._hot_update, *a, **k)

        def _hot_upsample(self, *a, **k):
            return self._t("liif", super()._hot_upsample, *a, **k)
    return Staged


def cpu_baseline(run):
    """The CPU oracle (same weights) timed on this host as BASELINE.md §4 states the protocol (the reference's only timing is the
    perf_counter pair of evaluation.py:248-250): fp32, eval, no_grad, threads = physical cores this process may use, 1 warm-up +
    3 timed full runs of 1 pair, the MEDIAN reported, ms per GRU iteration = (t32 - t8) / 24, and the per-stage split (build,
    lookup, update_block, LIIF) from timers on the oracle's hot-path hooks.  A host too slow for 3 full runs inside the bound
    gets fewer (stated in `sample`).  Returns (record, oracle output): the output of a full-length run is kept so the line can
    state this run's EPE against it."""
    from oracle.model import OracleIGEV
    wl = run.wl
    cpu = host_cpu_info()
    threads = cpu["threads_used"]
    torch.set_num_threads(threads)
    ref = _staged_oracle(OracleIGEV)(run.args).eval()
    ref.load_state_dict({k: v.cpu() for k, v in run.model.state_dict().items()})
    i1, i2, coord, sc = run.cpu_inputs

    def go(iters):
        ref.__dict__["stage_s"] = {}
        t = time.perf_counter()
        with torch.no_grad():
            o = ref(i1, i2, iters=iters, test_mode=True, hr_coord=coord.clone(), scale=sc)
        dt = time.perf_counter() - t
        st = dict(ref.__dict__["stage_s"])
        st["backbone_other"] = max(0.0, dt - sum(st.values()))
        return dt, o, st
    go(1)  # warm-up (thread pools, allocator)
    lo = max(1, wl.iters // 4)
    t_lo = go(lo)[0]
    first = go(wl.iters)
    runs = [first]
    budget = 60.0  # seconds of full-length runs
    while len(runs) < 3 and (len(runs) + 1) * first[0] <= budget:
        runs.append(go(wl.iters))
    runs.sort(key=lambda r: r[0])
    t_full, out, stages = runs[len(runs) // 2]
    per_iter = max((t_full - t_lo) / max(1, wl.iters - lo), 1e-6)
    sample = (f"1 pair, full workload ({wl.iters} iters), fp32, eval, no_grad, {threads} threads; 1 warm-up + {len(runs)} timed runs, "
              f"median (all: {', '.join('%.2f' % r[0] for r in runs)} s); ms per GRU iteration = (t{wl.iters} - t{lo}) / {wl.iters - lo}")
    return {"value": round(1.0 / t_full, 5), "unit": "pairs/s", "cores": threads, "kind": "port", "sample": sample,
            "config": wl.name, "s_per_pair": round(t_full, 3), "runs_s": [round(r[0], 3) for r in runs],
            "ms_per_gru_iter": round(per_iter * 1e3, 2),
            "stage_split_s": {k: round(v, 3) for k, v in stages.items()},
            "stage_split_note": "wall clock of the median run by hot-path stage: build = gwc volume + all-pairs / geometry pyramids + "
                                "disparity regression, lookup = all iterations' pyramid lookups, update_block = all iterations' "
                                "BasicMultiUpdateBlock, liif = the upsampler; backbone_other = the rest of the pass",
            "host_cpu": cpu, "torch": torch.__version__}, out.float()


def cpu_baseline_cfg1():
    """BASELINE.md §4 also asks for cfg 1 (corePrune_RAFT, 256x512, 8 GRU iterations) on the host cores."""
    from anystereo.harness import workloads as WL
    from oracle.model import OracleRAFT
    wl = WL.WORKLOADS["cfg1"]
    model, args = WL.build_model(wl)
    ref = OracleRAFT(args).eval()
    ref.load_state_dict(model.state_dict())
    i1, i2, coord, sc = WL.build_inputs(wl)
    threads = host_cpu_info()["threads_used"]
    torch.set_num_threads(threads)

    def go():
        t = time.perf_counter()
        with torch.no_grad():
            ref(i1, i2, iters=wl.iters, test_mode=True, hr_coord=coord.clone(), scale=sc)
        return time.perf_counter() - t
    go()
    ts = sorted(go() for _ in range(3))
    return {"value": round(1.0 / ts[1], 4), "unit": "pairs/s", "cores": threads, "kind": "port", "s_per_pair": round(ts[1], 3),
            "sample": f"{wl.what}, {wl.iters} GRU iters, Q={coord.shape[1]}, median of 3, fp32, {threads} threads"}


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    global torch, RANK_CPUS
    claim_stdout()
    RANK_CPUS = pin_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))  # before torch / HIP start any thread
    import torch as _torch
    torch = _torch
    if RANK_CPUS is not None and world > 1:
        torch.set_num_threads(max(1, len(RANK_CPUS)))
    if not torch.cuda.is_available():
        return dry_train_main(a, rank, world) if a.mode == "train" else dry_main(a, rank, world)
    if REHEARSAL:
        local = 0
    if a.mode == "train":
        return train_main(a, rank, world, local)
    return infer_main(a, rank, world, local)


if __name__ == "__main__":
    main()
