"""CPU, world_size 2, gloo: the N>1 launch protocol of bench.py (replicas over independent pairs):
rendezvous, round-robin sharding, barrier, max-over-ranks timing, summed metrics."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, os.path.join(root, "any-stereo_amd"))
    from anystereo.harness import dist
    r, w, _ = dist.init("gloo")
    mine = dist.shard_indices(7, r, w)
    dist.barrier()
    t = dist.max_over_ranks(1.0 + r)            # slowest rank defines the step time
    tot = dist.sum_over_ranks([len(mine), sum(mine)])
    dist.finalize()
    q.put((r, mine, t, tot))


def test_replica_protocol_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    assert all(r[2] == 2.0 for r in res)                  # MAX over ranks
    assert all(r[3] == [7.0, 21.0] for r in res)          # every pair processed exactly once


def _train_worker(rank, world, port, q, kind="raft", impl="ddp"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path[:0] = [root, os.path.join(root, "any-stereo_amd")]
    import numpy as np
    import torch.distributed as td
    from anystereo.harness import dist
    from anystereo.harness.synthetic import fill_module_deterministic, tiny_train_case
    from anystereo.harness.train import Trainer, shard_batch
    from anystereo.models import default_args
    from oracle.model import OracleIGEV, OracleRAFT  # differentiable CPU stand-ins with the product's module tree (test infrastructure)
    torch.set_num_threads(2)
    r, w, _ = dist.init("gloo")
    args = default_args("continuous_RAFTStereo" if kind == "raft" else "continuous_IGEVStereo")
    model = (OracleRAFT if kind == "raft" else OracleIGEV)(args)
    fill_module_deterministic(model, base_seed=1)
    tr = Trainer(model, num_steps=50, train_iters=3, max_disp=args.max_disp, ddp_impl=impl)
    _, _, img1, img2, coord, gt, scale = tiny_train_case(kind)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    loss, met = tr.step(shard_batch((img1, img2, coord, gt, scale), r, w))
    if impl == "ddp":
        # wrapped at the first step, after the probe pass that freezes gradient-less parameters: plain DDP, no unused-parameter walk
        assert isinstance(tr.module, torch.nn.parallel.DistributedDataParallel) and tr.ddp_mode.startswith("plain DDP")
        assert not tr.module.find_unused_parameters
    else:  # bare module, one all-reduce of the concatenated gradients between backward and the update
        assert tr.module is model and tr.ddp_mode.startswith("flat")
    z = np.load(os.path.join(root, "tests", "golden", f"train_{kind}.npz"))
    total = float(np.sqrt((z["norms"] ** 2).sum()))          # clip_grad_norm_(1.0) scaled every gradient by 1/total
    named = dict(model.named_parameters())
    worst = 0.0
    for i, n in enumerate(str(x) for x in z["full_names"]):
        ref = torch.from_numpy(z[f"g{i}"])
        got = named[n].grad * (total + 1e-6)
        worst = max(worst, ((got - ref).abs().max() / ref.abs().max()).item())
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double()
    sums = [torch.zeros(2, dtype=torch.float64) for _ in range(w)]
    td.all_gather(sums, torch.stack([flat.sum(), flat.abs().sum()]))
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in model.named_parameters())
    losses = dist.sum_over_ranks([float(loss)])
    dist.finalize()
    q.put((r, worst, [s.tolist() for s in sums], moved, losses[0] / w, float(z["loss"]), list(tr.frozen_unused)))


import pytest


@pytest.mark.parametrize("kind,impl", [("raft", "ddp"), ("igev", "ddp"), ("raft", "flat")])
def test_ddp_training_step_world2(kind, impl):
    """cfg 4 protocol on CPU: 2 ranks x 1 sample, DDP(gloo) gradient averaging -> the full-batch gradient of the
    reference (G8 fixture; both samples have the same number of valid queries), identical parameters on both ranks
    after the AdamW step.  IGEV: the probe pass freezes the classifier (it only feeds init_disp, which the loss does not
    see), so plain DDP runs without find_unused_parameters.  impl "flat": the bare module + ONE all-reduce of the concatenated
    gradients (the exchange a graphed multi-rank step uses) gives the same averaged gradient."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_train_worker, args=(r, 2, port, q, kind, impl)) for r in range(2)]
    for p in ps:
        p.start()
    res = []
    for _ in range(600):
        try:
            res.append(q.get(timeout=0.5))
        except Exception:
            assert all(p.exitcode in (None, 0) for p in ps), "a rank died"
        if len(res) == len(ps):
            break
    assert len(res) == len(ps)
    res.sort()
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r, worst, sums, moved, mean_loss, ref_loss, frozen in res:
        # IGEV: the classifier and the unused BatchNorm of the hourglass' last (bn=False) transposed conv get no gradient
        assert ("classifier.weight" in frozen and len(frozen) <= 4) if (kind == "igev" and impl == "ddp") else (frozen == []), frozen
        assert sums[0] == sums[1], "parameters diverged between ranks"
        assert moved > 200
        if kind == "raft":
            assert worst < 5e-3, f"rank {r}: averaged gradient differs from the reference full-batch gradient ({worst:.2e})"
            assert abs(mean_loss - ref_loss) < 1e-3 * ref_loss
        else:
            # IGEV's hourglass holds BatchNorm3d layers in training mode: each rank normalises with the statistics of ITS
            # samples (as each DataParallel replica does, train_continuous_IGEV.py:184), so a 2 x 1 split is a different
            # function from the fixture's single batch of 2 — only the protocol is checked here (G8 itself: test_hip_parity)
            assert worst == worst and mean_loss == mean_loss and mean_loss > 0


def _state_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path[:0] = [root, os.path.join(root, "any-stereo_amd")]
    import warnings
    import torch.distributed as td
    from anystereo import ops
    from anystereo.harness import dist
    from anystereo.harness.synthetic import fill_module_deterministic, tiny_train_case
    from anystereo.harness.train import Trainer, shard_batch
    from anystereo.models import default_args
    from oracle.model import OracleRAFT
    torch.set_num_threads(2)
    r, w, _ = dist.init("gloo")
    args = default_args("continuous_RAFTStereo")
    model = OracleRAFT(args)
    fill_module_deterministic(model, base_seed=1 + 17 * r)  # every rank starts from DIFFERENT weights (per-rank RNG / only rank 0 loaded)
    tr = Trainer(model, num_steps=50, train_iters=2, max_disp=args.max_disp, ddp_impl="flat")
    _, _, img1, img2, coord, gt, scale = tiny_train_case("raft")
    # the broadcast must be visible to every (data_ptr, _version)-keyed cache (weight packs, BN folds, graph fingerprints): a rank
    # that ran a forward before its first step holds packs of its pre-broadcast weights
    state = list(model.parameters()) + [b_ for b_ in model.buffers() if b_ is not None and b_.numel()]
    v0 = [t._version for t in state]
    tr._sync_module_state()
    bumped = all(t._version > a for t, a in zip(state, v0))
    tr.step(shard_batch((img1, img2, coord, gt, scale), r, w))
    flat = torch.cat([t.detach().reshape(-1).double() for t in list(model.parameters()) + list(model.buffers())])
    sums = [torch.zeros(2, dtype=torch.float64) for _ in range(w)]
    td.all_gather(sums, torch.stack([flat.sum(), flat.abs().sum()]))
    # overflow handling is a group decision: only rank 1 "saw" saturated operands, both ranks halve the scale at the same step
    tr.loss_scale = 4096.0
    ops_count = ops.split_overflow_count
    ops.split_overflow_count = lambda reset=True: 3 if r == 1 else 0
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            n = tr._poll_overflow()
            gate = tr._overflow_gate()
    finally:
        ops.split_overflow_count = ops_count
    dist.finalize()
    q.put((r, [s_.tolist() for s_ in sums], n, tr.loss_scale, gate, tr.skipped_steps, bumped))


def test_flat_exchange_broadcasts_rank0_state_and_overflow_is_collective_world2():
    """ddp_impl "flat" (what a graphed multi-rank step uses) on ranks built from different weights: rank 0's parameters and buffers
    are broadcast before the first step (the role of DistributedDataParallel's constructor), so the ranks hold identical
    parameters after it; a split-precision overflow seen by ONE rank halves the loss scale / drops the update on BOTH."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_state_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = []
    for _ in range(600):
        try:
            res.append(q.get(timeout=0.5))
        except Exception:
            assert all(p.exitcode in (None, 0) for p in ps), "a rank died"
        if len(res) == len(ps):
            break
    assert len(res) == len(ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r, sums, n, scale, gate, skipped, bumped in sorted(res):
        assert sums[0] == sums[1], "ranks that started from different weights were not synchronised"
        assert r == 0 or bumped, "the state broadcast did not bump the tensors' version counters (stale weight packs on this rank)"
        assert n == 3 and scale == 1024.0, (r, n, scale)      # two collective polls (poll + gate), each halving on both ranks
        assert gate is False and skipped == 1


def _check_train_leg(d, world):
    """`train_mode` of an N-rank `bench.py --gpus N` line: the training leg the same ranks run after the inference leg
    (bench.train_leg) — every key a SCALE record needs to answer north_star's training-scaling criterion."""
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, root)
    import bench
    tm = d["train_mode"]
    assert tm is not None and "error" not in tm, tm
    assert set(bench.TRAIN_LEG_KEYS) <= set(tm), set(bench.TRAIN_LEG_KEYS) - set(tm)
    assert tm["n_gpus"] == world and tm["global_batch"] == 4 * world
    assert len(tm["per_rank_samples_per_s"]) == world and len(tm["one_rank_ms_per_step"]) == world
    assert len(tm["ranks_seen"]) == world and sorted(r["rank"] for r in tm["ranks_seen"]) == list(range(world))
    assert tm["distinct_devices"] == world
    assert tm["exchange_ms"] is not None and tm["exchange_ms"] > 0 and tm["grad_bytes"] > 0
    assert tm["scaling"] is not None and tm["scaling"] > 0 and tm["ms_per_step"] > 0
    assert tm["trainer"]["gradient_exchange"].startswith("flat") and tm["loss_finite"] is True
    return tm


def test_bench_launcher_spawns_ranks_world2():
    """`python bench.py --gpus 2` without torchrun: the parent spawns 2 ranks itself (it never imports torch or touches a
    GPU), the ranks rendezvous on 127.0.0.1, and rank 0 prints ONE JSON line with n_gpus = 2 and one value per rank.  Here
    (no GPU) the ranks run the dry protocol over gloo; on a GPU box the same launcher starts one RCCL rank per GPU."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""   # also on a GPU box: protocol rehearsal only
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None
    assert len(d["per_rank_step_s"]) == 2 and d["config"]["parallelism"] == "replicas x2"
    tm = _check_train_leg(d, 2)  # the N-rank training leg rides in the same line
    assert tm["dry_run"] is True and tm["value"] is None
    # every rank is bound to its own core set before torch starts a thread, and sizes its pools to it (bench.pin_rank)
    sets = d["per_rank_cpus"]
    avail = len(os.sched_getaffinity(0))
    if avail >= 2:
        assert len(sets) == 2 and not set(sets[0]) & set(sets[1]), sets
        assert len(sets[0]) == len(sets[1]) == avail // 2
        assert d["torch_threads"] == avail // 2
    # the parent's code path does not import torch
    src = open(os.path.join(root, "bench.py")).read()
    assert "\nimport torch" not in src.split("def main():")[0]


def test_bench_train_launcher_dry_protocol_world2():
    """`python bench.py --mode train --gpus 2` without a GPU: the launcher spawns 2 ranks, they run the Trainer's flat gradient
    exchange over gloo on a stand-in module (every rank built with DIFFERENT weights) and rank 0 prints ONE JSON line with the
    fields the 8-GPU driver run reads: summed samples/s, one value per rank, `allreduce_overlap.exposed_allreduce_ms`, disjoint
    core sets — and the ranks hold identical parameters after the steps."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["metric"] == "train_samples_per_s" and d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None
    assert len(d["per_rank_samples_per_s"]) == 2 and d["config"]["global_batch"] == 8
    ov = d["allreduce_overlap"]
    assert {"ms_per_step_with_allreduce", "ms_per_step_no_sync", "exposed_allreduce_ms", "gradient_bytes"} <= set(ov)
    assert d["parameters_equal_across_ranks"] is True and d["trainer"]["gradient_exchange"].startswith("flat")
    assert "wgrad" in d["dtype"]
    sets = d["per_rank_cpus"]
    if len(os.sched_getaffinity(0)) >= 2:
        assert len(sets) == 2 and not set(sets[0]) & set(sets[1]), sets


def test_bench_under_torch_distributed_run_world2():
    """The DRIVER's launch form — `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 ...` — without a GPU: the two ranks take RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
    launcher's environment, bind themselves to disjoint core sets, rendezvous over gloo and rank 0 prints the one line."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and len(d["per_rank_step_s"]) == 2
    _check_train_leg(d, 2)
    sets = d["per_rank_cpus"]
    if len(os.sched_getaffinity(0)) >= 2:
        assert len(sets) == 2 and not set(sets[0]) & set(sets[1]), sets


def test_bench_train_leg_watchdog_prints_stashed_line():
    """The training leg of an N-rank run is bounded: a rank stuck in it (a peer died inside a collective) lets rank 0 print the
    headline it already has, with train_mode = {"error": ...}, and every rank ends (bench._LegWatchdog)."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = ("import sys, time; sys.path.insert(0, %r); sys.argv=['bench.py']; import bench\n"
            "g = bench._LegWatchdog(0, 0.3); g.rearm({'metric': 'm', 'value': 1.5}); time.sleep(30)\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["value"] == 1.5 and "timed out" in d["train_mode"]["error"]
    code2 = code.replace("_LegWatchdog(0, 0.3)", "_LegWatchdog(1, 0.3)")
    r2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=60)
    assert r2.returncode == 0 and r2.stdout.strip() == ""  # other ranks leave silently
