"""One-process-per-GPU helpers for the replica-parallel (independent stereo pairs) launch:
rank discovery from the torchrun environment, round-robin sharding of pair indices, and the
max-over-ranks timing reduction bench.py reports.  No data-path collective exists on this path
(SURVEY.md §8e): the only traffic is the scalar reduction of timings / metrics — the role of
`reduce_scalar_outputs` (metrics_utils/experiment.py:166-193)."""
from __future__ import annotations

import os

import torch
import torch.distributed as td


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend: str = "nccl", device=None):
    """Initialise the default process group when WORLD_SIZE > 1 (backend 'nccl' is RCCL on ROCm)."""
    rank, world, local = env_rank()
    if world > 1 and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        td.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int):
    """Round-robin assignment of independent pairs to ranks."""
    return list(range(rank, n_items, world))


def barrier():
    if td.is_initialized():
        td.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    if not td.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device="cpu"):
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if td.is_initialized():
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t.tolist()


def finalize():
    if td.is_initialized():
        td.barrier()
        td.destroy_process_group()
