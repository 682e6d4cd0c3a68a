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
