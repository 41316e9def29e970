"""CPU, world_size 2 over gloo: the N>1 path of bench.py shards clips with no data-path collective and
reports max-over-ranks time; gather_rows reassembles per-clip results in global order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sm_hpss_mtl_amd.sharding import gather_rows, shard_indices, shard_range


def test_shards_partition_the_batch():
    for n, w in ((1024, 8), (1000, 8), (7, 8), (0, 2), (513, 2)):
        idx = np.concatenate([shard_indices(n, r, w) for r in range(w)])
        assert sorted(idx.tolist()) == list(range(n))
        rngs = [shard_range(n, r, w) for r in range(w)]
        assert rngs[0][0] == 0 and rngs[-1][1] == n and all(a[1] == b[0] for a, b in zip(rngs, rngs[1:]))
        assert max(h - l for l, h in rngs) - min(h - l for l, h in rngs) <= 1
    with pytest.raises(ValueError):
        shard_indices(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = shard_indices(n, rank, world)
    # stand-in for the per-clip hot path: a deterministic per-clip row
    local = torch.stack([torch.arange(3, dtype=torch.float32) + 10.0 * i for i in idx]) if len(idx) else torch.zeros((0, 3))
    full = gather_rows(local, idx, n, dist)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py: max over ranks
    q.put((rank, full.numpy(), float(t)))
    dist.destroy_process_group()


def test_two_rank_gloo_gather_and_max_time():
    world, n = 2, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in range(world)]
    [p.join(60) for p in ps]
    want = np.stack([np.arange(3, dtype=np.float32) + 10.0 * i for i in range(n)])
    for rank, full, tmax in res:
        assert np.array_equal(full, want)
        assert abs(tmax - 0.2) < 1e-12
