"""N>1 host logic on CPU: 2 processes over gloo.  Pairs are sharded round-robin, every pair is processed
exactly once (the oracle stands in for the GPU worker: no GPU here), rank 0 gathers the results in pair
order and the job time is the max over ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_pkg_module, ROOT


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    shard = load_pkg_module("shard"); synth = load_pkg_module("synth")
    cols, rows = 128, 96

    def make_pair(i):
        L, R, blend = synth.make_pair_np(cols, rows, 1234 + i)
        return L, R, blend

    def process(L, R, blend):
        f0, f1 = orc.flow_bidir(L, R, 0)
        return torch.from_numpy(orc.combine_novel_views(L, R, f0, f1, blend))

    local = shard.run_sharded(n_pairs, rank, world, make_pair, process)
    like = torch.empty((rows, cols, 4), dtype=torch.uint8)
    got = shard.gather_to_rank0(local, n_pairs, rank, world, like)
    tmax = shard.max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put((sorted(local.keys()), [g.numpy().copy() for g in got], tmax))
    else:
        q.put((sorted(local.keys()), None, tmax))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [3, 4])
def test_two_ranks_shard_and_gather(n_pairs, orc, synth):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + n_pairs
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    keys = sorted(k for r in res for k in r[0])
    assert keys == list(range(n_pairs))                      # every pair exactly once
    assert all(abs(r[2] - 2.0) < 1e-9 for r in res)          # max over ranks of (1.0, 2.0)
    gathered = [r[1] for r in res if r[1] is not None][0]
    assert len(gathered) == n_pairs
    for i in range(n_pairs):                                 # gathered in pair order == single-process result
        L, R, blend = synth.make_pair_np(128, 96, 1234 + i)
        f0, f1 = orc.flow_bidir(L, R, 0)
        assert np.array_equal(gathered[i], orc.combine_novel_views(L, R, f0, f1, blend))


def test_round_robin_assignment():
    shard = load_pkg_module("shard")
    assert shard.pairs_for_rank(8, 0, 8) == [0] and shard.pairs_for_rank(8, 7, 8) == [7]
    assert shard.pairs_for_rank(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((shard.pairs_for_rank(11, r, 3) for r in range(3)), [])) == list(range(11))
    with pytest.raises(ValueError):
        shard.pairs_for_rank(4, 4, 4)


def _worker_overlap(rank, world, port, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = load_pkg_module("shard")
    like = torch.zeros((6, 5, 4), dtype=torch.uint8)
    og = shard.OverlappedGather(like, world, rank)
    seen = []
    for k in range(steps):
        buf = og.out_buffer()
        buf.fill_(10 * k + rank + 1)         # "compute" step k into the buffer no gather is reading
        og.submit()                          # gather k starts; gather k-1 was waited for inside
        if rank == 0 and k > 0:
            pass                             # (the previous result list may be consumed here)
    og.wait()
    if rank == 0:
        seen = [int(t[0, 0, 0]) for t in og.last()]
    q.put((rank, seen, og.k))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("steps", [1, 4])
def test_overlapped_gather_two_ranks(steps):
    """bench.py's N>1 path: the final gather of step k overlaps step k+1 (double-buffered); after wait() rank 0 holds
    every rank's LAST result."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000) + steps
    procs = [ctx.Process(target=_worker_overlap, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r[0], r) for r in (q.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][2] == steps and res[1][2] == steps
    assert res[0][1] == [10 * (steps - 1) + 1, 10 * (steps - 1) + 2]
