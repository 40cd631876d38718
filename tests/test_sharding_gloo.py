"""CPU, world_size 2 over gloo: the N>1 path of bench.py (prompt sharding, barrier, max-over-ranks time, summed units)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusers_amd.sharding import job_throughput, prompt_seed, shard_range


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                b, e = shard_range(total, world, r)
                assert 0 <= b <= e <= total
                got += list(range(b, e))
            assert got == list(range(total))
    assert shard_range(256, 8, 3) == (96, 128)
    assert prompt_seed(5, 3) == 8
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = shard_range(9, world, rank)            # 9 prompts over 2 ranks: 5 + 4
    dist.barrier()
    thr = job_throughput(dist, e - b, 1.0 + rank)  # rank 1 is the slow one (2.0 s)
    seeds = [prompt_seed(100, i) for i in range(b, e)]
    gathered = [None] * world
    dist.all_gather_object(gathered, seeds)
    q.put((rank, thr, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_throughput_and_seeds():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, thr, gathered in res:
        assert abs(thr - 9 / 2.0) < 1e-9                     # all units / max time
        assert sum(gathered, []) == list(range(100, 109))     # seeds independent of the split
