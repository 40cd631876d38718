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


# ---------------------------------------------------------------------------------------------- CFG-pair split
def test_pair_layout():
    from diffusers_amd.sharding import pair_layout
    assert [pair_layout(8, r) for r in range(8)] == [(r // 2, r % 2, 4) for r in range(8)]
    with pytest.raises(ValueError):
        pair_layout(3, 0)


def _fake_unet(x, ehs_half):
    """Any deterministic function of (model input, this half's embeddings) will do for the exchange logic."""
    return torch.tanh(x * 0.7 + ehs_half.mean(dim=(1, 2)).view(-1, 1, 1, 1))


def _pair_worker(rank, world, port, q):
    from diffusers_amd.sharding import CfgPair, pair_layout
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pair, half, n_pairs = pair_layout(world, rank)
    groups = [dist.new_group([2 * p, 2 * p + 1]) for p in range(n_pairs)]   # every rank creates every group
    cp = CfgPair(dist, groups[pair], half)
    B, g = 3, 7.5
    gen = torch.Generator().manual_seed(1000 + pair)                        # same inputs on both ranks of a pair
    lat = torch.randn(B, 4, 8, 8, generator=gen)
    ehs = torch.randn(2 * B, 5, 16, generator=gen)
    sig = [14.6, 9.0, 4.0, 1.5, 0.0]
    ref = lat.clone() * sig[0]
    mine = lat.clone() * sig[0]
    for i in range(4):
        s, s2 = sig[i], sig[i + 1]
        # single-process reference: both halves locally (impl_txt2img.rs:273-304 with an Euler step)
        xin = ref / (s * s + 1) ** 0.5
        u, t = _fake_unet(xin, ehs[:B]), _fake_unet(xin, ehs[B:])
        ref = ref + (u + g * (t - u)) * (s2 - s)
        # pair split: one half here, the other through the 2-rank all-gather
        xin = mine / (s * s + 1) ** 0.5
        full = torch.zeros(2, xin.numel())
        full[half] = _fake_unet(xin, ehs[half * B:(half + 1) * B]).reshape(-1)
        cp.gather(full)
        u, t = full[0].view_as(xin), full[1].view_as(xin)
        mine = mine + (u + g * (t - u)) * (s2 - s)
    q.put((rank, float((mine - ref).abs().max()), mine.sum().item()))
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_pair_split_over_gloo_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    world = 4                                     # two pairs
    procs = [ctx.Process(target=_pair_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in procs)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, err, _ in res:
        assert err == 0.0, (rank, err)            # identical arithmetic on identical inputs: bit-exact
    assert res[0][2] == res[1][2] and res[2][2] == res[3][2]   # both ranks of a pair hold the same latents
    assert res[0][2] != res[2][2]                 # pairs work on different prompts
