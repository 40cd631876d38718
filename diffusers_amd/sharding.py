"""Multi-GPU partitioning of the txt2img path: independent prompts shard contiguously over ranks,
one process per GPU, weights replicated, NO data-path collective.  torch.distributed is only used to line ranks up
(barrier) and to take the max elapsed time for whole-job throughput."""
from __future__ import annotations

from typing import Tuple


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous ceil(P/G) split; trailing ranks may get fewer (or zero) prompts. Returns [begin, end)."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard arguments")
    per = -(-total // world)
    b = min(total, rank * per)
    return b, min(total, b + per)


def prompt_seed(base_seed: int, prompt_index: int) -> int:
    """Per-prompt seeds make results independent of the number of GPUs."""
    return (base_seed + prompt_index) & 0xFFFFFFFFFFFFFFFF


def job_throughput(dist, local_units: int, local_elapsed_s: float, device=None) -> float:
    """units of ALL ranks / MAX elapsed over ranks (dist = torch.distributed or None for a single process)."""
    if dist is None or not dist.is_initialized():
        return local_units / local_elapsed_s
    import torch
    t = torch.tensor([local_elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item() / t.item())
