"""Multi-GPU partitioning of the txt2img path: independent prompts shard contiguously over ranks,
one process per GPU, weights replicated, NO data-path collective.  torch.distributed is only used to line ranks up
(barrier) and to take the max elapsed time for whole-job throughput.

Optional CFG-pair split (BASELINE config 3): GPUs are paired (ranks 2p, 2p+1 — xGMI neighbours); the even rank runs the
unconditional half of every UNet call, the odd rank the text half, and the two swap their [B,4,h,w] outputs once per
step with a 2-rank all-gather (RCCL over xGMI when the backend is nccl).  It halves per-image LATENCY when there are
fewer prompts than GPUs; for throughput plain prompt sharding is better (no exchange at all)."""
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


def pair_layout(world: int, rank: int) -> Tuple[int, int, int]:
    """CFG-pair split: returns (pair_index, half, n_pairs); half 0 = unconditional (even rank), 1 = text (odd rank).
    Prompts are then sharded over PAIRS: ``shard_range(total, n_pairs, pair_index)``."""
    if world <= 0 or world % 2 or not (0 <= rank < world):
        raise ValueError("the CFG-pair split needs an even number of ranks")
    return rank // 2, rank % 2, world // 2


class _DevicePtr:
    """Exposes a raw device address as __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class CfgPair:
    """Exchange object handed to ``StableDiffusionTxt2ImgOptions.run(cfg_pair=...)``.

    ``half``: 0 = this process computes the unconditional half, 1 = the text half.
    ``group``: a torch.distributed group holding exactly the two peers, the unconditional rank FIRST.
    With the nccl backend (RCCL) the all-gather runs device-to-device over xGMI; with gloo (CPU tests, or two ranks
    sharing one GPU) the halves are staged through host memory."""

    def __init__(self, dist, group, half: int, device=None):
        if half not in (0, 1):
            raise ValueError("half must be 0 or 1")
        self.dist, self.group, self.half, self.device = dist, group, half, device
        self.backend = dist.get_backend(group) if dist is not None else None
        if dist is not None and dist.get_world_size(group) != 2:
            raise ValueError("a CFG pair is a group of exactly two ranks")
        if dist is not None and dist.get_rank(group) != half:
            raise ValueError("group order must be (unconditional rank, text rank)")
        # RCCL collectives can be enqueued behind the engine's stream (sde_exchange_stream_fn): no host synchronisation per step
        self.stream_ordered = self.backend == "nccl"
        self._gather_buf = None

    def gather(self, full):
        """``full``: tensor [2, n] (device or CPU) whose row ``self.half`` is filled; fills the other row."""
        import torch
        mine = full[self.half].contiguous()
        if self.backend == "nccl":
            out = torch.empty_like(full)
            self.dist.all_gather_into_tensor(out.view(-1), mine.view(-1), group=self.group)
            full[1 - self.half].copy_(out[1 - self.half])
        else:
            host = mine.detach().to("cpu")
            parts = [torch.empty_like(host), torch.empty_like(host)]
            self.dist.all_gather(parts, host, group=self.group)
            full[1 - self.half].copy_(parts[1 - self.half])
        return full

    def exchange_on_stream(self, ptr: int, half_elems: int, my_half: int, stream: int):
        """The engine's sde_exchange_stream_fn (nccl backend): ``stream`` is the engine's hipStream_t.  The all-gather is issued with
        that stream current, so RCCL's own stream waits for the UNet call already enqueued on it, and ``work.wait()`` makes the
        engine's stream wait for the collective -- stream-side dependencies only, the host returns at once."""
        import torch
        if my_half != self.half:
            raise RuntimeError(f"engine runs half {my_half}, pair object was built for half {self.half}")
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        ext = torch.cuda.ExternalStream(int(stream), device=dev)
        with torch.cuda.stream(ext):
            full = torch.as_tensor(_DevicePtr(ptr, 2 * half_elems), device=dev).view(2, half_elems)
            if self._gather_buf is None or self._gather_buf.shape != full.shape:
                self._gather_buf = torch.empty_like(full)
            work = self.dist.all_gather_into_tensor(self._gather_buf.view(-1), full[self.half].contiguous().view(-1), group=self.group, async_op=True)
            work.wait()                                   # a stream wait (current stream <- RCCL's), not a host wait
            full[1 - self.half].copy_(self._gather_buf[1 - self.half])

    def exchange(self, ptr: int, half_elems: int, my_half: int):
        """The engine's sde_exchange_fn: ``ptr`` = device f32 [2][half_elems], engine stream idle on entry; returns
        after the peer's half is in place."""
        import torch
        if my_half != self.half:
            raise RuntimeError(f"engine runs half {my_half}, pair object was built for half {self.half}")
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        full = torch.as_tensor(_DevicePtr(ptr, 2 * half_elems), device=dev).view(2, half_elems)
        self.gather(full)
        torch.cuda.current_stream(dev).synchronize()
