"""Rank plumbing for multi-GPU runs (one process per GPU, torch.distributed;
backend "nccl" is RCCL on ROCm, "gloo" for CPU rehearsals).

Two things shard:
  * independent evaluations -- the hyperparameter optimiser's candidates (multi-start points,
    line-search trial points) dealt round-robin to the ranks, every rank holding the full X, y and
    evaluating its own candidates with no data-path collective; results meet in one all_gather of
    (LML, gradient) per batch.  That is what this module serves (my_candidates, gather_results) and what
    bench.py reports as `value` for BASELINE configs 1-3 on N > 1 GPUs;
  * ONE evaluation over all GPUs: the 2-D block-cyclic factorisation with RCCL inside the library
    (gogp_amd/sharded.py: ShardedGP; gogp_amd/csrc/dist2d.hip) -- BASELINE configs 4 and 5.
The reference has no counterpart (single process, goroutines only: gp/gp.go:165-213).
"""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device=None):
    import torch.distributed as dist
    rank, _, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if device is not None and backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend=backend, **kw)
    return rank, world


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    """MAX over ranks of a host scalar (the step-time reduction of bench.py)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def my_candidates(ncand: int, rank: int, world: int) -> List[int]:
    """Indices of the candidates evaluated by `rank` (round-robin deal)."""
    return list(range(rank, ncand, world))


def gather_results(local_idx: Sequence[int], local_vals: np.ndarray, ncand: int,
                   width: int, device="cpu") -> np.ndarray:
    """all_gather of per-candidate result rows (e.g. [LML, grad...]) into an
    (ncand x width) array identical on every rank."""
    import torch
    import torch.distributed as dist
    out = np.zeros((ncand, width))
    local_vals = np.asarray(local_vals, dtype=np.float64).reshape(len(local_idx), width)
    if not (dist.is_available() and dist.is_initialized()):
        out[list(local_idx)] = local_vals
        return out
    world = dist.get_world_size()
    per = (ncand + world - 1) // world
    buf = torch.zeros((per, width + 1), dtype=torch.float64, device=device)
    for r, (i, row) in enumerate(zip(local_idx, local_vals)):
        buf[r, 0] = float(i) + 1.0  # 0 marks an empty slot
        buf[r, 1:] = torch.from_numpy(row)
    allb = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(allb, buf)
    for b in allb:
        bb = b.cpu().numpy()
        for row in bb:
            if row[0] > 0:
                out[int(row[0]) - 1] = row[1:]
    return out
