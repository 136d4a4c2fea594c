"""Data-parallel contract of the IQL step, shared by the trainer and the CPU
(gloo) tests.  One process per GPU; parameters, Adam state and the target net are replicated;
each rank runs forward+backward on ITS rows with every batch mean divided by the GLOBAL row
count, so that a SUM all-reduce of the flat gradient (n_params floats + 4 tail words carrying
the three loss contributions) reproduces the single-device step at batch = world * local rows
(reference fixture g8).  The update then runs redundantly on every rank from bit-identical
inputs, which keeps the replicas in sync without a parameter broadcast.

The reference has no multi-device code (no NCCL call site exists to mirror); the one collective
here is `all_reduce(SUM)` on one flat fp32 buffer of ~1.15 MB — RCCL over xGMI on the GPU box
(torch.distributed backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


def inv_batch(local_rows: int, world: int) -> float:
    """Divisor of the batch means on every rank: 1 / (rows of the GLOBAL batch)."""
    return 1.0 / (local_rows * world)


def shard(global_rows: int, rank: int, world: int) -> slice:
    """Parity runs: rank r trains on slice [r*b, (r+1)*b) of one global index draw."""
    if global_rows % world:
        raise ValueError(f"global batch {global_rows} is not divisible by world size {world}")
    b = global_rows // world
    return slice(rank * b, (rank + 1) * b)


def rank_seed(seed: int, rank: int) -> int:
    """Throughput runs: every rank draws its own rows on the device from a rank-offset stream."""
    return (int(seed) + 0x9E3779B97F4A7C15 * rank) & 0xFFFFFFFFFFFFFFFF


def reduce_and_update(flat: torch.Tensor, apply_update: Callable[[torch.Tensor], None],
                      group: Optional[dist.ProcessGroup] = None) -> None:
    """The exchange step: SUM all-reduce of the flat gradient, then the (redundant) update."""
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    apply_update(flat)


def broadcast_state(tensors, group: Optional[dist.ProcessGroup] = None, src: int = 0) -> None:
    """Make rank `src`'s parameters / moments / target the common starting point."""
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def broadcast_state_host(tensors, group: Optional[dist.ProcessGroup] = None, src: int = 0) -> None:
    """The same through host copies, for process groups whose backend cannot move device tensors (gloo rehearsals
    of the peer-to-peer exchange with several ranks on one GPU)."""
    for t in tensors:
        h = t.detach().cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
