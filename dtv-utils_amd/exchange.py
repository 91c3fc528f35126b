"""The two tiny exchange steps of a sharded papr run, over
torch.distributed — RCCL ("nccl" backend) between GPUs on xGMI, gloo in the CPU
tests.  One process per GPU; bulk sample data never leaves its GPU.

  exchange 1 (after pass 1): all-gather one papr_stats record (104 bytes) per
      rank, then every rank folds the records in rank (= file) order with
      papr_stats_merge — arg-extrema with a first-index tie-break are not an
      RCCL reduction op, and a fixed fold order keeps the double sum identical
      on every rank.
  exchange 2 (after pass 2): all-reduce (sum) of the L per-level counters as
      int64.  Integer, hence exactly order-independent.

Messages are <= ~2.5 KB, i.e. latency-bound; link bandwidth is irrelevant.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np
import torch
import torch.distributed as dist

from . import Stats, stats_merge


def allgather_stats(local: Stats, device: torch.device, group=None) -> List[Stats]:
    """Every rank's pass-1 record, in rank order."""
    if not dist.is_initialized():
        return [local]
    world = dist.get_world_size(group)
    raw = np.frombuffer(local.to_bytes(), dtype=np.uint8).copy()
    mine = torch.from_numpy(raw).to(device)
    gathered = torch.empty(world * raw.size, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    flat = gathered.cpu().numpy()
    size = C.sizeof(Stats)
    return [Stats.from_bytes(flat[r * size:(r + 1) * size].tobytes()) for r in range(world)]


def merged_stats(local: Stats, device: torch.device, group=None) -> Stats:
    """Exchange 1: the whole file's pass-1 result, identical on every rank."""
    return stats_merge(allgather_stats(local, device, group))


def allreduce_counts(counts: np.ndarray, device: torch.device, group=None) -> np.ndarray:
    """Exchange 2: per-level counts summed over all shards."""
    if not dist.is_initialized() or counts.size == 0:
        return counts.astype(np.uint64, copy=True)
    t = torch.from_numpy(counts.astype(np.int64)).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy().astype(np.uint64)


def shard_range(nsamples: int, rank: int, world: int, align: int = 8192):
    """Contiguous, chunk-aligned split of the sample axis (same rule as host/papr_main.c)."""
    per = (nsamples + world - 1) // world
    per = (per + align - 1) // align * align
    first = min(rank * per, nsamples)
    return first, min(per, nsamples - first)
