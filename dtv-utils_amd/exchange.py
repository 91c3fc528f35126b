"""The two tiny exchange steps of a sharded papr run, over
torch.distributed — RCCL ("nccl" backend) between GPUs on xGMI, gloo in the CPU
tests.  One process per GPU; bulk sample data never leaves its GPU.

  exchange 1 (after pass 1): all-gather one papr_stats record (96 bytes) per
      rank, then every rank folds the records in rank (= file) order with
      papr_stats_merge — arg-extrema with a first-index tie-break are not an
      RCCL reduction op, and a fixed fold order keeps the double sum identical
      on every rank.
  exchange 2 (after pass 2): all-reduce (sum) of the L per-level counters as
      int64.  Integer, hence exactly order-independent.

Messages are <= ~2.5 KB (latency-bound; link bandwidth is irrelevant), so what
costs is host<->device hops: buffers are allocated once (pinned on the host
side), copies are asynchronous, and each exchange ends in ONE stream
synchronisation.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np
import torch
import torch.distributed as dist

from . import MAX_LEVELS, Stats, stats_merge

_STATS_BYTES = C.sizeof(Stats)


class Exchange:
    """Pre-allocated buffers for the two exchanges on one device / process group."""

    def __init__(self, device: torch.device, group=None):
        self.device = torch.device(device)
        self.group = group
        self.active = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.cuda = self.device.type == "cuda"
        if not self.active:
            return

        def host(n, dtype):
            t = torch.empty(n, dtype=dtype)
            return t.pin_memory() if self.cuda else t

        self.h_rec = host(_STATS_BYTES, torch.uint8)
        self.h_all = host(self.world * _STATS_BYTES, torch.uint8)
        self.d_rec = torch.empty(_STATS_BYTES, dtype=torch.uint8, device=self.device)
        self.d_all = torch.empty(self.world * _STATS_BYTES, dtype=torch.uint8, device=self.device)
        self.h_cnt = host(MAX_LEVELS, torch.int64)
        self.d_cnt = torch.empty(MAX_LEVELS, dtype=torch.int64, device=self.device)

    def _sync(self):
        if self.cuda:
            torch.cuda.current_stream(self.device).synchronize()

    def allgather_stats(self, local: Stats) -> List[Stats]:
        """Every rank's pass-1 record, in rank order."""
        if not self.active:
            return [local]
        self.h_rec.numpy()[:] = np.frombuffer(local.to_bytes(), dtype=np.uint8)
        self.d_rec.copy_(self.h_rec, non_blocking=True)
        dist.all_gather_into_tensor(self.d_all, self.d_rec, group=self.group)
        self.h_all.copy_(self.d_all, non_blocking=True)
        self._sync()
        flat = self.h_all.numpy()
        return [Stats.from_bytes(flat[r * _STATS_BYTES:(r + 1) * _STATS_BYTES].tobytes()) for r in range(self.world)]

    def merged_stats(self, local: Stats) -> Stats:
        """Exchange 1: the whole file's pass-1 result, identical on every rank."""
        return stats_merge(self.allgather_stats(local))

    def allgather_bytes(self, blob: bytes) -> List[bytes]:
        """Variable-size byte strings from every rank, in rank order (exact-sum programs:
        ~0.1-2 MB each).  Two collectives: sizes, then the padded payload."""
        if not self.active:
            return [blob]
        size = torch.tensor([len(blob)], dtype=torch.int64, device=self.device)
        sizes = torch.empty(self.world, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(sizes, size, group=self.group)
        sizes = sizes.cpu().tolist()
        cap = max(max(sizes), 1)
        mine = torch.zeros(cap, dtype=torch.uint8)
        mine[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        mine = mine.to(self.device)
        allb = torch.empty(self.world * cap, dtype=torch.uint8, device=self.device)
        dist.all_gather_into_tensor(allb, mine, group=self.group)
        flat = allb.cpu().numpy()
        return [flat[r * cap:r * cap + sizes[r]].tobytes() for r in range(self.world)]

    def allreduce_counts(self, counts: np.ndarray) -> np.ndarray:
        """Exchange 2: per-level counts summed over all shards."""
        n = int(counts.size)
        if not self.active or n == 0:
            return counts.astype(np.uint64, copy=True)
        self.h_cnt.numpy()[:n] = counts.astype(np.int64, copy=False)
        d = self.d_cnt[:n]
        d.copy_(self.h_cnt[:n], non_blocking=True)
        dist.all_reduce(d, op=dist.ReduceOp.SUM, group=self.group)
        self.h_cnt[:n].copy_(d, non_blocking=True)
        self._sync()
        return self.h_cnt.numpy()[:n].astype(np.uint64)


_cache: Dict[tuple, Exchange] = {}


def _get(device, group) -> Exchange:
    key = (str(device), id(group), dist.is_initialized())
    if key not in _cache:
        _cache[key] = Exchange(device, group)
    return _cache[key]


def allgather_stats(local: Stats, device, group=None) -> List[Stats]:
    return _get(device, group).allgather_stats(local)


def merged_stats(local: Stats, device, group=None) -> Stats:
    return _get(device, group).merged_stats(local)


def allreduce_counts(counts: np.ndarray, device, group=None) -> np.ndarray:
    return _get(device, group).allreduce_counts(counts)


def shard_range(nsamples: int, rank: int, world: int, align: int = 8192):
    """Contiguous, chunk-aligned split of the sample axis (same rule as host/papr_main.c)."""
    per = (nsamples + world - 1) // world
    per = (per + align - 1) // align * align
    first = min(rank * per, nsamples)
    return first, min(per, nsamples - first)
