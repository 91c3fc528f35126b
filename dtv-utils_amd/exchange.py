"""The exchange steps of a sharded papr run — ctypes binding of the C ABI's papr_exchange_*
(include/papr_hip.h; dtv-utils_amd/csrc/papr_exchange.cpp), one process per GPU, bulk sample data never leaves its GPU.

  stats(local)        all-gather of one 96-byte papr_stats per rank + the ordered merge (rank = file order), identical
                      on every rank; also the sum of the ranks in front of this one
  counts(c)           all-reduce (sum) of the per-level counters, 64-bit integers
  exact_sum(program)  all-gather of the shards' exact-sum programs + papr_exact_chain: the reference's sequential sum

Two transports behind the same C code: `Exchange.rccl(gpu)` — RCCL over xGMI: ncclAllGather / ncclAllReduce on the
context's own HIP stream, device staging, one stream synchronisation per exchange (the 128-byte ncclUniqueId is
created on rank 0 and handed round with torch.distributed, whatever its backend); `Exchange.over_torch()` — the
collectives of an initialised torch.distributed group (gloo in the CPU tests) passed in as C callbacks.  Nothing is
reduced in Python.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from . import PaprError, Stats, lib

ID_BYTES = 128


class ExchangeTiming(C.Structure):
    _fields_ = [("stats_calls", C.c_uint64), ("counts_calls", C.c_uint64), ("exact_calls", C.c_uint64),
                ("stats_us", C.c_double), ("counts_us", C.c_double), ("exact_us", C.c_double),
                ("in_stream_calls", C.c_uint64)]

    def as_dict(self) -> dict:
        d = {"in_stream_collectives": int(self.in_stream_calls)}
        for kind in ("stats", "counts", "exact"):
            calls = getattr(self, kind + "_calls")
            d[kind] = {"calls": int(calls), "us_per_call": (getattr(self, kind + "_us") / calls) if calls else None}
        return d


_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t)


class _Ops(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allgather", _ALLGATHER), ("allreduce_sum_u64", _ALLREDUCE)]


_bound = False


def _lib():
    global _bound
    L = lib()
    if not _bound:
        vp, i32 = C.c_void_p, C.c_int
        L.papr_exchange_unique_id.argtypes = [vp]
        L.papr_exchange_open_rccl.argtypes = [C.POINTER(vp), vp, vp, i32, i32]
        L.papr_exchange_open_ops.argtypes = [C.POINTER(vp), C.POINTER(_Ops), i32, i32]
        L.papr_exchange_open_local.argtypes = [C.POINTER(vp), i32]
        L.papr_exchange_open_local.restype = i32
        L.papr_exchange_open_rccl_local.argtypes = [C.POINTER(vp), i32]
        L.papr_exchange_open_rccl_local.restype = i32
        L.papr_exchange_bind.argtypes = [vp, vp]
        L.papr_exchange_bind.restype = i32
        L.papr_exchange_open_rccl_local_async.argtypes = [C.POINTER(vp), i32, C.POINTER(i32)]
        L.papr_exchange_open_rccl_local_async.restype = i32
        L.papr_exchange_adopt_rccl.argtypes = [vp, vp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.papr_exchange_adopt_rccl.restype = i32
        L.papr_exchange_is_rccl.argtypes = [vp]
        L.papr_exchange_is_rccl.restype = i32
        L.papr_exchange_abort.argtypes = [vp]
        L.papr_exchange_abort.restype = None
        L.papr_exchange_selftest.argtypes = [vp, vp, C.c_int]
        L.papr_exchange_selftest.restype = C.c_int
        L.papr_exchange_close.argtypes = [vp]
        L.papr_exchange_close.restype = None
        L.papr_exchange_last_error.argtypes = [vp]
        L.papr_exchange_last_error.restype = C.c_char_p
        L.papr_exchange_stats.argtypes = [vp, C.POINTER(Stats), C.POINTER(Stats), C.POINTER(C.c_double), vp]
        L.papr_exchange_counts.argtypes = [vp, vp, i32]
        L.papr_exchange_exact_sum.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_double)]
        L.papr_exchange_get_timing.argtypes = [vp, C.POINTER(ExchangeTiming), i32]
        for name in ("papr_exchange_unique_id", "papr_exchange_open_rccl", "papr_exchange_open_ops", "papr_exchange_stats",
                     "papr_exchange_counts", "papr_exchange_exact_sum", "papr_exchange_get_timing"):
            getattr(L, name).restype = i32
        _bound = True
    return L


ABI_SYMBOLS = ("papr_exchange_unique_id", "papr_exchange_open_rccl", "papr_exchange_open_ops", "papr_exchange_close",
               "papr_exchange_last_error", "papr_exchange_stats", "papr_exchange_counts", "papr_exchange_exact_sum",
               "papr_exchange_get_timing", "papr_exchange_open_local", "papr_exchange_abort",
               "papr_exchange_open_rccl_local", "papr_exchange_bind",
               "papr_exchange_open_rccl_local_async", "papr_exchange_adopt_rccl", "papr_exchange_is_rccl", "papr_exchange_selftest")


class Exchange:
    """One papr_exchange: this process's end of the exchanges of a sharded run."""

    def __init__(self, handle, rank: int, world: int, transport: str, keep=None):
        self._L = _lib()
        self._x = handle
        self.rank, self.world, self.transport = rank, world, transport
        self._keep = keep   # callbacks / contexts that must outlive the handle

    # ---- constructors ----
    @classmethod
    def single(cls) -> "Exchange":
        """world size 1: every exchange is the identity (no transport is touched)."""
        L = _lib()
        x = C.c_void_p()
        rc = L.papr_exchange_open_ops(C.byref(x), None, 0, 1)
        if rc:
            raise PaprError(rc, "papr_exchange_open_ops", L.papr_exchange_last_error(None).decode())
        return cls(x, 0, 1, "none")

    @classmethod
    def local(cls, n: int) -> List["Exchange"]:
        """n linked handles for n threads of this process (papr_exchange_open_local: what bin/papr's shard threads use)."""
        L = _lib()
        xs = (C.c_void_p * n)()
        rc = L.papr_exchange_open_local(xs, n)
        if rc:
            raise PaprError(rc, "papr_exchange_open_local", L.papr_exchange_last_error(None).decode())
        return [cls(C.c_void_p(xs[r]), r, n, "threads") for r in range(n)]

    @classmethod
    def rccl_local(cls, n: int) -> List["Exchange"]:
        """n linked handles for n threads of this process with RCCL underneath (papr_exchange_open_rccl_local); every
        thread calls bind(gpu) once its context is open."""
        L = _lib()
        xs = (C.c_void_p * n)()
        rc = L.papr_exchange_open_rccl_local(xs, n)
        if rc:
            raise PaprError(rc, "papr_exchange_open_rccl_local", L.papr_exchange_last_error(None).decode())
        return [cls(C.c_void_p(xs[r]), r, n, "threads+RCCL") for r in range(n)]

    @classmethod
    def rccl_local_async(cls, devices) -> List["Exchange"]:
        """papr_exchange_open_rccl_local_async: hub handles at once, the communicators coming up in threads of their own;
        adopt(gpu, timeout_s) later takes them — or leaves the handles the hub's."""
        L = _lib()
        n = len(devices)
        xs = (C.c_void_p * n)()
        rc = L.papr_exchange_open_rccl_local_async(xs, n, (C.c_int * n)(*devices))
        if rc:
            raise PaprError(rc, "papr_exchange_open_rccl_local_async", L.papr_exchange_last_error(None).decode())
        return [cls(C.c_void_p(xs[r]), r, n, "threads(+RCCL when it is up)") for r in range(n)]

    def adopt(self, gpu, timeout_s: float = 30.0):
        """papr_exchange_adopt_rccl: (set-up seconds of this rank's thread, seconds this call waited)."""
        setup, waited = C.c_double(), C.c_double()
        self._chk(self._L.papr_exchange_adopt_rccl(self._x, gpu._ctx, timeout_s, C.byref(setup), C.byref(waited)),
                  "papr_exchange_adopt_rccl")
        self._keep = gpu
        return setup.value, waited.value

    def bind(self, gpu):
        """papr_exchange_bind: ncclCommInitRank on gpu's device (all ranks at once); no-op for the other transports."""
        self._chk(self._L.papr_exchange_bind(self._x, gpu._ctx), "papr_exchange_bind")
        self._keep = gpu
        return self

    @property
    def is_rccl(self) -> bool:
        return bool(self._L.papr_exchange_is_rccl(self._x))

    def abort(self):
        self._L.papr_exchange_abort(self._x)

    def selftest(self, gpu=None, verbose: bool = True):
        """papr_exchange_selftest: every collective the sharded step uses, once, on tiny buffers with predictable contents
        (all ranks call it together; rank 0 prints one stderr line per collective with its microseconds)."""
        self._chk(self._L.papr_exchange_selftest(self._x, gpu._ctx if gpu is not None else None, int(verbose)),
                  "papr_exchange_selftest")
        return self

    @classmethod
    def rccl(cls, gpu, rank: int, world: int, group=None) -> "Exchange":
        """RCCL communicator over the ranks of an initialised torch.distributed group (used only to hand the
        ncclUniqueId round); the collectives run on `gpu`'s own stream inside libpaprhip."""
        import torch.distributed as dist
        L = _lib()
        uid = bytearray(ID_BYTES)
        if rank == 0:
            buf = (C.c_ubyte * ID_BYTES)()
            rc = L.papr_exchange_unique_id(buf)
            if rc:
                raise PaprError(rc, "papr_exchange_unique_id", L.papr_exchange_last_error(None).decode())
            uid = bytearray(buf)
        if world > 1 or dist.is_initialized():
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = bytearray(box[0])
        x = C.c_void_p()
        raw = (C.c_ubyte * ID_BYTES).from_buffer(uid)
        rc = L.papr_exchange_open_rccl(C.byref(x), gpu._ctx, raw, rank, world)
        if rc:
            raise PaprError(rc, "papr_exchange_open_rccl", L.papr_exchange_last_error(None).decode())
        return cls(x, rank, world, "RCCL", keep=gpu)

    @classmethod
    def over_torch(cls, group=None) -> "Exchange":
        """The same C exchange code over the collectives of a torch.distributed group (CPU tensors: gloo)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def allgather(_user, send, recv, nbytes):
            try:
                mine = torch.frombuffer((C.c_ubyte * nbytes).from_address(send), dtype=torch.uint8).clone()
                out = torch.empty(world * nbytes, dtype=torch.uint8)
                dist.all_gather_into_tensor(out, mine, group=group)
                C.memmove(recv, out.numpy().ctypes.data, world * nbytes)
                return 0
            except Exception:   # an exception must not unwind through the C frames
                return 1

        def allreduce(_user, buf, count):
            try:
                a = np.ctypeslib.as_array(buf, shape=(count,))
                t = torch.from_numpy(a.astype(np.int64))
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                a[:] = t.numpy().astype(np.uint64)
                return 0
            except Exception:
                return 1

        ops = _Ops(None, _ALLGATHER(allgather), _ALLREDUCE(allreduce))
        L = _lib()
        x = C.c_void_p()
        rc = L.papr_exchange_open_ops(C.byref(x), C.byref(ops), rank, world)
        if rc:
            raise PaprError(rc, "papr_exchange_open_ops", L.papr_exchange_last_error(None).decode())
        return cls(x, rank, world, "gloo", keep=ops)

    def close(self):
        if self._x:
            self._L.papr_exchange_close(self._x)
            self._x = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int, what: str):
        if rc:
            raise PaprError(rc, what, self._L.papr_exchange_last_error(self._x).decode())

    # ---- the exchanges ----
    def stats(self, local: Stats):
        """(total, sum of the ranks in front of this one, [every rank's record])."""
        total, before = Stats(), C.c_double()
        every = (Stats * self.world)()
        self._chk(self._L.papr_exchange_stats(self._x, C.byref(local), C.byref(total), C.byref(before), every),
                  "papr_exchange_stats")
        return total, before.value, list(every)

    def merged_stats(self, local: Stats) -> Stats:
        return self.stats(local)[0]

    def counts(self, counts: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(counts, dtype=np.uint64).copy()
        self._chk(self._L.papr_exchange_counts(self._x, a.ctypes.data_as(C.c_void_p), a.size), "papr_exchange_counts")
        return a

    def exact_sum(self, program: bytes) -> float:
        out = C.c_double()
        buf = C.create_string_buffer(program, len(program))
        self._chk(self._L.papr_exchange_exact_sum(self._x, buf, len(program), C.byref(out)), "papr_exchange_exact_sum")
        return out.value

    def timing(self, reset: bool = False) -> ExchangeTiming:
        t = ExchangeTiming()
        self._chk(self._L.papr_exchange_get_timing(self._x, C.byref(t), int(reset)), "papr_exchange_get_timing")
        return t


def shard_range(nsamples: int, rank: int, world: int, align: int = 8192):
    """Contiguous, chunk-aligned split of the sample axis (same rule as host/papr_main.c)."""
    per = (nsamples + world - 1) // world
    per = (per + align - 1) // align * align
    first = min(rank * per, nsamples)
    return first, min(per, nsamples - first)
