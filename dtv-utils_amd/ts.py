"""ctypes binding of the transport-stream packet scan in libpaprhip.so (include/ts_hip.h) for tests and bench.py.

Plumbing only, like the papr binding in __init__.py: nothing is computed here and there is no fallback — without the
HIP library or a GPU the calls raise.
"""
from __future__ import annotations

import ctypes as C
import weakref
import os

import numpy as np

from . import PaprError, lib

PIDS = 0x2000
MAX_SYNC_ERRORS = 4096
MAX_DISCONTINUITIES = 4096

ABI_SYMBOLS = ("ts_walk_init", "ts_walk_is_clean", "ts_walk", "ts_format_report", "ts_format_report_all", "ts_hip_open", "ts_hip_close",
               "ts_hip_last_error", "ts_hip_upload", "ts_hip_load_file", "ts_hip_adopt", "ts_hip_generate", "ts_hip_generate_damaged",
               "ts_hip_download", "ts_hip_scan", "ts_hip_sync_error_count", "ts_hip_get_sync_errors", "ts_hip_discontinuity_count",
               "ts_hip_get_discontinuities", "ts_hip_result_size", "ts_host_pool_selftest")


class SyncError(C.Structure):
    _fields_ = [("skipped", C.c_uint64), ("at_packet", C.c_uint64)]


class Discontinuity(C.Structure):
    _fields_ = [("at_packet", C.c_uint64), ("after_sync_errors", C.c_uint64), ("pid", C.c_uint32), ("received", C.c_uint8),
                ("expected", C.c_uint8), ("pad", C.c_uint8 * 2)]


class ScanResult(C.Structure):
    """ts_scan_result (include/ts_hip.h)."""
    _fields_ = [("packets", C.c_uint64), ("count", C.c_uint32 * PIDS), ("first", C.c_uint64 * PIDS),
                ("last", C.c_uint64 * PIDS), ("nsync_errors", C.c_uint64), ("sync_errors", SyncError * MAX_SYNC_ERRORS),
                ("ndiscontinuities", C.c_uint64), ("discontinuities", Discontinuity * MAX_DISCONTINUITIES),
                ("cc_state", C.c_uint8 * PIDS),
                ("bytes", C.c_uint64), ("gpu_packets", C.c_uint64), ("launches", C.c_uint32), ("walks", C.c_uint32),
                ("kernel_ms", C.c_double), ("merge_ms", C.c_double)]

    # Every sync error / discontinuity of the scan, when the inline lists are not all of them: fetched from the context
    # (ts_hip_get_sync_errors / _discontinuities) when first asked for — the context holds them until its next scan.
    _src = None          # (TsHip, the number of the scan) while the complete lists are still to be fetched
    _errs = None
    _discs = None

    def _complete(self):
        if self._src is None:
            return
        gpu, scan_id = self._src[0](), self._src[1]
        self._src = None
        if gpu is None or gpu._scan_id != scan_id or not gpu._ctx:
            raise PaprError(-6, "ts_hip_get_sync_errors", "the complete lists are the context's LAST scan's: this result is older")
        n, nd = int(self.nsync_errors), int(self.ndiscontinuities)
        if n > MAX_SYNC_ERRORS:
            errs = (SyncError * n)()
            gpu._chk(gpu._L.ts_hip_get_sync_errors(gpu._ctx, 0, n, errs), "ts_hip_get_sync_errors")
            self._errs = errs
        if nd > MAX_DISCONTINUITIES:
            discs = (Discontinuity * nd)()
            gpu._chk(gpu._L.ts_hip_get_discontinuities(gpu._ctx, 0, nd, discs), "ts_hip_get_discontinuities")
            self._discs = discs

    @property
    def _all_errors(self):
        self._complete()
        return self._errs

    @property
    def _all_discs(self):
        self._complete()
        return self._discs

    def report(self) -> bytes:
        """The reference's report lines (ts_format_report / ts_format_report_all): every sync error and discontinuity in
        the order they were printed, then the PIDs."""
        if self._all_errors is None and self._all_discs is None:
            if int(self.nsync_errors) > MAX_SYNC_ERRORS or int(self.ndiscontinuities) > MAX_DISCONTINUITIES:
                raise PaprError(-7, "ts_format_report", "the result holds only the first %d / %d of %d sync errors / %d discontinuities"
                                % (MAX_SYNC_ERRORS, MAX_DISCONTINUITIES, self.nsync_errors, self.ndiscontinuities))
            buf = C.create_string_buffer((1 << 20) + 96 * (int(self.nsync_errors) + int(self.ndiscontinuities)))
            n = _lib().ts_format_report(C.byref(self), buf, len(buf))
            return buf.raw[:n]
        errs = self._all_errors if self._all_errors is not None else self.sync_errors
        ne = len(self._all_errors) if self._all_errors is not None else int(self.nsync_errors)
        discs = self._all_discs if self._all_discs is not None else self.discontinuities
        nd = len(self._all_discs) if self._all_discs is not None else int(self.ndiscontinuities)
        buf = C.create_string_buffer((1 << 20) + 96 * (ne + nd))
        n = _lib().ts_format_report_all(C.byref(self), errs, ne, discs, nd, buf, len(buf))
        return buf.raw[:n]

    def discontinuity_list(self):
        """(at_packet, sync-error lines printed before it, pid, received, expected) — the oracle's tuples."""
        src = self._all_discs if self._all_discs is not None else self.discontinuities
        n = len(self._all_discs) if self._all_discs is not None else min(int(self.ndiscontinuities), MAX_DISCONTINUITIES)
        return [(int(src[k].at_packet), int(src[k].after_sync_errors), int(src[k].pid), int(src[k].received), int(src[k].expected))
                for k in range(n)]

    def tables(self):
        return (np.ctypeslib.as_array(self.count).copy(), np.ctypeslib.as_array(self.first).copy(),
                np.ctypeslib.as_array(self.last).copy())

    def sync_error_list(self):
        src = self._all_errors if self._all_errors is not None else self.sync_errors
        n = len(self._all_errors) if self._all_errors is not None else min(int(self.nsync_errors), MAX_SYNC_ERRORS)
        return [(int(src[k].skipped), int(src[k].at_packet)) for k in range(n)]


class WalkState(C.Structure):
    _fields_ = [("pos", C.c_uint64), ("skipped", C.c_uint64), ("stale_af", C.c_uint32), ("extra_pending", C.c_uint32),
                ("hdmv", C.c_int)]


_bound = False


def _lib():
    global _bound
    L = lib()
    if not _bound:
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
        L.ts_walk_init.argtypes = [C.POINTER(WalkState), i32]
        L.ts_walk_init.restype = None
        L.ts_walk_is_clean.argtypes = [C.POINTER(WalkState)]
        L.ts_walk_is_clean.restype = i32
        L.ts_walk.argtypes = [C.POINTER(WalkState), vp, u64, u64, i32, u64, C.POINTER(ScanResult)]
        L.ts_walk.restype = u64
        L.ts_format_report.argtypes = [C.POINTER(ScanResult), C.c_char_p, C.c_size_t]
        L.ts_format_report.restype = C.c_size_t
        L.ts_format_report_all.argtypes = [C.POINTER(ScanResult), vp, u64, vp, u64, C.c_char_p, C.c_size_t]
        L.ts_format_report_all.restype = C.c_size_t
        L.ts_hip_sync_error_count.argtypes = [vp]
        L.ts_hip_sync_error_count.restype = u64
        L.ts_hip_get_sync_errors.argtypes = [vp, u64, u64, vp]
        L.ts_hip_get_sync_errors.restype = i32
        L.ts_hip_discontinuity_count.argtypes = [vp]
        L.ts_hip_discontinuity_count.restype = u64
        L.ts_hip_get_discontinuities.argtypes = [vp, u64, u64, vp]
        L.ts_hip_get_discontinuities.restype = i32
        L.ts_hip_open.argtypes = [C.POINTER(vp), i32]
        L.ts_hip_close.argtypes = [vp]
        L.ts_hip_close.restype = None
        L.ts_hip_last_error.argtypes = [vp]
        L.ts_hip_last_error.restype = C.c_char_p
        L.ts_hip_upload.argtypes = [vp, vp, u64]
        L.ts_hip_load_file.argtypes = [vp, C.c_char_p]
        L.ts_hip_adopt.argtypes = [vp, vp, u64]
        L.ts_hip_generate.argtypes = [vp, u64, u64, i32]
        L.ts_hip_generate_damaged.argtypes = [vp, u64, u64, u64]
        L.ts_hip_generate_damaged.restype = i32
        L.ts_hip_download.argtypes = [vp, vp, u64, u64]
        L.ts_hip_scan.argtypes = [vp, i32, C.POINTER(ScanResult)]
        L.ts_host_pool_selftest.argtypes = [C.c_int, C.c_int]
        L.ts_host_pool_selftest.restype = C.c_int
        L.ts_hip_result_size.argtypes = []
        L.ts_hip_result_size.restype = C.c_size_t
        if L.ts_hip_result_size() != C.sizeof(ScanResult):   # (ts_hip_scan writes the LIBRARY's struct into the caller's buffer)
            raise RuntimeError(f"libpaprhip.so's ts_scan_result has {L.ts_hip_result_size()} bytes, this binding's {C.sizeof(ScanResult)}")
        for name in ("ts_hip_open", "ts_hip_upload", "ts_hip_load_file", "ts_hip_adopt", "ts_hip_generate",
                     "ts_hip_download", "ts_hip_scan", "ts_hip_get_sync_errors"):   # (ts_hip_sync_error_count returns a count: u64)
            getattr(L, name).restype = i32
        _bound = True
    return L


def walk(data: bytes, hdmv: bool = False, window: int = None, rng=None) -> ScanResult:
    """The host walker alone over a whole stream (GPU-free; what ts_hip_scan uses across irregular packets).  With
    `window` the stream is fed in windows of random size < window, as the runtime feeds it."""
    L = _lib()
    st, res = WalkState(), ScanResult()
    L.ts_walk_init(C.byref(st), int(hdmv))
    a = np.frombuffer(data, dtype=np.uint8)
    n = a.size
    if window is None:
        L.ts_walk(C.byref(st), a.ctypes.data if n else None, 0, n, 1, 2 ** 64 - 1, C.byref(res))
        return res
    for _ in range(10 ** 7):
        base = st.pos
        w = min(n - base, int(rng.integers(189, window)))
        eof = int(base + w >= n)
        sub = a[base:base + w]
        L.ts_walk(C.byref(st), sub.ctypes.data if w else None, base, w, eof, int(rng.integers(0, 5)), C.byref(res))
        if eof and (st.pos >= n or st.pos == base):
            if st.pos < n:
                L.ts_walk(C.byref(st), a[st.pos:].ctypes.data, st.pos, n - st.pos, 1, 2 ** 64 - 1, C.byref(res))
            return res
    raise RuntimeError("ts_walk made no progress")


class TsHip:
    """One ts_hip_ctx: one GPU, one transport stream resident in HBM."""

    def __init__(self, device: int = 0):
        self._L = _lib()
        self._ctx = C.c_void_p()
        rc = self._L.ts_hip_open(C.byref(self._ctx), device)
        if rc:
            detail = self._L.ts_hip_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise PaprError(rc, "ts_hip_open", detail)
        self._keepalive = None
        self._scan_id = 0
        self._last_lazy = None

    def _settle_last_result(self):
        """The complete lists of a result are the context's until its next scan: a result somebody still holds gets them now
        (a result nobody holds any more costs nothing)."""
        last = self._last_lazy() if self._last_lazy is not None else None
        self._last_lazy = None
        if last is not None and last._src is not None:
            last._complete()

    def close(self):
        if self._ctx:
            self._settle_last_result()
            self._L.ts_hip_close(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int, what: str):
        if rc:
            raise PaprError(rc, what, self._L.ts_hip_last_error(self._ctx).decode())

    def upload(self, data: bytes):
        a = np.frombuffer(data, dtype=np.uint8)
        self._chk(self._L.ts_hip_upload(self._ctx, a.ctypes.data if a.size else None, a.size), "ts_hip_upload")

    def load_file(self, path: str):
        self._chk(self._L.ts_hip_load_file(self._ctx, os.fsencode(path)), "ts_hip_load_file")

    def adopt(self, device_ptr: int, nbytes: int, keepalive=None):
        self._chk(self._L.ts_hip_adopt(self._ctx, C.c_void_p(device_ptr), nbytes), "ts_hip_adopt")
        self._keepalive = keepalive

    def generate(self, npackets: int, seed: int = 0x7500001, hdmv: bool = False):
        self._chk(self._L.ts_hip_generate(self._ctx, seed, npackets, int(hdmv)), "ts_hip_generate")

    def generate_damaged(self, npackets: int, period: int, seed: int = 0x7500001) -> int:
        """npackets of the synthetic stream with one damaged spot every `period` packets; returns the stream's size."""
        self._chk(self._L.ts_hip_generate_damaged(self._ctx, seed, npackets, period), "ts_hip_generate_damaged")
        return npackets * 188 - npackets // (4 * period)

    def download(self, first: int, nbytes: int) -> bytes:
        out = np.empty(nbytes, dtype=np.uint8)
        self._chk(self._L.ts_hip_download(self._ctx, out.ctypes.data if nbytes else None, first, nbytes), "ts_hip_download")
        return out.tobytes()

    def scan(self, hdmv: bool = False) -> ScanResult:
        self._settle_last_result()
        res = ScanResult()
        self._chk(self._L.ts_hip_scan(self._ctx, int(hdmv), C.byref(res)), "ts_hip_scan")
        self._scan_id += 1
        if int(res.nsync_errors) > MAX_SYNC_ERRORS or int(res.ndiscontinuities) > MAX_DISCONTINUITIES:
            # the result holds the first 4096 of either inline; the reference prints every one: the rest on demand
            assert int(self._L.ts_hip_sync_error_count(self._ctx)) == int(res.nsync_errors)
            assert int(self._L.ts_hip_discontinuity_count(self._ctx)) == int(res.ndiscontinuities)
            res._src = (weakref.ref(self), self._scan_id)
            self._last_lazy = weakref.ref(res)
        return res
