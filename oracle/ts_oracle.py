"""ctypes binding of oracle/libts_oracle.so — TEST INFRASTRUCTURE ONLY (see ts_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this, and only to check (or time,
as a baseline) — never as a compute path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libts_oracle.so")
CLI_PATH = os.path.join(_HERE, "ts_oracle")
REF_CLI = os.path.join(_HERE, "_ref", "xport")   # the real reference, when it was built
MAX_SYNC_ERRORS = 1 << 20
REF_PROGRAM = "70000"   # a program number no 16-bit PAT entry can carry: the reference then demultiplexes nothing


class SyncError(C.Structure):
    _fields_ = [("skipped", C.c_uint64), ("at_packet", C.c_uint64)]


class Discontinuity(C.Structure):
    _fields_ = [("at_packet", C.c_uint64), ("after_sync_errors", C.c_uint64), ("pid", C.c_uint32), ("received", C.c_uint32),
                ("expected", C.c_uint32), ("pad", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("packets", C.c_uint64), ("count", C.c_uint32 * 0x2000), ("first", C.c_uint64 * 0x2000),
                ("last", C.c_uint64 * 0x2000), ("nsync_errors", C.c_uint64), ("sync_errors", SyncError * MAX_SYNC_ERRORS),
                ("ndiscontinuities", C.c_uint64), ("discontinuities", Discontinuity * MAX_SYNC_ERRORS)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-C", _HERE, "libts_oracle.so", "ts_oracle"])
        L = C.CDLL(LIB_PATH)
        L.ts_oracle_scan_mem.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(Result)]
        L.ts_oracle_scan_mem.restype = None
        L.ts_oracle_scan_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(Result)]
        _lib = L
    return _lib


def _unpack(r: Result) -> dict:
    n = min(int(r.nsync_errors), MAX_SYNC_ERRORS)
    return {"packets": int(r.packets), "count": np.ctypeslib.as_array(r.count).copy(),
            "first": np.ctypeslib.as_array(r.first).copy(), "last": np.ctypeslib.as_array(r.last).copy(),
            "nsync_errors": int(r.nsync_errors),
            "sync_errors": [(int(r.sync_errors[k].skipped), int(r.sync_errors[k].at_packet)) for k in range(n)],
            "ndiscontinuities": int(r.ndiscontinuities),
            # (at_packet, sync-error lines printed before it, pid, received, expected)
            "discontinuities": [(int(d.at_packet), int(d.after_sync_errors), int(d.pid), int(d.received), int(d.expected))
                                for d in r.discontinuities[:min(int(r.ndiscontinuities), MAX_SYNC_ERRORS)]]}


def scan_mem(data: bytes, hdmv: bool = False) -> dict:
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
    r = Result()
    lib().ts_oracle_scan_mem(buf.ctypes.data_as(C.c_void_p) if buf.size else None, buf.size, int(hdmv), C.byref(r))
    return _unpack(r)


def scan_file(path: str, hdmv: bool = False) -> dict:
    r = Result()
    if lib().ts_oracle_scan_file(os.fsencode(path), int(hdmv), C.byref(r)) != 0:
        raise FileNotFoundError(path)
    return _unpack(r)


def report_lines(res: dict) -> bytes:
    """The reference's lines this scan is pinned on: sync errors and discontinuities in order of occurrence, then the
    per-PID report."""
    out, d, discs = [], 0, res.get("discontinuities", [])
    for k in range(len(res["sync_errors"]) + 1):   # the two kinds of line interleave, in stream order
        while d < len(discs) and discs[d][1] <= k:
            at, _, pid, got, want = discs[d]
            out.append(b"Discontinuity!, pid = %d <0x%04x>, received = %2d, expected = %2d, at %d\n" % (pid, pid, got, want, at))
            d += 1
        if k < len(res["sync_errors"]):
            s, a = res["sync_errors"][k]
            out.append(b"Transport Sync Error, skipped %d bytes, at %d\n" % (_i32(s), a))
    for pid in np.nonzero(res["count"])[0]:
        out.append(b"packets for pid %4d <0x%04x> = %d, first = %d, last = %d\n" %
                   (pid, pid, _i32(int(res["count"][pid])), int(res["first"][pid]), int(res["last"][pid])))
    return b"".join(out)


def _i32(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def reference_lines(path: str, hdmv: bool = False) -> bytes:
    """Run the compiled reference (oracle/_ref/xport) and keep the lines the scan is pinned on."""
    p = subprocess.run([REF_CLI, "-psh" if hdmv else "-ps", path, REF_PROGRAM, "1", "1"], capture_output=True)
    if p.returncode != 0:
        raise RuntimeError(f"reference xport failed on {path}: rc={p.returncode}")
    return filter_lines(p.stdout)


def filter_lines(stdout: bytes) -> bytes:
    sync = [l for l in stdout.split(b"\n") if l.startswith(b"Transport Sync Error") or l.startswith(b"Discontinuity!")]
    pids = [l for l in stdout.split(b"\n") if l.startswith(b"packets for pid")]
    return b"".join(l + b"\n" for l in sync + pids)
