"""ctypes binding of oracle/libpapr_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this, and only to check (or time, as a baseline) — never as a compute path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpapr_oracle.so")
CLI_PATH = os.path.join(_HERE, "papr_oracle")
MKCFILE = os.path.join(_HERE, "mkcfile")
REF_CLI = os.path.join(_HERE, "_ref", "papr")  # the real reference, when it was built
REF_CLI_VECTORISED = os.path.join(_HERE, "_ref", "papr_o3avx2")  # same source, gcc -O3 -mavx2 -ffp-contract=off


class Result(C.Structure):
    _fields_ = [("sum", C.c_double), ("n", C.c_int64), ("peak", C.c_float), ("peak_idx", C.c_int64),
                ("re_pos", C.c_float), ("re_neg", C.c_float), ("im_pos", C.c_float), ("im_neg", C.c_float),
                ("re_pos_idx", C.c_int64), ("re_neg_idx", C.c_int64), ("im_pos_idx", C.c_int64),
                ("im_neg_idx", C.c_int64), ("mean", C.c_double), ("papr", C.c_float), ("nlevels", C.c_int),
                ("level", C.POINTER(C.c_float)), ("count", C.POINTER(C.c_int64))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-C", _HERE, "libpapr_oracle.so", "papr_oracle", "mkcfile"])
        L = C.CDLL(LIB_PATH)
        L.papr_oracle_run_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(Result)]
        L.papr_oracle_run_mem.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(Result)]
        L.papr_oracle_count_mem.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
        L.papr_oracle_levels.argtypes = [C.POINTER(Result), C.c_int]
        L.papr_oracle_levels.restype = None
        L.papr_oracle_free.argtypes = [C.POINTER(Result)]
        L.papr_oracle_free.restype = None
        _lib = L
    return _lib


def _unpack(r: Result) -> dict:
    out = {name: getattr(r, name) for name, _ in Result._fields_ if name not in ("level", "count")}
    n = r.nlevels
    out["level"] = np.ctypeslib.as_array(r.level, shape=(n,)).copy() if n > 0 else np.zeros(0, np.float32)
    out["count"] = np.ctypeslib.as_array(r.count, shape=(n,)).copy() if n > 0 else np.zeros(0, np.int64)
    return out


def run_mem(floats: np.ndarray, graph: bool) -> dict:
    """Both passes over an in-memory float stream (even or odd length)."""
    a = np.ascontiguousarray(floats, dtype=np.float32).reshape(-1)
    r = Result()
    lib().papr_oracle_run_mem(a.ctypes.data_as(C.c_void_p), a.size, int(graph), C.byref(r))
    out = _unpack(r)
    lib().papr_oracle_free(C.byref(r))
    return out


def run_file(path: str, graph: bool) -> dict:
    r = Result()
    if lib().papr_oracle_run_file(os.fsencode(path), int(graph), C.byref(r)) != 0:
        raise FileNotFoundError(path)
    out = _unpack(r)
    lib().papr_oracle_free(C.byref(r))
    return out


def count_mem(floats: np.ndarray, level_table: np.ndarray) -> np.ndarray:
    """Pass 2 alone against a given level table."""
    a = np.ascontiguousarray(floats, dtype=np.float32).reshape(-1)
    lv = np.ascontiguousarray(level_table, dtype=np.float32)
    out = np.zeros(lv.size, dtype=np.int64)
    lib().papr_oracle_count_mem(a.ctypes.data_as(C.c_void_p), a.size, lv.ctypes.data_as(C.c_void_p), lv.size,
                                out.ctypes.data_as(C.c_void_p))
    return out


def levels_from(sum_: float, n: int, peak: float, graph: bool):
    """Host scalar stage alone (mean, papr, level table) from pass-1 values."""
    r = Result()
    r.sum, r.n, r.peak = sum_, n, peak
    lib().papr_oracle_levels(C.byref(r), int(graph))
    out = _unpack(r)
    lib().papr_oracle_free(C.byref(r))
    return out["mean"], out["papr"], out["level"]


def cli(args, binary: str = CLI_PATH):
    """Run the oracle (or reference) command line; returns (rc, stdout, stderr)."""
    p = subprocess.run([binary, *args], capture_output=True)
    return p.returncode, p.stdout, p.stderr
