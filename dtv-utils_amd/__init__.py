"""ctypes binding of libpaprhip.so (include/papr_hip.h) for tests and bench.py.

The product is the C ABI and the `bin/papr` host program; this module is only
plumbing so Python harnesses can drive the same entry points.  It never
computes anything itself and has no fallback: if the HIP library is missing or
no GPU is usable, calls raise.

The directory name (`dtv-utils_amd`) is not importable as-is; load it with
`__graft_entry__.load_package()`, which registers it as `dtv_utils_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("PAPR_LIB_PATH") or os.path.join(_HERE, "libpaprhip.so")   # (the override: A/B builds of tools/)
CLI_PATH = os.path.join(REPO_ROOT, "bin", "papr")

MAX_OVERRIDES = 8
NO_INDEX = 2**64 - 1
FLAG_NAN = 1
FLAG_ODD_TAIL = 2
MAX_LEVELS = 16384

# every symbol include/papr_hip.h declares (tests check the .so exports them all)
ABI_SYMBOLS = (
    "papr_hip_abi_version", "papr_hip_device_count", "papr_hip_open", "papr_hip_close",
    "papr_hip_last_error", "papr_hip_device_name", "papr_hip_set_tuning", "papr_hip_set_timing",
    "papr_hip_get_timing", "papr_hip_get_timing_launches", "papr_file_samples", "papr_hip_load_file", "papr_hip_load_stream", "papr_hip_stream_stats", "papr_exact_chain_continue", "papr_hip_get_ingest_timing", "papr_hip_upload",
    "papr_hip_adopt", "papr_hip_generate", "papr_hip_download", "papr_hip_stats",
    "papr_stats_init", "papr_stats_merge", "papr_levels", "papr_hip_ccdf",
    "papr_hip_set_exact", "papr_hip_exact_program", "papr_hip_ccdf_exact", "papr_exact_chain",
    "papr_hip_estimate", "papr_hip_stats_sweep", "papr_hip_get_sweep_info", "papr_hip_get_wg_finish", "papr_guess_levels",
    "papr_hip_estimate_file", "papr_hip_load_file_sweep", "papr_hip_shard_fits",
    "papr_level_key", "papr_sweep_bands", "papr_sweep_resolve", "papr_hip_set_exact_hint",
    "papr_sweep_band_for", "papr_hip_set_band", "papr_hip_analyze", "papr_hip_sweep_variant_built",
)


class SynthOverride(C.Structure):
    _fields_ = [("index", C.c_uint64), ("i", C.c_float), ("q", C.c_float)]


class SynthSpec(C.Structure):
    """papr_synth_spec (include/papr_synth.h)."""
    _fields_ = [("seed", C.c_uint64), ("scale", C.c_float), ("n_overrides", C.c_uint32),
                ("ov", SynthOverride * MAX_OVERRIDES)]

    ENVELOPES = {"gauss": 0, "constant": 1, "bursty": 2}

    @classmethod
    def make(cls, seed: int = 0x5EED0001, scale: float = 0.0,
             overrides: Sequence[tuple] = (), envelope: str = "gauss") -> "SynthSpec":
        sp = cls()
        sp.seed = seed
        sp.scale = scale
        sp.n_overrides = len(overrides) | (cls.ENVELOPES[envelope] << 8)
        for k, (idx, i, q) in enumerate(overrides):
            sp.ov[k].index, sp.ov[k].i, sp.ov[k].q = idx, i, q
        return sp

    @classmethod
    def spike(cls, n: int, seed: int = 0x5EED0001, envelope: str = "gauss") -> "SynthSpec":
        """papr_synth_spike_spec: the bench workload (two equal ~30 dB spikes)."""
        if n < 16:
            return cls.make(seed, envelope=envelope)
        a = (n // 1000) * 731 + ((n % 1000) * 731) // 1000
        b = (n // 10) * 9 + ((n % 10) * 9) // 10
        return cls.make(seed, 0.0, [(a, 36.9375, 0.0), (b, 36.9375, 0.0)], envelope=envelope)


class Stats(C.Structure):
    """papr_stats (include/papr_hip.h)."""
    _fields_ = [("sum", C.c_double), ("n", C.c_uint64), ("peak_idx", C.c_uint64),
                ("re_pos_idx", C.c_uint64), ("re_neg_idx", C.c_uint64),
                ("im_pos_idx", C.c_uint64), ("im_neg_idx", C.c_uint64),
                ("nan_first_idx", C.c_uint64),
                ("peak", C.c_float), ("re_pos", C.c_float), ("re_neg", C.c_float),
                ("im_pos", C.c_float), ("im_neg", C.c_float),
                ("nan_first_neg", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}

    def to_bytes(self) -> bytes:
        return bytes(memoryview(self))

    @classmethod
    def from_bytes(cls, raw: bytes) -> "Stats":
        return cls.from_buffer_copy(raw)


class Timing(C.Structure):
    _fields_ = [("stats_ms", C.c_double), ("stats_launches", C.c_uint64), ("stats_bytes", C.c_uint64),
                ("ccdf_ms", C.c_double), ("ccdf_launches", C.c_uint64), ("ccdf_bytes", C.c_uint64),
                ("exact_ms", C.c_double), ("exact_launches", C.c_uint64), ("exact_bytes", C.c_uint64),
                ("sweep_ms", C.c_double), ("sweep_launches", C.c_uint64), ("sweep_bytes", C.c_uint64),
                ("aux_ms", C.c_double), ("aux_launches", C.c_uint64), ("aux_bytes", C.c_uint64)]


SWEEP_REASONS = ("ok", "no sweep", "exact mode / not resident", "no band form", "out of band", "stash full")


class SweepInfo(C.Structure):
    """papr_hip_sweep_info: what the last one-sweep pass / papr_hip_ccdf did."""
    _fields_ = [("stash_samples", C.c_uint64), ("stash_capacity", C.c_uint64), ("estimate_samples", C.c_uint64),
                ("swept", C.c_int), ("resolved", C.c_int), ("reason", C.c_int), ("band_log2", C.c_int),
                ("exact_redo_tiles", C.c_uint32), ("gave_up", C.c_uint32), ("kernel_variant", C.c_int),
                ("xcd_first", C.c_int)]

    def as_dict(self) -> dict:
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["reason"] = SWEEP_REASONS[self.reason] if 0 <= self.reason < len(SWEEP_REASONS) else self.reason
        return d


class Result(C.Structure):
    """papr_result: what papr_hip_analyze returns (the whole file's pass-1 record, host scalars, how the step went)."""
    _fields_ = [("total", Stats), ("mean", C.c_double), ("papr", C.c_float), ("nlevels", C.c_int),
                ("exact_sum", C.c_int), ("swept", C.c_int), ("resolved", C.c_int), ("reason", C.c_int),
                ("pass2_reruns", C.c_int), ("exact_redo_tiles", C.c_uint32), ("band_log2", C.c_int), ("reserved", C.c_int)]


ANALYZE_TWO_PASS = 1
ANALYZE_SPOIL_GUESS = 2


class IngestTiming(C.Structure):
    _fields_ = [("total_s", C.c_double), ("setup_s", C.c_double), ("read_s", C.c_double),
                ("buffer_wait_s", C.c_double), ("issue_s", C.c_double), ("drain_s", C.c_double),
                ("bytes", C.c_uint64), ("chunks", C.c_uint64), ("reader_threads", C.c_int), ("resident", C.c_int),
                ("o_direct", C.c_int), ("numa_bound", C.c_int), ("io_uring", C.c_int), ("file_passes", C.c_int)]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class Tuning(C.Structure):
    """papr_hip_tuning; *_variant fields hold (variant id + 1), 0 = built-in default."""
    _fields_ = [("stats_blocks", C.c_int), ("stats_variant", C.c_int), ("stats_map", C.c_int),
                ("ccdf_blocks", C.c_int), ("ccdf_variant", C.c_int), ("ccdf_map", C.c_int),
                ("nontemporal", C.c_int), ("hist_copies", C.c_int), ("flags", C.c_int),
                ("sweep_blocks", C.c_int), ("sweep_variant", C.c_int), ("sweep_map", C.c_int),
                ("sweep_band_log2", C.c_int), ("estimate_ratio", C.c_int), ("reserved", C.c_int)]


class PaprError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str):
        super().__init__(f"{what} failed with code {code}: {detail}")
        self.code = code


_lib: Optional[C.CDLL] = None
SWEEP_VARIANT, SWEEP_EXACT_VARIANT = 111, 131   # the product's one-sweep kernel forms (csrc/papr_kernels.h)


def sweep_variant_built(variant: int) -> bool:
    """Does this build of libpaprhip.so carry the one-sweep kernel form with that id?  (The product has two; the
    laboratory's forms — csrc/measure/papr_sweep_lab.hip — only exist in a `make MEASURE=1` build.)"""
    return bool(lib().papr_hip_sweep_variant_built(int(variant)))



def lib() -> C.CDLL:
    """Load libpaprhip.so (built in-tree by `make lib` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `make lib` or __graft_entry__.build(); "
            "there is no fallback implementation")
    # One process, one ROCm runtime: PyTorch brings its own libamdhip64 / libhsa-runtime64, and the copy that is mapped
    # first is the one both sides end up sharing — a second HSA runtime in the process finds no GPU ("No HIP GPUs are
    # available" from whoever came second).  So torch, where it is installed, goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.papr_hip_abi_version.restype = i32
    L.papr_hip_device_count.restype = i32
    L.papr_hip_open.argtypes = [C.POINTER(vp), i32]
    L.papr_hip_close.argtypes = [vp]
    L.papr_hip_close.restype = None
    L.papr_hip_last_error.argtypes = [vp]
    L.papr_hip_last_error.restype = C.c_char_p
    L.papr_hip_device_name.argtypes = [vp, C.c_char_p, i32]
    L.papr_hip_set_tuning.argtypes = [vp, C.POINTER(Tuning)]
    L.papr_hip_set_timing.argtypes = [vp, i32]
    L.papr_hip_sweep_variant_built.argtypes = [i32]
    L.papr_hip_sweep_variant_built.restype = i32
    L.papr_hip_get_timing.argtypes = [vp, C.POINTER(Timing)]
    L.papr_hip_get_timing_launches.argtypes = [vp, i32, vp, i32]
    L.papr_hip_get_timing_launches.restype = i32
    L.papr_file_samples.argtypes = [C.c_char_p, C.POINTER(u64)]
    L.papr_hip_load_file.argtypes = [vp, C.c_char_p, u64, u64]
    L.papr_hip_load_stream.argtypes = [vp, i32, C.POINTER(u64)]
    L.papr_hip_load_stream.restype = i32
    L.papr_hip_stream_stats.argtypes = [vp, i32, C.POINTER(Stats), C.POINTER(i32), C.POINTER(u64)]
    L.papr_hip_stream_stats.restype = i32
    L.papr_exact_chain_continue.argtypes = [C.POINTER(C.c_double), C.POINTER(vp), C.POINTER(C.c_size_t), i32]
    L.papr_exact_chain_continue.restype = i32
    L.papr_hip_get_ingest_timing.argtypes = [vp, C.POINTER(IngestTiming)]
    L.papr_hip_get_ingest_timing.restype = i32
    L.papr_hip_upload.argtypes = [vp, vp, u64, u64]
    L.papr_hip_adopt.argtypes = [vp, vp, u64, u64]
    L.papr_hip_generate.argtypes = [vp, C.POINTER(SynthSpec), u64, u64]
    L.papr_hip_download.argtypes = [vp, vp, u64, u64]
    L.papr_hip_stats.argtypes = [vp, C.POINTER(Stats)]
    L.papr_stats_init.argtypes = [C.POINTER(Stats)]
    L.papr_stats_init.restype = None
    L.papr_stats_merge.argtypes = [C.POINTER(Stats), C.POINTER(Stats)]
    L.papr_stats_merge.restype = None
    L.papr_levels.argtypes = [C.POINTER(Stats), i32, C.POINTER(C.c_double), C.POINTER(C.c_float), vp, i32]
    L.papr_hip_ccdf.argtypes = [vp, vp, i32, vp]
    L.papr_hip_set_exact.argtypes = [vp, i32]
    L.papr_hip_exact_program.argtypes = [vp, C.c_double, u64, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.papr_hip_ccdf_exact.argtypes = [vp, vp, i32, vp, C.c_double, u64, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.papr_exact_chain.argtypes = [C.POINTER(vp), C.POINTER(C.c_size_t), i32, C.POINTER(C.c_double)]
    L.papr_hip_estimate.argtypes = [vp, C.POINTER(Stats)]
    L.papr_guess_levels.argtypes = [C.POINTER(Stats), i32, C.c_double, vp, i32]
    L.papr_guess_levels.restype = i32
    L.papr_hip_stats_sweep.argtypes = [vp, vp, i32, C.POINTER(Stats)]
    L.papr_hip_estimate_file.argtypes = [vp, C.c_char_p, u64, u64, C.POINTER(Stats)]
    L.papr_level_key.argtypes = [C.c_float]
    L.papr_level_key.restype = C.c_uint32
    L.papr_sweep_bands.argtypes = [vp, i32, i32, vp, vp]
    L.papr_sweep_bands.restype = i32
    L.papr_sweep_resolve.argtypes = [vp, i32, i32, vp, vp, i32, vp, vp]
    L.papr_sweep_resolve.restype = i32
    L.papr_hip_shard_fits.argtypes = [vp, u64]
    L.papr_hip_shard_fits.restype = i32
    L.papr_hip_load_file_sweep.argtypes = [vp, C.c_char_p, u64, u64, vp, i32]
    L.papr_hip_get_sweep_info.argtypes = [vp, C.POINTER(SweepInfo)]
    if hasattr(L, "papr_hip_get_wg_finish"):   # (an older library under PAPR_LIB_PATH, side by side with this one: tools/lib_abn.sh)
        L.papr_hip_get_wg_finish.argtypes = [vp, vp, i32]
        L.papr_hip_get_wg_finish.restype = i32
    L.papr_hip_analyze.argtypes = [vp, vp, i32, C.c_uint, C.POINTER(Result), vp, vp, i32]
    L.papr_hip_analyze.restype = i32
    L.papr_sweep_band_for.argtypes = [C.POINTER(Stats)]
    L.papr_sweep_band_for.restype = i32
    L.papr_hip_set_band.argtypes = [vp, i32]
    L.papr_hip_set_band.restype = i32
    L.papr_hip_set_exact_hint.argtypes = [vp, C.c_double]
    L.papr_hip_set_exact_hint.restype = i32
    for name in ("papr_hip_set_exact", "papr_hip_exact_program", "papr_hip_ccdf_exact", "papr_exact_chain",
                 "papr_hip_estimate", "papr_hip_stats_sweep", "papr_hip_get_sweep_info", "papr_hip_estimate_file",
                 "papr_hip_load_file_sweep"):
        getattr(L, name).restype = i32
    for name in ("papr_hip_open", "papr_hip_device_name", "papr_hip_set_tuning", "papr_hip_set_timing",
                 "papr_hip_get_timing", "papr_file_samples", "papr_hip_load_file", "papr_hip_get_ingest_timing", "papr_hip_upload",
                 "papr_hip_adopt", "papr_hip_generate", "papr_hip_download", "papr_hip_stats",
                 "papr_levels", "papr_hip_ccdf"):
        getattr(L, name).restype = i32
    _lib = L
    return L


# ---- GPU-free helpers (host side of the ABI) ---------------------------------

def stats_merge(parts: Sequence[Stats]) -> Stats:
    """Fold shard records in file order (papr_stats_merge)."""
    L = lib()
    acc = Stats()
    L.papr_stats_init(C.byref(acc))
    for p in parts:
        L.papr_stats_merge(C.byref(acc), C.byref(p))
    return acc


def levels(total: Stats, graph: bool):
    """(mean, papr, float32 level table) exactly as the reference's host scalars."""
    L = lib()
    mean, papr = C.c_double(), C.c_float()
    n = L.papr_levels(C.byref(total), int(graph), C.byref(mean), C.byref(papr), None, 0)
    table = np.zeros(max(n, 0), dtype=np.float32)
    if n > 0:
        L.papr_levels(C.byref(total), int(graph), None, None, table.ctypes.data_as(C.c_void_p), n)
    return mean.value, papr.value, table


def guess_levels(est_total: Stats, graph: bool, max_db: float = None) -> np.ndarray:
    """Speculative level table for the one-sweep mode (papr_guess_levels); default reach: 60 dB
    above the estimated mean (48 dB for the 0.1 dB table, whose band LUT is four times denser)."""
    if max_db is None:
        max_db = 48.0 if graph else 60.0
    buf = np.zeros(MAX_LEVELS, dtype=np.float32)
    n = lib().papr_guess_levels(C.byref(est_total), int(bool(graph)), float(max_db), buf.ctypes.data_as(C.c_void_p),
                                MAX_LEVELS)
    return buf[:n].copy()


def band_for(est_total: Stats) -> int:
    """Band half-width (log2) the estimate's own standard error asks for (papr_sweep_band_for)."""
    return int(lib().papr_sweep_band_for(C.byref(est_total)))


def sweep_bands(guess_table: np.ndarray, band_log2: int = 14):
    """(keys, edges) of the bands around the guessed thresholds (papr_sweep_bands), or None without a band form."""
    lv = np.ascontiguousarray(guess_table, dtype=np.float32)
    keys = np.zeros(max(lv.size, 1), dtype=np.uint32)
    edges = np.zeros(2 * max(lv.size, 1), dtype=np.uint32)
    m = lib().papr_sweep_bands(lv.ctypes.data_as(C.c_void_p), lv.size, band_log2, keys.ctypes.data_as(C.c_void_p),
                               edges.ctypes.data_as(C.c_void_p))
    return (keys[:m].copy(), edges[:2 * m].copy()) if m > 0 else None


def sweep_resolve(guess_keys: np.ndarray, band_log2: int, above_band: np.ndarray, level_table: np.ndarray,
                  stash_above: np.ndarray):
    """counts_above for the true table from a sweep's leftovers (papr_sweep_resolve), or None if out of band."""
    gk = np.ascontiguousarray(guess_keys, dtype=np.uint32)
    ab = np.ascontiguousarray(above_band, dtype=np.uint64)
    lv = np.ascontiguousarray(level_table, dtype=np.float32)
    sa = np.ascontiguousarray(stash_above, dtype=np.uint64)
    out = np.zeros(lv.size, dtype=np.uint64)
    ok = lib().papr_sweep_resolve(gk.ctypes.data_as(C.c_void_p), gk.size, band_log2, ab.ctypes.data_as(C.c_void_p),
                                  lv.ctypes.data_as(C.c_void_p), lv.size, sa.ctypes.data_as(C.c_void_p),
                                  out.ctypes.data_as(C.c_void_p))
    return out if ok else None


def exact_chain(programs: Sequence[bytes]) -> float:
    """papr_exact_chain: replay the shards' sum programs (file order) into the
    reference's sequential double sum."""
    L = lib()
    n = len(programs)
    bufs = [C.create_string_buffer(p, len(p)) for p in programs]
    ptrs = (C.c_void_p * max(n, 1))(*[C.cast(b, C.c_void_p) for b in bufs])
    sizes = (C.c_size_t * max(n, 1))(*[len(p) for p in programs])
    out = C.c_double()
    rc = L.papr_exact_chain(ptrs, sizes, n, C.byref(out))
    if rc:
        raise PaprError(rc, "papr_exact_chain", "malformed program" if rc == -3 else "exact-sum invariant violated")
    return out.value


def exact_chain_continue(start: float, programs: Sequence[bytes]) -> float:
    """papr_exact_chain_continue: the same replay from the accumulator `start` the earlier programs left."""
    L = lib()
    n = len(programs)
    bufs = [C.create_string_buffer(p, len(p)) for p in programs]
    ptrs = (C.c_void_p * max(n, 1))(*[C.cast(b, C.c_void_p) for b in bufs])
    sizes = (C.c_size_t * max(n, 1))(*[len(p) for p in programs])
    acc = C.c_double(start)
    rc = L.papr_exact_chain_continue(C.byref(acc), ptrs, sizes, n)
    if rc:
        raise PaprError(rc, "papr_exact_chain_continue", "malformed program" if rc == -3 else "exact-sum invariant violated")
    return acc.value


def file_samples(path: str) -> int:
    n = C.c_uint64()
    rc = lib().papr_file_samples(os.fsencode(path), C.byref(n))
    if rc:
        raise PaprError(rc, "papr_file_samples", path)
    return n.value


# ---- one GPU context ------------------------------------------------------------

class PaprHip:
    """One papr_hip_ctx: one GPU, one shard of the sample axis."""

    def __init__(self, device: int = 0):
        self._L = lib()
        self._ctx = C.c_void_p()
        rc = self._L.papr_hip_open(C.byref(self._ctx), device)
        if rc:
            detail = self._L.papr_hip_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise PaprError(rc, "papr_hip_open", detail)
        self.device = device
        self._keepalive = None

    def close(self):
        if self._ctx:
            self._L.papr_hip_close(self._ctx)
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

    def _chk_pos(self, rc: int, what: str) -> int:
        if rc < 0:
            self._chk(rc, what)
        return rc

    def _chk(self, rc: int, what: str):
        if rc:
            raise PaprError(rc, what, self._L.papr_hip_last_error(self._ctx).decode())

    @property
    def name(self) -> str:
        buf = C.create_string_buffer(160)
        self._chk(self._L.papr_hip_device_name(self._ctx, buf, 160), "papr_hip_device_name")
        return buf.value.decode()

    def set_tuning(self, blocks=0, variant=None, map=None, nontemporal=1, hist_copies=0, flags=0,
                   stats_blocks=None, stats_variant=None, stats_map=None,
                   ccdf_blocks=None, ccdf_variant=None, ccdf_map=None,
                   sweep_blocks=0, sweep_variant=None, sweep_map=None, sweep_band_log2=0, estimate_ratio=0):
        """`blocks` / `variant` / `map` apply to both passes unless the per-pass value is given;
        None (0 for blocks) keeps each pass's built-in default.  The sweep_* / estimate_ratio knobs
        belong to the one-sweep mode."""
        def pick(per_pass, both):
            v = both if per_pass is None else per_pass
            return 0 if v is None else v + 1
        t = Tuning(blocks if stats_blocks is None else stats_blocks, pick(stats_variant, variant),
                   pick(stats_map, map), blocks if ccdf_blocks is None else ccdf_blocks,
                   pick(ccdf_variant, variant), pick(ccdf_map, map), 1 if nontemporal else 2, hist_copies, flags,
                   sweep_blocks, pick(sweep_variant, None), pick(sweep_map, None), sweep_band_log2, estimate_ratio, 0)
        self._chk(self._L.papr_hip_set_tuning(self._ctx, C.byref(t)), "papr_hip_set_tuning")

    def set_timing(self, enabled):
        """False / 0 off, True / 1 every timed kernel, 2 only the kernels that read the shard (include/papr_hip.h)."""
        self._chk(self._L.papr_hip_set_timing(self._ctx, int(enabled)), "papr_hip_set_timing")

    def timing_launches(self, kind: int, cap: int = 4096) -> np.ndarray:
        """Durations (ms) of the timed launches of one class, in dispatch order (3 = the sweep kernel)."""
        buf = np.zeros(cap, dtype=np.float32)
        n = self._L.papr_hip_get_timing_launches(self._ctx, int(kind), buf.ctypes.data_as(C.c_void_p), cap)
        if n < 0:
            self._chk(n, "papr_hip_get_timing_launches")
        return buf[:min(n, cap)].copy()

    def timing(self) -> Timing:
        t = Timing()
        self._chk(self._L.papr_hip_get_timing(self._ctx, C.byref(t)), "papr_hip_get_timing")
        return t

    # shard residency
    def load_file(self, path: str, first_sample: int = 0, nsamples: int = NO_INDEX):
        self._chk(self._L.papr_hip_load_file(self._ctx, os.fsencode(path), first_sample, nsamples),
                  "papr_hip_load_file")

    def load_stream(self, fd: int) -> int:
        """papr_hip_load_stream: a pipe / FIFO / socket read once, to its end, into the (growing) shard; returns the samples."""
        n = C.c_uint64(0)
        self._chk(self._L.papr_hip_load_stream(self._ctx, fd, C.byref(n)), "papr_hip_load_stream")
        return int(n.value)

    def stream_stats(self, fd: int):
        """papr_hip_stream_stats: pass 1 over a pipe / FIFO / socket of any length, window by window; returns
        (the stream's Stats, whether .sum is the reference's sequential sum, windows reduced)."""
        st, ex, w = Stats(), C.c_int(0), C.c_uint64(0)
        self._chk(self._L.papr_hip_stream_stats(self._ctx, fd, C.byref(st), C.byref(ex), C.byref(w)), "papr_hip_stream_stats")
        return st, bool(ex.value), int(w.value)

    def estimate_file(self, path: str, first_sample: int = 0, nsamples: int = NO_INDEX) -> Stats:
        """papr_hip_estimate for a file range that is not loaded (yet)."""
        s = Stats()
        self._chk(self._L.papr_hip_estimate_file(self._ctx, os.fsencode(path), first_sample, nsamples, C.byref(s)),
                  "papr_hip_estimate_file")
        return s

    def load_file_sweep(self, path: str, guess_table: np.ndarray, first_sample: int = 0, nsamples: int = NO_INDEX):
        """load_file with pass 1 AND the banded pass 2 riding along with the ingest (one read of the file)."""
        lv = np.ascontiguousarray(guess_table, dtype=np.float32)
        self._chk(self._L.papr_hip_load_file_sweep(self._ctx, os.fsencode(path), first_sample, nsamples,
                                                   lv.ctypes.data_as(C.c_void_p), lv.size), "papr_hip_load_file_sweep")

    def wg_finish(self):
        """papr_hip_get_wg_finish: (us after the first one, XCD) for each workgroup of the last sweep launch."""
        t = np.zeros(4096, dtype=np.uint32)
        n = self._chk_pos(self._L.papr_hip_get_wg_finish(self._ctx, t.ctypes.data_as(C.c_void_p), t.size), "papr_hip_get_wg_finish")
        t = t[:min(n, t.size)]
        if t.size == 0:
            return np.zeros(0), np.zeros(0, dtype=np.uint32)
        xcd = t >> 28
        ticks = (t & 0x0FFFFFFF).astype(np.int64)
        d = (ticks - ticks.min()) % (1 << 28)
        if d.max() > (1 << 27):   # (the 28-bit counter wrapped inside the launch: measure from the largest gap)
            d = (ticks - ticks.max()) % (1 << 28)
        return d.astype(np.float64) / 100.0, xcd

    def wg_finish_us(self) -> np.ndarray:
        return self.wg_finish()[0]

    def shard_fits(self, nsamples: int) -> bool:
        return bool(self._chk_pos(self._L.papr_hip_shard_fits(self._ctx, nsamples), "papr_hip_shard_fits"))

    def ingest_timing(self) -> IngestTiming:
        t = IngestTiming()
        self._chk(self._L.papr_hip_get_ingest_timing(self._ctx, C.byref(t)), "papr_hip_get_ingest_timing")
        return t

    def upload(self, iq: np.ndarray, base_index: int = 0):
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        if iq.size % 2:
            raise ValueError("upload takes whole IQ pairs")
        self._chk(self._L.papr_hip_upload(self._ctx, iq.ctypes.data_as(C.c_void_p), iq.size // 2, base_index),
                  "papr_hip_upload")

    def adopt(self, device_ptr: int, nsamples: int, base_index: int = 0, keepalive=None):
        self._chk(self._L.papr_hip_adopt(self._ctx, C.c_void_p(device_ptr), nsamples, base_index), "papr_hip_adopt")
        self._keepalive = keepalive

    def generate(self, spec: SynthSpec, first_index: int, nsamples: int):
        self._chk(self._L.papr_hip_generate(self._ctx, C.byref(spec), first_index, nsamples), "papr_hip_generate")

    def download(self, first: int, nsamples: int) -> np.ndarray:
        out = np.empty(2 * nsamples, dtype=np.float32)
        self._chk(self._L.papr_hip_download(self._ctx, out.ctypes.data_as(C.c_void_p), first, nsamples),
                  "papr_hip_download")
        return out

    # bit-exact mean
    def set_exact(self, enabled: bool = True):
        self._chk(self._L.papr_hip_set_exact(self._ctx, int(enabled)), "papr_hip_set_exact")

    def set_band(self, band_log2: int = 0):
        """Half-width (log2 of bit patterns) of the threshold bands of the next sweeps; 0 = built-in default."""
        self._chk(self._L.papr_hip_set_band(self._ctx, int(band_log2)), "papr_hip_set_band")

    def set_exact_hint(self, estimated_sum_before_shard: float):
        """Exact one-read sweep: the estimated sum of everything before this shard (0 for the first)."""
        self._chk(self._L.papr_hip_set_exact_hint(self._ctx, float(estimated_sum_before_shard)), "papr_hip_set_exact_hint")

    def exact_program(self, before: float = 0.0, n_total: int = 0) -> bytes:
        """This shard's serialised sum program (needs set_exact(True) + stats() first)."""
        ptr, size = C.c_void_p(), C.c_size_t()
        self._chk(self._L.papr_hip_exact_program(self._ctx, before, n_total, C.byref(ptr), C.byref(size)),
                  "papr_hip_exact_program")
        return C.string_at(ptr, size.value)

    def ccdf_exact(self, level_table: np.ndarray, before: float = 0.0, n_total: int = 0):
        """Pass 2 and the sum program from ONE sweep over the samples: (counts, program)."""
        lv = np.ascontiguousarray(level_table, dtype=np.float32)
        out = np.zeros(lv.size, dtype=np.uint64)
        ptr, size = C.c_void_p(), C.c_size_t()
        self._chk(self._L.papr_hip_ccdf_exact(self._ctx, lv.ctypes.data_as(C.c_void_p), lv.size,
                                              out.ctypes.data_as(C.c_void_p), before, n_total, C.byref(ptr),
                                              C.byref(size)), "papr_hip_ccdf_exact")
        return out, C.string_at(ptr, size.value)

    # the two passes
    def stats(self) -> Stats:
        s = Stats()
        self._chk(self._L.papr_hip_stats(self._ctx, C.byref(s)), "papr_hip_stats")
        return s

    # one-sweep mode: both passes in one read (papr_sweep.hip)
    def estimate(self) -> Stats:
        """sum / n over a pseudo-random 1/64 sample of the shard (only those two fields are set)."""
        s = Stats()
        self._chk(self._L.papr_hip_estimate(self._ctx, C.byref(s)), "papr_hip_estimate")
        return s

    def stats_sweep(self, guess_table: np.ndarray) -> Stats:
        """papr_hip_stats + banded pass 2 around the guessed thresholds; a following ccdf() is then
        answered from the sweep whenever the true thresholds fall inside the bands."""
        lv = np.ascontiguousarray(guess_table, dtype=np.float32)
        s = Stats()
        self._chk(self._L.papr_hip_stats_sweep(self._ctx, lv.ctypes.data_as(C.c_void_p), lv.size, C.byref(s)),
                  "papr_hip_stats_sweep")
        return s

    def analyze(self, xch=None, graph: bool = False, two_pass: bool = False, spoil_guess: bool = False):
        """papr_hip_analyze: the whole result in one call — (Result, level table, counts above each level).
        `xch`: a dtv_utils_amd.exchange.Exchange (None: single shard)."""
        if not hasattr(self, "_an_levels"):
            self._an_levels = np.zeros(MAX_LEVELS, dtype=np.float32)
            self._an_counts = np.zeros(MAX_LEVELS, dtype=np.uint64)
        res = Result()
        self._chk(self._L.papr_hip_analyze(self._ctx, xch._x if xch is not None else None, int(bool(graph)),
                                           (ANALYZE_TWO_PASS if two_pass else 0) | (ANALYZE_SPOIL_GUESS if spoil_guess else 0),
                                           C.byref(res),
                                           self._an_levels.ctypes.data_as(C.c_void_p),
                                           self._an_counts.ctypes.data_as(C.c_void_p), MAX_LEVELS), "papr_hip_analyze")
        n = res.nlevels
        return res, self._an_levels[:n].copy(), self._an_counts[:n].copy()

    def sweep_info(self) -> SweepInfo:
        i = SweepInfo()
        self._chk(self._L.papr_hip_get_sweep_info(self._ctx, C.byref(i)), "papr_hip_get_sweep_info")
        return i

    def ccdf(self, level_table: np.ndarray) -> np.ndarray:
        lv = np.ascontiguousarray(level_table, dtype=np.float32)
        out = np.zeros(lv.size, dtype=np.uint64)
        self._chk(self._L.papr_hip_ccdf(self._ctx, lv.ctypes.data_as(C.c_void_p), lv.size,
                                        out.ctypes.data_as(C.c_void_p)), "papr_hip_ccdf")
        return out


def format_report(total: Stats, mean: float, papr: float, counts: np.ndarray, graph: bool) -> str:
    """The reference's stdout (papr.c:132-135,154-161 / 186-190) from ABI results.

    Only used by tests to compare ABI-level results with golden stdout; the
    shipped formatter is host/papr_main.c.  Uses C printf via ctypes so the
    digits come from the same libc.
    """
    libc = C.CDLL(None)
    buf = C.create_string_buffer(256)

    def fmt(f: bytes, *args):
        libc.snprintf(buf, 256, f, *args)
        return buf.value.decode()

    n32 = np.float32(np.int64(total.n))
    with np.errstate(all="ignore"):
        pct = [float(np.float32(np.float32(np.int64(c)) / n32)) * 100.0 for c in counts]
    d = C.c_double
    out = []
    if graph:
        for p in pct:
            out.append(fmt(b"%0.8f\n", d(p)))
        return "".join(out)
    out.append(fmt(b"Peak magnitude = %f\n", d(float(np.sqrt(np.float64(total.peak))))))
    out.append(fmt(b"average power = %lf, peak power = %f @ %lld\n\n", d(mean), d(total.peak),
                   C.c_longlong(total.peak_idx * 8)))
    out.append(fmt(b"Maximum PAPR = %f\n", d(papr)))
    for j, p in enumerate(pct):
        out.append(fmt(b"percentage above %d dB = %0.8f\n", C.c_int(j), d(p)))
    out.append("\n")
    out.append(fmt(b"peak real positive = %f, peak imaginary positive = %f\n", d(total.re_pos), d(total.im_pos)))
    out.append(fmt(b"peak real negative = %f, peak imaginary negative = %f\n\n", d(total.re_neg), d(total.im_neg)))
    out.append(fmt(b"peak real positive @ %lld, peak imaginary positive @ %lld\n",
                   C.c_longlong(total.re_pos_idx * 8), C.c_longlong(total.im_pos_idx * 8 + 1)))
    out.append(fmt(b"peak real negative @ %lld, peak imaginary negative @ %lld\n",
                   C.c_longlong(total.re_neg_idx * 8), C.c_longlong(total.im_neg_idx * 8 + 1)))
    return "".join(out)
