"""GPU-free checks of the drop-in boundary: libpaprhip.so loads and exports
every symbol include/papr_hip.h declares, the host-side ABI functions
(papr_stats_merge, papr_levels, papr_file_samples) agree with the oracle, and
both the library and bin/papr fail loudly — no CPU fallback — without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_names, golden_path

try:
    import torch
    HAVE_GPU = torch.cuda.is_available()
except Exception:  # pragma: no cover
    HAVE_GPU = False


HEADERS = ("papr_hip.h", "papr_exchange.h", "papr_hip_measure.h")   # the papr path; the exchange between shards; measurement


def declared_functions(headers=HEADERS):
    out = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(papr_[a-z_]+)\s*\(", text))
    return sorted(out)


def test_the_product_header_is_the_papr_path_only():
    """include/papr_hip.h is what a caller of the papr path binds: no tuning, timing or probe entry
    points — those live in papr_hip_measure.h."""
    product = declared_functions(("papr_hip.h",))
    assert len(product) <= 28, product   # (28: papr_hip_stream_stats + papr_exact_chain_continue, a stream of any length)
    for name in ("papr_hip_set_tuning", "papr_hip_set_timing", "papr_hip_get_timing", "papr_hip_generate", "papr_hip_adopt",
                 "papr_sweep_bands", "papr_hip_get_sweep_info"):
        assert name not in product and name in declared_functions(("papr_hip_measure.h",))


def test_header_and_binding_list_agree(pkg):
    from dtv_utils_amd import exchange
    assert declared_functions() == sorted(pkg.ABI_SYMBOLS + exchange.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    for name in declared_functions():
        assert hasattr(L, name), f"libpaprhip.so does not export {name}"
    assert L.papr_hip_abi_version() == 6
    # struct layouts the binding assumes
    assert C.sizeof(pkg.Stats) == 96
    assert C.sizeof(pkg.SynthSpec) == 16 + 16 * 8


def test_no_torch_or_hip_types_in_the_abi():
    for h in HEADERS + ("ts_hip.h",):
        text = open(os.path.join(ROOT, "include", h)).read()
        code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        assert "hip/" not in code and "torch" not in code and "hipStream" not in code, h


def test_library_does_not_link_the_oracle(pkg):
    out = subprocess.run(["ldd", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "papr_oracle" not in syms
    out = subprocess.run(["ldd", pkg.CLI_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "libpaprhip" in out


def _stats_from_oracle(pkg, r, base):
    s = pkg.Stats()
    pkg.lib().papr_stats_init(C.byref(s))
    s.sum, s.n = r["sum"], r["n"]
    for k in ("peak", "re_pos", "re_neg", "im_pos", "im_neg"):
        v = r[k]
        setattr(s, k, v)
        setattr(s, k + "_idx", (r[k + "_idx"] + base) if v != 0 else 0)
    return s


@pytest.mark.parametrize("name", ["g1m", "ties", "spike20k", "tiny", "zeros", "k8"])
def test_stats_merge_of_ordered_shards_equals_whole(pkg, orc, name):
    """papr_stats_merge is what makes sharding legal: folding per-shard records
    in file order must reproduce the sequential trackers (first index wins)."""
    floats = np.fromfile(golden_path(name), dtype=np.float32)
    n = floats.size // 2
    whole = orc.run_mem(floats, False)
    rng = np.random.default_rng(7)
    for nshards in (1, 2, 3, 8):
        cuts = sorted(rng.integers(0, n + 1, size=nshards - 1).tolist())
        bounds = [0, *cuts, n]
        parts = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            r = orc.run_mem(floats[2 * a:2 * b], False)
            parts.append(_stats_from_oracle(pkg, r, a))
        m = pkg.stats_merge(parts)
        assert m.n == whole["n"]
        for k in ("peak", "re_pos", "re_neg", "im_pos", "im_neg"):
            assert getattr(m, k) == whole[k], (name, nshards, k)
            assert getattr(m, k + "_idx") == whole[k + "_idx"], (name, nshards, k)
        assert m.sum == pytest.approx(whole["sum"], rel=1e-12, abs=0)


def test_stats_merge_nan_is_sticky_and_first_wins(pkg):
    L = pkg.lib()
    a, b, c = pkg.Stats(), pkg.Stats(), pkg.Stats()
    for s in (a, b, c):
        L.papr_stats_init(C.byref(s))
    a.sum, a.n = 10.0, 5
    b.sum, b.n, b.flags, b.nan_first_idx, b.nan_first_neg = float("nan"), 5, pkg.FLAG_NAN, 7, 1
    b.sum = -abs(b.sum)  # keep whatever sign; only stickiness is checked via flags below
    c.sum, c.n, c.flags, c.nan_first_idx, c.nan_first_neg = float("nan"), 5, pkg.FLAG_NAN, 12, 0
    m = pkg.stats_merge([a, b, c])
    assert m.n == 15 and m.flags & pkg.FLAG_NAN and m.nan_first_idx == 7 and m.nan_first_neg == 1
    assert m.sum != m.sum


def test_levels_match_oracle_host_scalars(pkg, orc):
    """papr_levels vs the oracle's restatement of papr.c:131-141 / 164-173 on a
    sweep of (sum, n, peak), including the degenerate cases of SURVEY A12."""
    rng = np.random.default_rng(11)
    cases = [(39.5, 8, 25.0), (20.32, 20, 5.0), (9.0, 1, 9.0), (0.0, 0, 0.0), (0.0, 1000, 0.0),
             (float("inf"), 10, float("inf")), (float("nan"), 10, 3.0), (1.0e-30, 7, 1.0e-31),
             (1789569.7, 1342177, 1364.3789)]
    for _ in range(200):
        n = int(rng.integers(1, 10**9))
        mean = float(10 ** rng.uniform(-12, 6))
        peak = float(np.float32(mean * 10 ** rng.uniform(0, 4.5)))
        cases.append((mean * n, n, peak))
    for s, n, peak in cases:
        st = pkg.Stats()
        pkg.lib().papr_stats_init(C.byref(st))
        st.sum, st.n, st.peak = s, n, peak
        for graph in (False, True):
            mean, papr, table = pkg.levels(st, graph)
            o_mean, o_papr, o_table = orc.levels_from(s, n, peak, graph)
            assert (mean == o_mean) or (mean != mean and o_mean != o_mean)
            assert (papr == o_papr) or (papr != papr and o_papr != o_papr)
            assert np.array_equal(table, o_table), (s, n, peak, graph)


def test_levels_nan_sign_is_preserved(pkg):
    st = pkg.Stats()
    pkg.lib().papr_stats_init(C.byref(st))
    st.n, st.peak = 10, 1.0
    for neg in (False, True):
        st.sum = np.copysign(np.nan, -1.0 if neg else 1.0)
        mean, papr, table = pkg.levels(st, False)
        assert np.signbit(mean) == neg and np.signbit(papr) == neg and table.size == 0


@pytest.mark.parametrize("name", golden_names())
def test_file_samples(pkg, orc, name):
    assert pkg.file_samples(golden_path(name)) == orc.run_file(golden_path(name), False)["n"]


def test_file_samples_errors(pkg):
    with pytest.raises(pkg.PaprError):
        pkg.file_samples("/nonexistent/x.cfile")
    with pytest.raises(pkg.PaprError):
        pkg.file_samples("/tmp")


def test_cli_usage_and_open_errors_need_no_gpu(pkg):
    """These exits happen before any GPU work, exactly as papr.c:53-98."""
    usage = b"usage: papr -g <infile>\nOptions:\n\tg = graph suitable output\n"
    for args in ([], ["a", "b"], ["a", "b", "c"]):
        p = subprocess.run([pkg.CLI_PATH, *args], capture_output=True)
        assert (p.returncode, p.stdout, p.stderr) == (255, b"", usage)
    p = subprocess.run([pkg.CLI_PATH, "/nonexistent/file.cfile"], capture_output=True)
    assert (p.returncode, p.stdout, p.stderr) == (255, b"", b"Cannot open bitstream file </nonexistent/file.cfile>\n")
    p = subprocess.run([pkg.CLI_PATH, "-g", "/nonexistent/file.cfile"], capture_output=True)
    assert p.returncode == 255 and p.stderr == b"Cannot open bitstream file </nonexistent/file.cfile>\n"


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_gpu(pkg):
    """No CPU fallback anywhere in the product path."""
    with pytest.raises(pkg.PaprError) as e:
        pkg.PaprHip(0)
    assert e.value.code == -1
    p = subprocess.run([pkg.CLI_PATH, golden_path("k8")], capture_output=True)
    assert p.returncode == 254 and p.stdout == b"" and b"no usable GPU" in p.stderr


def test_guess_levels_matches_levels_where_they_overlap(pkg):
    """host helper: the guess table is papr_levels' table for the same mean, carried on past the peak"""
    st = pkg.Stats()
    st.sum, st.n, st.peak = 123456.789, 100000, 50.0
    for graph in (False, True):
        mean, papr, table = pkg.levels(st, graph)
        guess = pkg.guess_levels(st, graph)
        assert guess.size > table.size and np.array_equal(guess[:table.size], table)
        assert np.all(np.diff(guess) > 0)
    st.n = 0
    assert pkg.guess_levels(st, False).size == 0
