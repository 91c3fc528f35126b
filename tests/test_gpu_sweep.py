"""One-sweep mode (papr_sweep.hip): pass 1 and pass 2 in ONE read of the shard.

The sweep speculates on the level table (mean estimated from a 1/64 sample, thresholds widened into
bands, in-band powers stashed) — so the tests pin down that speculation can never change a result:
every count equals the oracle's `power > level` count, whether the sweep resolved the table from its
stash or fell back to reading the shard again, and its pass-1 record equals papr_hip_stats'."""
import os

import numpy as np
import pytest

from test_gpu_parity import TRACKERS, check_stats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(pkg):
    g = pkg.PaprHip(0)
    yield g
    g.set_tuning()
    g.close()


def one_sweep(pkg, g, graph, guess=None):
    """estimate -> guess table -> sweep -> true table -> counts; returns (stats, table, counts, info)"""
    if guess is None:
        guess = pkg.guess_levels(g.estimate(), graph)
    st = g.stats_sweep(guess)
    swept = g.sweep_info().swept
    mean, papr, table = pkg.levels(st, graph)
    counts = g.ccdf(table)
    info = g.sweep_info()
    assert info.swept == swept
    return st, table, counts, info


def same_record(a, b):
    assert a.n == b.n and a.flags == b.flags
    for k in TRACKERS:
        assert getattr(a, k) == getattr(b, k), k
        assert getattr(a, k + "_idx") == getattr(b, k + "_idx"), k
    if a.sum != a.sum:
        assert b.sum != b.sum and np.signbit(a.sum) == np.signbit(b.sum)
        assert a.nan_first_idx == b.nan_first_idx and a.nan_first_neg == b.nan_first_neg
    elif np.isinf(b.sum):
        assert a.sum == b.sum
    else:
        assert abs(a.sum - b.sum) <= 1e-12 * abs(b.sum)


SIZES = [1, 2, 63, 64, 255, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 12289, 100003, 1048576 + 5, 5 * 1048576 + 4099]


@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
@pytest.mark.parametrize("n", SIZES)
def test_sweep_equals_oracle_sizes(pkg, orc, gpu, n, graph):
    rng = np.random.default_rng(1000 + n)
    iq = rng.standard_normal(2 * n).astype(np.float32)
    gpu.upload(iq)
    st, table, counts, info = one_sweep(pkg, gpu, graph)
    check_stats(st, orc.run_mem(iq, graph))
    assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table))
    same_record(st, gpu.stats())
    if n >= 2048 and table.size:   # enough samples for an estimate: the sweep must have run ...
        assert info.swept == 1, info.as_dict()
        assert 0 < info.stash_samples <= n
    if n >= 100003:                # ... and with a decent estimate the speculation must have paid off
        assert info.resolved == 1 and info.stash_samples <= n // 4, info.as_dict()


# The laboratory's kernel forms (csrc/measure/, `make MEASURE=1`) are collected only when the library under test IS that
# build (PAPR_LIB_PATH=build_measure/libpaprhip.so): the product build does not carry them.
LAB = "build_measure" in os.environ.get("PAPR_LIB_PATH", "")
LAB_GEOMETRIES = [(40, 0, 0), (40, 7, 0), (40, 1, 0), (40, 700, 2), (40, 300, 1),
                  (1, 96, 1), (1, 1000, 2), (4, 7, 0), (4, 1, 0), (4, 0, 2), (4, 0, 0), (8, 0, 0),
                  (8, 64, 2), (13, 0, 0), (13, 1536, 2), (13, 300, 1), (14, 0, 0),
                  (14, 8, 2), (20, 0, 0), (20, 96, 1), (24, 0, 0), (24, 7, 2),
                  (32, 0, 0), (32, 3, 0), (41, 0, 0), (41, 100, 0)] if LAB else []


@pytest.mark.parametrize("tune", [dict(sweep_variant=v, sweep_blocks=b, sweep_map=m)
                                  for v, b, m in [(111, 0, 0), (111, 7, 0), (111, 1, 0), (111, 700, 2), (111, 300, 1)] + LAB_GEOMETRIES] +
                         [dict(sweep_band_log2=b) for b in (13, 15, 16, 17)] +
                         [dict(estimate_ratio=r) for r in (1, 7, 1000)] + [dict(hist_copies=1), dict(hist_copies=8)],
                         ids=str)
def test_sweep_geometries_agree(pkg, orc, gpu, tune):
    if "sweep_variant" in tune and not pkg.sweep_variant_built(tune["sweep_variant"]):
        pytest.skip("a laboratory kernel form: built by `make MEASURE=1` only")
    n = 3 * 1048576 + 4099
    gpu.generate(pkg.SynthSpec.spike(n, seed=78), 0, n)
    iq = gpu.download(0, n)
    ref = orc.run_mem(iq, True)
    try:
        gpu.set_tuning(**tune)
        for graph in (False, True):
            st, table, counts, info = one_sweep(pkg, gpu, graph)
            check_stats(st, ref)
            assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table))
            if tune.get("sweep_band_log2", 14) >= 14 or 20 <= tune.get("sweep_variant", 111) < 40 or tune.get("sweep_variant", 111) == 41:   # (narrower bands need a finer one-edge LUT than fits: plain pass 1)
                assert info.swept == 1 and info.resolved == 1, info.as_dict()
    finally:
        gpu.set_tuning()


def test_sweep_first_index_wins(pkg, orc, gpu):
    """equal extremes planted in different lanes, waves, workgroups and loop iterations, plus the
    sub-tile remainder: the sweep's per-tile trackers must still report the FIRST occurrence"""
    n = 6 * 1048576 + 777
    rng = np.random.default_rng(5)
    iq = (rng.standard_normal(2 * n) * 0.25).astype(np.float32)
    spots = [n - 3, 5 * 1048576 + 11, 4096 * 300 + 2049, 4096 * 300 + 1, 777777, 4097, 64, 63]
    for s in spots:
        iq[2 * s] = 9.5
        iq[2 * s + 1] = -9.5
    gpu.upload(iq)
    for tune in (dict(), dict(sweep_variant=111, sweep_blocks=3), dict(sweep_variant=40, sweep_blocks=3), dict(sweep_variant=4),
                 dict(sweep_variant=1, sweep_blocks=3), dict(sweep_variant=8, sweep_map=2), dict(sweep_variant=20),
                 dict(sweep_variant=32), dict(sweep_variant=41, sweep_blocks=5)):
        if "sweep_variant" in tune and not pkg.sweep_variant_built(tune["sweep_variant"]):
            continue   # (a laboratory form: `make MEASURE=1` builds)
        gpu.set_tuning(**tune)
        st, table, counts, info = one_sweep(pkg, gpu, False)
        gpu.set_tuning()
        assert st.peak_idx == 63 and st.re_pos_idx == 63 and st.im_neg_idx == 63
        check_stats(st, orc.run_mem(iq, False))
        assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table))


def test_sweep_bad_guess_falls_back(pkg, orc, gpu):
    """a guess that is off by 3 % puts the true thresholds outside the bands: the counts are still the oracle's"""
    n = 2 * 1048576 + 33
    rng = np.random.default_rng(11)
    iq = rng.standard_normal(2 * n).astype(np.float32)
    gpu.upload(iq)
    est = gpu.estimate()
    for factor, want_resolved in ((1.03, 0), (0.5, 0), (1.0002, 1)):
        bad = pkg.Stats.from_bytes(est.to_bytes())
        bad.sum = est.sum * factor
        for graph in (False, True):
            st, table, counts, info = one_sweep(pkg, gpu, graph, pkg.guess_levels(bad, graph))
            assert info.swept == 1 and info.resolved == want_resolved, info.as_dict()
            if not want_resolved:
                assert info.as_dict()["reason"] == "out of band"
            assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table))


def test_sweep_guess_tables_without_band_form(pkg, orc, gpu):
    n = 300007
    rng = np.random.default_rng(12)
    iq = rng.standard_normal(2 * n).astype(np.float32)
    gpu.upload(iq)
    ref = orc.run_mem(iq, False)
    for guess in (np.zeros(0, np.float32), np.array([np.nan], np.float32), np.array([0.0, 1.0], np.float32),
                  np.array([1e-42, 2.0], np.float32), np.array([2.0, 2.0001], np.float32),
                  np.array([np.inf], np.float32), np.array([-1.0, 3e38], np.float32)):
        st = gpu.stats_sweep(guess)
        info = gpu.sweep_info()
        assert info.swept == 0 and info.as_dict()["reason"] == "no band form", (guess, info.as_dict())
        check_stats(st, ref)
        mean, papr, table = pkg.levels(st, False)
        assert np.array_equal(gpu.ccdf(table).astype(np.int64), orc.count_mem(iq, table))
    # duplicated and unsorted guesses are fine
    st = gpu.stats_sweep(np.array([8.0, 2.0, 2.0, 4.0], np.float32))
    assert gpu.sweep_info().swept == 1
    tab = np.array([4.0001, 1.9999, 8.0, 2.0], np.float32)
    assert np.array_equal(gpu.ccdf(tab).astype(np.int64), orc.count_mem(iq, tab))
    assert gpu.sweep_info().resolved == 1
    # a table that reaches outside the bands is answered by reading the shard
    tab = np.array([2.0, 3.0], np.float32)
    assert np.array_equal(gpu.ccdf(tab).astype(np.int64), orc.count_mem(iq, tab))
    assert gpu.sweep_info().resolved == 0


def test_sweep_constant_envelope_overflows_the_stash(pkg, orc, gpu):
    """every sample has the mean power: all of them land in the 0 dB band, far more than the stash holds"""
    n = 1048576 + 9
    ph = np.random.default_rng(13).uniform(0, 2 * np.pi, n)
    iq = np.empty(2 * n, np.float32)
    iq[0::2] = np.cos(ph) * 3
    iq[1::2] = np.sin(ph) * 3
    gpu.upload(iq)
    for graph in (False, True):
        st, table, counts, info = one_sweep(pkg, gpu, graph)
        assert info.swept == 1 and info.resolved == 0 and info.as_dict()["reason"] == "stash full", info.as_dict()
        # (the workgroups give up once most of what they fold is in band — papr_sweep.hip sweep_give_up — and mark
        # their segments overfull: the number of stashed samples is no longer a count)
        assert info.stash_samples > info.stash_capacity
        check_stats(st, orc.run_mem(iq, graph))
        assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table))


def test_sweep_nan_and_inf_inputs(pkg, orc, gpu):
    n = 400000
    rng = np.random.default_rng(14)
    base = rng.standard_normal(2 * n).astype(np.float32)
    for kind in ("nan_i", "nan_q_neg", "inf", "both"):
        iq = base.copy()
        if kind in ("nan_i", "both"):
            iq[2 * 123457] = np.nan
        if kind == "nan_q_neg":
            iq[2 * 99 + 1] = -np.nan
        if kind in ("inf", "both"):
            iq[2 * 200001 + 1] = np.inf
        gpu.upload(iq)
        guess = pkg.guess_levels(gpu.estimate(), False)
        st = gpu.stats_sweep(guess if guess.size else np.array([2.0], np.float32))
        check_stats(st, orc.run_mem(iq, False))
        same_record(st, gpu.stats())
        tab = np.array([0.5, 2.0, 8.0, np.inf, 1e30], np.float32)
        assert np.array_equal(gpu.ccdf(tab).astype(np.int64), orc.count_mem(iq, tab))


def test_sweep_state_is_tied_to_the_shard(pkg, orc, gpu):
    rng = np.random.default_rng(15)
    a = rng.standard_normal(2 * 500000).astype(np.float32)
    b = (rng.standard_normal(2 * 500000) * 1.0005).astype(np.float32)
    gpu.upload(a)
    guess = pkg.guess_levels(gpu.estimate(), False)
    st = gpu.stats_sweep(guess)
    mean, papr, table = pkg.levels(st, False)
    assert np.array_equal(gpu.ccdf(table).astype(np.int64), orc.count_mem(a, table)) and gpu.sweep_info().resolved == 1
    # a second table against the same sweep
    assert np.array_equal(gpu.ccdf(table[:3]).astype(np.int64), orc.count_mem(a, table[:3]))
    assert gpu.sweep_info().resolved == 1
    # new samples: the old sweep must not be used
    gpu.upload(b)
    assert np.array_equal(gpu.ccdf(table).astype(np.int64), orc.count_mem(b, table)) and gpu.sweep_info().resolved == 0
    # plain stats() after a sweep drops it as well
    gpu.stats_sweep(guess)
    gpu.stats()
    assert np.array_equal(gpu.ccdf(table).astype(np.int64), orc.count_mem(b, table)) and gpu.sweep_info().resolved == 0
    # exact-sum mode: the sweep call is just pass 1
    gpu.set_exact(True)
    try:
        st = gpu.stats_sweep(guess)
        assert gpu.sweep_info().swept == 0 and gpu.sweep_info().as_dict()["reason"].startswith("exact mode")
        check_stats(st, orc.run_mem(b, False))
    finally:
        gpu.set_exact(False)


def test_sweep_shards_merge_like_the_two_pass_path(pkg, orc, gpu):
    """three shards of one stream, estimates merged before guessing, stats merged before the true table"""
    n = 3 * 700001
    rng = np.random.default_rng(16)
    iq = rng.standard_normal(2 * n).astype(np.float32)
    cuts = [0, 700001, 1500000, n]
    ests = []
    for s0, s1 in zip(cuts, cuts[1:]):
        gpu.upload(iq[2 * s0:2 * s1], base_index=s0)
        ests.append(gpu.estimate())
    for graph in (False, True):
        guess = pkg.guess_levels(pkg.stats_merge(ests), graph)
        ctxs = []
        try:
            parts = []
            for s0, s1 in zip(cuts, cuts[1:]):
                g = pkg.PaprHip(0)
                ctxs.append(g)
                g.upload(iq[2 * s0:2 * s1], base_index=s0)
                parts.append(g.stats_sweep(guess))
            tot = pkg.stats_merge(parts)
            check_stats(tot, orc.run_mem(iq, graph))
            mean, papr, table = pkg.levels(tot, graph)
            counts = sum(g.ccdf(table).astype(np.int64) for g in ctxs)
            assert all(g.sweep_info().resolved == 1 for g in ctxs)
            assert np.array_equal(counts, orc.count_mem(iq, table))
        finally:
            for g in ctxs:
                g.close()


def test_sweep_randomised(pkg, orc, gpu):
    """random sizes, scales, offsets and heavy tails; the sweep may or may not resolve — counts must not care"""
    rng = np.random.default_rng(2024)
    resolved = 0
    for case in range(60):
        n = int(rng.integers(1, 600000))
        kind = case % 4
        if kind == 0:
            iq = rng.standard_normal(2 * n) * 10.0 ** rng.uniform(-3, 3)
        elif kind == 1:
            iq = rng.standard_t(3, 2 * n) * 0.1
        elif kind == 2:
            iq = rng.standard_normal(2 * n) * np.repeat(rng.uniform(0.0, 2.0, n // 4096 + 1), 8192)[:2 * n]   # bursty
        else:
            iq = rng.uniform(-1, 1, 2 * n) + rng.uniform(-0.5, 0.5)
        iq = iq.astype(np.float32)
        base = int(rng.integers(0, 1 << 40))
        gpu.upload(iq, base_index=base)
        graph = bool(case & 4)
        st, table, counts, info = one_sweep(pkg, gpu, graph)
        check_stats(st, orc.run_mem(iq, graph), base=base)
        assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table)), (case, info.as_dict())
        resolved += info.resolved
    assert resolved >= 30


# ---- one-sweep INGEST: both passes ride along with the file on its way into HBM (or through it) ----------------

def test_file_ingest_sweep_resident_and_streamed(pkg, orc, tmp_path, monkeypatch):
    """papr_hip_estimate_file + papr_hip_load_file_sweep: odd float count + stray bytes, many chunks, resident and
    re-streamed (HBM budget too small) shards, one and two shards — counts without a second read of the file"""
    import subprocess
    from dtv_utils_amd import exchange
    n = 3 * 1048576 + 12345
    path = str(tmp_path / "sweep.cfile")
    subprocess.check_call([orc.MKCFILE, path, str(n), "--spike", "--extra-floats", "1", "--extra-bytes", "2"])
    monkeypatch.setenv("PAPR_CHUNK_MB", "1")           # many chunks => many sweep launches into the same stash segments
    for graph in (False, True):
        ref = orc.run_file(path, graph)
        for budget in (None, "4"):                    # resident, then streamed (4 MiB HBM budget)
            if budget:
                monkeypatch.setenv("PAPR_HBM_BUDGET_MB", budget)
            else:
                monkeypatch.delenv("PAPR_HBM_BUDGET_MB", raising=False)
            with pkg.PaprHip(0) as g:
                est = g.estimate_file(path)
                assert est.n > 0 and abs(est.sum / est.n - ref["sum"] / ref["n"]) < 2e-3 * ref["sum"] / ref["n"]
                g.load_file_sweep(path, pkg.guess_levels(est, graph))
                assert g.ingest_timing().resident == (0 if budget else 1)
                st = g.stats()
                check_stats(st, ref)
                assert st.flags & pkg.FLAG_ODD_TAIL and g.sweep_info().swept == 1
                mean, papr, table = pkg.levels(st, graph)
                assert np.array_equal(table, ref["level"])
                assert np.array_equal(g.ccdf(table).astype(np.int64), ref["count"])
                assert g.sweep_info().resolved == 1, g.sweep_info().as_dict()
                # a table outside the bands still gets the right answer, by reading again
                odd = (table[:4] * np.float32(1.05)).astype(np.float32)
                got = g.ccdf(odd)
                assert g.sweep_info().resolved == 0
                with pkg.PaprHip(0) as plain:           # (the two-pass path, itself checked against the oracle elsewhere)
                    plain.load_file(path)
                    assert np.array_equal(got, plain.ccdf(odd))
                # two shards of the same file, estimates merged, then merged stats
                total_n = pkg.file_samples(path)
                ranges = [exchange.shard_range(total_n, r, 2) for r in range(2)]
                guess = pkg.guess_levels(pkg.stats_merge([g.estimate_file(path, f, c) for f, c in ranges]), graph)
                parts, counts = [], np.zeros(table.size, np.uint64)
                ctxs = [pkg.PaprHip(0) for _ in ranges]
                try:
                    for c, (first, cnt) in zip(ctxs, ranges):
                        c.load_file_sweep(path, guess, first, cnt)
                        parts.append(c.stats())
                    check_stats(pkg.stats_merge(parts), ref)
                    for c in ctxs:
                        counts += c.ccdf(table)
                        assert c.sweep_info().resolved == 1
                finally:
                    for c in ctxs:
                        c.close()
                assert np.array_equal(counts.astype(np.int64), ref["count"])


def test_file_ingest_sweep_fallbacks(pkg, orc, tmp_path, monkeypatch):
    import subprocess
    monkeypatch.setenv("PAPR_CHUNK_MB", "1")
    # NaN in the file: the ingest is redone the plain way, first-NaN index and sign as the reference has them
    path = str(tmp_path / "nan.cfile")
    subprocess.check_call([orc.MKCFILE, path, "600000", "--set", "500000", "nan", "1", "--set", "300001", "1", "-nan"])
    ref = orc.run_file(path, False)
    for budget in (None, "2"):
        if budget:
            monkeypatch.setenv("PAPR_HBM_BUDGET_MB", budget)
        with pkg.PaprHip(0) as g:
            g.load_file_sweep(path, np.array([1.0, 2.0, 4.0], np.float32))
            st = g.stats()
            check_stats(st, ref)
            assert st.nan_first_idx == 300001 and st.nan_first_neg == 1 and g.sweep_info().swept == 0
    monkeypatch.delenv("PAPR_HBM_BUDGET_MB", raising=False)
    # exact-sum mode, empty guess, tiny and empty files: plain ingest, same results
    small = str(tmp_path / "small.cfile")
    subprocess.check_call([orc.MKCFILE, small, "1000"])
    empty = str(tmp_path / "empty.cfile")
    open(empty, "wb").close()
    with pkg.PaprHip(0) as g:
        assert g.estimate_file(small).n == 0 and g.estimate_file(empty).n == 0
        for p in (small, empty):
            ref = orc.run_file(p, False)
            g.load_file_sweep(p, np.array([2.0], np.float32))
            st = g.stats()
            check_stats(st, ref)
            mean, papr, table = pkg.levels(st, False)
            assert np.array_equal(g.ccdf(table).astype(np.int64), ref["count"])
        g.set_exact(True)
        g.load_file_sweep(small, np.array([2.0], np.float32))
        assert g.sweep_info().swept == 0 and g.sweep_info().as_dict()["reason"].startswith("exact mode")
        check_stats(g.stats(), orc.run_file(small, False))


@pytest.mark.parametrize("shards", [1, 3])
def test_cli_tree_sum_mode_reads_the_file_once(pkg, orc, tmp_path, shards):
    """bin/papr with PAPR_EXACT_SUM=0: estimate + one-sweep ingest.  Same stdout as with PAPR_ONE_SWEEP=0 (two
    passes), resident or streamed; PAPR_STATS tells that every shard was answered from its sweep."""
    import json
    import os
    import subprocess
    n = 2 * 1048576 + 777
    path = str(tmp_path / "cli.cfile")
    subprocess.check_call([orc.MKCFILE, path, str(n), "--spike"])
    base = dict(os.environ, PAPR_EXACT_SUM="0", PAPR_STATS="1", PAPR_GPUS=str(shards), PAPR_OVERSUBSCRIBE="1",
                PAPR_CHUNK_MB="1")
    for graph in (False, True):
        args = [pkg.CLI_PATH] + (["-g"] if graph else []) + [path]
        two = subprocess.run(args, capture_output=True, env=dict(base, PAPR_ONE_SWEEP="0"))
        assert two.returncode == 0 and json.loads(two.stderr.decode().splitlines()[-1])["shards_swept"] == 0
        for budget, force in ((None, None), (None, "1"), ("2", None)):
            env = dict(base, **({"PAPR_HBM_BUDGET_MB": budget} if budget else {}), **({"PAPR_ONE_SWEEP": force} if force else {}))
            one = subprocess.run(args, capture_output=True, env=env)
            info = json.loads(one.stderr.decode().splitlines()[-1])
            assert one.returncode == 0 and one.stdout == two.stdout, (graph, budget)
            want = shards if (budget or force) else 0      # by default only shards that do not fit in HBM are swept
            assert info["shards_swept"] == want and info["shards_resolved_from_sweep"] == want, info
            assert info["gpu0_ingest"]["resident"] == (0 if budget else 1)


@pytest.mark.parametrize("hint", [0.0, 3.0e5], ids=["good-hint", "bad-hint"])
@pytest.mark.parametrize("redo_cap", [None, "0"], ids=["read-back", "second-pass"])
def test_exact_one_sweep_ingest_of_a_streamed_shard(pkg, orc, tmp_path, monkeypatch, redo_cap, hint):
    """papr_hip_estimate_file + papr_hip_load_file_sweep in exact-sum mode on a shard that does not stay in HBM: the
    sequential sum and the counts of the reference from ONE pass over the file (the tiles whose binade speculation
    failed and the unprovable ones are read back); with the read-back switched off the file is streamed once more."""
    import subprocess
    n = 5 * 1048576 + 4099
    path = str(tmp_path / "x.cfile")
    subprocess.check_call([orc.MKCFILE, path, str(n), "--spike", "--extra-floats", "1", "--extra-bytes", "2"])
    monkeypatch.setenv("PAPR_HBM_BUDGET_MB", "2")
    monkeypatch.setenv("PAPR_CHUNK_MB", "1")
    if redo_cap is not None:
        monkeypatch.setenv("PAPR_STREAM_REDO_CAP", redo_cap)
    for graph in (False, True):
        ref = orc.run_file(path, graph)
        with pkg.PaprHip(0) as g:
            g.set_exact(True)
            est = g.estimate_file(path)
            g.set_band(pkg.band_for(est))
            g.set_exact_hint(hint)    # (a wrong hint shifts every speculated prefix: hundreds of tiles to rebuild)
            g.load_file_sweep(path, pkg.guess_levels(est, graph))
            assert g.ingest_timing().resident == 0 and g.sweep_info().swept == 1
            res, table, counts = g.analyze(None, graph)
            check_stats(res.total, ref)
            assert res.exact_sum == 1 and res.total.sum == ref["sum"] and res.mean == ref["mean"] and res.papr == ref["papr"]
            assert np.array_equal(table, ref["level"]) and np.array_equal(counts.astype(np.int64), ref["count"])
            redone = res.exact_redo_tiles
            assert (redone > 100) if hint else (redone < 100), redone
            # (nothing to rebuild and nothing unprovable would need no second pass either way)
            assert g.ingest_timing().file_passes == (2 if (redo_cap is not None and redone) else 1), g.ingest_timing().as_dict()


@pytest.mark.parametrize("shards", [1, 3])
def test_cli_exact_sum_streams_the_file_once(pkg, orc, tmp_path, shards):
    """bin/papr in its default (exact-sum) mode on a file that does not fit the HBM budget: estimate sample + ONE pass
    over the file — the sweep builds the rounding functions for speculated binades, the tiles that need another look
    are read back from the file — and the reference's stdout, sequential sum and all; also forced on resident shards."""
    import json
    import os
    import subprocess
    n = 6 * 1048576 + 12345          # odd float count: the phantom sample sits in the last tile's tail
    path = str(tmp_path / "cli.cfile")
    subprocess.check_call([orc.MKCFILE, path, str(n), "--spike", "--extra-floats", "1"])
    base = dict(os.environ, PAPR_STATS="1", PAPR_GPUS=str(shards), PAPR_OVERSUBSCRIBE="1", PAPR_CHUNK_MB="1")
    for graph in (False, True):
        args = (["-g"] if graph else []) + [path]
        want = subprocess.run([orc.REF_CLI if os.path.exists(orc.REF_CLI) else orc.CLI_PATH] + args, capture_output=True)
        for budget, force in (("2", None), (None, "1")):
            env = dict(base, **({"PAPR_HBM_BUDGET_MB": budget} if budget else {}), **({"PAPR_ONE_SWEEP": force} if force else {}))
            got = subprocess.run([pkg.CLI_PATH] + args, capture_output=True, env=env)
            assert got.returncode == 0 and got.stdout == want.stdout, (graph, budget, got.stderr[-500:])
            info = json.loads(got.stderr.decode().splitlines()[-1])
            assert info["exact_sum"] == 1 and info["shards_swept"] == shards, info
            assert info["shards_resolved_from_sweep"] == shards and info["gpu0_ingest"]["file_passes"] == 1, info
            assert info["gpu0_ingest"]["resident"] == (0 if budget else 1)
        # the two-pass form of the same (PAPR_ONE_SWEEP=0) reads the streamed file twice, same output
        two = subprocess.run([pkg.CLI_PATH] + args, capture_output=True, env=dict(base, PAPR_HBM_BUDGET_MB="2", PAPR_ONE_SWEEP="0"))
        info = json.loads(two.stderr.decode().splitlines()[-1])
        assert two.stdout == want.stdout and info["exact_sum"] == 1 and info["gpu0_ingest"]["file_passes"] == 2, info


# ---- the committed golden fixtures through the one-sweep ingest ------------------------------------------------

from conftest import golden_names, golden_path, golden_text  # noqa: E402


@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
@pytest.mark.parametrize("name", golden_names())
def test_golden_fixtures_through_the_one_sweep_ingest(pkg, gpu, name, graph):
    """every fixture (odd tails, NaN/Inf, denormals, ties, producer formats, bursts with exact silence) ingested
    with papr_hip_estimate_file + papr_hip_load_file_sweep: the report is the reference's recorded stdout"""
    path = golden_path(name)
    gpu.load_file_sweep(path, pkg.guess_levels(gpu.estimate_file(path), graph))
    st = gpu.stats()
    mean, papr, table = pkg.levels(st, graph)
    counts = gpu.ccdf(table)
    assert pkg.format_report(st, mean, papr, counts, graph).encode() == golden_text(name, graph)


# ---- papr_hip_analyze: the whole result in one call, alone and with peers ------------------------------------------

@pytest.mark.parametrize("exact", [False, True], ids=["tree", "exact"])
@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
def test_analyze_is_the_reference_result(pkg, orc, graph, exact):
    """papr_hip_analyze (one call: estimate, sweep, host scalars, recount, exact chain) against the oracle, also with
    the speculation switched off or spoiled — the result may not depend on how many reads it took."""
    n = 3 * 1048576 + 4099
    with pkg.PaprHip(0) as g:
        g.set_exact(exact)
        g.generate(pkg.SynthSpec.spike(n, seed=321), 0, n)
        iq = g.download(0, n)
        ref = orc.run_mem(iq, graph)
        for kw in (dict(), dict(two_pass=True), dict(spoil_guess=True)):
            res, table, counts = g.analyze(None, graph, **kw)
            check_stats(res.total, ref)
            assert res.nlevels == table.size == ref["level"].size and res.exact_sum == int(exact)
            if exact:
                assert res.total.sum == ref["sum"] and res.mean == ref["mean"] and res.papr == ref["papr"]
                assert np.array_equal(table, ref["level"]) and np.array_equal(counts.astype(np.int64), ref["count"])
            else:
                assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table))
            if not kw:
                assert res.swept == 1 and res.resolved == 1
            elif "two_pass" in kw:
                assert res.swept == 0 and res.resolved == 0
            else:
                assert res.swept == 1 and res.resolved == 0   # out of band: the counts came from a second read


@pytest.mark.parametrize("exact", [False, True], ids=["tree", "exact"])
def test_analyze_with_peers_over_the_in_process_exchange(pkg, orc, exact):
    """three shards of one stream, three threads, three contexts on one GPU, papr_exchange_open_local between them
    (what bin/papr does): every thread must return the oracle's whole-file result."""
    import threading
    from dtv_utils_amd import exchange
    n = 3 * 700000 + 123
    with pkg.PaprHip(0) as g0:
        g0.generate(pkg.SynthSpec.spike(n, seed=17), 0, n)
        iq = g0.download(0, n)
    ref = orc.run_mem(iq, True)
    world = 3
    xs = exchange.Exchange.local(world)
    out, errs = [None] * world, []

    def work(r):
        try:
            first, cnt = exchange.shard_range(n, r, world)
            with pkg.PaprHip(0) as g:
                g.set_exact(exact)
                g.upload(iq[2 * first:2 * (first + cnt)], base_index=first)
                out[r] = g.analyze(xs[r], True)
        except Exception as e:   # noqa: BLE001
            errs.append(e)
            pkg.lib().papr_exchange_abort(xs[r]._x)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for r in range(world):
        res, table, counts = out[r]
        check_stats(res.total, ref)
        if exact:
            assert res.total.sum == ref["sum"] and np.array_equal(table, ref["level"])
            assert np.array_equal(counts.astype(np.int64), ref["count"])
        else:
            assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table))
        assert res.total.to_bytes() == out[0][0].total.to_bytes() and np.array_equal(counts, out[0][2])
    for x in xs:
        x.close()


@pytest.mark.parametrize("exact", [False, True], ids=["tree", "exact"])
def test_analyze_constant_envelope_gives_up_the_sweep(pkg, orc, exact):
    """a constant-envelope capture (every power next to the mean: no band can be narrow enough) through the whole
    analyze call, big enough that every workgroup runs into its give-up rule: the sweep must step aside and the
    result be the reference's all the same"""
    n = 16 * 1048576 + 77
    with pkg.PaprHip(0) as g:
        g.set_exact(exact)
        g.generate(pkg.SynthSpec.spike(n, seed=99, envelope="constant"), 0, n)
        iq = g.download(0, n)
        for graph in (False, True):
            ref = orc.run_mem(iq, graph)
            res, table, counts = g.analyze(None, graph)
            check_stats(res.total, ref)
            assert res.nlevels == table.size == ref["level"].size
            if not exact:   # (the exact path's kernel can use bands narrow enough to leave the carrier outside them)
                info = g.sweep_info()
                assert res.resolved == 0 and info.gave_up > 0 and info.as_dict()["reason"] == "stash full"
            if exact:
                assert res.total.sum == ref["sum"] and np.array_equal(table, ref["level"])
                assert np.array_equal(counts.astype(np.int64), ref["count"])
            else:
                assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table))


@pytest.mark.parametrize("variant", [131] + ([48, 56] if LAB else []))
def test_exact_sweep_variants_agree_with_the_reference(pkg, orc, variant):
    """the exact-sum forms of the sweep kernel (the product's papr_sweep3_kernel; from the laboratory its predecessor with
    the returning-atomic and the ballot ring stash), whole result against the oracle"""
    if not pkg.sweep_variant_built(variant):
        pytest.skip("a laboratory kernel form: built by `make MEASURE=1` only")
    n = 7 * 1048576 + 2049
    with pkg.PaprHip(0) as g:
        g.set_exact(True)
        g.set_tuning(sweep_variant=variant)
        g.generate(pkg.SynthSpec.spike(n, seed=4242 + variant), 0, n)
        iq = g.download(0, n)
        for graph in (False, True):
            ref = orc.run_mem(iq, graph)
            res, table, counts = g.analyze(None, graph)
            check_stats(res.total, ref)
            assert res.exact_sum == 1 and res.total.sum == ref["sum"] and res.swept == 1 and res.resolved == 1
            assert np.array_equal(table, ref["level"]) and np.array_equal(counts.astype(np.int64), ref["count"])


@pytest.mark.parametrize("exact", [False, True], ids=["tree", "exact"])
@pytest.mark.parametrize("envelope", ["gauss", "bursty", "constant"])
def test_device_side_guess_equals_the_host_side_guess(pkg, orc, monkeypatch, envelope, exact):
    """papr_hip_analyze without peers builds the guessed bands on the device between the estimate and the sweep
    (papr_guess_bands_kernel: no host round trip) and runs the stash recount speculatively against the level table the
    device expects (papr_true_table_kernel; taken only if the host's libm table is that table bit for bit);
    PAPR_FUSED_GUESS=0 is the host path, PAPR_SPEC_RECOUNT=0 recounts on the host's table.  Same result every way —
    and the reference's — at sizes where the sample is everything and where it is not, with the guess spoiled, both
    tables."""
    for n in (2047, 300007, 20 * 1048576 + 333):
        with pkg.PaprHip(0) as g:
            g.set_exact(exact)
            g.generate(pkg.SynthSpec.spike(n, seed=1000 + n % 97, envelope=envelope), 0, n)
            iq = g.download(0, n)
            for graph in (False, True):
                ref = orc.run_mem(iq, graph)
                got = {}
                want_counts = {}   # (the oracle's counts per level table: the six ways arrive at one or two tables)

                def oracle_counts(table):
                    key = table.tobytes()
                    if key not in want_counts:
                        want_counts[key] = orc.count_mem(iq, table)
                    return want_counts[key]
                for fused in ("1", "0", "1 without the speculated recount"):
                    monkeypatch.setenv("PAPR_FUSED_GUESS", fused[0])
                    monkeypatch.setenv("PAPR_FUSED_EXACT", fused[0])
                    monkeypatch.setenv("PAPR_SPEC_RECOUNT", "0" if len(fused) > 1 else "1")
                    for kw in (dict(), dict(spoil_guess=True)):
                        res, table, counts = g.analyze(None, graph, **kw)
                        check_stats(res.total, ref)
                        assert np.array_equal(counts.astype(np.int64), oracle_counts(table))
                        if exact:
                            assert res.exact_sum == 1 and res.total.sum == ref["sum"] and np.array_equal(table, ref["level"])
                        got[(fused, bool(kw))] = (res.total.sum, res.nlevels, tuple(counts.tolist()), res.swept, res.resolved,
                                                  res.band_log2 if res.swept else None)
                for spoiled in (False, True):
                    a, b, c = got[("1", spoiled)], got[("0", spoiled)], got[("1 without the speculated recount", spoiled)]
                    assert a[:3] == b[:3] == c[:3], (n, graph, spoiled)       # sum, levels, counts: bit for bit
                    assert a[3:] == b[3:] == c[3:], (n, graph, spoiled, a, b)  # ... and the same decisions on the way


def test_analyze_hostile_inputs_through_the_single_wait_step(pkg, orc):
    """papr_hip_analyze without peers (estimate, device-side guess, sweep, device-side table, recount: one wait) on
    inputs that break its assumptions one at a time — NaN and Inf in the data, all-zero and denormal-power streams
    (no band form), an amplitude whose powers overflow float, a peak that makes the table longer than the device
    speculates (> 1024 levels) — against the oracle, both tables."""
    n = 600011
    rng = np.random.default_rng(2024)
    base = rng.standard_normal(2 * n).astype(np.float32)
    cases = {}
    x = base.copy(); x[2 * 123457] = np.nan; cases["nan"] = x
    x = base.copy(); x[2 * 77 + 1] = -np.nan; x[2 * 500000] = np.inf; cases["nan and inf"] = x
    x = base.copy(); x[2 * 400001 + 1] = -np.inf; cases["inf"] = x
    cases["zeros"] = np.zeros(2 * n, np.float32)
    cases["denormal powers"] = (base * np.float32(1e-21)).astype(np.float32)
    cases["overflowing powers"] = (base * np.float32(3e19)).astype(np.float32)
    x = (base * np.float32(1e-3)).astype(np.float32); x[2 * (n - 5)] = np.float32(3e5); cases["170 dB peak"] = x   # -g: > 1024 levels
    # (the peak near the end: a sequential double sum that met it early would drop the low bits of everything after it)
    with pkg.PaprHip(0) as g:
        for name, iq in cases.items():
            g.upload(iq)
            for graph in (False, True):
                ref = orc.run_mem(iq, graph)
                res, table, counts = g.analyze(None, graph)
                check_stats(res.total, ref)
                assert res.nlevels == table.size == ref["level"].size, (name, graph)
                # (the tree sum may differ from the reference's sequential sum in the last places: count against OUR table)
                assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table)), (name, graph)


@pytest.mark.parametrize("exact", [False, True], ids=["tree", "exact"])
def test_counters_through_the_finalize_kernel_equal_the_copied_ones(pkg, orc, monkeypatch, exact):
    """The sweep's bins and segment counters reach the host through papr_stats_finalize (mapped host memory) — in the
    single-wait step and in the with-peers path (estimate / stats_sweep / ccdf called one by one) — or, with
    PAPR_FUSED_COPIES=1, by a D2H copy as before.  Same results, same decisions, and the reference's."""
    for n in (4099, 5 * 1048576 + 77):
        with pkg.PaprHip(0) as g:
            g.set_exact(exact)
            g.generate(pkg.SynthSpec.spike(n, seed=4242 + n % 89), 0, n)
            iq = g.download(0, n)
            for graph in (False, True):
                ref = orc.run_mem(iq, graph)
                got = {}
                for copies in ("0", "1"):
                    monkeypatch.setenv("PAPR_FUSED_COPIES", copies)
                    res, table, counts = g.analyze(None, graph)
                    check_stats(res.total, ref)
                    assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table))
                    # the with-peers sequence, by hand
                    est = g.estimate()
                    g.set_band(pkg.band_for(est))
                    st = g.stats_sweep(pkg.guess_levels(est, graph))
                    mean, papr, table2 = pkg.levels(st, graph)
                    counts2 = g.ccdf(table2)
                    info = g.sweep_info()
                    check_stats(st, ref)
                    assert np.array_equal(counts2.astype(np.int64), orc.count_mem(iq, table2))
                    got[copies] = (res.total.sum, tuple(counts.tolist()), res.swept, res.resolved, st.sum,
                                   tuple(counts2.tolist()), info.swept, info.resolved, info.stash_samples)
                assert got["0"] == got["1"], (n, graph)


def test_timing_levels(pkg):
    """papr_hip_set_timing: 1 times every bracketed kernel, 2 only the kernels that read the shard (the estimate and
    recount kernels are left alone: a timed dispatch costs the stream a few microseconds on either side), 0 nothing.
    The events are bound to the dispatches, so the sweep's figure is the kernel's own duration."""
    n = 8 * 1048576
    with pkg.PaprHip(0) as g:
        g.generate(pkg.SynthSpec.spike(n, seed=77), 0, n)
        g.analyze(None, False)
        seen = {}
        for level in (1, 2, 0):
            g.set_timing(level)
            for _ in range(3):
                res, _, _ = g.analyze(None, False)
                assert res.swept and res.resolved
            tm = g.timing()
            seen[level] = (tm.sweep_launches, tm.aux_launches, tm.sweep_ms, tm.aux_ms, tm.sweep_bytes)
        g.set_timing(0)
    assert seen[1][0] == 3 and seen[1][1] >= 6 and seen[1][2] > 0 and seen[1][3] > 0  # (estimate + recount per step)
    assert seen[2][0] == 3 and seen[2][1] == 0 and seen[2][2] > 0 and seen[2][3] == 0
    assert seen[0][:2] == (0, 0)
    assert seen[1][4] == seen[2][4] == 3 * n * 8
    # 64 MiB at < 8 TB/s takes > 8 us per launch; a duration that included the neighbours' launches would be far above 1 ms
    for level in (1, 2):
        per_launch = seen[level][2] / 3
        assert 0.008 < per_launch < 1.0, seen


_SKEW_SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as ge
pkg = ge.load_package()
n = (1 << 26) + 12345
out = {}
with pkg.PaprHip(0) as g:
    g.generate(pkg.SynthSpec.spike(n, seed=2718), 0, n)
    for exact in (False, True):
        g.set_exact(exact)
        for graph in (False, True):
            for rep in range(2):   # (the second step's parity comes out of the first one's record)
                res, table, counts = g.analyze(None, graph)
            us, xcd = g.wg_finish()
            info = g.sweep_info()
            t = res.total
            # (the same through the calls one by one — papr_hip_stats_sweep's own launch takes the skew too)
            est = g.estimate()
            g.set_band(pkg.band_for(est))
            st = g.stats_sweep(pkg.guess_levels(est, graph))
            counts2 = g.ccdf(pkg.levels(st, graph)[2])
            out["%d%d" % (exact, graph)] = dict(
                sum=float(t.sum).hex(), mean=float(res.mean).hex(), papr=float(res.papr).hex(), swept=res.swept, resolved=res.resolved,
                trackers=[[float(getattr(t, k)).hex(), int(getattr(t, k + "_idx"))] for k in ("peak", "re_pos", "re_neg", "im_pos", "im_neg")],
                table=table.tobytes().hex()[:4096], counts=counts.tolist(), xcd=xcd.tolist(), xcd_first=info.xcd_first,
                by_hand=[None if exact else float(st.sum).hex(), int(st.peak_idx), counts2.tolist()])   # (exact mode: this call's sum is the kernel's
                # own running total of the segments' pairs, whatever order its workgroups met them in — the result's sum is the program's)
print(json.dumps(out))
"""


def test_the_xcd_skew_changes_no_result_and_follows_the_queue(pkg, tmp_path):
    """The sweep kernels' walk gives the workgroups on the odd XCDs one round in R less (papr_skew_walk.h) — which workgroups those
    are is the queue's business (workgroup 0 on XCD 6 in a plain process, 5 with RCCL's queues beside: profiles/r05_xcd_skew.txt),
    so the runtime asks (papr_hip_sweep_info.xcd_first) and follows every sweep's record.  Whatever the period and whichever
    parity takes the skew, every result is the one of the plain grid stride, bit for bit; workgroups land on the XCDs round-robin
    from the one the runtime believes in."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "skew.py"
    script.write_text(_SKEW_SCRIPT)
    runs = {}
    for name, env in (("plain", dict(PAPR_XCD_SKEW="0")), ("default", dict()), ("odd slow", dict(PAPR_XCD_SKEW="3", PAPR_XCD_PARITY="0")),
                      ("even slow", dict(PAPR_XCD_SKEW="3", PAPR_XCD_PARITY="1")), ("short period", dict(PAPR_XCD_SKEW="2"))):
        p = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert p.returncode == 0, (name, p.stderr[-2000:])
        runs[name] = json.loads(p.stdout.strip().splitlines()[-1])
    for name, got in runs.items():
        for mode, r in got.items():
            xcd = np.array(r.pop("xcd"))
            first = r.pop("xcd_first")
            assert r["swept"] == 1 and r["resolved"] == 1, (name, mode)
            assert np.array_equal(xcd, (np.arange(xcd.size) + xcd[0]) % 8), (name, mode, xcd[:16])
            assert first == xcd[0], (name, mode, first, xcd[:8])
            want = dict(runs["plain"][mode])
            want.pop("xcd", None), want.pop("xcd_first", None)
            assert r == want, (name, mode)
