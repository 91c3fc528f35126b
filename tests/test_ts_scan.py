"""The transport-stream packet scan (include/ts_hip.h).

CPU tier: the closed-form packet walker (dtv-utils_amd/csrc/ts_walk_core.h compiled into ts_host.c's ts_walk — the very
step the scan kernel runs across irregular packets) against the recorded lines of the real reference and against the
byte-wise oracle on random damaged streams, whole and fed in windows; the library exports every symbol
include/ts_hip.h declares.
GPU tier: ts_hip_scan through the C ABI against the same goldens / oracle — one span per CU and, on the small
fixtures, many small spans (the speculated entries, the chain check and the re-launch of a span that guessed wrong) —
the synthetic bench stream against the host generator, more sync errors than the result holds inline."""
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import FULL_TIER, GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ts_oracle  # noqa: E402
import ts_streams  # noqa: E402
from test_ts_oracle import MANIFEST, golden_lines, random_stream_kwargs  # noqa: E402


@pytest.fixture(scope="module")
def ts(pkg):
    from dtv_utils_amd import ts as mod
    return mod


def test_library_exports_every_symbol_of_ts_hip_h(ts):
    header = open(os.path.join(ROOT, "include", "ts_hip.h")).read()
    declared = set(re.findall(r"\b(ts_(?:hip_|host_|walk|format)\w*)\s*\(", header))
    assert declared == set(ts.ABI_SYMBOLS), declared ^ set(ts.ABI_SYMBOLS)
    L = ts._lib()
    for name in ts.ABI_SYMBOLS:
        assert hasattr(L, name), name


def test_the_line_pool_runs_every_job_exactly_once(ts):
    """The host threads that lay out a damaged stream's report lines take their jobs from one atomic ticket, round after round
    without sleeping in between.  A ticket drawn late in one round once ran a job of the NEXT round a second time (its job count
    and function were read after the main thread had moved on): holes of `skipped 0 bytes, at 0` in one report in fifty and, when
    the double count overshot, a scan that never returned.  Rounds of 1 ... 64 jobs, jumping about, every job exactly once."""
    L = ts._lib()
    for threads, rounds in ((8, 150000), (3, 60000), (2, 30000), (1, 100)):
        assert L.ts_host_pool_selftest(threads, rounds) == 0, (threads, rounds)


@pytest.mark.parametrize("name", sorted(ts_streams.FIXTURES))
def test_walker_reproduces_reference_lines(ts, name):
    data = ts_streams.fixture_bytes(name)
    assert hashlib.sha256(data).hexdigest() == MANIFEST[name]["sha256"]
    hd = ts_streams.is_hdmv(name)
    assert ts.walk(data, hd).report() == golden_lines(name)
    assert ts.walk(data, hd, window=3000, rng=np.random.default_rng(3)).report() == golden_lines(name)


def test_walker_equals_oracle_on_random_damaged_streams(ts):
    rng = np.random.default_rng(99)
    for t in range(150):
        kw = random_stream_kwargs(t, rng)
        data = ts_streams.make_stream(**kw)
        want = ts_oracle.report_lines(ts_oracle.scan_mem(data, kw["hdmv"]))
        assert ts.walk(data, kw["hdmv"]).report() == want, kw
        assert ts.walk(data, kw["hdmv"], window=int(rng.integers(400, 40000)), rng=rng).report() == want, kw


def test_synthetic_stream_is_pinned_and_regular(tmp_path):
    """include/ts_synth.h through oracle/mkts: fixed bytes, and a stream the reference reads without a single
    sync error or continuity complaint."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "mkts"])
    path = str(tmp_path / "s.ts")
    subprocess.check_call([os.path.join(ROOT, "oracle", "mkts"), path, "20000"])
    data = open(path, "rb").read()
    assert len(data) == 20000 * 188 and hashlib.sha256(data).hexdigest() == \
        "5d2c099c6bcb7696b86137cddc499d84bde9d0f3134ea1395d42ad0a4681e7ac"
    res = ts_oracle.scan_mem(data)
    assert res["packets"] == 20000 and res["nsync_errors"] == 0
    if os.path.exists(ts_oracle.REF_CLI):
        p = subprocess.run([ts_oracle.REF_CLI, "-ps", path, ts_oracle.REF_PROGRAM, "1", "1"], capture_output=True)
        assert b"Discontinuity" not in p.stdout and b"Sync Error" not in p.stdout
        assert ts_oracle.filter_lines(p.stdout) == ts_oracle.report_lines(res)


# ---- GPU ------------------------------------------------------------------------------------------------

def _with_env(ts, env):
    """a context opened under `env` (the scan's geometry and form are fixed when it is opened)"""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return ts.TsHip(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# Every GPU test runs the scan three ways: "auto" (the default: the full-table form — 1024 threads, 8192-entry tables, one span
# per CU — which gives a damaged stream up for the slot form), "slots" (512 threads, per-slot tables, two spans per CU; hands a
# stream with more PIDs in a span than it has slots back to the full tables) and "full" (the full-table form, whatever comes)
@pytest.fixture(scope="module", params=["auto", "slots", "full"])
def gpu(ts, request):
    g = _with_env(ts, {"TS_SCAN_FORM": request.param})
    yield g
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ts_streams.FIXTURES))
def test_scan_reproduces_reference_lines(ts, gpu, name):
    data = ts_streams.fixture_bytes(name)
    gpu.upload(data)
    res = gpu.scan(ts_streams.is_hdmv(name))
    assert res.report() == golden_lines(name)
    assert res.bytes == len(data) and res.packets == ts_oracle.scan_mem(data, ts_streams.is_hdmv(name))["packets"]


@pytest.mark.gpu
def test_scan_equals_oracle_on_random_damaged_streams(ts, gpu):
    rng = np.random.default_rng(4242)
    for t in range(120 if FULL_TIER else 60):
        kw = random_stream_kwargs(t, rng)
        kw["npackets"] = int(kw["npackets"] * rng.choice([1, 1, 8, 40]))   # some streams long enough for every workgroup
        data = ts_streams.make_stream(**kw)
        ref = ts_oracle.scan_mem(data, kw["hdmv"])
        gpu.upload(data)
        res = gpu.scan(kw["hdmv"])
        assert res.report() == ts_oracle.report_lines(ref), kw
        cnt, first, last = res.tables()
        assert res.packets == ref["packets"] and np.array_equal(cnt, ref["count"]) and \
            np.array_equal(first, ref["first"]) and np.array_equal(last, ref["last"]), kw
        assert res.sync_error_list() == ref["sync_errors"]
        assert res.discontinuity_list() == ref["discontinuities"], kw


@pytest.mark.gpu
def test_scan_of_the_synthetic_stream_and_handover_counts(ts, gpu, tmp_path):
    """The bench stream generated ON the device equals the host generator's bytes; a regular stream is one launch
    and no walk; one inserted byte costs a walk on the device, not a hand-over or a rescan."""
    n = 300000
    gpu.generate(n)
    path = str(tmp_path / "s.ts")
    subprocess.check_call([os.path.join(ROOT, "oracle", "mkts"), path, str(n)])
    host = open(path, "rb").read()
    assert gpu.download(0, n * 188) == host
    res = gpu.scan()
    ref = ts_oracle.scan_mem(host)
    assert res.report() == ts_oracle.report_lines(ref) and res.packets == n
    assert res.launches == 1 and res.walks == 0 and res.gpu_packets == n
    gpu.generate(1000, hdmv=True)
    subprocess.check_call([os.path.join(ROOT, "oracle", "mkts"), path, "1000", "--hdmv"])
    host_h = open(path, "rb").read()
    assert gpu.download(0, 192000) == host_h
    assert gpu.scan(hdmv=True).report() == ts_oracle.report_lines(ts_oracle.scan_mem(host_h, True))
    damaged = host[:188 * 1234] + b"\x00" + host[188 * 1234:]
    gpu.upload(damaged)
    res = gpu.scan()
    assert res.report() == ts_oracle.report_lines(ts_oracle.scan_mem(damaged))
    # one byte too many puts every later packet on an odd offset: about one in 4096 of them then ends a byte past a
    # 16384-byte read of the reference (a `skipped 1 bytes` line each, xport.c:4302) — the kernel reports those itself
    errs = res.sync_error_list()
    assert errs == ts_oracle.scan_mem(damaged)["sync_errors"] and errs[0] == (1, 1234) and len(errs) > 60
    assert res.launches <= 2 and res.walks <= 8 and res.gpu_packets >= n - 12
    # file ingest and caller-owned device memory
    gpu.load_file(path)
    assert gpu.scan(hdmv=True).report() == ts_oracle.report_lines(ts_oracle.scan_mem(host_h, True))
    with pytest.raises(Exception):
        gpu.load_file(str(tmp_path / "missing.ts"))


@pytest.fixture(scope="module", params=["auto", "slots", "full", "full, bridges worked out at once", "slots, bridges walked by the merge",
                                        "slots, no look-ahead", "full, no look-ahead"])
def gpu_small_spans(ts, request):
    """a context that cuts even the small fixtures into many spans (4 KiB each: 21 packets).  The bridges between the spans are
    worked out by ts_bridge_kernel in front of the merge for the slot form and for re-scanned spans, by the merge kernel itself
    otherwise: the last two parameters swap that (TS_SCAN_BRIDGES)."""
    form, _, how = request.param.partition(", ")
    env = {"TS_SCAN_SPANS": "256", "TS_SCAN_MIN_SPAN": "4096", "TS_SCAN_FORM": form}
    if "bridges" in how:
        env["TS_SCAN_BRIDGES"] = "1" if "at once" in how else "0"
    if "look-ahead" in how:   # (a damaged spot's window and the headers behind it asked for only when they are needed)
        env["TS_SCAN_LOOKAHEAD"] = "0"
    g = _with_env(ts, env)
    yield g
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ts_streams.FIXTURES))
def test_scan_in_many_small_spans_reproduces_reference_lines(ts, gpu_small_spans, name):
    """every fixture cut into 4 KiB spans: damage next to and across span boundaries, spans without any packet grid
    (garbage), spans the chain has run past — the speculated entries, the chain check, the re-launches"""
    data = ts_streams.fixture_bytes(name)
    gpu_small_spans.upload(data)
    res = gpu_small_spans.scan(ts_streams.is_hdmv(name))
    assert res.report() == golden_lines(name)
    ref = ts_oracle.scan_mem(data, ts_streams.is_hdmv(name))
    cnt, first, last = res.tables()
    assert res.packets == ref["packets"] and np.array_equal(cnt, ref["count"]) and np.array_equal(first, ref["first"]) \
        and np.array_equal(last, ref["last"])


@pytest.mark.gpu
def test_scan_in_small_spans_equals_oracle_on_random_damaged_streams(ts, gpu_small_spans):
    rng = np.random.default_rng(777)
    relaunched = 0
    for t in range(120 if FULL_TIER else 60):
        kw = random_stream_kwargs(t, rng)
        kw["npackets"] = int(kw["npackets"] * rng.choice([1, 4, 16]))
        data = ts_streams.make_stream(**kw)
        ref = ts_oracle.scan_mem(data, kw["hdmv"])
        gpu_small_spans.upload(data)
        res = gpu_small_spans.scan(kw["hdmv"])
        assert res.report() == ts_oracle.report_lines(ref), kw
        cnt, first, last = res.tables()
        assert res.packets == ref["packets"] and np.array_equal(cnt, ref["count"]) and \
            np.array_equal(first, ref["first"]) and np.array_equal(last, ref["last"]), kw
        relaunched += res.launches > 1
    assert relaunched > 0   # (some span did guess wrong: the re-launch path was taken)


@pytest.mark.gpu
@pytest.mark.parametrize("env,many", [({"TS_SCAN_SLOT_LIMIT": "3"}, False),
                                      ({"TS_SCAN_SLOT_LIMIT": "3", "TS_SCAN_SPANS": "64", "TS_SCAN_MIN_SPAN": "4096"}, False),
                                      ({"TS_SCAN_SPANS": "4"}, True)],
                         ids=["three slots", "three slots, small spans", "3000 PIDs in a span"])
def test_more_pids_than_slots_falls_back_to_the_full_tables(ts, env, many):
    """The slot form keeps per-PID state for the PIDs a span actually meets (1024 slots).  A span that meets more — garbage
    read as packets carries any PID — says so, and the scan is done again with the 8192-entry tables: same report, same
    tables, at the price of the launches that were dropped."""
    rng = np.random.default_rng(99)
    with _with_env(ts, dict(env, TS_SCAN_FORM="slots")) as g:  # (the slot form first: the default would not take it for these)
        fell_back = 0
        for t in range(12):
            if not many:
                kw = random_stream_kwargs(t, rng)
                kw["npackets"] = int(kw["npackets"] * rng.choice([1, 8]))
            else:  # 20000 packets in four spans, 3000 PIDs among them
                kw = dict(seed=int(rng.integers(1, 1 << 30)), npackets=20000, hdmv=bool(t & 1),
                          pids=tuple(int(v) for v in rng.choice(np.arange(1, 0x1FFF), size=3000, replace=False)))
            data = ts_streams.make_stream(**kw)
            ref = ts_oracle.scan_mem(data, kw["hdmv"])
            g.upload(data)
            res = g.scan(kw["hdmv"])
            assert res.report() == ts_oracle.report_lines(ref), kw
            cnt, first, last = res.tables()
            assert res.packets == ref["packets"] and np.array_equal(cnt, ref["count"]) and \
                np.array_equal(first, ref["first"]) and np.array_equal(last, ref["last"]), kw
            assert res.discontinuity_list() == ref["discontinuities"], kw
            fell_back += res.launches >= 2
        assert fell_back >= (12 if many else 4)


@pytest.mark.gpu
def test_a_damaged_stream_is_given_up_for_the_slot_form(ts, tmp_path):
    """The default scan starts in the full-table form; spans that meet damage more often than once in 6144 packets stop it
    and the slot form does the stream (one launch more, the same report); an occasional damaged spot does not."""
    with _with_env(ts, {"TS_SCAN_FORM": "auto"}) as g, _with_env(ts, {"TS_SCAN_FORM": "full"}) as f:
        for n, period, gives_up in ((2_000_000, 500, True), (2_000_000, 50000, False)):
            size = g.generate_damaged(n, period)
            assert f.generate_damaged(n, period) == size
            a, b = g.scan(), f.scan()
            assert a.report() == b.report() and a.packets == b.packets
            assert a.sync_error_list() == b.sync_error_list() and a.discontinuity_list() == b.discontinuity_list()
            ca, cb = a.tables(), b.tables()
            assert all(np.array_equal(x, y) for x, y in zip(ca, cb))
            assert (a.launches > b.launches) == gives_up, (period, a.launches, b.launches)
            if gives_up:   # the context remembers: the next scan starts in the slot form (and goes back when a stream is in order)
                again = g.scan()
                assert again.launches < a.launches and again.report() == a.report() and again.packets == a.packets
                assert all(np.array_equal(x, y) for x, y in zip(again.tables(), ca))
        path = str(tmp_path / "d.ts")
        subprocess.check_call([os.path.join(ROOT, "oracle", "mkts"), path, "400000", "--damage", "500"])
        host = open(path, "rb").read()
        g.upload(host)
        assert g.scan().report() == ts_oracle.report_lines(ts_oracle.scan_mem(host))
    # Looking ahead across a damaged spot (the walker's window and the headers of the block behind asked for while the partial
    # block is committed) is a matter of when bytes are asked for: with and without, in both forms, on a stream of the bench's
    # kind whose spans are long enough for whole blocks on either side of every spot — and against the oracle
    want = None
    for form in ("slots", "full"):
        for ahead in ("1", "0"):
            with _with_env(ts, {"TS_SCAN_FORM": form, "TS_SCAN_LOOKAHEAD": ahead}) as g:
                g.generate_damaged(8_000_000, 1000)
                r = g.scan()
                got = (r.report(), r.packets, r.sync_error_list(), r.discontinuity_list(), [x.tobytes() for x in r.tables()])
                assert want is None or got == want, (form, ahead)
                want = got
                g.upload(host)
                assert g.scan().report() == ts_oracle.report_lines(ts_oracle.scan_mem(host)), (form, ahead)


@pytest.mark.gpu
def test_quirk_packets_through_the_walker_give_the_same_report(ts, monkeypatch):
    monkeypatch.setenv("TS_SCAN_QUIRK_EVENTS", "0")
    with ts.TsHip(0) as g:
        for name in ("ts_quirk", "ts_quirk_af", "ts_hdmv_quirk", "ts_quirk_then_garbage"):
            g.upload(ts_streams.fixture_bytes(name))
            assert g.scan(ts_streams.is_hdmv(name)).report() == golden_lines(name)


@pytest.mark.gpu
def test_more_sync_errors_than_the_result_holds_inline(ts, gpu, tmp_path):
    """The reference prints EVERY `Transport Sync Error` line (xport.c:4325-4327); ts_scan_result holds 4096 inline
    and ts_hip_get_sync_errors the rest.  A synthetic stream with a damaged spot every 40 packets: 7000+ lines."""
    n, period = 280000, 40
    size = gpu.generate_damaged(n, period)
    path = str(tmp_path / "d.ts")
    subprocess.check_call([os.path.join(ROOT, "oracle", "mkts"), path, str(n), "--damage", str(period)])
    host = open(path, "rb").read()
    assert len(host) == size and gpu.download(0, size) == host
    res = gpu.scan()
    ref = ts_oracle.scan_mem(host)
    assert ref["nsync_errors"] > ts.MAX_SYNC_ERRORS and res.nsync_errors == ref["nsync_errors"]
    assert res.sync_error_list() == ref["sync_errors"]
    # ... and every `Discontinuity!` line (xport.c:2876-2884): each lost packet leaves its PID's counter one short
    assert res.ndiscontinuities == ref["ndiscontinuities"] > 1000 and res.discontinuity_list() == ref["discontinuities"]
    assert res.report() == ts_oracle.report_lines(ref)
    if os.path.exists(ts_oracle.REF_CLI):
        assert res.report() == ts_oracle.reference_lines(path)
    # damage costs walks on the device — in the span it sits in, or, right behind a span boundary, in the merge's bridge
    # up to the span's speculated entry; a launch more only where the chain arrives BEHIND the place a span assumed
    # (measured on this stream: 18 launches; before the bridges: one per damaged boundary, 50+)
    assert res.walks >= n // period and res.launches <= 1 + 40


@pytest.mark.gpu
def test_scan_error_states(ts):
    with ts.TsHip(0) as g:
        with pytest.raises(Exception) as e:
            g.scan()
        assert e.value.code == -6
        g.upload(b"")
        res = g.scan()
        assert res.packets == 0 and res.report() == b"" and res.launches == 0
        with pytest.raises(Exception):
            g.generate_damaged(1000, 30)   # not a multiple of 4 * period


@pytest.mark.gpu
@pytest.mark.parametrize("hdmv", [False, True], ids=["188", "192"])
def test_long_spans_with_random_damage_equal_the_oracle(ts, hdmv):
    """Spans as long as the bench's (thousands of packets: whole blocks on either side of a damaged spot, so the look-ahead has
    something to be right or wrong about) with damage of every kind at random places — bytes inserted (1 ... 400, some of them
    0x47), bytes missing (1 ... 400), sync bytes overwritten, bytes overwritten with false sync bytes, whole packets gone — in
    all three forms, looking ahead and not, against the oracle."""
    rng = np.random.default_rng(20260930 + int(hdmv))
    unit = 192 if hdmv else 188
    npackets = 1_200_000
    with ts.TsHip(0) as g:
        g.generate(npackets, seed=0x7500001 + 77, hdmv=hdmv)
        clean = np.frombuffer(g.download(0, npackets * unit), dtype=np.uint8)
    places = np.sort(rng.choice(npackets - 8, size=900, replace=False) + 4)
    parts, at = [], 0
    for k in places:
        cut = int(k) * unit + int(rng.integers(0, unit))
        parts.append(clean[at:cut])
        kind = int(rng.integers(0, 6))
        if kind == 0:     # garbage inserted
            junk = rng.integers(0, 256, size=int(rng.integers(1, 401)), dtype=np.uint8)
            parts.append(junk)
            at = cut
        elif kind == 1:   # bytes missing
            at = cut + int(rng.integers(1, 401))
        elif kind == 2:   # the next sync byte overwritten
            nxt = (int(k) + 1) * unit + (4 if hdmv else 0)
            parts.append(clean[cut:nxt])
            parts.append(np.array([int(rng.integers(0, 256))], dtype=np.uint8))
            at = nxt + 1
        elif kind == 3:   # a run overwritten, false sync bytes in it
            n = int(rng.integers(1, 600))
            junk = rng.integers(0, 256, size=n, dtype=np.uint8)
            junk[rng.integers(0, n, size=max(1, n // 40))] = 0x47
            parts.append(junk)
            at = cut + n
        elif kind == 4:   # whole packets gone (the grid stays)
            at = cut + unit * int(rng.integers(1, 4))
        else:             # a few bytes inserted: the grid moves by 1 ... 7
            parts.append(rng.integers(0, 256, size=int(rng.integers(1, 8)), dtype=np.uint8))
            at = cut
    parts.append(clean[at:])
    data = np.concatenate(parts).tobytes()
    ref = ts_oracle.scan_mem(data, hdmv)
    want = ts_oracle.report_lines(ref)
    for form in ("auto", "slots", "full"):
        for ahead in ("1", "0"):
            for spans in ("256", "24"):   # (4 700 packets a span: a few blocks; 50 000: dozens of whole blocks between the spots)
                with _with_env(ts, {"TS_SCAN_FORM": form, "TS_SCAN_LOOKAHEAD": ahead, "TS_SCAN_SPANS": spans}) as g:
                    g.upload(data)
                    res = g.scan(hdmv)
                    how = (form, ahead, spans)
                    assert res.report() == want, how
                    cnt, first, last = res.tables()
                    assert res.packets == ref["packets"] and np.array_equal(cnt, ref["count"]) and \
                        np.array_equal(first, ref["first"]) and np.array_equal(last, ref["last"]), how
                    assert res.sync_error_list() == ref["sync_errors"] and res.discontinuity_list() == ref["discontinuities"], how


@pytest.mark.gpu
def test_a_span_the_one_in_front_reached_into_is_not_scanned_again(ts):
    """A span that ends un-clean on its boundary takes, by the reference's rules, the first packet(s) that start in the next span
    along.  That span counted them too: instead of scanning it again from the chain's state (one workgroup over the whole span),
    the merge counts them once (ts_overlap_packets: the chain must stand whole packets behind the span's entry, on its grid,
    having consumed exactly those packets back to back).  Streams of the bench's kind where that happens (and one where the
    chain merely ENDS on the next span's grid, which must not be taken for it: period 40): same report, same tables as with
    TS_SCAN_OVERLAP=0 and as the oracle's where the oracle is quick; fewer launches."""
    fewer = 0
    for period, n in ((1500, 56_490_000), (500, 36_732_000), (700, 34_459_600), (1000, 20_224_000), (40, 280_000)):
        n = n // (4 * period) * (4 * period)
        got = {}
        for overlap in ("1", "0"):
            with _with_env(ts, {"TS_SCAN_FORM": "slots" if period > 40 else "full", "TS_SCAN_OVERLAP": overlap}) as g:
                g.generate_damaged(n, period, seed=0x7500001 + n % 97)
                g.scan()
                r = g.scan()
                got[overlap] = (r.report(), r.packets, [x.tobytes() for x in r.tables()], r.sync_error_list(), r.discontinuity_list())
                launches = r.launches
                if overlap == "1":
                    with_overlap = launches
                    if n < 1_000_000:
                        host = g.download(0, n * 188 - n // (4 * period))
                        ref = ts_oracle.scan_mem(host)
                        assert r.report() == ts_oracle.report_lines(ref) and r.discontinuity_list() == ref["discontinuities"]
        assert got["1"] == got["0"], (period, n)
        assert with_overlap <= launches
        fewer += with_overlap < launches
    assert fewer >= 3
