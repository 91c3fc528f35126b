"""Parity tests proper: the HIP path, called through the C ABI (ctypes) and
through the bin/papr host program, against the CPU oracle on the same inputs,
against the committed golden stdout of the real reference, and — at
BASELINE.json's full 10 GiB size — against the reference's recorded stdout and
size-independent properties.  Integer/index results must be bit-exact; the
double sum is order-dependent in the reference (serial `sum +=`, papr.c:104)
and is held to 1e-12 relative here (north_star: dB values within 1e-5)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import FULL_TIER, ROOT, golden_names, golden_path, golden_text

pytestmark = pytest.mark.gpu

SUM_RTOL = 1e-12
TRACKERS = ("peak", "re_pos", "re_neg", "im_pos", "im_neg")


@pytest.fixture(scope="module")
def gpu(pkg):
    g = pkg.PaprHip(0)
    yield g
    g.close()


def check_stats(st, ref, base=0):
    assert st.n == ref["n"]
    for k in TRACKERS:
        assert getattr(st, k) == ref[k], k
        want = ref[k + "_idx"] + base if ref[k] != 0 else 0
        assert getattr(st, k + "_idx") == want, k
    if ref["sum"] != ref["sum"]:
        assert st.sum != st.sum and np.signbit(st.sum) == np.signbit(ref["sum"])
    elif np.isinf(ref["sum"]):
        assert st.sum == ref["sum"]
    else:
        assert abs(st.sum - ref["sum"]) <= SUM_RTOL * abs(ref["sum"])


def run_both_passes(pkg, orc, g, floats_for_oracle, graph):
    st = g.stats()
    mean, papr, table = pkg.levels(st, graph)
    o_mean, o_papr, o_table = orc.levels_from(st.sum, st.n, st.peak, graph)
    assert np.array_equal(table, o_table)
    counts = g.ccdf(table)
    assert np.array_equal(counts.astype(np.int64), orc.count_mem(floats_for_oracle, table))
    return st, mean, papr, table, counts


# ---- committed golden fixtures: ABI path and CLI path ---------------------------

@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
@pytest.mark.parametrize("name", golden_names())
def test_abi_reproduces_reference_stdout(pkg, gpu, name, graph):
    gpu.load_file(golden_path(name))
    st = gpu.stats()
    mean, papr, table = pkg.levels(st, graph)
    counts = gpu.ccdf(table)
    assert pkg.format_report(st, mean, papr, counts, graph).encode() == golden_text(name, graph)


@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
@pytest.mark.parametrize("name", golden_names())
def test_cli_reproduces_reference_stdout(pkg, manifest, name, graph):
    args = [pkg.CLI_PATH] + (["-g"] if graph else []) + [golden_path(name)]
    p = subprocess.run(args, capture_output=True)
    want = manifest[name]["graph" if graph else "default"]
    assert p.returncode == want["rc"], p.stderr
    assert p.stderr.decode() == want["stderr"]
    assert p.stdout == golden_text(name, graph)


@pytest.mark.parametrize("exact", ["1", "0"], ids=["exact sum (the default)", "tree sum"])
def test_cli_over_rccl_reproduces_reference_stdout(pkg, manifest, exact):
    """bin/papr with its shard's exchanges as RCCL collectives on device buffers, queued on the GPU's stream (PAPR_XCH=rccl:
    papr_exchange_open_rccl_local + papr_exchange_bind — what it uses by default on a multi-GPU node; here a communicator
    of ONE rank, which still executes every collective): every fixture, both tables, stdout / stderr / exit status the
    reference's; and the step did cross RCCL (PAPR_STATS says which transport carried it)."""
    env = dict(os.environ, PAPR_GPUS="1", PAPR_XCH="rccl", PAPR_EXACT_SUM=exact, PAPR_STATS="1")

    def one(case):
        name, graph = case
        args = [pkg.CLI_PATH] + (["-g"] if graph else []) + [golden_path(name)]
        return case, subprocess.run(args, capture_output=True, env=env)

    # (a run is ~2 s of communicator set-up around milliseconds of work: six of them side by side on the one GPU)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=6) as pool:
        names = golden_names()
        if not FULL_TIER:   # every third fixture + the ones with a path of their own (the CLI test above runs all of them)
            names = sorted(set(names[::3]) | {n for n in names if n in ("empty", "odd", "oddb", "stray3", "ties", "g1m", "zeros", "nan_order", "inf_nan", "one", "overflow")})
        results = list(pool.map(one, [(name, graph) for name in names for graph in (False, True)]))
    for (name, graph), p in results:
        want = manifest[name]["graph" if graph else "default"]
        assert p.returncode == want["rc"], (name, graph, p.stderr)
        assert p.stdout == golden_text(name, graph), (name, graph, p.stderr)
        lines = p.stderr.decode().splitlines()
        stats = json.loads(lines[-1])
        assert "\n".join(lines[:-1]) + ("\n" if len(lines) > 1 else "") == want["stderr"]
        assert stats["exchange"] == "rccl" and stats["gpus"] == 1


@pytest.mark.parametrize("env,why", [
    (dict(PAPR_GPUS="1", PAPR_XCH="rccl", PAPR_XCH_BIND_FAIL="all"), "an injected failure"),
    (dict(PAPR_GPUS="1", PAPR_XCH="auto", PAPR_XCH_BIND_DELAY_MS="4000"), "was not done when the shards were loaded"),
    (dict(PAPR_GPUS="1", PAPR_XCH="rccl", PAPR_XCH_BIND_DELAY_MS="4000", PAPR_XCH_BIND_TIMEOUT_S="0.5"), "within PAPR_XCH_BIND_TIMEOUT_S"),
    (dict(PAPR_GPUS="3", PAPR_OVERSUBSCRIBE="1", PAPR_XCH="rccl", PAPR_XCH_BIND_SHARED_OK="1", PAPR_XCH_BIND_FAIL="all"), "an injected failure"),
    (dict(PAPR_GPUS="3", PAPR_OVERSUBSCRIBE="1", PAPR_XCH="rccl", PAPR_XCH_BIND_SHARED_OK="1", PAPR_XCH_BIND_FAIL="1", PAPR_XCH_BIND_TIMEOUT_S="5"),
     "an injected failure"),
    (dict(PAPR_GPUS="3", PAPR_OVERSUBSCRIBE="1", PAPR_XCH_BIND_SHARED_OK="1", PAPR_XCH_BIND_FAIL="all"), None),
], ids=["the set-up fails", "auto: not up in time", "waited for, too slow", "three shards, all fail", "three shards, one fails: the others are left inside ncclCommInitRank",
        "three shards, RCCL not asked for by name: the fall-back is silent"])
def test_cli_survives_an_rccl_setup_that_fails_or_is_late(pkg, manifest, env, why):
    """The communicators come up in threads of their own beside the ingest (papr_exchange_open_rccl_local_async) and are
    taken when the shards are loaded (papr_exchange_adopt_rccl) — all shards or none.  A set-up that fails (injected behind
    PAPR_XCH_BIND_FAIL), is not up when it is needed (PAPR_XCH=auto never waits) or not within PAPR_XCH_BIND_TIMEOUT_S is the
    in-process hub — and ONE line on stderr when RCCL was asked for by name, none by default: stdout, exit status and the rest
    of stderr are the reference's."""
    for name in ("g1m", "odd"):
        for graph in (False, True):
            p = subprocess.run([pkg.CLI_PATH] + (["-g"] if graph else []) + [golden_path(name)], capture_output=True,
                               env=dict(os.environ, PAPR_STATS="1", **env), timeout=120)
            want = manifest[name]["graph" if graph else "default"]
            assert p.returncode == want["rc"] and p.stdout == golden_text(name, graph), (name, graph, p.stderr)
            lines = p.stderr.decode().splitlines()
            said = [l for l in lines if l.startswith("papr: RCCL set-up did not complete")]
            if why is None:   # (the default with more than one GPU: a drop-in's stderr stays the reference's)
                assert said == [], lines
            else:
                assert len(said) == 1 and why in said[0] and "in-process hub" in said[0], lines
            stats = json.loads(lines[-1])
            assert stats["exchange"] != "rccl" and stats["exact_sum"] == 1
            rest = [l for l in lines[:-1] if l not in said]
            assert "\n".join(rest) + ("\n" if rest else "") == want["stderr"]


def test_cli_takes_rccl_when_it_is_up_and_says_what_the_set_up_cost(pkg, manifest):
    """... and when the set-up completes — waited for (PAPR_XCH=rccl), or simply there in time (PAPR_XCH=auto with an ingest
    that takes longer than ncclCommInitRank: here made so with PAPR_XCH_BIND_TIMEOUT_S unused and a generous file) — the
    step's exchanges cross RCCL and PAPR_STATS prices the set-up: how long the thread ran, how long the shard waited."""
    p = subprocess.run([pkg.CLI_PATH, golden_path("g1m")], capture_output=True, timeout=120,
                       env=dict(os.environ, PAPR_STATS="1", PAPR_GPUS="1", PAPR_XCH="rccl"))
    assert p.returncode == 0 and p.stdout == golden_text("g1m", False), p.stderr
    stats = json.loads(p.stderr.decode().splitlines()[-1])
    assert stats["exchange"] == "rccl" and stats["exchange_setup_s"] > 0 and 0 <= stats["exchange_wait_s"] <= stats["exchange_setup_s"] + 0.05
    assert stats["exchange_wait_s"] < stats["total_s"]


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_cli_multi_shard_path(pkg, manifest, shards):
    """The multi-GPU code path of bin/papr (one context + thread per shard, ordered merge, chained
    exact-sum programs, summed counts), exercised by oversubscribing the one GPU of the test box."""
    env = dict(os.environ, PAPR_GPUS=str(shards), PAPR_OVERSUBSCRIBE="1")

    def one(case):
        name, graph = case
        args = [pkg.CLI_PATH] + (["-g"] if graph else []) + [golden_path(name)]
        return case, subprocess.run(args, capture_output=True, env=env)

    # (a run is a second of context set-up per shard around milliseconds of work: four of them side by side on the one GPU)
    from concurrent.futures import ThreadPoolExecutor
    cases = [(name, graph) for name in ("g1m", "ties", "spike20k", "chunk3odd", "tiny", "nan_order", "ofdm_dvbt2_clipped", "one", "empty")
             for graph in (False, True)]
    with ThreadPoolExecutor(max_workers=4) as pool:
        for (name, graph), p in pool.map(one, cases):
            assert p.returncode == 0 and p.stdout == golden_text(name, graph), (name, graph, shards, p.stderr)


def test_cli_option_grammar_on_gpu(pkg):
    p = subprocess.run([pkg.CLI_PATH, "-xGy", golden_path("k8")], capture_output=True)
    assert p.returncode == 0 and p.stderr == b"Unsupported Option: x\nUnsupported Option: y\n"
    assert p.stdout == golden_text("k8", True)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "papr")), reason="no compiled reference")
def test_cli_vs_reference_binary_on_fresh_files(pkg, orc, tmp_path):
    """Live diff against the real reference program on files no fixture covers."""
    rng = np.random.default_rng(5)
    for n, extra in [(300000, ["--spike"]), (1 << 20, []), (777777, ["--extra-floats", "1", "--extra-bytes", "3"]),
                     (50000, ["--scale", "3.3717e-8"]), (2500000, ["--spike", "--seed", "99"])]:
        path = str(tmp_path / "f.cfile")
        subprocess.check_call([orc.MKCFILE, path, str(n), *extra, *(["--seed", str(int(rng.integers(1, 1 << 30)))]
                                                                   if "--seed" not in extra else [])])
        for mode in ([], ["-g"]):
            got = subprocess.run([pkg.CLI_PATH, *mode, path], capture_output=True)
            want = subprocess.run([orc.REF_CLI, *mode, path], capture_output=True)
            assert (got.returncode, got.stdout, got.stderr) == (want.returncode, want.stdout, want.stderr), (n, extra, mode)


def _through_a_fifo(cmd, path, fifo, writer_chunk=1 << 16, env=None):
    """Run `cmd <fifo>` with the bytes of `path` written into the FIFO by a thread (in pieces no larger than a pipe's
    buffer, like a producer would)."""
    import threading
    if os.path.exists(fifo):
        os.unlink(fifo)
    os.mkfifo(fifo)

    def feed():
        with open(path, "rb") as src, open(fifo, "wb") as dst:
            while True:
                b = src.read(writer_chunk)
                if not b:
                    break
                try:
                    dst.write(b)
                except BrokenPipeError:
                    break
    t = threading.Thread(target=feed, daemon=True)
    t.start()
    p = subprocess.run(cmd + [fifo], capture_output=True, timeout=120, env=env)
    t.join(30)
    os.unlink(fifo)
    return p


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "papr")), reason="no compiled reference")
def test_cli_on_a_stream_that_cannot_be_rewound_prints_what_the_reference_prints(pkg, orc, tmp_path):
    """papr.c fopen()s anything (:62, :93).  On a FIFO its pass 1 sees every sample; its fseeko (:142 / :174) fails, EOF
    stays set, pass 2 counts nothing: every percentage prints 0.00000000.  bin/papr reads the stream once into HBM
    (papr_hip_load_stream) and prints the same — stdout, stderr and exit status against the reference binary fed through
    a FIFO of its own; even, odd-float and stray-byte tails, an empty stream, both modes."""
    fifo = str(tmp_path / "in.fifo")
    cases = [(300000, ["--spike"]), (16384 * 3 + 77, ["--extra-floats", "1"]), (777777, ["--extra-floats", "1", "--extra-bytes", "3"]),
             (5000, ["--extra-floats", "1", "--extra-bytes", "2"]), (2500000, ["--spike", "--seed", "99"]), (0, [])]
    for n, extra in cases:
        path = str(tmp_path / "f.cfile")
        if n:
            subprocess.check_call([orc.MKCFILE, path, str(n), *extra, *(["--seed", "4711"] if "--seed" not in extra else [])])
        else:
            open(path, "wb").close()
        for mode in ([], ["-g"]):
            want = _through_a_fifo([orc.REF_CLI, *mode], path, fifo)
            got = _through_a_fifo([pkg.CLI_PATH, *mode], path, fifo)
            assert (got.returncode, got.stdout, got.stderr) == (want.returncode, want.stdout, want.stderr), (n, extra, mode)
            if n:   # the stream's pass 1 is the file's pass 1: same header lines, zero percentages
                on_file = subprocess.run([orc.REF_CLI, *mode, path], capture_output=True).stdout
                assert got.stdout != on_file and len(got.stdout.splitlines()) == len(on_file.splitlines())
                assert b"0.00000000" in got.stdout


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "papr")), reason="no compiled reference")
def test_cli_on_a_stream_longer_than_the_hbm_budget_prints_what_the_reference_prints(pkg, orc, tmp_path):
    """papr.c:100-129 reads a FIFO of ANY length in 64 KiB of memory; so does bin/papr in one window of HBM
    (papr_hip_stream_stats): 40 MiB through a FIFO under a 16 MiB budget (three windows), both modes — even, odd-float and
    stray-byte tails, a stream that ends ON a window's end, and a last window of ONE float whose phantom partner lies in the
    chunk kept from the window before — stdout, stderr and exit status against the reference binary fed through a FIFO of
    its own, the mean from the reference's sequential sum (`exact_sum: 1`) carried from window to window."""
    import json
    fifo = str(tmp_path / "in.fifo")
    mib = (1 << 20) // 8
    cases = [(40 * mib, ["--spike"]), (40 * mib, ["--extra-floats", "1"]), (40 * mib + 12345, ["--extra-floats", "1", "--extra-bytes", "3"]),
             (28 * mib, []), (14 * mib, ["--extra-floats", "1"]), (14 * mib, ["--extra-floats", "1", "--extra-bytes", "1"]),
             (33 * mib + 1, ["--extra-bytes", "2"])]
    env = dict(os.environ, PAPR_HBM_BUDGET_MB="16", PAPR_STATS="1")
    for n, extra in cases:
        path = str(tmp_path / "f.cfile")
        subprocess.check_call([orc.MKCFILE, path, str(n), *extra, "--seed", "3"])
        for mode in ([], ["-g"]):
            want = _through_a_fifo([orc.REF_CLI, *mode], path, fifo)
            got = _through_a_fifo([pkg.CLI_PATH, *mode], path, fifo, env=env)
            err = got.stderr.decode().splitlines()
            info = json.loads(err[-1])
            assert (got.returncode, got.stdout, "\n".join(err[:-1]).encode()) == (want.returncode, want.stdout, want.stderr), (n, extra, mode)
            assert info["exact_sum"] == 1 and info["stream_windows"] >= 2 and info["bytes"] >= n * 8, info
            on_file = subprocess.run([orc.REF_CLI, *mode, path], capture_output=True).stdout
            assert len(got.stdout.splitlines()) == len(on_file.splitlines()) and b"0.00000000" in got.stdout
            if not mode:   # pass 1's lines are the file's, the percentages are the stream's zeros
                assert got.stdout.splitlines()[:4] == on_file.splitlines()[:4]


def test_stream_stats_window_by_window_is_the_files_pass1(pkg, orc, tmp_path):
    """papr_hip_stream_stats through the ABI: a pipe fed 70 MiB + an odd tail under windows of 8 MiB gives the record
    papr_hip_stats gives for the same bytes as one resident shard (trackers and indices bit for bit), and in exact-sum
    mode the ORACLE's sequential sum bit for bit; with NaN in the stream it says the sum is not the emulation's."""
    import threading
    n = (70 << 20) // 8 + 4321
    rng = np.random.default_rng(12)
    iq = (rng.standard_normal(2 * n + 1) * 0.7).astype(np.float32)   # odd float count
    for k in (n // 3, n // 3 + 7, 2 * (n // 3)):   # a peak in the middle of a window, and its ties further on and three windows
        iq[2 * k], iq[2 * k + 1] = 9.5, 0.0       # later: the first index wins across windows too
    path = str(tmp_path / "same.cfile")
    iq.tofile(path)
    want = orc.run_file(path, False)

    def through_a_pipe(data, exact):
        r, w = os.pipe()

        def feed():
            try:
                with os.fdopen(w, "wb") as f:
                    f.write(data)
            except BrokenPipeError:
                pass
        t = threading.Thread(target=feed, daemon=True)   # (a failure below must not leave the session waiting for the writer)
        t.start()
        os.environ["PAPR_STREAM_WINDOW_MB"] = "8"
        try:
            with pkg.PaprHip(0) as g:
                g.set_exact(exact)
                out = g.stream_stats(r)
        finally:
            del os.environ["PAPR_STREAM_WINDOW_MB"]
            os.close(r)      # (a writer still blocked on a full pipe gets EPIPE and ends)
        t.join(30)
        return out
    st, exact, windows = through_a_pipe(iq.tobytes(), True)
    assert exact and windows == 9 and st.n == n + 1 == want["n"]
    assert st.sum == want["sum"], (st.sum.hex(), float(want["sum"]).hex())
    with pkg.PaprHip(0) as g:
        g.load_file(path)
        whole = g.stats()
    for k in ("n", "peak", "peak_idx", "re_pos", "re_neg", "im_pos", "im_neg", "re_pos_idx", "re_neg_idx", "im_pos_idx", "im_neg_idx", "flags"):
        assert getattr(st, k) == getattr(whole, k), k
    assert st.peak_idx == n // 3
    st2, exact2, _ = through_a_pipe(iq.tobytes(), False)
    assert not exact2 and abs(st2.sum - want["sum"]) <= 1e-12 * want["sum"]
    iq[2 * (n // 2) + 1] = np.nan
    st3, exact3, _ = through_a_pipe(iq.tobytes(), True)
    assert not exact3 and np.isnan(st3.sum) and (st3.flags & 1)


@pytest.mark.parametrize("env,lines", [(dict(PAPR_GPUS="1", PAPR_XCH="rccl"), 5), (dict(PAPR_GPUS="3", PAPR_OVERSUBSCRIBE="1"), 2),
                                       (dict(PAPR_GPUS="3", PAPR_OVERSUBSCRIBE="1", PAPR_XCH_IN_STREAM="2"), 5)],
                         ids=["RCCL, one rank", "three threads at the hub", "three threads, the hub standing in for the in-stream collectives"])
def test_cli_runs_the_exchange_self_test_when_asked(pkg, env, lines):
    """PAPR_XCH_SELFTEST=1: in front of its first step with peers bin/papr runs every collective the step uses once, on tiny
    buffers with predictable contents, and rank 0 says so on stderr, one line per collective with its microseconds — what a
    first run on N GPUs prints before anything else; stdout stays the reference's."""
    p = subprocess.run([pkg.CLI_PATH, "-g", golden_path("g1m")], capture_output=True,
                       env=dict(os.environ, PAPR_XCH_SELFTEST="1", PAPR_XCH_TIMEOUT_S="60", **env))
    assert p.returncode == 0 and p.stdout == golden_text("g1m", True), p.stderr
    said = [l for l in p.stderr.decode().splitlines() if l.startswith("papr exchange self-test:")]
    assert len(said) == lines and all(" ok " in l and " us " in l for l in said), p.stderr.decode()
    assert any("host-level all-gather" in l for l in said) and (lines < 5 or any("all-gather-v" in l for l in said))


def test_load_stream_grows_the_shard_and_keeps_the_bytes(pkg, gpu, tmp_path):
    """papr_hip_load_stream through the ABI: a pipe fed 300 MiB + an odd tail (the shard is re-allocated on the way: it
    starts at 256 MiB) holds exactly the file's samples, the phantom sample included."""
    n = (300 << 20) // 8 + 12345
    rng = np.random.default_rng(11)
    iq = rng.standard_normal(2 * n + 1).astype(np.float32)   # odd float count
    r, w = os.pipe()
    import threading

    def feed():
        try:
            with os.fdopen(w, "wb") as f:
                f.write(iq.tobytes())
        except BrokenPipeError:
            pass
    t = threading.Thread(target=feed, daemon=True)
    t.start()
    try:
        got_n = gpu.load_stream(r)
    finally:
        os.close(r)
    t.join(30)
    path = str(tmp_path / "same.cfile")
    iq.tofile(path)
    assert got_n == n + 1
    a = gpu.download(0, got_n)
    st_stream = gpu.stats()
    gpu.load_file(path)
    assert np.array_equal(a.view(np.uint32), gpu.download(0, got_n).view(np.uint32))
    assert st_stream.to_bytes() == gpu.stats().to_bytes()


def test_load_stream_beyond_the_budget_is_an_error_that_leaves_the_context_usable(pkg, orc, tmp_path):
    """papr_hip_load_stream keeps the WHOLE stream resident (for a caller who wants pass 2 of it): beyond the HBM budget it is
    PAPR_E_NOMEM — through its clean-up (the growing shard and the events are given back) — and the context then
    loads and analyses a file as if nothing had happened; papr_hip_stream_stats takes the same stream in windows."""
    import threading
    n = (40 << 20) // 8
    path = str(tmp_path / "f.cfile")
    subprocess.check_call([orc.MKCFILE, path, str(n), "--seed", "5", "--spike"])
    data = open(path, "rb").read()
    want = orc.run_file(path, False)

    def pipe_with(data):
        r, w = os.pipe()

        def feed():
            try:
                with os.fdopen(w, "wb") as f:
                    f.write(data)
            except BrokenPipeError:
                pass
        t = threading.Thread(target=feed, daemon=True)
        t.start()
        return r, t
    os.environ["PAPR_HBM_BUDGET_MB"] = "16"
    try:
        with pkg.PaprHip(0) as g:
            g.set_exact(True)
            r, t = pipe_with(data)
            try:
                with pytest.raises(pkg.PaprError) as e:
                    g.load_stream(r)
            finally:
                os.close(r)
            t.join(30)
            assert e.value.code == -4 and "HBM budget" in str(e.value)
            r, t = pipe_with(data)
            try:
                st, exact, windows = g.stream_stats(r)
            finally:
                os.close(r)
            t.join(30)
            assert exact and windows == 3 and st.n == n and st.sum == want["sum"] and st.peak_idx == want["peak_idx"]
            g.load_file(golden_path("g1m"))          # (a shard again, in the same context)
            res, table, counts = g.analyze(None, False)
            ref = orc.run_file(golden_path("g1m"), False)
            assert res.total.n == ref["n"] and res.total.sum == ref["sum"] and res.exact_sum == 1
    finally:
        del os.environ["PAPR_HBM_BUDGET_MB"]


# ---- seeded random inputs vs the oracle, every launch geometry --------------------

SIZES = [1, 2, 63, 64, 255, 4095, 4096, 4097, 8191, 12289, 100003, 1048576 + 5]


@pytest.mark.parametrize("n", SIZES)
def test_upload_vs_oracle_sizes(pkg, orc, gpu, n):
    rng = np.random.default_rng(n)
    iq = rng.standard_normal(2 * n).astype(np.float32)
    gpu.upload(iq)
    for graph in (False, True):
        st, *_ = run_both_passes(pkg, orc, gpu, iq, graph)
        check_stats(st, orc.run_mem(iq, graph))


@pytest.mark.parametrize("tune", [dict(blocks=1), dict(blocks=7), dict(blocks=256, map=1), dict(blocks=2048, map=2),
                                  dict(blocks=1024, map=2, nontemporal=0, variant=0), dict(blocks=64, map=1, hist_copies=1),
                                  dict(blocks=4096, map=0, hist_copies=2), dict(blocks=300, map=2), dict(flags=1)]
                         + [dict(variant=v, map=m, blocks=b) for v in (0, 1, 4, 6, 13, 14) for m, b in ((0, 0), (1, 96), (2, 1000))],
                         ids=str)
def test_launch_geometries_agree(pkg, orc, gpu, tune):
    n = 3 * 1048576 + 4099
    gpu.generate(pkg.SynthSpec.spike(n, seed=77), 0, n)
    iq = gpu.download(0, n)
    ref = orc.run_mem(iq, True)
    try:
        gpu.set_tuning(**tune)
        st, *_ = run_both_passes(pkg, orc, gpu, iq, True)
        check_stats(st, ref)
    finally:
        gpu.set_tuning()


def test_generate_matches_host_generator(pkg, orc, gpu, tmp_path):
    n, first = 200001, 123456789012
    path = str(tmp_path / "g.cfile")
    sp = pkg.SynthSpec.make(seed=4242, scale=0.0, overrides=[(first + 5, 9.0, -9.0), (first + n - 1, float("inf"), 1.0)])
    gpu.generate(sp, first, n)
    got = gpu.download(0, n)
    # host side: the same inline generator through mkcfile on a window is not
    # index-offsettable from the CLI, so compare a low window of the stream
    gpu.generate(pkg.SynthSpec.make(seed=4242), 0, 5000)
    low = gpu.download(0, 5000)
    subprocess.check_call([orc.MKCFILE, path, "5000", "--seed", "4242"])
    assert np.array_equal(low, np.fromfile(path, dtype=np.float32))
    assert got[10] == 9.0 and got[11] == -9.0 and np.isinf(got[2 * (n - 1)])
    # shard-independence: a shard generated at an offset equals that range of a longer one
    gpu.generate(pkg.SynthSpec.make(seed=4242), 1000, 3000)
    assert np.array_equal(gpu.download(0, 3000), low[2000:8000])


def test_global_indices_follow_base_index(pkg, orc, gpu):
    n, base = 50000, (1 << 33) + 8192
    rng = np.random.default_rng(3)
    iq = rng.standard_normal(2 * n).astype(np.float32)
    gpu.upload(iq, base_index=base)
    check_stats(gpu.stats(), orc.run_mem(iq, False), base=base)


def test_shards_merge_to_whole_bit_exactly(pkg, orc, gpu):
    """Sharding invariance (what the multi-GPU path relies on): any ordered
    split, merged with papr_stats_merge / summed counts, equals the single-shard
    run — trackers, indices and counts exactly."""
    n = 2 * 1048576 + 77
    gpu.generate(pkg.SynthSpec.spike(n, seed=5), 0, n)
    iq = gpu.download(0, n)
    whole = gpu.stats()
    mean, papr, table = pkg.levels(whole, True)
    whole_counts = gpu.ccdf(table)
    for world in (2, 3, 8):
        from dtv_utils_amd import exchange
        parts, counts = [], np.zeros(table.size, dtype=np.uint64)
        for r in range(world):
            first, cnt = exchange.shard_range(n, r, world)
            gpu.upload(iq[2 * first:2 * (first + cnt)], base_index=first)
            parts.append(gpu.stats())
            counts += gpu.ccdf(table)
        m = pkg.stats_merge(parts)
        for k in TRACKERS:
            assert getattr(m, k) == getattr(whole, k) and getattr(m, k + "_idx") == getattr(whole, k + "_idx")
        assert m.n == whole.n and abs(m.sum - whole.sum) <= SUM_RTOL * whole.sum
        assert np.array_equal(counts, whole_counts)


def test_first_index_wins_across_lanes_waves_and_blocks(pkg, orc, gpu):
    """Equal extremes planted in different lanes / waves / workgroups / tiles."""
    n = 5 * 4096 * 64 + 1000
    iq = np.full(2 * n, 0.25, dtype=np.float32)
    spots = [4096 * 40 + 2049, 4096 * 40 + 2048, 17, 4096 * 64 * 4 + 999, n - 1, 4096 * 3 + 1]
    for s in spots:
        iq[2 * s], iq[2 * s + 1] = 3.0, -4.0
    gpu.upload(iq)
    for tune in (dict(), dict(blocks=3), dict(blocks=64, map=1), dict(blocks=64, map=2)):
        gpu.set_tuning(**tune)
        st = gpu.stats()
        assert st.peak == 25.0 and st.peak_idx == 17 and st.re_pos_idx == 17 and st.im_neg_idx == 17
        assert st.re_neg == 0.0 and st.re_neg_idx == 0 and st.im_pos == 0.25 and st.im_pos_idx == 0
    gpu.set_tuning()


# ---- pass 2 with arbitrary level tables ---------------------------------------------

def test_ccdf_arbitrary_level_tables(pkg, orc, gpu):
    n = 300007
    rng = np.random.default_rng(9)
    iq = (rng.standard_normal(2 * n) * 0.7).astype(np.float32)
    iq[2 * 5], iq[2 * 6] = np.nan, np.inf
    iq[2 * 7:2 * 9] = 0.0
    gpu.upload(iq)
    tables = {
        "unsorted_dups": np.array([2.0, 0.5, 0.5, 1.0, 4.0, 0.25, 1.0], np.float32),
        "special": np.array([-1.0, 0.0, -0.0, np.nan, np.inf, -np.inf, 1e-45, 1e-39, 3.4e38, 1.0], np.float32),
        "dense": np.linspace(0.9, 1.1, 700).astype(np.float32),            # closer than one LUT cell: search kernel
        "huge": (10 ** np.linspace(-6, 3, 5000)).astype(np.float32),       # 5000 levels
        "one": np.array([1.0], np.float32),
        "wide": (10 ** np.linspace(-30, 30, 200)).astype(np.float32),
        "denormals": np.array([1e-44, 1e-42, 1e-40, 1e-38], np.float32),
        # a level equal to FLT_MAX puts the top LUT cell past +Inf: nothing but +Inf itself is above it
        "flt_max": np.array([1.0, np.finfo(np.float32).max, 2.0], np.float32),
        "flt_max_only": np.array([np.finfo(np.float32).max, np.float32(3.0e38)], np.float32),
    }
    for tag, tab in tables.items():
        got = gpu.ccdf(tab)
        assert np.array_equal(got.astype(np.int64), orc.count_mem(iq, tab)), tag
    assert gpu.ccdf(np.zeros(0, np.float32)).size == 0
    with pytest.raises(pkg.PaprError) as e:
        gpu.ccdf(np.ones(pkg.MAX_LEVELS + 1, np.float32))
    assert e.value.code == -7


def test_no_fma_contraction(pkg, orc, gpu):
    """fma(I,I,Q*Q) differs from fl(fl(I*I)+fl(Q*Q)) on ~16 % of samples; the
    exact peak and exact counts against thresholds placed ON sample powers
    would expose a contracted multiply-add."""
    n = 1 << 18
    rng = np.random.default_rng(1234)
    iq = rng.standard_normal(2 * n).astype(np.float32)
    pw = (iq[0::2] * iq[0::2]) + (iq[1::2] * iq[1::2])          # numpy float32: separate roundings
    fused = (iq[0::2].astype(np.float64) ** 2 + (iq[1::2] * iq[1::2]).astype(np.float64)).astype(np.float32)
    assert np.count_nonzero(pw != fused) > n // 20               # the test has teeth
    table = np.sort(pw[rng.integers(0, n, 400)])
    gpu.upload(iq)
    assert np.array_equal(gpu.ccdf(table).astype(np.int64), orc.count_mem(iq, table))
    st = gpu.stats()
    assert st.peak == pw.max() and st.peak_idx == int(np.argmax(pw))


def test_adopt_external_device_memory(pkg, orc, gpu):
    import torch
    n = 70001
    t = torch.randn(2 * n, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    gpu.adopt(t.data_ptr(), n, base_index=0, keepalive=t)
    iq = t.cpu().numpy()
    st, *_ = run_both_passes(pkg, orc, gpu, iq, False)
    check_stats(st, orc.run_mem(iq, False))
    with pytest.raises(pkg.PaprError):
        gpu.adopt(t.data_ptr() + 4, 10)


def test_error_states(pkg):
    with pkg.PaprHip(0) as g:
        with pytest.raises(pkg.PaprError) as e:
            g.stats()
        assert e.value.code == -6
        with pytest.raises(pkg.PaprError) as e:
            g.ccdf(np.ones(3, np.float32))
        assert e.value.code == -6
        with pytest.raises(pkg.PaprError) as e:
            g.load_file("/nonexistent/x.cfile")
        assert e.value.code == -5
    with pytest.raises(pkg.PaprError) as e:
        pkg.PaprHip(9999)
    assert e.value.code == -1


# ---- ingest: chunked, sharded, and re-streamed when the shard exceeds the budget ----

def test_file_ingest_chunking_and_sharding(pkg, orc, tmp_path, monkeypatch):
    n = 3 * 1048576 + 12345
    path = str(tmp_path / "big.cfile")
    subprocess.check_call([orc.MKCFILE, path, str(n), "--spike", "--extra-floats", "1", "--extra-bytes", "2"])
    ref = orc.run_file(path, True)
    monkeypatch.setenv("PAPR_CHUNK_MB", "1")           # many chunks
    for budget in (None, "4"):                        # resident, then re-streamed (4 MiB HBM budget)
        if budget:
            monkeypatch.setenv("PAPR_HBM_BUDGET_MB", budget)
        with pkg.PaprHip(0) as g:
            g.load_file(path)
            st = g.stats()
            check_stats(st, ref)
            assert st.flags & pkg.FLAG_ODD_TAIL
            mean, papr, table = pkg.levels(st, True)
            assert np.array_equal(g.ccdf(table).astype(np.int64), ref["count"]) and np.array_equal(table, ref["level"])
            # two shards of the same file, merged
            from dtv_utils_amd import exchange
            total_n = pkg.file_samples(path)
            parts, counts = [], np.zeros(table.size, np.uint64)
            for r in range(2):
                first, cnt = exchange.shard_range(total_n, r, 2)
                g.load_file(path, first, cnt)
                parts.append(g.stats())
                counts += g.ccdf(table)
            check_stats(pkg.stats_merge(parts), ref)
            assert np.array_equal(counts.astype(np.int64), ref["count"])


def _disk_dir():
    """A directory on a real block-device filesystem (O_DIRECT is refused by tmpfs), or None."""
    for cand in (os.environ.get("PAPR_TEST_DISK_DIR"), os.path.join(ROOT, "gpurun_out"), "/var/tmp", "/root"):
        if not cand:
            continue
        try:
            os.makedirs(cand, exist_ok=True)
            probe = os.path.join(cand, f".papr_odirect_probe_{os.getpid()}")
            with open(probe, "wb") as f:
                f.write(b"\0" * 4096)
            try:
                fd = os.open(probe, os.O_RDONLY | os.O_DIRECT)
                os.close(fd)
                return cand
            except OSError:
                pass
            finally:
                os.unlink(probe)
        except OSError:
            continue
    return None


def _io_uring_offered():
    """does io_uring_setup work here (kernel + seccomp profile)?  (425 = __NR_io_uring_setup on x86-64)"""
    import ctypes
    libc = ctypes.CDLL(None, use_errno=True)
    params = (ctypes.c_ubyte * 120)()
    fd = libc.syscall(425, 4, params)
    if fd < 0:
        return False
    os.close(fd)
    return True


@pytest.mark.parametrize("uring", ["1", "0"], ids=["io_uring", "reader-threads"])
@pytest.mark.parametrize("budget", [None, "4"], ids=["resident", "streamed"])
def test_file_ingest_o_direct(pkg, orc, monkeypatch, budget, uring):
    """PAPR_O_DIRECT=1 (what a cold file >= 64 MiB gets automatically), through one io_uring where the kernel offers it
    and through the reader threads (PAPR_IO_URING=0): 4 KiB-aligned direct reads into the pinned
    staging buffers, the unaligned end of the file finished by the buffered descriptor, odd-float tail + stray
    bytes patched after a direct read, a shard whose start is not 4 KiB-aligned (falls through to buffered reads),
    many chunks; resident and re-streamed.  Everything against the oracle on the same file."""
    d = _disk_dir()
    if d is None:
        pytest.skip("no filesystem here accepts O_DIRECT")
    n = 3 * 1048576 + 12345
    path = os.path.join(d, f"papr_odirect_{os.getpid()}.cfile")
    try:
        subprocess.check_call([orc.MKCFILE, path, str(n), "--spike", "--extra-floats", "1", "--extra-bytes", "3"])
        ref = orc.run_file(path, True)
        monkeypatch.setenv("PAPR_O_DIRECT", "1")
        monkeypatch.setenv("PAPR_IO_URING", uring)
        monkeypatch.setenv("PAPR_CHUNK_MB", "1")
        monkeypatch.setenv("PAPR_READ_THREADS", "5")    # slices that are not 4 KiB multiples of each other
        if budget:
            monkeypatch.setenv("PAPR_HBM_BUDGET_MB", budget)
        with pkg.PaprHip(0) as g:
            g.load_file(path)
            t = g.ingest_timing()
            assert t.o_direct == 1 and t.resident == (0 if budget else 1), t.as_dict()
            assert t.io_uring == (int(uring) if _io_uring_offered() else 0), t.as_dict()
            st = g.stats()
            check_stats(st, ref)
            assert st.flags & pkg.FLAG_ODD_TAIL
            mean, papr, table = pkg.levels(st, True)
            assert np.array_equal(table, ref["level"])
            assert np.array_equal(g.ccdf(table).astype(np.int64), ref["count"])
            if not budget:
                # the resident bytes are the file's bytes (plus the phantom partner), nothing read past `want` leaked in
                got = g.download(0, st.n)
                want = np.fromfile(path, dtype=np.float32, count=2 * st.n - 1)
                assert np.array_equal(got[:-1].view(np.uint32), want.view(np.uint32))
            # shards that start off a 4 KiB boundary (sample 8192 * k is aligned; 8192 * k + 100 is not) and a
            # one-sweep ingest through the same descriptor
            from dtv_utils_amd import exchange
            parts, counts = [], np.zeros(table.size, np.uint64)
            cuts = [0, 8192 * 37 + 100, 8192 * 200, pkg.file_samples(path)]
            for a, b in zip(cuts[:-1], cuts[1:]):
                g.load_file(path, a, b - a)
                assert g.ingest_timing().o_direct == 1
                parts.append(g.stats())
                counts += g.ccdf(table)
            check_stats(pkg.stats_merge(parts), ref)
            assert np.array_equal(counts.astype(np.int64), ref["count"])
            est = g.estimate_file(path)
            g.load_file_sweep(path, pkg.guess_levels(est, True))
            assert g.ingest_timing().o_direct == 1
            st2 = g.stats()
            check_stats(st2, ref)
            assert np.array_equal(g.ccdf(pkg.levels(st2, True)[2]).astype(np.int64), ref["count"])
        # the CLI on the same file, same switch
        for graph in (False, True):
            want = subprocess.run([orc.REF_CLI if os.path.exists(orc.REF_CLI) else orc.CLI_PATH] +
                                  (["-g"] if graph else []) + [path], capture_output=True)
            got = subprocess.run([pkg.CLI_PATH] + (["-g"] if graph else []) + [path], capture_output=True,
                                 env=dict(os.environ, PAPR_STATS="1"))
            assert got.stdout == want.stdout and got.returncode == 0
            assert b'"o_direct": 1' in got.stderr
    finally:
        if os.path.exists(path):
            os.unlink(path)


def test_nan_first_index_and_sign_when_streamed(pkg, orc, tmp_path, monkeypatch):
    path = str(tmp_path / "nan.cfile")
    subprocess.check_call([orc.MKCFILE, path, "600000", "--set", "500000", "nan", "1", "--set", "300001", "1", "-nan"])
    ref = orc.run_file(path, False)
    monkeypatch.setenv("PAPR_CHUNK_MB", "1")
    for budget in (None, "2"):
        if budget:
            monkeypatch.setenv("PAPR_HBM_BUDGET_MB", budget)
        with pkg.PaprHip(0) as g:
            g.load_file(path)
            st = g.stats()
            check_stats(st, ref)
            assert st.nan_first_idx == 300001 and st.nan_first_neg == 1 and st.flags & pkg.FLAG_NAN


# ---- BASELINE.json full size: 10 GiB resident shard -----------------------------------

@pytest.fixture(scope="module")
def big(pkg):
    n = 1342177280  # 10 GiB of gr_complex
    g = pkg.PaprHip(0)
    g.generate(pkg.SynthSpec.spike(n), 0, n)
    yield g, n
    g.close()


@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
def test_full_size_matches_reference_stdout(pkg, manifest, big, graph):
    """configs[1]/[2]: the 10 GiB workload, generated on the device from the
    shared generator, against the stdout the real reference printed for the same
    stream (tests/golden/big_spike10g.*.txt)."""
    if "big_spike10g" not in manifest:
        pytest.skip("full-size golden not recorded")
    g, n = big
    assert manifest["big_spike10g"]["nsamples"] == n
    st = g.stats()
    mean, papr, table = pkg.levels(st, graph)
    counts = g.ccdf(table)
    got = pkg.format_report(st, mean, papr, counts, graph).encode()
    assert got == golden_text("big_spike10g", graph)


@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
def test_full_size_one_sweep_matches_reference_stdout(pkg, manifest, big, graph):
    """The path bench.py times by default — estimate -> guessed bands -> ONE sweep -> stash recount — on the
    10 GiB workload against the stdout the real reference printed for it, answered from the sweep (no second read)."""
    if "big_spike10g" not in manifest:
        pytest.skip("full-size golden not recorded")
    g, n = big
    est = g.estimate()
    st = g.stats_sweep(pkg.guess_levels(est, graph))
    info = g.sweep_info()
    assert info.swept == 1, info.as_dict()
    mean, papr, table = pkg.levels(st, graph)
    counts = g.ccdf(table)
    info = g.sweep_info()
    assert info.resolved == 1, info.as_dict()
    assert pkg.format_report(st, mean, papr, counts, graph).encode() == golden_text("big_spike10g", graph)
    # and the sweep's pass-1 record equals the two-pass kernel's, indices included
    two = g.stats()
    for k in TRACKERS:
        assert getattr(st, k) == getattr(two, k) and getattr(st, k + "_idx") == getattr(two, k + "_idx"), k
    assert np.array_equal(g.ccdf(table), counts) and g.sweep_info().resolved == 0   # plain pass 2: same counts


def test_full_size_properties(pkg, orc, big):
    g, n = big
    st = g.stats()
    sp = pkg.SynthSpec.spike(n)
    # the two equal spikes: the first one must win
    assert st.peak == np.float32(36.9375) ** 2 and st.peak_idx == min(sp.ov[0].index, sp.ov[1].index)
    assert st.re_pos == np.float32(36.9375) and st.re_pos_idx == st.peak_idx
    mean, papr, table = pkg.levels(st, True)
    assert 300 <= table.size <= 303 and abs(papr - 30.1) < 0.05
    counts = g.ccdf(table)
    assert counts[0] <= n and np.all(np.diff(counts.astype(np.int64)) <= 0) and counts[-1] <= 2
    # determinism: repeated runs are bit-identical, sum included
    st2 = g.stats()
    assert st2.to_bytes() == st.to_bytes() and np.array_equal(g.ccdf(table), counts)
    # geometry independence of everything but the last bits of the sum
    g.set_tuning(blocks=1536, map=2)
    st3 = g.stats()
    g.set_tuning()
    for k in TRACKERS:
        assert getattr(st3, k) == getattr(st, k) and getattr(st3, k + "_idx") == getattr(st, k + "_idx")
    assert abs(st3.sum - st.sum) <= SUM_RTOL * st.sum
    # oracle on a window of the very same device data
    w0, wn = 987654321, 1 << 20
    iq = g.download(w0, wn)
    tab = table[::10]
    sub = pkg.PaprHip(0)
    try:
        sub.upload(iq, base_index=w0)
        check_stats(sub.stats(), orc.run_mem(iq, False), base=w0)
        assert np.array_equal(sub.ccdf(tab).astype(np.int64), orc.count_mem(iq, tab))
    finally:
        sub.close()


# ---- bit-exact mean: the reference's sequential double sum, reproduced on the GPU ------

def exact_sum_one_shard(pkg, g, n_total=None):
    st = g.stats()
    prog = g.exact_program(0.0, n_total or st.n)
    return st, pkg.exact_chain([prog]), prog


@pytest.fixture(scope="module")
def xgpu(pkg):
    g = pkg.PaprHip(0)
    g.set_exact(True)
    yield g
    g.close()


@pytest.mark.parametrize("n", [1, 5, 2047, 2048, 2049, 6000, 300007, 128 * 2048 + 17, 4 * 128 * 2048, 5000011])
def test_exact_sum_equals_sequential_sum_sizes(pkg, orc, xgpu, n):
    rng = np.random.default_rng(1000 + n)
    iq = rng.standard_normal(2 * n).astype(np.float32)
    xgpu.upload(iq)
    st, exact, _ = exact_sum_one_shard(pkg, xgpu)
    ref = orc.run_mem(iq, False)
    assert exact == ref["sum"], (exact.hex(), ref["sum"].hex(), st.sum.hex())
    check_stats(st, ref)   # exact mode leaves pass 1's other results unchanged


@pytest.mark.parametrize("case", ["spike", "tiny", "huge", "zeros_first", "zeros_mixed", "growing", "shrinking",
                                  "constant", "denormal", "sparse"])
def test_exact_sum_hard_inputs(pkg, orc, xgpu, case):
    import zlib
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    n = 1_500_000
    iq = rng.standard_normal(2 * n).astype(np.float32)
    if case == "spike":
        iq[2 * 777777] = 3.0e4
    elif case == "tiny":
        iq *= np.float32(3.3717e-8)
    elif case == "huge":
        iq *= np.float32(1e15)
    elif case == "zeros_first":
        iq[: 2 * 700000] = 0
    elif case == "zeros_mixed":
        iq[2 * 100000: 2 * 900000] = 0
        iq[2 * 1200000: 2 * 1200100] = 0
    elif case == "growing":      # amplitude x 2^40 over the file: the sum changes binade constantly
        iq *= np.exp2(np.repeat(np.linspace(-20, 20, n), 2)).astype(np.float32)
    elif case == "shrinking":
        iq *= np.exp2(np.repeat(np.linspace(20, -20, n), 2)).astype(np.float32)
    elif case == "constant":     # every addition is a potential tie
        iq[0::2], iq[1::2] = 0.75, 0.25
    elif case == "denormal":
        iq *= np.float32(4e-21)
    elif case == "sparse":
        mask = rng.random(n) < 0.999
        iq[0::2][mask] = 0
        iq[1::2][mask] = 0
    xgpu.upload(iq)
    st, exact, prog = exact_sum_one_shard(pkg, xgpu)
    ref = orc.run_mem(iq, False)
    assert exact == ref["sum"], (case, exact.hex(), ref["sum"].hex(), st.sum.hex())
    assert len(prog) < 8 * n // 2, "the program must stay a small fraction of the data"


def test_exact_sum_golden_fixtures(pkg, orc):
    """Exact mode through the file ingest on every fixture with a finite sum."""
    with pkg.PaprHip(0) as g:
        g.set_exact(True)
        for name in golden_names():
            ref = orc.run_file(golden_path(name), False)
            if not np.isfinite(ref["sum"]):
                continue
            g.load_file(golden_path(name))
            st = g.stats()
            assert pkg.exact_chain([g.exact_program(0.0, st.n)]) == ref["sum"], name


def test_exact_sum_across_shards(pkg, orc, xgpu):
    """Multi-GPU form: every shard builds its program knowing only the ACCURATE sum of the
    shards before it; the chained result is the sequential sum of the whole stream."""
    from dtv_utils_amd import exchange
    n = 3_000_017
    xgpu.generate(pkg.SynthSpec.spike(n, seed=31), 0, n)
    iq = xgpu.download(0, n)
    ref = orc.run_mem(iq, False)
    for world in (1, 2, 5):
        progs, before = [], 0.0
        for r in range(world):
            first, cnt = exchange.shard_range(n, r, world)
            xgpu.upload(iq[2 * first:2 * (first + cnt)], base_index=first)
            st = xgpu.stats()
            progs.append(xgpu.exact_program(before, n))
            before += st.sum
        assert pkg.exact_chain(progs) == ref["sum"], world


def test_exact_program_states(pkg):
    with pkg.PaprHip(0) as g:
        g.upload(np.ones(20000, np.float32))
        g.stats()
        with pytest.raises(pkg.PaprError) as e:
            g.exact_program()          # exact mode was not enabled
        assert e.value.code == -6
        g.set_exact(True)
        with pytest.raises(pkg.PaprError):
            g.exact_program()          # stats not re-run in exact mode
        g.stats()
        assert pkg.exact_chain([g.exact_program()]) == 20000.0
    with pytest.raises(pkg.PaprError):
        pkg.exact_chain([b"garbage" * 10])


def test_exact_sum_full_size(pkg, manifest, big):
    """10 GiB: the GPU-reproduced sequential sum equals the oracle's, bit for bit."""
    want = manifest.get("big_spike10g", {}).get("oracle_sequential_sum_hex")
    if not want:
        pytest.skip("full-size sequential sum not recorded")
    g, n = big
    g.set_exact(True)
    try:
        st = g.stats()
        g.set_timing(True)
        prog = g.exact_program(0.0, n)
        tm = g.timing()
        g.set_timing(False)
        exact = pkg.exact_chain([prog])
        assert exact == float.fromhex(want), (exact.hex(), want, st.sum.hex())
        # the tree sum is the accurate one: at 1.3e9 terms the reference's serial accumulator has drifted
        # ~1e-11 relative from it, which is why the exact emulation exists
        assert abs(st.sum - exact) <= 1e-10 * exact
        print(f"exact program: {len(prog)} bytes, device {tm.exact_ms:.3f} ms")
    finally:
        g.set_exact(False)


@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
@pytest.mark.parametrize("n", [1000, 2048 * 3 + 5, 700001, 3 * 1048576 + 123])
def test_fused_pass2_and_sum_program(pkg, orc, xgpu, n, graph):
    """papr_hip_ccdf_exact: ONE sweep yields pass 2's counts and the sum program."""
    xgpu.generate(pkg.SynthSpec.spike(n, seed=900 + n % 97), 0, n)
    iq = xgpu.download(0, n)
    st = xgpu.stats()
    mean, papr, table = pkg.levels(st, graph)
    counts, prog = xgpu.ccdf_exact(table, 0.0, n)
    ref = orc.run_mem(iq, graph)
    assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, table))
    assert pkg.exact_chain([prog]) == ref["sum"]
    st.sum = ref["sum"]
    assert np.array_equal(pkg.levels(st, graph)[2], ref["level"])


def test_fused_pass2_falls_back_for_unusual_tables(pkg, orc, xgpu):
    n = 200003
    rng = np.random.default_rng(77)
    iq = rng.standard_normal(2 * n).astype(np.float32)
    xgpu.upload(iq)
    xgpu.stats()
    ref_sum = orc.run_mem(iq, False)["sum"]
    for tab in (np.linspace(0.9, 1.1, 700).astype(np.float32),              # search form
                np.array([-1.0, 0.0, np.nan, 1.0], np.float32), np.zeros(0, np.float32),
                (10 ** np.linspace(-6, 3, 5000)).astype(np.float32)):
        counts, prog = xgpu.ccdf_exact(tab, 0.0, n)
        assert np.array_equal(counts.astype(np.int64), orc.count_mem(iq, tab))
        assert pkg.exact_chain([prog]) == ref_sum


def test_exact_program_host_assembly_fallback(pkg, orc, xgpu, monkeypatch):
    """The host-driven program assembly (used when a shard has more mixed groups / raw tiles than the
    device-side lists hold) produces the same program as the device-side gather."""
    n = 2_000_003
    xgpu.generate(pkg.SynthSpec.spike(n, seed=4), 0, n)
    iq = xgpu.download(0, n)
    xgpu.stats()
    dev = xgpu.exact_program(0.0, n)
    monkeypatch.setenv("PAPR_EXACT_HOST_ASSEMBLY", "1")
    host = xgpu.exact_program(0.0, n)
    assert host == dev
    assert pkg.exact_chain([host]) == orc.run_mem(iq, False)["sum"]


def test_exact_sum_on_restreamed_shard(pkg, orc, tmp_path, monkeypatch):
    """Shard larger than the HBM budget (BASELINE configs[4]): pass 1 streams the file leaving per-tile sums,
    pass 2 streams it again with the fused sweep; counts and the sequential sum must still be exact."""
    n = 5 * 1048576 + 4321
    path = str(tmp_path / "big.cfile")
    subprocess.check_call([orc.MKCFILE, path, str(n), "--spike", "--extra-floats", "1", "--extra-bytes", "1"])
    ref = orc.run_file(path, True)
    monkeypatch.setenv("PAPR_CHUNK_MB", "2")
    monkeypatch.setenv("PAPR_HBM_BUDGET_MB", "8")
    with pkg.PaprHip(0) as g:
        g.set_exact(True)
        g.load_file(path)
        assert not g.ingest_timing().resident
        st = g.stats()
        check_stats(st, ref)
        mean, papr, table = pkg.levels(st, True)
        counts, prog = g.ccdf_exact(table, 0.0, st.n)
        assert pkg.exact_chain([prog]) == ref["sum"]
        st.sum = ref["sum"]
        assert np.array_equal(pkg.levels(st, True)[2], ref["level"])
        if np.array_equal(table, ref["level"]):
            assert np.array_equal(counts.astype(np.int64), ref["count"])
    # and the CLI end to end under the same budget
    for graph in (False, True):
        want = subprocess.run([orc.REF_CLI if os.path.exists(orc.REF_CLI) else orc.CLI_PATH] + (["-g"] if graph else []) + [path],
                              capture_output=True)
        got = subprocess.run([pkg.CLI_PATH] + (["-g"] if graph else []) + [path], capture_output=True,
                             env=dict(os.environ, PAPR_STATS="1"))
        assert got.stdout == want.stdout and got.returncode == 0
        assert b'"exact_sum": 1' in got.stderr and b'"resident": 0' in got.stderr


def test_shard_beyond_2_to_32_samples(pkg):
    """A 34 GiB shard (4.56e9 samples > 2^32) on one GPU: in-shard sample numbers need all 64 bits.
    No CPU oracle at this size; checked through planted extremes, shard-split invariance and the
    agreement of the exact-sum chain with and without a cut."""
    n = 34 * (1 << 30) // 8
    first, second = (1 << 32) + 5, (1 << 32) + 123456789
    sp = pkg.SynthSpec.make(seed=2026, overrides=[(second, 50.0, -50.0), (first, 50.0, -50.0), (n - 1, -60.0, 1.0),
                                                  (7, 0.0, 55.0)])
    with pkg.PaprHip(0) as g:
        g.set_exact(True)
        g.generate(sp, 0, n)
        st = g.stats()
        assert st.n == n and st.peak == np.float32(5000.0) and st.peak_idx == first      # first of the two equal peaks
        assert (st.re_pos, st.re_pos_idx) == (50.0, first) and (st.im_neg, st.im_neg_idx) == (-50.0, first)
        assert (st.re_neg, st.re_neg_idx) == (-60.0, n - 1) and (st.im_pos, st.im_pos_idx) == (55.0, 7)
        mean, papr, table = pkg.levels(st, True)
        counts, prog = g.ccdf_exact(table, 0.0, n)
        whole = pkg.exact_chain([prog])
        assert abs(whole - st.sum) <= 1e-9 * st.sum and np.all(np.diff(counts.astype(np.int64)) <= 0)
        assert counts[0] < n and counts[-1] <= 4
        # the one-sweep path on the same 34 GiB shard: same record (indices beyond 2^32), same counts
        g.set_exact(False)
        two = g.stats()
        sw = g.stats_sweep(pkg.guess_levels(g.estimate(), True))
        assert g.sweep_info().swept == 1 and sw.n == two.n and abs(sw.sum - two.sum) <= SUM_RTOL * two.sum
        for k in TRACKERS:
            assert getattr(sw, k) == getattr(two, k) and getattr(sw, k + "_idx") == getattr(two, k + "_idx"), k
        assert sw.peak_idx == first and sw.re_neg_idx == n - 1 and sw.im_pos_idx == 7
        mean_t, papr_t, table_t = pkg.levels(sw, True)
        swept_counts = g.ccdf(table_t)
        assert g.sweep_info().resolved == 1, g.sweep_info().as_dict()
        g.stats()                                   # drops the sweep: the next ccdf reads the shard
        assert np.array_equal(swept_counts, g.ccdf(table_t)) and g.sweep_info().resolved == 0
        g.set_exact(True)
        # the same stream as two shards cut beyond 2^32: merged record, summed counts and chained exact sum agree
        cut = ((1 << 32) + 99999) // 8192 * 8192
        parts, progs, total_counts, before = [], [], np.zeros(table.size, np.uint64), 0.0
        for a, b in ((0, cut), (cut, n)):
            g.generate(sp, a, b - a)
            s = g.stats()
            c, p = g.ccdf_exact(table, before, n)
            parts.append(s)
            progs.append(p)
            total_counts += c
            before += s.sum
        m = pkg.stats_merge(parts)
        for k in TRACKERS:
            assert getattr(m, k) == getattr(st, k) and getattr(m, k + "_idx") == getattr(st, k + "_idx"), k
        assert np.array_equal(total_counts, counts)
        assert pkg.exact_chain(progs) == whole


def test_ccdf_randomised_level_tables(pkg, orc, gpu):
    """300 random level tables (log-uniform, clustered, adjacent bit patterns, duplicates, huge spans,
    sample powers themselves as thresholds) against the oracle's float `>` loop: exercises the LUT cell
    choice, the LUT/search switch and the key construction at their edges."""
    rng = np.random.default_rng(20260929)
    n = 60011
    iq = (rng.standard_normal(2 * n) * rng.choice([1e-3, 0.7, 40.0], size=2 * n, p=[0.2, 0.7, 0.1])).astype(np.float32)
    iq[2 * 11:2 * 13] = 0.0
    gpu.upload(iq)
    pw = (iq[0::2] * iq[0::2]) + (iq[1::2] * iq[1::2])
    for trial in range(300):
        kind = trial % 6
        m = int(rng.integers(1, 400))
        if kind == 0:      # log-uniform over a random span
            lo, hi = sorted(rng.uniform(-12, 6, 2))
            tab = 10.0 ** rng.uniform(lo, hi + 1e-3, m)
        elif kind == 1:    # geometric ladder like the reference's, random ratio
            tab = rng.uniform(1e-4, 2.0) * rng.uniform(1.001, 1.6) ** np.arange(m)
        elif kind == 2:    # clusters of adjacent float bit patterns
            base = np.float32(10.0 ** rng.uniform(-3, 2))
            bits = np.frombuffer(base.tobytes(), np.uint32)[0] + rng.integers(0, 40, m).astype(np.uint32)
            tab = bits.view(np.float32)
        elif kind == 3:    # thresholds that ARE sample powers (strictness of `>`), plus duplicates
            tab = np.concatenate([pw[rng.integers(0, n, m)], pw[rng.integers(0, n, 3)].repeat(2)])
        elif kind == 4:    # two far-apart clusters: a coarse LUT would be huge
            tab = np.concatenate([10.0 ** rng.uniform(-9, -8.9, m // 2 + 1), 10.0 ** rng.uniform(3, 3.1, m // 2 + 1)])
        else:              # mixed signs, zeros and specials sprinkled in
            tab = np.concatenate([10.0 ** rng.uniform(-4, 3, m), [0.0, -0.0, -1.0, np.inf, np.nan][: int(rng.integers(0, 6))]])
        with np.errstate(over="ignore"):          # ladders may run past FLT_MAX: +inf thresholds are part of the test
            tab = np.asarray(tab, dtype=np.float32)
        rng.shuffle(tab)
        got = gpu.ccdf(tab)
        want = orc.count_mem(iq, tab)
        assert np.array_equal(got.astype(np.int64), want), (trial, kind, tab[:8])


def test_pass1_randomised_inputs(pkg, orc, xgpu):
    """150 random streams (random length, amplitude law, planted ties / zeros / NaN / Inf at random
    places, random launch geometry): pass-1 record, NaN bookkeeping and — when finite — the exact
    sequential sum against the oracle."""
    rng = np.random.default_rng(77001)
    for trial in range(150):
        n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 9000), rng.integers(9000, 120000)]))
        iq = (rng.standard_normal(2 * n) * 10.0 ** rng.uniform(-6, 3)).astype(np.float32)
        if rng.random() < 0.5 and n > 8:               # equal extremes in several places
            spots = rng.integers(0, n, int(rng.integers(2, 6)))
            iq[2 * spots], iq[2 * spots + 1] = np.float32(77.0), np.float32(-77.0)
        if rng.random() < 0.3:
            a = int(rng.integers(0, n))
            iq[2 * a:2 * min(n, a + int(rng.integers(1, 5000)))] = 0.0
        special = rng.random()
        if special < 0.15:
            iq[int(rng.integers(0, 2 * n))] = rng.choice([np.nan, -np.nan])
        if 0.1 < special < 0.25:
            iq[int(rng.integers(0, 2 * n))] = rng.choice([np.inf, -np.inf])
        xgpu.set_tuning(blocks=int(rng.choice([0, 1, 3, 64])), map=int(rng.integers(0, 3)))
        base = int(rng.integers(0, 1 << 40)) // 8192 * 8192
        xgpu.upload(iq, base_index=base)
        st = xgpu.stats()
        ref = orc.run_mem(iq, False)
        check_stats(st, ref, base=base)
        if np.isfinite(ref["sum"]):
            assert pkg.exact_chain([xgpu.exact_program(0.0, n)]) == ref["sum"], trial
        else:
            assert bool(st.flags & pkg.FLAG_NAN) == bool(np.isnan(ref["sum"]))
    xgpu.set_tuning()
