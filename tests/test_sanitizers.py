"""Host-side C under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has
no sanitizer story; the build gets one for the code that parses bytes).  papr_exact_chain replays sum
programs that, in a multi-GPU run, arrive from other processes — it must reject damaged ones without ever
reading outside them — and the oracle's chunk feeder is run over every fixture the same way."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_names, golden_path, golden_text
from test_exact_chain_cpu import build_program

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


@pytest.fixture(scope="module")
def san_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "dtv-utils_amd", "csrc")]
    subprocess.check_call(["gcc", *SAN, "-ffp-contract=off", *inc, os.path.join(ROOT, "tests", "c", "chain_harness.c"),
                           os.path.join(ROOT, "dtv-utils_amd", "csrc", "papr_host.c"), "-o", str(d / "chain"), "-lm"])
    subprocess.check_call(["gcc", *SAN, "-ffp-contract=off", *inc, os.path.join(ROOT, "tests", "c", "sweep_harness.c"),
                           os.path.join(ROOT, "dtv-utils_amd", "csrc", "papr_host.c"), "-o", str(d / "sweep"), "-lm"])
    subprocess.check_call(["gcc", *SAN, "-ffp-contract=off", "-DPAPR_ORACLE_MAIN", *inc,
                           os.path.join(ROOT, "oracle", "papr_oracle.c"), "-o", str(d / "oracle"), "-lm"])
    return d


def run(binary, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([str(binary), *map(str, args)], capture_output=True, env=env)
    assert b"ERROR: AddressSanitizer" not in p.stderr and b"runtime error" not in p.stderr, p.stderr.decode()[-3000:]
    return p


def test_chain_under_sanitizers_valid_and_damaged_programs(orc, san_dir):
    rng = np.random.default_rng(99)
    iq = rng.standard_normal(2 * 300001).astype(np.float32)
    prog, _ = build_program(iq)
    good = san_dir / "good.bin"
    good.write_bytes(prog)
    p = run(san_dir / "chain", good)
    rc, hexsum = p.stdout.split()
    assert int(rc) == 0 and float.fromhex(hexsum.decode()) == orc.run_mem(iq, False)["sum"]
    # truncations, random byte flips, inflated counts: any return code is fine, a sanitizer report is not
    bad = san_dir / "bad.bin"
    for trial in range(300):
        b = bytearray(prog)
        kind = trial % 4
        if kind == 0:
            b = b[: int(rng.integers(0, len(b)))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 20))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 2:      # header fields only (counts, sizes)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(8, 48))] = int(rng.integers(0, 256))
        else:                # huge counts
            off = int(rng.choice([16, 24, 32, 36, 40]))
            b[off:off + 4] = (0xFFFFFFF0 + int(rng.integers(0, 15))).to_bytes(4, "little")
        bad.write_bytes(bytes(b))
        p = run(san_dir / "chain", bad)
        assert p.returncode == 0 and p.stdout.split()[0] in (b"0", b"-3", b"-8"), (trial, p.stdout)


def test_sweep_host_logic_under_sanitizers(san_dir):
    """papr_guess_levels / papr_sweep_bands / papr_sweep_resolve (the host half of the one-sweep mode) on random and
    hostile tables — NaN, +-0, denormal, huge, duplicated, unsorted levels, every band width — in exact-size heap
    blocks: no sanitizer report, and all three outcomes (resolved, refused, no band form) occur."""
    totals = np.zeros(3, dtype=np.int64)
    for seed in (1, 2, 3, 20260929):
        p = run(san_dir / "sweep", seed, 400)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        totals += np.array(p.stdout.split(), dtype=np.int64)
    assert np.all(totals > 0), totals


def test_oracle_under_sanitizers(san_dir):
    for name in golden_names():
        for graph in (False, True):
            p = run(san_dir / "oracle", *(["-g"] if graph else []), golden_path(name))
            want = golden_text(name, graph)
            if b"nan" not in want:   # which NaN sign a compiler's operand order produces differs between -O1 and
                assert p.stdout == want, name   # the reference's -O2 build; the sanitizer run is about memory safety


# ---- the threaded host code under ThreadSanitizer ------------------------------------------------------------------------
TSAN = ["-fsanitize=thread", "-g", "-O1", "-std=c++17"]


def run_tsan(cmd, timeout=600, **env):
    p = subprocess.run([str(c) for c in cmd], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **env))
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-4000:]
    return p


def test_line_pool_under_thread_sanitizer(tmp_path):
    """The packet scan's host threads (ts_line_pool.h: spinning workers, jobs from one atomic ticket) under ThreadSanitizer:
    rounds of 1 ... 64 jobs, every job exactly once, no report.  (The pool's first form failed this by never returning.)"""
    exe = tmp_path / "pool_tsan"
    subprocess.check_call(["g++", *TSAN, "-I" + os.path.join(ROOT, "dtv-utils_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "pool_harness.cpp"), "-o", str(exe), "-lpthread"])
    for threads, rounds in ((8, 20000), (3, 20000), (2, 5000)):
        p = run_tsan([exe, threads, rounds])
        assert p.returncode == 0 and p.stdout.strip() == "0", (threads, rounds, p.stdout, p.stderr[-2000:])


def test_in_process_exchange_under_thread_sanitizer(tmp_path):
    """The in-process exchange (the hub that bin/papr's per-GPU threads meet at) under ThreadSanitizer: host-level collectives
    checked round after round, the self-test's collectives, then one thread cancels the exchange while the others wait in
    one.  (Without PAPR_XCH_TIMEOUT_S: gcc 11's libtsan does not know pthread_cond_clockwait and reports every timed wait.)"""
    rocm_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(rocm_inc, "rccl", "rccl.h")) or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("no ROCm headers here")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "dtv-utils_amd", "csrc"), "-I" + rocm_inc]
    objs = []
    for src, cc, std in (("papr_exchange.cpp", "g++", ["-std=c++17"]), ("papr_host.c", "gcc", [])):
        obj = tmp_path / (src + ".o")
        subprocess.check_call([cc, "-fsanitize=thread", "-g", "-O1", *std, "-fPIC", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", *inc, "-c",
                               os.path.join(ROOT, "dtv-utils_amd", "csrc", src), "-o", str(obj)])
        objs.append(str(obj))
    exe = tmp_path / "hub_tsan"
    subprocess.check_call(["g++", *TSAN, "-D__HIP_PLATFORM_AMD__", *inc, os.path.join(ROOT, "tests", "c", "hub_harness.cpp"), *objs, "-o", str(exe),
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-lpthread", "-ldl", "-lm"])
    env = {k: v for k, v in os.environ.items() if k != "PAPR_XCH_TIMEOUT_S"}
    for threads, rounds in ((8, 1500), (3, 1500), (2, 500)):
        p = subprocess.run([str(exe), str(threads), str(rounds)], capture_output=True, text=True, timeout=600, env=env)
        assert "ThreadSanitizer" not in p.stderr, p.stderr[-4000:]
        assert p.returncode == 0 and p.stdout.strip() == "ok", (threads, rounds, p.stdout, p.stderr[-2000:])
    # round 6: the handles of papr_exchange_open_rccl_local_async — n set-up threads of their own beside the callers — with the
    # set-up made to fail (no RCCL, no GPU touched): every caller adopts, all agree through the hub that nobody has a
    # communicator, rank 0 says so in one line, and the handles are the hub's for the same rounds of collectives and the abort
    for threads, rounds in ((8, 300), (2, 300), (1, 50)):
        p = subprocess.run([str(exe), str(threads), str(rounds), "async"], capture_output=True, text=True, timeout=600,
                           env=dict(env, PAPR_XCH_BIND_FAIL="all"))
        assert "ThreadSanitizer" not in p.stderr, p.stderr[-4000:]
        assert p.returncode == 0 and p.stdout.strip() == "ok", (threads, rounds, p.stdout, p.stderr[-2000:])
        said = [l for l in p.stderr.splitlines() if l.startswith("papr: RCCL set-up did not complete")]
        assert len(said) == 1 and "injected" in said[0], p.stderr[-1000:]
