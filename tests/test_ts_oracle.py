"""The TS packet-scan oracle (oracle/ts_oracle.c) against the REAL reference: the lines
`oracle/_ref/xport -ps[h]` printed for every fixture of tests/ts_streams.py (recorded by
tests/golden/make_golden_ts.py), and — where the compiled reference is present — a live differential fuzz."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ts_oracle  # noqa: E402
import ts_streams  # noqa: E402

TS_GOLDEN = os.path.join(GOLDEN, "ts")
MANIFEST = json.load(open(os.path.join(TS_GOLDEN, "manifest.json")))


def golden_lines(name):
    with open(os.path.join(TS_GOLDEN, name + ".txt"), "rb") as f:
        return f.read()


@pytest.mark.parametrize("name", sorted(ts_streams.FIXTURES))
def test_oracle_reproduces_reference_lines(name):
    data = ts_streams.fixture_bytes(name)
    assert len(data) == MANIFEST[name]["bytes"] and hashlib.sha256(data).hexdigest() == MANIFEST[name]["sha256"], \
        "the seeded generator no longer produces the stream the golden lines were recorded for"
    res = ts_oracle.scan_mem(data, ts_streams.is_hdmv(name))
    assert ts_oracle.report_lines(res) == golden_lines(name)
    assert res["nsync_errors"] == MANIFEST[name]["sync_error_lines"]
    assert res["ndiscontinuities"] == MANIFEST[name]["discontinuity_lines"]


def test_oracle_cli_and_file_path(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ts_oracle", "libts_oracle.so"])
    for name in ("ts_insert_sync", "ts_hdmv_damaged"):
        path = str(tmp_path / (name + ".ts"))
        with open(path, "wb") as f:
            f.write(ts_streams.fixture_bytes(name))
        hd = ts_streams.is_hdmv(name)
        p = subprocess.run([ts_oracle.CLI_PATH] + (["-h"] if hd else []) + [path], capture_output=True)
        assert p.returncode == 0 and p.stdout == golden_lines(name)
        assert ts_oracle.report_lines(ts_oracle.scan_file(path, hd)) == golden_lines(name)
    assert subprocess.run([ts_oracle.CLI_PATH, str(tmp_path / "missing.ts")], capture_output=True).returncode == 255


def random_stream_kwargs(t, rng):
    kw = dict(seed=5000 + t, npackets=int(rng.integers(0, 700)), hdmv=bool(rng.integers(0, 4) == 0),
              pat=bool(rng.integers(0, 2)), af_rate=float(rng.random()),
              bad_af_rate=float(rng.random() * 0.2 * rng.integers(0, 2)), tei_rate=float(rng.random() * 0.2),
              offset_garbage=int(rng.integers(0, 20000)) if rng.integers(0, 2) else 0,
              garbage_has_sync=bool(rng.integers(0, 2)), truncate=int(rng.integers(0, 300)) if rng.integers(0, 2) else 0)
    n = kw["npackets"]
    dm = []
    for _ in range(int(rng.integers(0, 4))):
        if n > 2:
            dm.append((int(rng.integers(1, n)), "insert" if rng.integers(0, 2) else "delete",
                       int(rng.integers(1, 188 if rng.integers(0, 2) else 3000))))
    dm = [d for d in dm if not (d[1] == "delete" and d[2] > 187)]
    kw["damage"] = list({d[0]: d for d in dm}.values())
    if n > 4 and rng.integers(0, 2):   # continuity counters: jumps, lost and repeated packets, packets without a payload
        pick = lambda m: tuple(int(v) for v in rng.integers(0, n, int(rng.integers(0, m))))
        kw["cc_jump_at"] = {k: int(rng.integers(1, 16)) for k in pick(12)}
        kw["drop_at"], kw["repeat_at"], kw["no_payload_at"] = pick(6), pick(6), pick(20)
    if rng.integers(0, 3) == 0 and n > 130:   # a packet that ends one byte past a 16384-byte read (the reference's quirk)
        k = int(rng.integers(88, 130))
        kw["offset_garbage"] = ts_streams.quirk_offset(k, 192 if kw["hdmv"] else 188)
        if rng.integers(0, 2):
            kw["last_byte_sync_at"] = k
    return kw


@pytest.mark.skipif(not os.path.exists(ts_oracle.REF_CLI), reason="no compiled reference (oracle/_ref/xport)")
def test_live_fuzz_against_reference_binary(tmp_path):
    rng = np.random.default_rng(20260929)
    path = str(tmp_path / "f.ts")
    for t in range(120):
        kw = random_stream_kwargs(t, rng)
        data = ts_streams.make_stream(**kw)
        with open(path, "wb") as f:
            f.write(data)
        want = ts_oracle.reference_lines(path, kw["hdmv"])
        assert ts_oracle.report_lines(ts_oracle.scan_mem(data, kw["hdmv"])) == want, kw
