#!/usr/bin/env python3
"""Record the REAL reference's report lines for the TS packet-scan fixtures (build container only).

Compiles /root/reference/xport.c where it lies into oracle/_ref/xport (oracle/Makefile `ref`, the reference
README's own gcc line), generates every stream of tests/ts_streams.py:FIXTURES, runs

    oracle/_ref/xport -ps[h] <stream> 70000 1 1

(parse only, no rate output, a program number no PAT can carry) and stores the `Transport Sync Error`, `Discontinuity!` and
`packets for pid` lines of its stdout as tests/golden/ts/<name>.txt, plus the stream's size and sha256 in
tests/golden/ts/manifest.json — the streams themselves are regenerated from the seeded generator by the tests,
which check that hash first.  Only data (expected outputs) is stored.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ts_oracle  # noqa: E402
import ts_streams  # noqa: E402


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"])
    if not os.path.exists(ts_oracle.REF_CLI):
        sys.exit("oracle/_ref/xport missing: this script needs /root/reference")
    out_dir = os.path.join(HERE, "ts")
    os.makedirs(out_dir, exist_ok=True)
    manifest = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name in ts_streams.FIXTURES:
            data = ts_streams.fixture_bytes(name)
            path = os.path.join(tmp, name + ".ts")
            with open(path, "wb") as f:
                f.write(data)
            lines = ts_oracle.reference_lines(path, ts_streams.is_hdmv(name))
            with open(os.path.join(out_dir, name + ".txt"), "wb") as f:
                f.write(lines)
            manifest[name] = {"bytes": len(data), "sha256": hashlib.sha256(data).hexdigest(),
                              "hdmv": ts_streams.is_hdmv(name), "lines": lines.count(b"\n"),
                              "sync_error_lines": lines.count(b"Transport Sync Error"),
                              "discontinuity_lines": lines.count(b"Discontinuity!")}
    json.dump(manifest, open(os.path.join(out_dir, "manifest.json"), "w"), indent=1, sort_keys=True)
    print(f"recorded {len(manifest)} TS fixtures")


if __name__ == "__main__":
    main()
