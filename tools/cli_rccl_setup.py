#!/usr/bin/env python3
"""What the RCCL set-up costs bin/papr end to end, now that it runs BESIDE the ingest (papr_exchange_open_rccl_local_async +
papr_exchange_adopt_rccl): the bench workload as a 10 GiB file in /dev/shm through bin/papr with PAPR_XCH=threads (no RCCL in
the process), =rccl (the communicator waited for) and =auto (taken only if it is up when the shard is loaded), a few runs
each, interleaved; per run the wall time (process start -> exit), and from PAPR_STATS the set-up thread's duration and how
long the shard's thread waited for it.  Also a stream through a FIFO (no RCCL is started for those).

  python tools/cli_rccl_setup.py [--gib 10] [--runs 7] > profiles/r06_cli_rccl_setup.txt

Measurement aid: nothing of the product imports it."""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=10.0)
    ap.add_argument("--runs", type=int, default=7)
    args = ap.parse_args()
    import __graft_entry__ as ge
    import mkcfile_gpu
    pkg = ge.load_package()
    n = int(args.gib * (1 << 30)) // 8 // 8192 * 8192
    path = f"/dev/shm/papr_rccl_setup_{os.getpid()}.cfile"
    golden = None
    gpath = os.path.join(ROOT, "tests", "golden", "big_spike10g.default.txt")
    if abs(args.gib - 10.0) < 1e-12 and os.path.exists(gpath):
        golden = open(gpath, "rb").read()
    try:
        print(f"# {args.gib:g} GiB spike workload in /dev/shm, written by the device generator in {mkcfile_gpu.make(path, n):.1f} s; "
              f"{args.runs} runs per setting, interleaved; bin/papr default table, exact sum")
        for _ in range(2):   # (the first two reads of a fresh tmpfs file are slow: bench.py e2e_block)
            subprocess.run([pkg.CLI_PATH, path], capture_output=True)
        settings = [("threads", dict(PAPR_GPUS="1", PAPR_XCH="threads")), ("rccl", dict(PAPR_GPUS="1", PAPR_XCH="rccl")),
                    ("auto", dict(PAPR_GPUS="1", PAPR_XCH="auto")),
                    ("rccl, set-up fails", dict(PAPR_GPUS="1", PAPR_XCH="rccl", PAPR_XCH_BIND_FAIL="all"))]
        rows = {name: [] for name, _ in settings}
        for _ in range(args.runs):
            for name, env in settings:
                time.sleep(0.75)   # (the previous process's 10 GiB of HBM are still being unmapped: bench.py e2e_block)
                t0 = time.perf_counter()
                p = subprocess.run([pkg.CLI_PATH, path], capture_output=True, env=dict(os.environ, PAPR_STATS="1", **env))
                wall = time.perf_counter() - t0
                st = json.loads(p.stderr.decode().splitlines()[-1])
                rows[name].append((wall, st, p.returncode == 0 and (golden is None or p.stdout == golden)))
        print(f"{'PAPR_XCH':22s} {'wall s (median)':>16s} {'min':>8s} {'max':>8s} {'exchange':>9s} {'setup_s':>9s} {'wait_s':>9s} {'open_s':>8s} {'ingest_s':>9s} {'analysis_s':>10s}  stdout = reference's")
        med = {}
        for name, _ in settings:
            r = sorted(rows[name], key=lambda e: e[0])
            m = r[len(r) // 2]
            med[name] = m[0]
            print(f"{name:22s} {m[0]:16.4f} {r[0][0]:8.4f} {r[-1][0]:8.4f} {m[1]['exchange']:>9s} {m[1]['exchange_setup_s']:9.4f} "
                  f"{m[1]['exchange_wait_s']:9.4f} {m[1]['open_s']:8.4f} {m[1]['ingest_pass1_s']:9.4f} {m[1]['analysis_s']:10.4f}  {all(e[2] for e in r)}")
        print(f"# rccl - threads: {1e3 * (med['rccl'] - med['threads']):+.1f} ms; auto - threads: {1e3 * (med['auto'] - med['threads']):+.1f} ms "
              f"(medians of {args.runs})")
        print("# all set-up seconds of the rccl runs: " + " ".join(f"{e[1]['exchange_setup_s']:.3f}" for e in rows["rccl"]))
        print("# all waits of the rccl runs:          " + " ".join(f"{e[1]['exchange_wait_s']:.3f}" for e in rows["rccl"]))
        print("# exchange of the auto runs:           " + " ".join(e[1]["exchange"] for e in rows["auto"]))
    finally:
        if os.path.exists(path):
            os.unlink(path)


if __name__ == "__main__":
    main()
