#!/usr/bin/env python3
"""Launch-geometry sweep on one GPU: kernel-only time of both passes (HIP events
inside libpaprhip) for each tuning, interleaved rounds, median and min reported.

    python tools/sweep.py --gib 10 --rounds 5 "blocks=2048,map=0,nt=1" "blocks=1024,map=2,nt=1" ...
"""
import argparse
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def parse(spec):
    d = dict(blocks=0, variant=None, map=None, nontemporal=1, hist_copies=0, flags=0)
    alias = {"nt": "nontemporal", "copies": "hist_copies", "search": "flags", "sblocks": "stats_blocks",
             "svariant": "stats_variant", "smap": "stats_map", "cblocks": "ccdf_blocks", "cvariant": "ccdf_variant",
             "cmap": "ccdf_map"}
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        d[alias.get(k, k)] = int(v)
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=10.0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--mode", default="both")
    ap.add_argument("specs", nargs="*", default=[""])
    a = ap.parse_args()
    pkg = ge.load_package()
    n = int(a.gib * (1 << 30)) // 8 // 8192 * 8192
    g = pkg.PaprHip(0)
    g.generate(pkg.SynthSpec.spike(n), 0, n)
    st = g.stats()
    tables = {m: pkg.levels(st, m == "graph")[2] for m in ("default", "graph")}
    modes = ["default", "graph"] if a.mode == "both" else [a.mode]
    res = {s: {"stats": [], **{m: [] for m in modes}} for s in a.specs}
    for r in range(a.rounds + 1):
        for s in a.specs:
            g.set_tuning(**parse(s))
            g.set_timing(True)
            g.stats()
            t = g.timing()
            if r:
                res[s]["stats"].append(t.stats_ms)
            for m in modes:
                g.set_timing(True)
                g.ccdf(tables[m])
                t = g.timing()
                if r:
                    res[s][m].append(t.ccdf_ms)
    gb = n * 8 / 1e9
    for s in a.specs:
        row = {"spec": s}
        for k, v in res[s].items():
            med, mn = statistics.median(v), min(v)
            row[k] = {"med_ms": round(med, 4), "min_ms": round(mn, 4), "GB/s_med": round(gb / med * 1e3, 1)}
        print(json.dumps(row), flush=True)
    g.close()


if __name__ == "__main__":
    main()
