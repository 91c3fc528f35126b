#!/usr/bin/env python3
"""Does the pass-1 kernel time depend on WHERE the 10 GiB shard was allocated?  Allocates several shards
(all kept alive, so they land in different places), fills each with the same stream and times pass 1 with
the XCD-contiguous (map=2) and grid-stride (map=0) tile mappings, pass 2, and the one-sweep kernel with
each mapping, on each of them."""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    pkg = ge.load_package()
    nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n = 10 * (1 << 30) // 8
    bufs = [torch.empty(n * 8 + 65536, dtype=torch.uint8, device="cuda:0") for _ in range(nbuf)]
    g = pkg.PaprHip(0)
    table = None
    for k, b in enumerate(bufs):
        g.adopt(b.data_ptr(), n, 0, keepalive=b)
        g.generate(pkg.SynthSpec.spike(n), 0, n)
        row = {"buffer": k, "addr": hex(b.data_ptr())}
        for name, tune in (("stats_map2", dict(stats_map=2)), ("stats_map0", dict(stats_map=0))):
            g.set_tuning(**tune)
            g.stats()
            ms = []
            for _ in range(6):
                g.set_timing(True)
                st = g.stats()
                ms.append(g.timing().stats_ms)
            row[name] = round(statistics.median(ms), 4)
        if table is None:
            table = pkg.levels(st, True)[2]
        g.set_tuning()
        g.ccdf(table)
        ms = []
        for _ in range(6):
            g.set_timing(True)
            g.ccdf(table)
            ms.append(g.timing().ccdf_ms)
        row["ccdf_map0"] = round(statistics.median(ms), 4)
        guess = pkg.guess_levels(g.estimate(), False)
        for m in (0, 1, 2):
            g.set_tuning(sweep_map=m)
            g.stats_sweep(guess)
            ms = []
            for _ in range(6):
                g.set_timing(True)
                g.stats_sweep(guess)
                ms.append(g.timing().sweep_ms)
            row[f"sweep_map{m}"] = round(statistics.median(ms), 4)
        g.set_tuning()
        print(json.dumps(row), flush=True)
    g.close()


if __name__ == "__main__":
    main()
