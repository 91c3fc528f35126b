#!/usr/bin/env python3
"""exact_sweep_probe.py — the ONE-read exact-sum path (papr_sweep2_kernel<EXACT> + speculated binades) on the GPU.

For a few small sizes: the chained sum must equal the oracle's sequential sum bit for bit and the counts the
oracle's.  For the 10 GiB bench stream: the sum must be 0x1.aaab438591900p+30 (tests/golden/manifest.json), and the
kernel / wall times of the one-read step are printed next to the two-read exact step.

  python tools/exact_sweep_probe.py [--gib 10] [--variants 48,49,50] [--reps 10]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def one_read_exact(pkg, g, graph, n_total, before_est=0.0, before=0.0):
    est = g.estimate()
    g.set_exact_hint(before_est)
    st = g.stats_sweep(pkg.guess_levels(est, graph))
    info = g.sweep_info()
    mean, papr, table = pkg.levels(st, graph)
    counts, prog = g.ccdf_exact(table, before, n_total)
    info2 = g.sweep_info()
    return st, table, counts, prog, info, info2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=10.0)
    ap.add_argument("--variants", default="48,49,50,51")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--skip-small", action="store_true")
    args = ap.parse_args()
    pkg = ge.load_package()
    orc = ge.load_oracle()
    g = pkg.PaprHip(0)
    g.set_exact(True)
    ok_all = True
    if not args.skip_small:
        for n in (5, 2047, 2048 * 3 + 5, 300007, 128 * 2048 + 17, 4 * 128 * 2048, 5000011, 40 * 1048576 + 999):
            g.generate(pkg.SynthSpec.spike(n, seed=7 + n), 0, n)
            iq = g.download(0, n)
            for graph in (False, True):
                ref = orc.run_mem(iq, graph)
                st, table, counts, prog, info, info2 = one_read_exact(pkg, g, graph, n)
                seq = pkg.exact_chain([prog])
                st.sum = seq
                table2 = pkg.levels(st, graph)[2]
                if not np.array_equal(table2, table):
                    counts = g.ccdf(table2)
                ok = seq == ref["sum"] and np.array_equal(table2, ref["level"]) and \
                    np.array_equal(counts.astype(np.int64), ref["count"])
                ok_all = ok_all and ok
                print(f"n={n} graph={graph}: swept={info.swept} reason={info.as_dict()['reason']} resolved={info2.resolved} "
                      f"redo_tiles={info2.exact_redo_tiles} sum_equal={seq == ref['sum']} counts_ok={ok}", flush=True)
    # full size
    n = int(args.gib * (1 << 30)) // 8 // 8192 * 8192
    shard = torch.empty(n * 8 + 65536, dtype=torch.uint8, device="cuda:0")
    g.adopt(shard.data_ptr(), n, base_index=0, keepalive=shard)
    g.generate(pkg.SynthSpec.spike(n), 0, n)
    want_hex = None
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
    if abs(args.gib - 10.0) < 1e-9:
        want_hex = man["big_spike10g"]["oracle_sequential_sum_hex"]
    for graph in (False, True):
        # two-read exact (round-1 form) for reference
        g.set_tuning()
        g.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            st = g.stats()
            mean, papr, table = pkg.levels(st, graph)
            counts, prog = g.ccdf_exact(table, 0.0, n)
            seq = pkg.exact_chain([prog])
        wall = (time.perf_counter() - t0) / args.reps * 1e3
        tm = g.timing()
        g.set_timing(False)
        print(f"mode={'graph' if graph else 'default'} two-read exact: stats {tm.stats_ms / max(tm.stats_launches, 1):.3f} + "
              f"exact {tm.exact_ms / max(tm.exact_launches, 1):.3f} ms, wall {wall:.3f} ms, sum {seq.hex()}", flush=True)
        ref_counts, ref_table, ref_seq = counts, table, seq
        for v in [int(x) for x in args.variants.split(",")]:
            g.set_tuning(sweep_variant=v)
            g.set_timing(True)
            t0 = time.perf_counter()
            for _ in range(args.reps):
                st, table, counts, prog, info, info2 = one_read_exact(pkg, g, graph, n)
                seq = pkg.exact_chain([prog])
            wall = (time.perf_counter() - t0) / args.reps * 1e3
            tm = g.timing()
            g.set_timing(False)
            ok = seq == ref_seq and (want_hex is None or seq.hex() == want_hex) and np.array_equal(table, ref_table) and \
                np.array_equal(counts, ref_counts)
            ok_all = ok_all and ok
            print(f"  v={v}: swept={info.swept} reason={info.as_dict()['reason']} resolved={info2.resolved} "
                  f"redo_tiles={info2.exact_redo_tiles} sweep {tm.sweep_ms / max(tm.sweep_launches, 1):.3f} ms "
                  f"({n * 8 / (tm.sweep_ms / max(tm.sweep_launches, 1)) / 1e6:.0f} GB/s) exact-helpers "
                  f"{tm.exact_ms / max(tm.exact_launches, 1):.3f} ms aux {tm.aux_ms / args.reps:.3f} ms/step wall {wall:.3f} ms "
                  f"sum {seq.hex()} OK={ok} program={len(prog)} B", flush=True)
    print("ALL OK" if ok_all else "MISMATCH", flush=True)
    g.close()


if __name__ == "__main__":
    main()
