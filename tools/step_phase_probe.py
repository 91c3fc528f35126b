#!/usr/bin/env python3
"""Where a default step's time outside the sweep kernel goes: wall time of every phase of the one-read step (the calls
papr_hip_analyze makes, one by one) on the 10 GiB bench workload, against the kernel times the library measures."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(float(sys.argv[1]) * (1 << 30)) // 8 if len(sys.argv) > 1 else 1342177280
with pkg.PaprHip(0) as g:
    g.generate(pkg.SynthSpec.spike(n), 0, n)
    for graph in (False, True):
        acc = {}
        g.set_timing(True)
        reps = 40
        for r in range(reps + 5):
            if r == 5:
                acc = {}
                t_before = g.timing()
                k0 = (t_before.sweep_ms, t_before.aux_ms)
            t = [time.perf_counter()]
            est = g.estimate(); t.append(time.perf_counter())
            guess = pkg.guess_levels(est, graph); g.set_band(pkg.band_for(est)); t.append(time.perf_counter())
            st = g.stats_sweep(guess); t.append(time.perf_counter())
            mean, papr, table = pkg.levels(st, graph); t.append(time.perf_counter())
            counts = g.ccdf(table); t.append(time.perf_counter())
            for name, a, b in zip(("estimate (kernel + sync)", "guess table (host)", "sweep (prepare + kernel + sync)", "levels (host)", "ccdf (resolve + recount + sync)"), t[:-1], t[1:]):
                acc[name] = acc.get(name, 0.0) + (b - a)
            acc["step"] = acc.get("step", 0.0) + (t[-1] - t[0])
        tm = g.timing()
        print("mode", "graph" if graph else "default", "resolved", g.sweep_info().resolved)
        for k, v in acc.items():
            print("  %-34s %.4f ms" % (k, v / reps * 1e3))
        print("  kernels: sweep %.4f ms, estimate + recount %.4f ms" % ((tm.sweep_ms - k0[0]) / reps, (tm.aux_ms - k0[1]) / reps))
