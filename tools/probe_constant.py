#!/usr/bin/env python3
"""GPU probe: what the one-read path does with a constant-envelope stream (sweep info per mode)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 1048576 + 77
for exact in (False, True):
    with pkg.PaprHip(0) as g:
        g.set_exact(exact)
        g.generate(pkg.SynthSpec.spike(n, seed=99, envelope="constant"), 0, n)
        iq = g.download(0, min(n, 1 << 20))
        pw = iq[0::2].astype(np.float32) ** 2 + iq[1::2].astype(np.float32) ** 2
        print("powers: min %r max %r distinct %d" % (pw.min(), np.sort(pw)[-3], np.unique(pw).size))
        for graph in (False, True):
            res, table, counts = g.analyze(None, graph)
            print("exact" if exact else "tree", "graph" if graph else "default", "mean", res.mean, "level0", table[0],
                  g.sweep_info().as_dict(), "resolved", res.resolved, "counts[:3]", counts[:3])
