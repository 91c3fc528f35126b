#!/usr/bin/env python3
"""A randomised soak of the one-sweep step ON THE GPU BOX: N random shards (20 ... 400 million samples, every envelope of the
generator, the bench's spikes or none, a random scale) analysed by papr_hip_analyze in its one-read form — whose stash spills
are timed by the chip's real-time counter, i.e. take a different path on every run — and in its two-pass form (two plain
kernels, no speculation), tree sum and exact sum, both level tables, three times each: the record, the level table and every
count must be the same from both forms and from run to run, and in exact-sum mode the sum is one and the same double.
   gpurun -- 'python tools/papr_soak.py 60'"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package()


def key(res, table, counts):
    t = res.total
    return (t.to_bytes(), float(res.mean).hex(), float(res.papr).hex(), table.tobytes(), counts.tobytes())


def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(20260930)
    envelopes = sorted(pkg.SynthSpec.ENVELOPES)
    bad = runs = 0
    with pkg.PaprHip(0) as g:
        for it in range(n_iter):
            n = int(rng.integers(20_000_000, 400_000_000)) | int(rng.integers(0, 2))
            env = envelopes[int(rng.integers(0, len(envelopes)))]
            seed = int(rng.integers(1, 1 << 31))
            spec = pkg.SynthSpec.spike(n, seed=seed, envelope=env) if rng.integers(0, 3) else pkg.SynthSpec.make(seed, float(rng.choice([0.0, 0.01, 3.0])), envelope=env)
            g.generate(spec, 0, n)
            for exact in (False, True):
                g.set_exact(exact)
                for graph in (False, True):
                    want = key(*g.analyze(None, graph, two_pass=True))
                    for rep in range(3):
                        res, table, counts = g.analyze(None, graph)
                        runs += 1
                        got = key(res, table, counts)
                        if exact:
                            same = got == want
                        else:
                            # (the tree sum's grouping is each kernel form's own: the trackers must agree with the two-pass form's
                            # whatever the sums do, table and counts too where the sums agree — and the one-read form with itself)
                            same = got[0][8:] == want[0][8:] and (got[0][:8] != want[0][:8] or got == want)
                            if rep == 0:
                                first = got
                            same = same and got == first
                        if not same:
                            bad += 1
                            print("MISMATCH", it, n, env, seed, "exact" if exact else "tree", "graph" if graph else "default", "rep", rep,
                                  "swept", res.swept, "resolved", res.resolved, flush=True)
            if it % 10 == 9:
                print("...", it + 1, "shards,", runs, "one-read steps,", bad, "mismatches", flush=True)
    print("shards", n_iter, "one-read steps", runs, "mismatches", bad)


if __name__ == "__main__":
    main()
