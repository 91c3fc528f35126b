#!/usr/bin/env python3
"""kernel_ab.py — sweep-kernel forms side by side on ONE box, in ONE process, interleaved (boxes and processes differ
by +-3 %; only an interleaved A/B says which form is faster).

For every round, every form, both tables: `--steps` whole steps (papr_hip_analyze) on the 10 GiB bench shard with HIP
events on the sweep kernel; prints the median kernel time per (form, table) and per round, the step time, and checks
that every form produced the same result (sum bits, counts CRC).

  python tools/kernel_ab.py --exact --variants 135,131 [--rounds 3] [--steps 12] [--gib 10]
  python tools/kernel_ab.py --variants 111,...           (tree-sum forms)
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=10.0)
    ap.add_argument("--variants", default="135,131")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--exact", action="store_true")
    ap.add_argument("--env", default="", help="NAME=a|b|...: an environment knob the library reads per launch, as one more axis")
    args = ap.parse_args()
    pkg = ge.load_package()
    n = int(args.gib * (1 << 30)) // 8 // 8192 * 8192
    shard = torch.empty(n * 8 + 65536, dtype=torch.uint8, device="cuda:0")
    g = pkg.PaprHip(0)
    g.adopt(shard.data_ptr(), n, keepalive=shard)
    g.generate(pkg.SynthSpec.spike(n), 0, n)
    g.set_exact(args.exact)
    forms = [int(v) for v in args.variants.split(",")]
    forms = [v for v in forms if pkg.sweep_variant_built(v)]
    results = {}
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:   # the GPU needs ~20 ms of load to reach its steady state
        g.analyze(None, False)
    rows = {}
    env_name, env_values = (args.env.split("=")[0], args.env.split("=")[1].split("|")) if "=" in args.env else ("", [""])
    for rnd in range(args.rounds):
        for v, ev in [(v, ev) for v in forms for ev in env_values]:
            g.set_tuning(sweep_variant=v)
            if env_name:
                os.environ[env_name] = ev
            for graph in (False, True):
                for _ in range(3):
                    g.analyze(None, graph)
                g.set_timing(2)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    res, table, counts = g.analyze(None, graph)
                torch.cuda.synchronize()
                step_ms = (time.perf_counter() - t0) / args.steps * 1e3
                k = g.timing_launches(3)
                g.set_timing(False)
                info = g.sweep_info()
                sig = (float(res.total.sum).hex(), zlib.crc32(np.ascontiguousarray(counts, dtype=np.uint64).tobytes()),
                       int(res.nlevels), int(res.exact_sum), int(res.resolved))
                results.setdefault(graph, set()).add(sig)
                rows.setdefault((v, ev, graph), []).append((float(np.median(k)), float(k.min()), step_ms, int(info.kernel_variant)))
                print(json.dumps({"round": rnd, "variant": v, "env": ev, "ran": int(info.kernel_variant), "graph": graph,
                                  "kernel_ms_median": round(float(np.median(k)), 4), "kernel_ms_min": round(float(k.min()), 4),
                                  "step_ms": round(step_ms, 4), "frac_of_8TBs": round(float(n * 8 / float(np.median(k)) / 1e6 / 8000), 4),
                                  "sum": sig[0], "crc": sig[1], "exact_sum": sig[3], "resolved": sig[4]}), flush=True)
    print("# summary (median over rounds of the per-round medians)")
    for (v, ev, graph), r in sorted(rows.items()):
        km = float(np.median([x[0] for x in r]))
        print(f"# variant {v:4d} {env_name}={ev:6s} {'-g' if graph else '  '}  kernel {km:.4f} ms  frac {n * 8 / km / 1e6 / 8000:.4f}  "
              f"step {float(np.median([x[2] for x in r])):.4f} ms")
    same = all(len(s) == 1 for s in results.values())
    print("# every form produced the same result:", same)
    g.close()
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
