#!/usr/bin/env python3
"""ablation_probe.py — what each ingredient of the one-sweep kernel costs: papr_sweep_kernel<1024, 4> with one part
left out (variants 60-69, measurement only: their results are meaningless), kernel time by HIP events.

  python tools/ablation_probe.py [--gib 10] [--reps 10]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

NAMES = {4: "full", 60: "no stash", 61: "no histogram", 62: "no stash, no histogram", 63: "no LUT lookup", 64: "no trackers",
         65: "no sum", 66: "nothing (loads + power only)", 67: "no spill check", 68: "no LUT, stash, histogram", 69: "no trackers, no sum"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=10.0)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--family", type=int, default=0, help="0: <1024, 4> (ids 60-69, base 4); 30: <256, 8> (ids 90-99, base 0); 60: round 2's <512, 8> (ids 120-129, base 40); 100: the product form (111 and ids 140-152)")
    ap.add_argument("--blocks", default="1024")
    args = ap.parse_args()
    pkg = ge.load_package()
    n = int(args.gib * (1 << 30)) // 8 // 8192 * 8192
    shard = torch.empty(n * 8 + 65536, dtype=torch.uint8, device="cuda:0")
    g = pkg.PaprHip(0)
    g.adopt(shard.data_ptr(), n, base_index=0, keepalive=shard)
    g.generate(pkg.SynthSpec.spike(n), 0, n)
    for graph in (False, True):
        est = g.estimate()
        guess = pkg.guess_levels(est, graph)
        for rnd in range(3):
          for blocks in [int(b) for b in args.blocks.split(",")]:
            ids = ((4, 60, 61, 62, 63, 64, 65, 67, 68, 69, 66) if args.family == 0 else
                   (0, 90, 91, 92, 94, 95, 97, 98, 99, 80, 81, 82, 2) if args.family == 30 else
                   (40, 120, 121, 122, 124, 125, 128, 129) if args.family == 60 else
                   (111, 140, 150, 151, 152, 141, 142, 143, 110, 144, 145, 147, 148, 153, 149, 146, 154))
            for v in ids:
                g.set_tuning(sweep_variant=v, sweep_blocks=blocks)
                g.set_timing(True)
                for _ in range(args.reps):
                    try:
                        g.stats_sweep(guess)
                    except pkg.PaprError:  # (an ablated launch breaks the sweep's invariants: only its time is of interest)
                        pass
                tm = g.timing()
                g.set_timing(False)
                ms = tm.sweep_ms / max(tm.sweep_launches, 1)
                name = NAMES.get(v) or NAMES.get(v - args.family) or {111: "product (spills in step)", 140: "the same, every wave spills when ITS slice is full (round 3 start)", 145: "true double buffer", 146: "spills every 32nd tile, chip-wide", 147: "spills every 82 us, chip-wide", 148: "every 41 us", 149: "every 164 us", 153: "every 20 us", 154: "every 82 us, no prefetch", 110: "half the slice per wave (6 KiB)", 144: "nontemporal spill stores", 150: "no stash", 151: "no histogram", 152: "no stash, no histogram", 141: "spill without the stores", 142: "spills over the same 12 KiB, write-through", 143: "spills over the same 12 KiB, plain stores", 40: "full <512, 8> (default)", 0: "full <256, 8>", 2: "<256, 8> next-tile prefetch", 80: "<256, 8> ballot stash", 81: "<256, 8> ballot + 16-byte spills", 82: "<256, 8> prefetch + ballot"}.get(v, "?")
                print(f"mode={'graph' if graph else 'default'} round {rnd} blocks {blocks} v={v:2d} {name:34s} {ms:.3f} ms  {n * 8 / ms / 1e6:.0f} GB/s", flush=True)
    g.close()


if __name__ == "__main__":
    main()
