#!/usr/bin/env python3
"""When do the workgroups of the sweep kernels finish?  The product kernels are ONE persistent workgroup per CU over a static
share of the shard, so a launch lasts as long as its slowest workgroup.  For the tree-sum and the exact-sum step, both tables:
the spread of the workgroups' finish times (papr_hip_get_wg_finish) over a few steps on the 10 GiB shard — overall, per XCD
(workgroup i runs on XCD i mod 8) — next to the kernel's duration.
  gpurun -- 'python tools/wg_finish_probe.py'"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    pkg = ge.load_package()
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    n = int(gib * (1 << 30)) // 8 // 8192 * 8192
    shard = torch.empty(n * 8 + 65536, dtype=torch.uint8, device="cuda:0")
    with pkg.PaprHip(0) as gpu:
        gpu.adopt(shard.data_ptr(), n, keepalive=shard)
        gpu.generate(pkg.SynthSpec.spike(n), 0, n)
        for exact in (False, True):
            gpu.set_exact(exact)
            for graph in (False, True):
                gpu.set_timing(2)
                for _ in range(24):
                    gpu.analyze(None, graph)
                gpu.set_timing(2)
                rows = []
                for _ in range(6):
                    gpu.analyze(None, graph)
                    us, xcd = gpu.wg_finish()
                    rows.append(us)
                k_ms = gpu.timing_launches(3)
                d = np.stack(rows)                     # steps x workgroups, us after the step's first finisher
                last = d.max(axis=1)
                per_xcd = np.stack([d[:, x::8].mean(axis=1) for x in range(8)], axis=1).mean(axis=0)
                slow = np.argsort(d.mean(axis=0))[-5:][::-1]
                where = "workgroup i on XCD (i + %d) mod 8" % int((xcd[0] - 0) % 8) if np.array_equal((xcd - xcd[0]) % 8, np.arange(xcd.size) % 8) else "XCDs of workgroups 0..15: %s" % xcd[:16].tolist()
                print(f"[{where}] {'exact' if exact else 'tree '} sum, {'0.1 dB' if graph else '1 dB  '} table: kernel {np.mean(k_ms) / 1e3:.4f} ms; "
                      f"first -> last workgroup done: {last.mean():.1f} us (max {last.max():.1f}), median workgroup {np.median(d, axis=1).mean():.1f} us, "
                      f"p90 {np.percentile(d, 90, axis=1).mean():.1f} us; mean finish per XCD {np.round(per_xcd, 1).tolist()}; "
                      f"latest workgroups {slow.tolist()} at {np.round(d.mean(axis=0)[slow], 1).tolist()} us", flush=True)


if __name__ == "__main__":
    main()
