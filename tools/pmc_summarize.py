#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 --pmc passes tools/pmc_probe.sh made: one line per kernel, one column per counter."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:60]


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    grids = {}
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row.get("Kernel_Name", "?")) + " g" + row.get("Grid_Size", "?")
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    for k in sorted(acc):
        if not any(t in k for t in ("papr_sweep", "papr_stats_kernel", "papr_ccdf_kernel", "papr_exact_seg", "ts_scan_kernel")):
            continue
        print(k)
        c = {n: v[0] / max(v[1], 1) for n, v in acc[k].items()}
        for n in sorted(c):
            print(f"    {n:40s} {c[n]:18.1f}   (n={acc[k][n][1]})")
        if "SQ_INST_LEVEL_VMEM" in c and c.get("SQ_INSTS_VMEM_RD"):
            print(f"    -> avg VMEM latency (LEVEL/INSTS_RD)      {c['SQ_INST_LEVEL_VMEM'] / c['SQ_INSTS_VMEM_RD']:10.1f}")
        if "SQ_INST_LEVEL_LDS" in c and c.get("SQ_INSTS_LDS"):
            print(f"    -> avg LDS latency (LEVEL/INSTS_LDS)      {c['SQ_INST_LEVEL_LDS'] / c['SQ_INSTS_LDS']:10.1f}")
        if c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                      "SQ_ACTIVE_INST_VMEM"):
                if n in c:
                    print(f"    -> {n}/WAVE_CYCLES                  {c[n] / w:8.3f}")
        if c.get("TCP_TCC_READ_REQ_sum"):
            print(f"    -> TCP->TCC read latency                   {c['TCP_TCC_READ_REQ_LATENCY_sum'] / c['TCP_TCC_READ_REQ_sum']:10.1f}")
        if c.get("TCC_EA0_RDREQ_sum"):
            print(f"    -> EA read latency (LEVEL/RDREQ)           {c['TCC_EA0_RDREQ_LEVEL_sum'] / c['TCC_EA0_RDREQ_sum']:10.1f}")


if __name__ == "__main__":
    main()
