#!/usr/bin/env python3
"""Distil gpurun_out/<tag>/ (made by tools/collect_profiles.sh on the GPU box)
into the small, tracked files under profiles/:

  <tag>_kernel_stats_{default,graph}.csv   rocprofv3 --kernel-trace --stats summary
  <tag>_pmc_summary.json                   per-kernel averages of every PMC pass, with the
                                           gfx950 FETCH_SIZE correction applied and stated
  <tag>_bench_{default,graph}.json         the bench.py lines of the same session
  <tag>_hbm_read_probe.txt                 empirical read ceiling
  pmc_traffic.json                         what bench.py reports as roofline.traffic
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for k in ("papr_ccdf_kernel", "papr_stats_kernel", "papr_stats_finalize", "papr_generate_kernel",
              "papr_first_nan_kernel"):
        if k in name:
            return k
    return name.split("(")[0]


def newest(pattern):
    """gpurun merges every call's files into the same local directory: keep the latest run only."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": len(next(iter(cs.values())))}
            for k, cs in out.items() if k.startswith("papr_")}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for name in ("bench_exact.json", "bench_exact_graph.json", "bench_torchrun1.json", "cli_e2e.txt"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
    for f in newest(os.path.join(src, "stats_exact", "*", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats_exact.csv"))
    for mode in ("default", "graph"):
        for f in newest(os.path.join(src, f"stats_{mode}", "*", "*_kernel_stats.csv")):
            shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats_{mode}.csv"))
        for name in (f"bench_{mode}.json", f"stats_{mode}.json"):
            p = os.path.join(src, name)
            if os.path.exists(p) and os.path.getsize(p):
                shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
    p = os.path.join(src, "hbm_read_probe.txt")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_hbm_read_probe.txt"))

    summary = {"note": "per-dispatch averages over the bench run of each rocprofv3 --pmc pass (one pass per counter set, "
                       "--kernel-trace only). FETCH_SIZE is in KiB and, on gfx950 with this rocprofv3, tallies each 128-B "
                       "request of a 16 B/lane coalesced stream at 64 B (MI355X_MICROARCH.md, HBM section): "
                       "hbm_read_bytes = 2 * FETCH_SIZE * 1024; cross-check = TCC_EA0_RDREQ_sum * 128."}
    traffic = {"source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE, separate pass, x2 gfx950 correction)"}
    for mode in ("default", "graph"):
        c = counters(os.path.join(src, f"pmc_fetch_{mode}"))
        for k, v in c.items():
            if "FETCH_SIZE" in v:
                v["hbm_read_bytes_corrected"] = 2 * v["FETCH_SIZE"] * 1024
        summary[f"fetch_{mode}"] = c
        bj = os.path.join(src, f"pmc_fetch_{mode}.json")
        gib = 10.0
        try:
            gib = json.load(open(bj))["config"]["samples_per_gpu"] * 8 / (1 << 30)
        except Exception:
            pass
        for k in ("papr_ccdf_kernel", "papr_stats_kernel"):
            if k in c and "hbm_read_bytes_corrected" in c[k]:
                key = mode if k == "papr_ccdf_kernel" else "any"
                traffic.setdefault(k, {})[key] = {"hbm_bytes_per_launch": c[k]["hbm_read_bytes_corrected"],
                                                  "gib_per_gpu": gib}
    for name in ("pmc_tcc", "pmc_sq", "pmc_grbm"):
        c = counters(os.path.join(src, name))
        if name == "pmc_tcc":
            for k, v in c.items():
                if "TCC_EA0_RDREQ_sum" in v:
                    v["rdreq_x128_bytes"] = v["TCC_EA0_RDREQ_sum"] * 128
                if v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0) > 0:
                    v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        summary[name + "_graph"] = c
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    if len(traffic) > 1:
        json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
