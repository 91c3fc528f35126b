#!/usr/bin/env python3
"""Distil gpurun_out/<tag>/ (made by tools/collect_profiles.sh on the GPU box)
into the small, tracked files under profiles/:

  <tag>_kernel_stats_[twopass_]{default,graph}.csv   rocprofv3 --kernel-trace --stats summary (one-sweep / --two-pass)
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
    for k in ("papr_ccdf_power_kernel", "papr_ccdf_kernel", "papr_stats_kernel", "papr_stats_finalize",
              "papr_generate_kernel", "papr_first_nan_kernel", "papr_sweep_kernel", "papr_estimate_kernel"):
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
    for name in ("bench_exact.json", "bench_exact_graph.json", "bench_torchrun1.json", "cli_e2e.txt",
                 "bench_twopass_default.json", "bench_twopass_graph.json"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
    for f in newest(os.path.join(src, "stats_exact", "*", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats_exact.csv"))
    for mode in ("default", "graph", "twopass_default", "twopass_graph"):
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
    traffic = {"source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE, separate pass, x2 gfx950 correction; "
                         "sweep kernel: + its stash writes, WRITE_SIZE pass calibrated on the generate kernel)"}
    gib = 10.0
    for way in ("", "twopass_"):
        for mode in ("default", "graph"):
            c = counters(os.path.join(src, f"pmc_fetch_{way}{mode}"))
            for k, v in c.items():
                if "FETCH_SIZE" in v:
                    v["hbm_read_bytes_corrected"] = 2 * v["FETCH_SIZE"] * 1024
            summary[f"fetch_{way}{mode}"] = c
            try:
                gib = json.load(open(os.path.join(src, f"pmc_fetch_{way}{mode}.json")))["config"]["samples_per_gpu"] * 8 / (1 << 30)
            except Exception:
                pass
            for k in ("papr_ccdf_kernel", "papr_stats_kernel", "papr_sweep_kernel"):
                if k in c and "hbm_read_bytes_corrected" in c[k]:
                    key = "any" if k == "papr_stats_kernel" else mode
                    traffic.setdefault(k, {})[key] = {"hbm_bytes_per_launch": c[k]["hbm_read_bytes_corrected"],
                                                      "gib_per_gpu": gib}
    # WRITE_SIZE is uncalibrated on gfx950: calibrate it on papr_generate_kernel, which writes exactly the shard
    for mode in ("default", "graph"):
        c = counters(os.path.join(src, f"pmc_write_{mode}"))
        gen = c.get("papr_generate_kernel", {}).get("WRITE_SIZE")
        if gen:
            factor = gib * (1 << 30) / (gen * 1024)
            for k, v in c.items():
                if "WRITE_SIZE" in v:
                    v["hbm_write_bytes_calibrated"] = v["WRITE_SIZE"] * 1024 * factor
            c["calibration"] = {"WRITE_SIZE_KiB_of_generate_kernel": gen, "bytes_it_wrote": gib * (1 << 30), "factor": factor}
            sw = c.get("papr_sweep_kernel", {}).get("hbm_write_bytes_calibrated")
            if sw is not None and "papr_sweep_kernel" in traffic and mode in traffic["papr_sweep_kernel"]:
                ent = traffic["papr_sweep_kernel"][mode]
                ent["hbm_read_bytes_per_launch"] = ent["hbm_bytes_per_launch"]
                ent["hbm_write_bytes_per_launch"] = sw
                ent["hbm_bytes_per_launch"] = ent["hbm_read_bytes_per_launch"] + sw
        summary[f"write_{mode}"] = c
    for name in ("pmc_tcc_", "pmc_sq_", "pmc_lds_", "pmc_tcc_twopass_", "pmc_sq_twopass_", "pmc_lds_twopass_", "pmc_grbm"):
        c = counters(os.path.join(src, name))
        if name.startswith("pmc_tcc"):
            for k, v in c.items():
                if "TCC_EA0_RDREQ_sum" in v:
                    v["rdreq_x128_bytes"] = v["TCC_EA0_RDREQ_sum"] * 128
                if v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0) > 0:
                    v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        summary[name.rstrip("_") + "_graph"] = c
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    if len(traffic) > 1:
        json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
