#!/usr/bin/env python3
"""Distil gpurun_out/<tag>/ (made by tools/collect_profiles.sh on the GPU box) into the tracked files under profiles/:

  <tag>_bench_<run>.json             the full records of the session's bench.py runs (full = the driver's command with every
                                     leg; default, exact, twopass, ts, ts_damage, torchrun1, torchrun1_hostpath, bursty,
                                     constant, miss); <tag>_bench_full.line = the compact stdout line of the driver's command
  <tag>_kernel_stats_<run>.csv       rocprofv3 --kernel-trace --stats summary of `bench.py [flags] --steps 20 --warmup 3`
  <tag>_stats_<run>.json             the bench line printed under the profiler in that very run
  <tag>_kernel_durations.json        the dominant kernel's average duration per MODE out of the kernel trace (bench.py
                                     runs its default and its -g leg in one process: the first half of the dispatches is
                                     the default leg), next to the HIP-event figure of the same run
  <tag>_pmc_summary.json             FETCH_SIZE / WRITE_SIZE per kernel and mode with the gfx950 corrections written out
  pmc_traffic.json                   what bench.py reports as roofline.traffic
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KERNELS = ("papr_ccdf_power_kernel", "papr_ccdf_kernel", "papr_stats_kernel", "papr_stats_finalize", "papr_generate_kernel",
           "papr_first_nan_kernel", "papr_sweep3_kernel", "papr_sweep_kernel", "papr_estimate_kernel", "papr_exact_",
           "ts_scan_kernel", "ts_merge_kernel", "ts_generate_kernel")


def short(name):
    for k in KERNELS:
        if k in name:
            if k == "papr_exact_":
                return name.split("(")[0].replace("void ", "")
            return k
    return name.split("(")[0]


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


def by_mode(rows, value):
    """rows of ONE kernel in dispatch order -> {'default': avg of the first half, 'graph': avg of the second}."""
    vals = [value(r) for r in rows]
    if len(vals) < 2 or len(vals) % 2:
        return {"any": sum(vals) / len(vals), "dispatches": len(vals)} if vals else {}
    h = len(vals) // 2
    return {"default": sum(vals[:h]) / h, "graph": sum(vals[h:]) / h, "dispatches": len(vals)}


def trace_durations(d, skip=0, timed=0):
    """Per kernel and leg: the average duration over ALL its dispatches, and — for a kernel launched once per step — over the
    dispatches of the TIMED steps alone (`skip` pre-heat + warm-up steps in front, then `timed` steps; bench.py's own
    figure is the average of exactly those)."""
    out = collections.defaultdict(list)
    for f in newest(os.path.join(d, "*", "*_kernel_trace.csv")):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            out[short(r["Kernel_Name"])].append(r)
    ms = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    res = {}
    for k, v in out.items():
        res[k] = by_mode(v, ms)
        n = len(v)
        if timed and n % 2 == 0 and n // 2 >= skip + timed:
            h = n // 2
            legs = {"default": v[skip:skip + timed], "graph": v[h + skip:h + skip + timed]}
            res[k]["timed_steps_only"] = {m: sum(ms(r) for r in rr) / len(rr) for m, rr in legs.items()}
    return res


def counters(d, counter):
    out = collections.defaultdict(list)
    for f in newest(os.path.join(d, "*", "*_counter_collection.csv")):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            if r["Counter_Name"] == counter:
                out[short(r["Kernel_Name"])].append(r)
    return {k: by_mode(v, lambda r: float(r["Counter_Value"])) for k, v in out.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    if not glob.glob(os.path.join(src, "stats_*")):  # (a gpurun call that found no box leaves nothing: do not write empty summaries)
        sys.exit(f"{src}: no collection to distil")
    for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "bench_full.line")):
        if os.path.getsize(f):
            shutil.copy(f, os.path.join(dst, f"{tag}_{os.path.basename(f)}"))
    durations, summary = {}, {
        "note": "per-dispatch averages of the separate rocprofv3 --pmc passes (one counter per pass, --kernel-trace only) over "
                "`bench.py --steps 3 --warmup 1` (default leg, then the -g leg: split by dispatch order).  FETCH_SIZE is in KiB "
                "and, on gfx950 with this rocprofv3, tallies each 128-B request of a 16 B/lane coalesced stream at 64 B "
                "(MI355X_MICROARCH.md, HBM section): hbm_read_bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE is calibrated on the "
                "generate kernel of the same run, which writes exactly the stream."}
    traffic_path = os.path.join(dst, "pmc_traffic.json")
    traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
    old_src = traffic.pop("source", None)
    traffic.pop("papr_sweep2_kernel<EXACT>", None)  # (the exact-sum sweep's first form: the laboratory's now)
    for k, v in traffic.items():
        for ent in v.values():
            ent.setdefault("source", old_src)
    for run, dom, gen in (("default", "papr_sweep_kernel", "papr_generate_kernel"),
                          ("exact", "papr_sweep3_kernel", "papr_generate_kernel"),
                          ("ts", "ts_scan_kernel", "ts_generate_kernel")):
        for f in newest(os.path.join(src, f"stats_{run}", "*", "*_kernel_stats.csv")):
            shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats_{run}.csv"))
        p = os.path.join(src, f"stats_{run}.json")
        line = None
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, f"{tag}_stats_{run}.json"))
            line = json.load(open(p))
        skip = timed = 0
        if line and run != "ts":
            skip = int((line.get("config", {}).get("preheat") or {}).get("steps", 0)) + int(line.get("warmup", 0))
            timed = int(line.get("steps", 0))
        tr = trace_durations(os.path.join(src, f"stats_{run}"), skip, timed)
        ent = {"kernel": dom, "rocprofv3_kernel_trace_avg_ms": tr.get(dom)}
        if line:
            ent["hip_events_same_run_ms"] = {"default": line["roofline"].get("kernel_ms"),
                                             "graph": (line.get("graph") or {}).get("roofline", {}).get("kernel_ms")}
            nbytes = line["roofline"].get("algorithmic_bytes_per_launch")
        else:
            nbytes = None
        durations[run] = ent
        fetch = counters(os.path.join(src, f"pmc_fetch_{run}"), "FETCH_SIZE")
        write = counters(os.path.join(src, f"pmc_write_{run}"), "WRITE_SIZE")
        try:
            pl = json.load(open(os.path.join(src, f"pmc_fetch_{run}.json")))
            # (TS: the bytes of the stream — what the generate kernel wrote — not the header lines the scan is priced on)
            nbytes = pl["config"]["bytes_per_gpu"] if run == "ts" else pl["roofline"]["algorithmic_bytes_per_launch"]
        except Exception:
            pass
        g = write.get(gen, {})
        gen_kib = g.get("any") or g.get("default")
        factor = (nbytes / (gen_kib * 1024)) if (gen_kib and nbytes) else None
        summary[run] = {"algorithmic_bytes_per_launch": nbytes, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                        "write_calibration": {"kernel": gen, "WRITE_SIZE_KiB": gen_kib, "bytes_it_wrote": nbytes,
                                              "factor": factor}}
        name = dom
        variant = (line or {}).get("roofline", {}).get("kernel_variant")  # the kernel FORM these counters belong to
        for mode in ("default", "graph", "any"):
            if mode in fetch.get(dom, {}):
                rd = 2 * fetch[dom][mode] * 1024
                wr = write.get(dom, {}).get(mode)
                wr = wr * 1024 * factor if (wr is not None and factor) else 0.0
                traffic.setdefault(name, {})[mode] = {
                    "gib_per_gpu": nbytes / (1 << 30) if nbytes else None,
                    "hbm_bytes_per_launch": rd + wr, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                    "kernel_variant": variant,
                    "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE, separate pass, x2 gfx950 "
                              "correction; + the kernel's writes from the WRITE_SIZE pass calibrated on the generate kernel)"}
    # TS scan: requests the L2 sent to memory and its hit / miss counts (one miss = one 128-byte line)
    tcc = {}
    for d, names in (("pmc_tcc_ts", ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum")), ("pmc_hit_ts", ("TCC_HIT_sum", "TCC_MISS_sum"))):
        for c in names:
            v = counters(os.path.join(src, d), c).get("ts_scan_kernel")
            if v:
                tcc[c] = v.get("any", v.get("default"))
    if tcc:
        if "TCC_EA0_RDREQ_sum" in tcc:
            tcc["hbm_read_bytes_if_128B_requests"] = tcc["TCC_EA0_RDREQ_sum"] * 128
        tcc["note"] = ("one memory-side request per L2 miss, none of them 32-byte: every request fills one 128-byte L2 line, "
                       "the line that holds a packet header (1.03125 lines per 188-byte packet)")
        summary["ts_tcc"] = tcc
    json.dump(durations, open(os.path.join(dst, f"{tag}_kernel_durations.json"), "w"), indent=1, sort_keys=True)
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    json.dump(traffic, open(traffic_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(durations, indent=1))
    print(json.dumps({k: {m: {f: e[f] for f in e if f != "source"} for m, e in v.items()} for k, v in traffic.items()}, indent=1))


if __name__ == "__main__":
    main()
