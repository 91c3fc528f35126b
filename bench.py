#!/usr/bin/env python3
"""bench.py — the papr hot path on N MI355X GPUs of one node.

One "step" = one complete papr run over the HBM-resident shard(s).  By default
it takes ONE read of the samples (papr_sweep.hip): a 1/64-sample mean estimate
(+ exchange), the guessed level table, the sweep kernel — pass 1 (power, double
sum, first-index peak + I/Q extrema) plus pass-2 counting against bands around
the guessed thresholds — the stats exchange, the host scalar stage (mean, PAPR,
the true level table), the recount of the few in-band samples and the count
exchange.  `--two-pass` runs the classic pass 1 / pass 2 kernels instead (two
reads; also what a step falls back to if the speculation misses), `--exact`
the same single read with the reference's sequential double sum reproduced
bit for bit (bin/papr's default; `--exact --exact-two-pass`: in two reads).
Every step recomputes everything from the samples; nothing carries over
between steps.  The plain invocation — what the driver runs — prints the
configs[1] line with the other legs as members: `graph` (configs[2]), `exact`
(both tables), `ts` / `ts_damaged` (the transport-stream packet scan of `--workload ts`, clean and with one damaged
spot per 1000 packets),
`cpu_baseline` / `cpu_baseline_graph` (the reference binary on one host core)
and `e2e` (bin/papr on the same workload from /dev/shm, PCIe-inclusive);
`--headline-only` leaves the members out.  Workload at N=1 is BASELINE.json configs[1]: 10 GiB of
synthetic gr_complex IQ, default mode ("peak+mean+1 dB histogram"); `--mode
graph` gives configs[2] (0.1 dB CCDF, ~301 bins).  With N ranks every rank owns
its own 10 GiB shard of an N x 10 GiB stream (weak scaling; N=8 is configs[3]).

Launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
one rank per GPU; the step's exchanges between the ranks are RCCL collectives on device buffers, queued by libpaprhip on
its own communicator and stream (--backend nccl, the default).  The bench's own bookkeeping — the barrier around the
timed steps, the max of the ranks' times, handing the RCCL id round — goes over gloo (--control), because a second RCCL
communicator (torch's) in the process costs the sweep kernel 1.3 % and the step 10 us.  Rank 0 prints ONE JSON line.  Inputs are already resident in HBM when the timed
region starts (generated on the device by the shared counter-hash generator).
"""
from __future__ import annotations

import argparse
import gc
import hashlib
import json
import os
import subprocess
import sys
import time
import zlib

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

PREHEAT_STEPS = 32   # default of --preheat: untimed steps (~50-70 ms) in front of every leg's warm-up (see run_mode)
LINE_LIMIT = 6000    # bytes of the ONE stdout line (the driver keeps the last 8000 characters of stdout + stderr)
E2E_RUNS = 5         # timed runs of bin/papr per table in the e2e leg: the median is reported
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def make_cfile(orc, path, n):
    """The bench workload (spike stream of n samples) as a file: oracle/mkcfile, one process per 1/16 of it."""
    workers = max(1, min(16, os.cpu_count() or 1, n // (1 << 22)))
    per = (n // workers + 8191) // 8192 * 8192
    procs = [subprocess.Popen([orc.MKCFILE, path, str(n), "--spike", "--part", str(w * per), str(min(per, n - w * per))])
             for w in range(workers) if w * per < n]
    if any(p.wait() != 0 for p in procs) or os.path.getsize(path) != n * 8:
        raise RuntimeError("mkcfile failed")


def cpu_baseline(pkg, mode: str, sample_gib: float):
    """Time the reference program (oracle/_ref/papr, compiled from the reference's
    own papr.c) — or, if it is absent, the oracle port — on a bounded sample of the
    same synthetic workload, single thread (the reference is single-threaded)."""
    orc = ge.load_oracle()
    ref = orc.REF_CLI if os.path.exists(orc.REF_CLI) else None
    binary, kind = (ref, "reference") if ref else (orc.CLI_PATH, "port")
    if not os.path.exists(binary):
        return None
    n = int(sample_gib * (1 << 30)) // 8
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(tmpdir, f"papr_bench_sample_{os.getpid()}.cfile")
    try:
        t0 = time.perf_counter()
        make_cfile(orc, path, n)
        gen_s = time.perf_counter() - t0
        args = [binary] + (["-g"] if mode == "graph" else []) + [path]
        # (no warm-up run: the file was written a moment ago and sits in the page cache)
        t0 = time.perf_counter()
        p = subprocess.run(args, capture_output=True)
        cpu_s = time.perf_counter() - t0
        # same file through the GPU product (ABI path) for an in-run parity check
        with pkg.PaprHip(int(os.environ.get("LOCAL_RANK", "0"))) as g:
            g.load_file(path)
            st = g.stats()
            mean, papr, table = pkg.levels(st, mode == "graph")
            text = pkg.format_report(st, mean, papr, g.ccdf(table), mode == "graph").encode()
        best = None
        if kind == "reference" and os.path.exists(orc.REF_CLI_VECTORISED):
            # the same reference source built for speed (auto-vectorised threshold loop); same stdout required
            t0 = time.perf_counter()
            pv = subprocess.run([orc.REF_CLI_VECTORISED] + args[1:], capture_output=True)
            best_s = time.perf_counter() - t0
            best = {"value": n / best_s / 1e6, "flags": "gcc -O3 -mavx2 -ffp-contract=off", "cores": 1,
                    "stdout_identical_to_O2": pv.stdout == p.stdout}
        return {"value": n / cpu_s / 1e6, "unit": "Msamples/s", "cores": 1, "kind": kind, "best_effort_build": best,
                "sample": f"{sample_gib:g} GiB ({n} samples) of the same spike workload, mode={mode}, "
                          f"levels={int(table.size)}, file in {tmpdir} (page cache warm), {cpu_s:.2f} s wall",
                "sample_short": f"{sample_gib:g} GiB ({n} samples) of the same stream, mode={mode}, {int(table.size)} levels, "
                                f"page cache warm, {cpu_s:.2f} s wall",
                "nproc": os.cpu_count(), "gpu_stdout_identical": text == p.stdout, "mkcfile_s": round(gen_s, 2)}
    finally:
        if os.path.exists(path):
            os.unlink(path)


def e2e_block(pkg, gib: float):
    """End to end (PCIe-inclusive; never `value`): `bin/papr` — the drop-in — on a file of the bench workload in
    /dev/shm: process start -> last byte of stdout, both modes, against the reference's recorded stdout."""
    orc = ge.load_oracle()
    n = int(gib * (1 << 30)) // 8 // 8192 * 8192
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(tmpdir, f"papr_bench_e2e_{os.getpid()}.cfile")
    out = {"file": f"{gib:g} GiB spike workload in {tmpdir} (page cache: written, then read twice)", "bytes": n * 8}
    try:
        # what the link gives a plain pinned hipMemcpy on this box: the ingest's H2D leg is priced against it
        import torch
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        src = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
        dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        ceiling = 0.0
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dst.copy_(src, non_blocking=True)
            b.record()
            b.synchronize()
            ceiling = max(ceiling, (1 << 30) / (a.elapsed_time(b) * 1e-3) / 1e9)
        del src, dst
        out["h2d_pinned_ceiling_GBps"] = round(ceiling, 2)
        make_cfile(orc, path, n)
        # "In the page cache" means read before: the first TWO reads of a freshly written tmpfs file run at 21-28 GB/s — its
        # pages go through the LRU lists, referenced on the first touch, activated on the second, under a lock 16 reader
        # threads fight over — every later one at the link's rate.  Two untimed runs, then the timed ones.
        for _ in range(2):
            subprocess.run([pkg.CLI_PATH, path], capture_output=True)
        for graph, tag in ((False, "default"), (True, "graph")):
            runs = []
            for _ in range(E2E_RUNS):
                # (a process that has just exited is still being taken down by the driver — 10 GiB of HBM to unmap — and a
                # runtime that starts during that takes 0.2 s instead of 0.08 to come up: every run starts on a quiet GPU)
                time.sleep(0.75)
                t0 = time.perf_counter()
                p = subprocess.run([pkg.CLI_PATH] + (["-g"] if graph else []) + [path], capture_output=True,
                                   env=dict(os.environ, PAPR_STATS="1"))
                runs.append((time.perf_counter() - t0, p))
            runs.sort(key=lambda r: r[0])
            dt, p = runs[len(runs) // 2]   # the MEDIAN run, and its own PAPR_STATS line
            golden, name = golden_report(1, gib, graph)
            info = {}
            try:
                info = json.loads(p.stderr.decode().splitlines()[-1])
            except Exception:
                pass
            out[tag] = {"seconds": dt, "seconds_is": f"median of {E2E_RUNS} runs (after 2 untimed ones)",
                        "seconds_all": [round(r[0], 4) for r in runs],
                        "all_stdout_identical": None if golden is None else all(r[1].stdout == golden for r in runs),
                        "msamples_per_s": n / dt / 1e6, "rc": p.returncode,
                        "stdout_identical_to_reference": None if golden is None else p.stdout == golden,
                        "ingest_GBps": info.get("ingest_GBps"),
                        "ingest_frac_of_h2d_ceiling": (round(info["ingest_GBps"] / ceiling, 3)
                                                       if info.get("ingest_GBps") and ceiling else None),
                        "open_s": info.get("open_s"),
                        "ingest_pass1_s": info.get("ingest_pass1_s"), "analysis_s": info.get("analysis_s"),
                        "exact_sum": info.get("exact_sum"), "gpus": info.get("gpus")}
    finally:
        if os.path.exists(path):
            os.unlink(path)
    return out


def step_summary(step_s):
    """min / median / p90 / max of the timed steps in ms, and which step was the slowest."""
    ms = np.asarray(step_s, dtype=np.float64) * 1e3
    if ms.size == 0:
        return None
    med = float(np.median(ms))
    stalled = ms > 2.0 * med   # a step several times its median: the host's wait came back late (seen: 8-9 ms, one box in three)
    return {"min": round(float(ms.min()), 4), "median": round(med, 4),
            "p90": round(float(np.percentile(ms, 90)), 4), "max": round(float(ms.max()), 4),
            "slowest": int(ms.argmax()), "n": int(ms.size),
            "stalled": int(stalled.sum()), "stalled_ms": round(float((ms[stalled] - med).sum()), 3),
            "all": [round(float(v), 3) for v in ms[:64]]}


def brief(summary):
    """A step summary without its list (the legs' one-liners inside the headline's roofline)."""
    return None if summary is None else {k: v for k, v in summary.items() if k != "all"}


def mmm(summary):
    """[min, median, max] of a step summary: the legs' one-liners."""
    return None if summary is None else [summary["min"], summary["median"], summary["max"]]


def leg_summary(line):
    """What the driver's record needs of a member leg (it keeps `roofline` and `config` whole, nothing else of a member)."""
    r = line["roofline"]
    return {"ms_per_step": round(line["ms_per_step"], 4), "value": round(line["value"], 1), "kernel": r["kernel"],
            "kernel_ms": round(r["kernel_ms"], 4), "frac": round(r["frac"], 4), "step_ms": mmm(r.get("step_ms")), "kernel_launch_ms": mmm(r.get("kernel_launch_ms")),
            "host_and_exchange_ms_per_step": round(r["host_and_exchange_ms_per_step"], 4),
            "all_kernels_frac_of_peak": round(line["kernels"]["all_kernels_frac_of_peak"], 4),
            "parity_in_run": line["parity_in_run"], "steps": line["steps"], "warmup": line["warmup"]}


def _r(v, nd=4):
    """Round floats for the compact line (the full record keeps every digit)."""
    return round(float(v), nd) if isinstance(v, (float, np.floating)) else v


def compact_line(full, full_path):
    """The ONE stdout line: the contract keys, `config`, `roofline` (with every other leg as a one-liner under
    `roofline.legs`, and the fractions the verdicts ask about repeated as plain numbers so that a record which keeps only
    the scalars of `roofline` still has them), `cpu_baseline[_graph]`, the parity flags — and, LAST, `headline`: the four
    numbers of the run again, so that the final characters of stdout carry them whatever is cut off in front.  Everything
    else (member legs whole, per-step arrays, kernel tables, the exchange's timing) is in the file `full` names."""
    c, r = full["config"], full["roofline"]
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(full["value"], 1), _r(full["ms_per_step"], 5)
    pre = c.get("preheat") or {}
    sw = c.get("one_sweep") or None
    line["config"] = {
        "workload": c["workload"], "mode": c.get("mode"), "signal": c.get("signal"), "forced_miss": c.get("forced_miss"),
        "samples_per_gpu": c.get("samples_per_gpu"), "samples_total": c.get("samples_total"), "levels": c.get("levels"),
        "papr_db": c.get("papr_db"), "reads_of_the_shard_per_step": c.get("reads_of_the_shard_per_step"),
        "exact_sequential_sum": c.get("exact_sequential_sum"), "sum_hex": c.get("sum_hex"),
        "counts_crc32": c.get("counts_crc32"), "sharding": c.get("sharding"), "exchange": c.get("exchange"),
        # untimed steps in front of the --warmup steps (an idle MI355X runs its first ~10 launches 5-7 % slow), and what the
        # first five untimed steps cost on the host clock: the price of a FIRST papr_hip_analyze on an idle GPU
        "preheat_steps": pre.get("steps"), "cold_first_steps_ms": pre.get("cold_first_steps_ms"),
        "one_sweep": (None if sw is None else {"resolved": sw.get("steps_resolved_from_the_sweep"), "stash": sw.get("stash_samples"),
                                               "band_log2": sw.get("band_log2"), "gave_up": sw.get("gave_up")}),
    }
    legs = r.get("legs") or {}
    k = full.get("kernels") or {}
    line["roofline"] = {
        "bound": r["bound"], "achieved": _r(r["achieved"], 2), "peak": r["peak"], "unit": r["unit"],
        "frac": _r(r["achieved"], 2) / r["peak"],   # (of the rounded figure beside it, exactly)
        "traffic": _r(r.get("traffic"), 0), "kernel": r["kernel"], "kernel_ms": _r(r["kernel_ms"], 5),
        "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch"), "kernel_variant": r.get("kernel_variant"),
        "traffic_source": (r.get("traffic_source") or "")[:90] or None, "traffic_stale": r.get("traffic_stale"),
        "traffic_over_algorithmic": _r(r.get("traffic_over_algorithmic"), 4), "traffic_measured_s": r.get("traffic_measured_s"),
        "step_ms": brief(r.get("step_ms")), "kernel_launch_ms": brief(r.get("kernel_launch_ms")),
        "step_ms_median": (r.get("step_ms") or {}).get("median"), "step_ms_max": (r.get("step_ms") or {}).get("max"),
        # steps that took more than twice the median (the host's wait coming back late: 8-9 ms now and then on some boxes,
        # the GPU clocking down behind it) and what they cost the timed region: `value` = total / elapsed includes them
        "stalled_steps": (r.get("step_ms") or {}).get("stalled"), "stalled_ms": (r.get("step_ms") or {}).get("stalled_ms"),
        "value_at_median_step": (_r(full["config"]["samples_total"] / ((r.get("step_ms") or {}).get("median") * 1e-3) / 1e6, 1)
                                 if (r.get("step_ms") or {}).get("median") else None),
        "kernels_ms_per_step": _r(r.get("kernels_ms_per_step"), 5),
        "host_and_exchange_ms_per_step": _r(r.get("host_and_exchange_ms_per_step"), 5),
        "all_kernels_frac_of_peak": _r(k.get("all_kernels_frac_of_peak"), 4),
        "wg_finish_median_last_us": ([(r.get("wg_finish_us") or {}).get("median"), (r.get("wg_finish_us") or {}).get("last")]
                                     if r.get("wg_finish_us") else None),
        "two_pass_equiv_frac": _r((r.get("survey_two_pass_equiv") or {}).get("frac"), 4),
        # the member legs' fractions of peak as plain numbers ...
        **{f"{name}_frac": leg.get("frac") for name, leg in legs.items() if leg.get("frac") is not None},
        **{f"{name}_ms_per_step": leg.get("ms_per_step") for name, leg in legs.items() if leg.get("ms_per_step") is not None},
        # ... and their one-liners
        "legs": legs,
    }
    for key in ("cpu_baseline", "cpu_baseline_graph"):
        cb = full.get(key)
        if isinstance(cb, dict):
            if "error" in cb:
                line[key] = {"error": str(cb["error"])[:200]}
                continue
            best = cb.get("best_effort_build") or {}
            line[key] = {"value": _r(cb["value"], 3), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                         "nproc": cb.get("nproc"), "sample": cb.get("sample_short") or cb["sample"][:120],
                         "gpu_stdout_identical": cb.get("gpu_stdout_identical"),
                         "O3_avx2_value": _r(best.get("value"), 3) if best else None}
    for key in ("weak", "strong"):   # more than one rank: both readings of the scaling target
        if isinstance(full.get(key), dict):
            line[key] = {k: v for k, v in full[key].items() if k not in ("step_ms", "kernel_launch_ms")}
    for key in ("parity_in_run", "parity_golden", "report_sha256", "device"):
        line[key] = full.get(key)
    line["full"] = full_path
    cb = line.get("cpu_baseline") or {}
    line["headline"] = {"value": line["value"], "ms_per_step": line["ms_per_step"], "roofline_frac": line["roofline"]["frac"],
                        "cpu_baseline_value": cb.get("value"), "parity_in_run": line["parity_in_run"]}
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:   # never again a line the driver cannot keep: drop the nested summaries first
        for leg in legs.values():
            for key in ("step_ms", "kernel_launch_ms"):
                leg.pop(key, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:
        line["roofline"].pop("legs", None)
        text = json.dumps(line, separators=(",", ":"))
    return text


def emit(full, full_path, real_stdout):
    """Rank 0: the full record to `full_path` (a file: the driver keeps the last 8000 characters of stdout AND stderr
    together, so nothing long may follow the line on either), the compact line — ONE line, <= LINE_LIMIT bytes — to stdout."""
    shown = None
    try:
        os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
        with open(full_path, "w") as f:
            json.dump(full, f)
            f.write("\n")
        shown = os.path.relpath(full_path, ROOT) if os.path.abspath(full_path).startswith(ROOT + os.sep) else full_path
    except OSError as e:
        print(f"bench.py: full record not written ({e})", file=sys.stderr)
    text = compact_line(full, shown) if full["unit"] == "Msamples/s" else compact_ts_line(full, shown)
    assert len(text) <= LINE_LIMIT and "\n" not in text, len(text)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(text, flush=True)
    os.dup2(2, 1)


def compact_ts_line(full, full_path):
    """--workload ts: the scan's own line without its per-step array."""
    line = json.loads(json.dumps(full))
    line["roofline"]["step_ms"] = brief(line["roofline"].get("step_ms"))
    line["full"] = full_path
    cb = line.get("cpu_baseline") or {}
    line["headline"] = {"value": round(line["value"], 1), "ms_per_step": round(line["ms_per_step"], 5),
                        "roofline_frac": round(line["roofline"]["frac"], 5), "cpu_baseline_value": cb.get("value")}
    return json.dumps(line, separators=(",", ":"))


def golden_report(world: int, gib: float, graph: bool):
    """The reference's recorded stdout for this run's global stream (tests/golden/, produced by the reference
    binary in the build container), or None when no golden exists for the configuration."""
    if abs(gib - 10.0) > 1e-12:
        return None, None
    name = f"big_spike{10 * world}g.{'graph' if graph else 'default'}.txt"
    path = os.path.join(ROOT, "tests", "golden", name)
    if not os.path.exists(path):
        return None, None
    with open(path, "rb") as f:
        return f.read(), name


def run_mode(args, mode, env):
    """Warm up, time exactly --steps steps of one mode, and (rank 0) build its JSON line."""
    pkg, gpu, xch = env["pkg"], env["gpu"], env["xch"]
    world, rank, device, use_dist = env["world"], env["rank"], env["device"], env["use_dist"]
    per_gpu, total, one_sweep = env["per_gpu"], env["total"], env["one_sweep"]
    graph = mode == "graph"
    args = argparse.Namespace(**{**vars(args), "mode": mode})

    result = {}

    def step():
        # ONE call into the library (papr_hip_analyze, include/papr_hip.h): mean estimate + its exchange, guessed
        # bands, the sweep (pass 1 + banded pass 2, with --exact also the rounding functions of the sequential
        # sum), exchange of the pass-1 records + ordered merge, mean / PAPR / level table as the reference derives
        # them, stash recount (or pass 2 if the speculation missed), with --exact the chained sequential sum, and
        # the all-reduce of the counters.  Nothing carries over between steps.
        res, table, counts = gpu.analyze(xch, graph, two_pass=not one_sweep, spoil_guess=args.force_miss)
        result["resolved"] = result.get("resolved", 0) + int(res.resolved)
        result["redo_tiles"] = result.get("redo_tiles", 0) + int(res.exact_redo_tiles)
        result["reruns"] = result.get("reruns", 0) + int(res.pass2_reruns)
        result["exact_done"] = result.get("exact_done", 0) + int(res.exact_sum)
        result.update(total=res.total, mean=res.mean, papr=res.papr, table=table, counts=counts)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The timing mode of the timed steps is on BEFORE the warm-up: a kernel launch with a pair of events bound to it
    # (hipExtLaunchKernelGGL) is its own path through the runtime — its first use, like every other first use of the step,
    # belongs to the warm-up, and the event pairs exist before the first of them is queued (papr_hip_set_timing).
    # HIP events on the kernels that read the shard (the roofline's kernel among them), bound to their dispatches; the
    # small estimate / recount kernels are timed in extra steps behind the timed region — a timed kernel carries a
    # completion signal of its own, which costs the stream ~5 us on either side (profiles/r02_step_timeline.txt)
    gc_was_on = gc.isenabled()
    gc.collect()   # (before the warm-up: nothing may leave the GPU idle between the warm-up and the timed steps)
    gc.disable()
    gpu.set_timing(2)
    # Pre-heat, in front of the W warm-up steps and like them untimed: an idle MI355X needs ~15-20 ms of load before its
    # kernels run at their steady speed (profiles/r04_driver_command_repeat.txt: with five 1.6 ms warm-up steps alone, the
    # first six TIMED sweeps took 1.64, 1.64, 1.62, 1.60, 1.57, 1.55 ms before settling at 1.53), and every leg of this
    # script starts on a GPU that has been idle while the host set the leg up.  Disclosed in config.preheat.
    preheat_steps = args.preheat   # (a COUNT, not a duration: with peers every step holds collectives all ranks must enter)
    cold = []                      # the first untimed steps on the idle GPU, each on the host clock: what a FIRST papr_hip_analyze costs
    t_heat = time.perf_counter()
    heat_s = 0.0
    for i in range(preheat_steps + args.warmup):
        t_c = time.perf_counter()
        step()
        if i < 5:
            cold.append(round((time.perf_counter() - t_c) * 1e3, 3))
        if i + 1 == preheat_steps:
            heat_s = time.perf_counter() - t_heat
    result["preheat"] = {"steps": preheat_steps, "seconds": round(heat_s, 4), "cold_first_steps_ms": cold,
                         "what": "untimed steps in front of the warm-up steps: the idle GPU reaches its steady speed after ~20 ms of load"}
    for k in ("resolved", "redo_tiles", "reruns", "exact_done"):
        result.pop(k, None)
    xch.timing(reset=True)
    gpu.set_timing(2)
    # Every step ends in its one wait, so the host clock around a step is that step's time: kept per step, so that the
    # line can say whether `value` (total / elapsed, below) is twenty equal steps or nineteen and a stall.
    step_s = [0.0] * args.steps
    fence()
    t0 = time.perf_counter()
    t_prev = t0
    for i in range(args.steps):
        step()
        t_now = time.perf_counter()
        step_s[i] = t_now - t_prev
        t_prev = t_now
    fence()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    tm = gpu.timing()
    launch_ms = {kind: gpu.timing_launches(kind) for kind in (0, 1, 2, 3)}   # per launch, in dispatch order
    xt = xch.timing().as_dict()
    if one_sweep:
        result["sweep_info"] = gpu.sweep_info().as_dict()
        try:   # when the last sweep launch's persistent workgroups were done (papr_hip_get_wg_finish), us after the first
            wf, where = gpu.wg_finish()
            if wf.size:
                result["wg_finish_us"] = {"workgroups": int(wf.size), "median": round(float(np.median(wf)), 1), "last": round(float(wf.max()), 1),
                                          "per_xcd_mean": [round(float(wf[where == x].mean()), 1) if (where == x).any() else None for x in range(8)],
                                          "xcd_of_first_workgroups": where[:8].tolist()}
        except Exception:
            pass
    aux_steps = min(5, args.steps)
    kept = dict(result)
    gpu.set_timing(1)
    for _ in range(aux_steps):
        step()
    tm_aux = gpu.timing()
    gpu.set_timing(False)
    result.clear()
    result.update(kept)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.control == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total * args.steps / elapsed / 1e6
        k_stats = tm.stats_ms / max(tm.stats_launches, 1)
        k_ccdf = tm.ccdf_ms / max(tm.ccdf_launches, 1)
        b_stats = tm.stats_bytes / max(tm.stats_launches, 1)
        b_ccdf = tm.ccdf_bytes / max(tm.ccdf_launches, 1)
        gbs_stats = b_stats / (k_stats * 1e-3) / 1e9 if k_stats else 0.0
        gbs_ccdf = b_ccdf / (k_ccdf * 1e-3) / 1e9 if k_ccdf else 0.0
        # exact mode: pass 2 runs inside the fused sweep (timed with its five small helper kernels)
        # (the one-read exact step has only helper kernels in this class: timed in the extra steps)
        tm_x, steps_x = (tm, args.steps) if tm.exact_launches else (tm_aux, aux_steps)
        k_exact = tm_x.exact_ms / max(tm_x.exact_launches, 1)
        b_exact = tm_x.exact_bytes / max(tm_x.exact_launches, 1)
        gbs_exact = b_exact / (k_exact * 1e-3) / 1e9 if k_exact else 0.0
        k_sweep = tm.sweep_ms / max(tm.sweep_launches, 1)
        b_sweep = tm.sweep_bytes / max(tm.sweep_launches, 1)
        gbs_sweep = b_sweep / (k_sweep * 1e-3) / 1e9 if k_sweep else 0.0
        aux_ms = tm_aux.aux_ms / aux_steps    # estimate + stash recount kernels, per step (from the extra steps)
        aux_bytes = tm_aux.aux_bytes / aux_steps
        # with --exact the sweep is the wave-private-segment kernel that also builds the rounding-function pairs (papr_sweep3_kernel)
        sweep_name = "papr_sweep3_kernel" if (args.exact and not args.exact_two_pass) else "papr_sweep_kernel"
        dom, dom_gbs, dom_ms = max([("papr_stats_kernel", gbs_stats, k_stats), ("papr_ccdf_kernel", gbs_ccdf, k_ccdf),
                                    ("papr_exact_seg_kernel<CCDF>", gbs_exact, k_exact),
                                    (sweep_name, gbs_sweep, k_sweep)], key=lambda e: e[2])
        kernel_ms_per_step = (tm.stats_ms + tm.ccdf_ms + tm.sweep_ms) / args.steps + tm_x.exact_ms / steps_x + aux_ms
        bytes_per_step = ((tm.stats_bytes + tm.ccdf_bytes + tm.sweep_bytes) / args.steps + tm_x.exact_bytes / steps_x
                          + aux_bytes)
        # HBM traffic from the PMC pass of the same command (profiles/pmc_traffic.json), taken over only if that pass ran
        # the kernel FORM this run ran (the library reports the form's id; a stale file says so instead of a number)
        kernel_variant = int(result.get("sweep_info", {}).get("kernel_variant", -1)) if one_sweep else None
        traffic, traffic_src, traffic_stale = None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(dom, {}).get(args.mode if args.mode in tj.get(dom, {}) else "any")
                if ent and abs(ent.get("gib_per_gpu", 0) - args.gib) < 1e-9:
                    if kernel_variant is None or ent.get("kernel_variant") == kernel_variant:
                        traffic, traffic_src = ent["hbm_bytes_per_launch"], ent.get("source", tj.get("source"))
                    else:
                        traffic_stale = (f"profiles/pmc_traffic.json was measured on kernel form {ent.get('kernel_variant')}, "
                                         f"this run used form {kernel_variant}")
            except Exception:
                pass
        # in-run parity: what this run would print, against what the REFERENCE printed for the same global stream
        report = pkg.format_report(result["total"], result["mean"], result["papr"], result["counts"], graph).encode()
        golden, golden_name = (golden_report(env.get("golden_world", world), args.gib, graph)
                               if args.signal == "gauss" and env.get("golden_world", world) else (None, None))
        parity = None if golden is None else (report == golden)
        line = {
            "metric": "IQ Msamples/s + achieved HBM GB/s (% of peak), 10 GiB cfile, 1/2/4/8 GPU",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": env.get("scaling", "weak"), "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"papr {'-g ' if graph else ''}on {per_gpu * 8 / (1 << 30):g} GiB synthetic gr_complex IQ per GPU "
                                   f"({'0.1 dB CCDF' if graph else 'peak+mean+1 dB histogram'}), HBM-resident, "
                                   f"{world}xMI355X",
                       "mode": args.mode, "signal": args.signal, "forced_miss": bool(args.force_miss),
                       "preheat": result.get("preheat"),
                       "samples_per_gpu": per_gpu, "samples_total": total,
                       "levels": int(result["table"].size), "papr_db": round(float(result["papr"]), 6),
                       "reads_of_the_shard_per_step": 1 if one_sweep and result.get("resolved", 0) == args.steps else 2,
                       "one_sweep": ({"steps_resolved_from_the_sweep": int(result.get("resolved", 0)),
                                      "exact_redo_tiles_per_step": (result.get("redo_tiles", 0) / args.steps
                                                                    if args.exact else None),
                                      **{k: result.get("sweep_info", {}).get(k) for k in
                                         ("stash_samples", "stash_capacity", "estimate_samples", "band_log2", "reason", "gave_up")}}
                                     if one_sweep else None),
                       "exact_sequential_sum": bool(args.exact) and result.get("exact_done", 0) == args.steps,
                       "sum_hex": float(result["total"].sum).hex(),
                       "exact_pass2_reruns": int(result.get("reruns", 0)),
                       "counts_crc32": zlib.crc32(np.ascontiguousarray(result["counts"], dtype=np.uint64).tobytes()),
                       "sharding": f"sample axis, {world} contiguous shard(s)",
                       "exchange": ("RCCL" if args.backend == "nccl" else "gloo") +
                                   (" all-gather(estimate) +" if one_sweep else "") +
                                   " all-gather(stats) + all-reduce(counts)" if use_dist else "none"},
            "roofline": {"bound": "hbm", "achieved": dom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": dom_gbs / HBM_PEAK_GBS, "traffic": traffic, "kernel": dom,
                         "kernel_ms": dom_ms, "algorithmic_bytes_per_launch": per_gpu * 8,
                         "kernel_variant": kernel_variant, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         # the timed region step by step (host clock around each papr_hip_analyze, which ends in the step's
                         # one wait): `value` = total / elapsed, i.e. the MEAN — a stall shows as max >> median here
                         "step_ms": step_summary(step_s),
                         # the dominant kernel launch by launch (HIP events bound to the dispatches): is it the kernel or
                         # the host that makes a slow step slow
                         "kernel_launch_ms": step_summary(launch_ms[{"papr_stats_kernel": 0, "papr_ccdf_kernel": 1,
                                                                     "papr_exact_seg_kernel<CCDF>": 2}.get(dom, 3)] / 1e3),
                         "kernels_ms_per_step": kernel_ms_per_step,
                         "host_and_exchange_ms_per_step": ms_per_step - kernel_ms_per_step,
                         # how evenly the launch ends: its persistent workgroups' finish times after the first one's (the
                         # kernels walk the shard with a skew between the XCDs, which do not read at one speed)
                         "wg_finish_us": result.get("wg_finish_us"),
                         # SURVEY.md 8(d) prices a papr result at 16 B/sample (two reads: 500 000 Msamples/s = 100 %).  The
                         # sweep kernel does the work of both passes in ONE read, so `frac` above is priced on the 8 B/sample
                         # it actually moves; on the survey's two-pass convention the same launch retires 16 B/sample:
                         "survey_two_pass_equiv": ({"achieved": 2 * dom_gbs, "frac": 2 * dom_gbs / HBM_PEAK_GBS,
                                                    "note": "2 x achieved: one read doing both passes' work, on SURVEY 8(d)'s "
                                                            "16 B/sample scale (may exceed 1; not a bandwidth claim)"}
                                                   if one_sweep and result.get("resolved", 0) == args.steps else None)},
            # stdout of this run (last step) == the reference program's recorded stdout for the same stream
            # (tests/golden/<name>; null: no golden for this size / rank count)
            "parity_in_run": parity, "parity_golden": golden_name,
            "report_sha256": hashlib.sha256(report).hexdigest(),
            "kernels": {"papr_stats_kernel": {"avg_ms": k_stats, "GB/s": gbs_stats, "launches": int(tm.stats_launches)},
                        "papr_ccdf_kernel": {"avg_ms": k_ccdf, "GB/s": gbs_ccdf, "launches": int(tm.ccdf_launches)},
                        sweep_name: {"avg_ms": k_sweep, "GB/s": gbs_sweep, "launches": int(tm.sweep_launches)},
                        "estimate_and_recount_kernels": {"ms_per_step": aux_ms, "launches": int(tm_aux.aux_launches),
                                                         "bytes_per_step": aux_bytes,
                                                         "timed_in": f"{aux_steps} extra steps behind the timed region"},
                        "all_kernels_frac_of_peak": bytes_per_step / (kernel_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
                        if kernel_ms_per_step else 0.0,
                        "papr_exact_kernels": {"avg_ms": k_exact, "GB/s": gbs_exact, "launches": int(tm_x.exact_launches)},
                        "host_and_exchange_ms_per_step": ms_per_step - kernel_ms_per_step},
            # host wall time inside the C-ABI exchanges (H2D + collective + D2H + one stream sync each), rank 0
            "exchange": {"transport": xch.transport, "world": world, "bench_bookkeeping_over": args.control if use_dist else None, **xt},
            "device": gpu.name,
        }
        return line
    return None


def run_ts(args, rank, world, local_rank, use_dist):
    """--workload ts: the transport-stream packet scan (include/ts_hip.h).  One step = one
    complete scan of the HBM-resident stream: sync lock, per-PID count / first / last.  Every rank owns its own
    10 GiB stream (independent streams: no exchange)."""
    pkg = ge.load_package()
    from dtv_utils_amd import ts
    npackets = int(args.gib * (1 << 30)) // 188
    gpu = ts.TsHip(local_rank)
    period = int(round(1.0 / args.damage)) if args.damage > 0 else 0
    if period:   # one damaged spot (inserted / missing bytes, an overwritten sync byte) every `period` packets
        npackets = npackets // (4 * period) * (4 * period)
        gpu.generate_damaged(npackets, period, seed=0x7500001 + rank)
    else:
        gpu.generate(npackets, seed=0x7500001 + rank)
    res = None
    first_launches = None
    preheat_steps = args.preheat
    cold = []
    for i in range(preheat_steps + args.warmup):   # (as run_mode: the idle GPU's first ~20 ms of load are slower)
        t_c = time.perf_counter()
        res = None
        res = gpu.scan()
        if i < 5:
            cold.append(round((time.perf_counter() - t_c) * 1e3, 3))
        if i == 0:
            first_launches = int(res.launches)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    step_s = [0.0] * args.steps
    gc_was_on = gc.isenabled()
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    t_prev = t0
    kernel_ms = merge_ms = 0.0
    for i in range(args.steps):
        res = None   # (a result somebody still holds is given its complete lists before the next scan: nobody holds this one)
        res = gpu.scan()
        kernel_ms += res.kernel_ms
        merge_ms += res.merge_ms
        t_now = time.perf_counter()
        step_s[i] = t_now - t_prev
        t_prev = t_now
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", local_rank) if args.control == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        gpu.close()
        return None
    nbytes = npackets * 188 - (npackets // (4 * period) if period else 0)
    k_ms = kernel_ms / args.steps
    # Algorithmic bytes of the scan kernel: the 128-byte lines that hold a packet header (its sync byte, PID,
    # adaptation_field_control and adaptation_field_length: an aligned 8-byte window that starts at the sync byte) —
    # HBM cannot deliver less than a line, and at a 188-byte stride no two packets share one.  The reference reads
    # every byte of the stream; `stream_GBps` is the rate on that scale.
    starts = (np.arange(min(npackets, 1 << 16), dtype=np.int64) * 188) & ~3   # the pattern repeats every 32 packets
    lines_per_packet = float(np.mean(1 + ((starts % 128) + 7) // 128))
    line_bytes = npackets * lines_per_packet * 128
    traffic, traffic_src = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        ent = tj.get("ts_scan_kernel", {}).get("default")
        if ent and abs(ent.get("gib_per_gpu", 0) - args.gib) < 1e-3:
            traffic, traffic_src = ent["hbm_bytes_per_launch"], ent.get("source")
    except Exception:
        pass
    line = {
        "metric": "TS Mpackets/s + achieved HBM GB/s (% of peak), 10 GiB transport stream per GPU",
        "value": npackets * world * args.steps / elapsed / 1e6, "unit": "Mpackets/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"xport -p packet scan (sync lock, per-PID count/first/last, continuity counters) on {args.gib:g} GiB synthetic "
                               f"MPEG-2 TS per GPU, HBM-resident, {world}xMI355X", "packets_per_gpu": npackets,
                   "bytes_per_gpu": nbytes, "launches_per_scan": int(res.launches), "walks_per_scan": int(res.walks),
                   "launches_of_the_first_scan": first_launches,
                   "form": "a context starts a scan in the form its last scan ended in: the first scan of a damaged stream tries the full "
                           "tables and is given up for the slot form (launches_of_the_first_scan), the timed ones start in the slot form",
                   "damage": (f"one damaged spot every {period} packets (include/ts_synth.h: ts_synth_damaged_byte)" if period else None),
                   "preheat": {"steps": preheat_steps, "cold_first_steps_ms": cold, "what": "untimed scans in front of the warm-up"},
                   "sync_error_lines": int(res.nsync_errors), "discontinuity_lines": int(res.ndiscontinuities),
                   "packets_counted": int(res.packets),
                   "pids_seen": int(np.count_nonzero(res.tables()[0])), "sharding": "independent streams, no exchange",
                   "report_sha256": hashlib.sha256(res.report()).hexdigest(),
                   "stream_GBps": nbytes * world * args.steps / elapsed / 1e9},
        "roofline": {"bound": "hbm", "achieved": line_bytes / (k_ms * 1e-3) / 1e9 if k_ms else 0.0, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": (line_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms else 0.0,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "ts_scan_kernel", "kernel_ms": k_ms, "merge_kernel_ms": merge_ms / args.steps,
                     "step_ms": step_summary(step_s),
                     "algorithmic_bytes_per_launch": line_bytes,
                     "algorithmic_bytes_note": f"{lines_per_packet:.4f} 128-byte lines per packet (the header's); the stream "
                                               f"itself is {nbytes} bytes"},
        "device": None,
    }
    if not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = ts_cpu_baseline(gpu, ts, args.cpu_sample_gib or 2.0, period)
        except Exception as e:
            line["cpu_baseline"] = {"error": repr(e)}
    gpu.close()
    return line


def ts_cpu_baseline(gpu, ts, sample_gib: float, period: int = 0):
    """The reference's xport (oracle/_ref/xport, compiled from its own source) — or the oracle port — on a bounded
    sample of the same synthetic stream, single thread, and the GPU scan's report against it on that sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ts_oracle
    n = int(sample_gib * (1 << 30)) // 188
    if period:
        n = n // (4 * period) * (4 * period)
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(tmpdir, f"ts_bench_sample_{os.getpid()}.ts")
    try:
        # (the sample file comes from the device-side generator — byte-identical to oracle/mkts's, which the tests pin —
        # because the host tool makes 0.1 GB/s)
        size = gpu.generate_damaged(n, period) if period else (gpu.generate(n) or n * 188)
        with open(path, "wb") as f:
            for off in range(0, size, 1 << 28):
                f.write(gpu.download(off, min(1 << 28, size - off)))
        use_ref = os.path.exists(ts_oracle.REF_CLI)
        cmd = [ts_oracle.REF_CLI, "-ps", path, ts_oracle.REF_PROGRAM, "1", "1"] if use_ref else [ts_oracle.CLI_PATH, path]
        subprocess.run(cmd, capture_output=True)   # page-cache warm, untimed
        runs, cpu_total = 0, 0.0
        while cpu_total < 10.0 and runs < 64:       # about 10 s of CPU work: the same file again and again
            t0 = time.perf_counter()
            p = subprocess.run(cmd, capture_output=True)
            cpu_total += time.perf_counter() - t0
            runs += 1
        cpu_s = cpu_total / runs
        want = ts_oracle.filter_lines(p.stdout)
        gpu.load_file(path)
        got = gpu.scan().report()
        return {"value": n / cpu_s / 1e6, "unit": "Mpackets/s", "cores": 1, "kind": "reference" if use_ref else "port",
                "GB/s": n * 188 / cpu_s / 1e9,
                "sample": f"{sample_gib:g} GiB ({n} packets) of the same synthetic stream, file in {tmpdir} (page cache "
                          f"warm), {runs} runs, {cpu_s:.2f} s wall each", "nproc": os.cpu_count(), "gpu_report_identical": got == want}
    finally:
        if os.path.exists(path):
            os.unlink(path)


def traffic_child(args):
    """`bench.py --traffic-child` (run by live_traffic under rocprofv3 --pmc): the step itself and nothing else — the shard
    generated on the device (its generate kernel writes exactly the stream: the WRITE_SIZE calibration), one untimed step and
    two counted ones of the 1 dB table, then the same of the 0.1 dB table.  Prints nothing."""
    pkg = ge.load_package()
    from dtv_utils_amd import exchange
    per_gpu = int(args.gib * (1 << 30)) // 8 // 8192 * 8192
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    shard = torch.empty(per_gpu * 8 + 65536, dtype=torch.uint8, device=device)
    gpu = pkg.PaprHip(0)
    gpu.adopt(shard.data_ptr(), per_gpu, base_index=0, keepalive=shard)
    gpu.generate(pkg.SynthSpec.spike(per_gpu, envelope=args.signal), 0, per_gpu)
    if args.exact:
        gpu.set_exact(True)
    xch = exchange.Exchange.single()
    for graph in (False, True):
        for _ in range(3):
            gpu.analyze(xch, graph)
    torch.cuda.synchronize()
    gpu.close()


def live_traffic(args, kernel, nbytes):
    """HBM traffic of the dominant kernel measured IN THIS RUN (outside the timed region, outside `value`): the same step in
    a child process under `rocprofv3 --pmc`, one counter per pass with --kernel-trace only, collected and corrected as
    MI355X_MICROARCH.md's HBM section prescribes — FETCH_SIZE (KiB) tallies each 128-byte request of a 16 B/lane stream at
    64 bytes on gfx950: x2; WRITE_SIZE is calibrated on the generate kernel of the same pass, which writes exactly the
    stream.  Returns {"default": bytes per launch, "graph": ...} + the source text, or None (no rocprofv3 on the box, or a
    pass that left no counters: the caller keeps profiles/pmc_traffic.json's constant and says so)."""
    import csv
    import glob
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rp or any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_")) for k in os.environ):
        return None   # (no profiler on the box — or this run is itself being profiled: no profiler inside a profiler)
    tmp = tempfile.mkdtemp(prefix="papr_traffic_", dir="/tmp")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [rp, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--traffic-child", "--gib", str(args.gib), "--signal", args.signal]
            if args.exact:
                cmd.append("--exact")
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
            p = subprocess.run(cmd, capture_output=True, timeout=600, cwd="/tmp", env=dict(env, TMPDIR="/tmp"))
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                rows += [r for r in csv.DictReader(open(f, newline="")) if r.get("Counter_Name") == counter]
            if p.returncode != 0 or not rows:
                return None
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            dom = [float(r["Counter_Value"]) for r in rows if kernel + "<" in r["Kernel_Name"] or kernel + "(" in r["Kernel_Name"]]
            gen = [float(r["Counter_Value"]) for r in rows if "papr_generate_kernel" in r["Kernel_Name"]]
            if len(dom) != 6 or not gen:
                return None
            # (three launches per table: the first one, on the idle GPU, is not counted)
            vals[counter] = {"default": sum(dom[1:3]) / 2, "graph": sum(dom[4:6]) / 2, "generate": gen[0]}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    wfac = nbytes / (vals["WRITE_SIZE"]["generate"] * 1024) if vals["WRITE_SIZE"]["generate"] else 0.0
    out = {}
    for m in ("default", "graph"):
        rd = 2.0 * vals["FETCH_SIZE"][m] * 1024
        wr = vals["WRITE_SIZE"][m] * 1024 * wfac
        out[m] = {"traffic": rd + wr, "read": rd, "write": wr}
    out["source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, a pass each (--kernel-trace only) over a child "
                     "process running the same step; x2 gfx950 correction on the reads, writes calibrated on the generate kernel")
    out["write_calibration_factor"] = wfac
    return out


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args) -> int:
    """`python bench.py --gpus N ...` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): this process
    becomes the launcher.  N copies of the same command line, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
    torch.distributed.run would set them, rendezvous on 127.0.0.1), rank 0 on this process's stdout — it prints the ONE line —
    and every other rank's stdout on stderr.  Fewer visible GPUs than ranks is refused in one stderr line before anything is
    started (unless --backend gloo: ranks share GPUs round-robin, the code path and not a measurement).  A rank that fails
    takes the others down with it (exact PIDs), and the exchange's watchdog (PAPR_XCH_TIMEOUT_S) turns a lost rank into that."""
    import signal
    n = args.gpus
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible == 0:
        print("bench.py: needs a GPU: the papr product has no CPU path", file=sys.stderr)
        return 3
    if args.backend != "gloo" and visible < n:
        print(f"bench.py: --gpus {n} but {visible} GPU(s) visible: one rank per GPU over RCCL needs {n} devices "
              f"(--backend gloo lets ranks share a GPU: the N>1 code path, not a measurement)", file=sys.stderr)
        return 3
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               PAPR_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=None if r == 0 else sys.stderr.fileno()))

    def stop_all(sig=signal.SIGTERM):
        for q in procs:
            if q.poll() is None:
                try:
                    q.send_signal(sig)
                except OSError:
                    pass

    def on_signal(signum, _frame):
        stop_all(signum)
        raise SystemExit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, on_signal)
    deadline = time.monotonic() + float(os.environ.get("PAPR_BENCH_LAUNCH_TIMEOUT_S", "3600"))
    rc = 0
    try:
        while True:
            states = [q.poll() for q in procs]
            bad = [(r, c) for r, c in enumerate(states) if c not in (None, 0)]
            if bad:
                r, rc = bad[0]
                print(f"bench.py: rank {r} of {n} ended with status {rc}: stopping the other ranks", file=sys.stderr)
                break
            if all(c == 0 for c in states):
                return 0
            if time.monotonic() > deadline:
                print("bench.py: the ranks did not finish inside PAPR_BENCH_LAUNCH_TIMEOUT_S: stopping them", file=sys.stderr)
                rc = 124
                break
            time.sleep(0.05)
    finally:
        stop_all()
        t_end = time.monotonic() + 5.0
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()
    return rc if 0 < rc < 256 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["default", "graph", "both"], default="both",
                    help="default = configs[1] (1 dB table), graph = configs[2] (papr -g, 0.1 dB CCDF); both (the "
                         "default) times configs[1] as the headline and adds configs[2] under \"graph\" in the same line")
    ap.add_argument("--gib", type=float, default=10.0, help="GiB of IQ per GPU")
    ap.add_argument("--signal", choices=["gauss", "bursty", "constant"], default="gauss",
                    help="envelope of the synthetic IQ (include/papr_synth.h): gauss = the BASELINE workload; bursty "
                         "(on/off keying) and constant (constant envelope: every sample ON the 0 dB threshold) are the "
                         "one-sweep speculation's hard cases")
    ap.add_argument("--force-miss", action="store_true",
                    help="feed the sweep a guess that is 3 %% off, so that every step pays the second read")
    ap.add_argument("--workload", choices=["papr", "ts"], default="papr",
                    help="papr = BASELINE.json's metric (the default); ts = the transport-stream packet scan of "
                         "xport.c (SURVEY.md 8(f) N4), a second line of its own")
    ap.add_argument("--damage", type=float, default=0.0,
                    help="--workload ts: damaged spots per packet (1e-4 = one every 10^4 packets): the scan's unfriendly case")
    ap.add_argument("--exact", action="store_true",
                    help="also reproduce the reference's sequential double sum bit for bit every step "
                         "(rounding functions computed in the pass-2 sweep + a short host chain); off by default")
    ap.add_argument("--exact-two-pass", action="store_true",
                    help="with --exact: the round-1 form (pass 1 leaving per-tile sums, then the fused rounding-function + "
                         "pass-2 sweep: two reads) instead of the one-read form")
    ap.add_argument("--two-pass", action="store_true",
                    help="read the shard twice (papr_stats_kernel + papr_ccdf_kernel) instead of the one-sweep path")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend; gloo (+ ranks sharing GPUs round-robin) exists so the N>1 "
                         "code path can be exercised on a box with fewer GPUs than ranks")
    ap.add_argument("--control", choices=["gloo", "nccl"], default="gloo",
                    help="what carries the bench's OWN bookkeeping between the ranks (the barrier around the timed steps, the "
                         "max of the ranks' times, the RCCL id): gloo by default — the data path's collectives are RCCL either "
                         "way (--backend nccl), but a second RCCL communicator, torch's, beside the one the step uses costs "
                         "the sweep kernel 1.3 %% and the step 10 us (measured at world size 1)")
    ap.add_argument("--preheat", type=int, default=PREHEAT_STEPS,
                    help="untimed steps in front of the --warmup steps of every leg (an idle MI355X runs its first ~10 launches "
                         "5-7 %% slow); 0 = the warm-up steps alone.  The first five untimed steps are reported either way "
                         "(config.cold_first_steps_ms)")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="where the full record goes (every member leg whole, per-step arrays): stdout carries only the compact "
                         "line the driver keeps (<= 6000 bytes)")
    ap.add_argument("--cpu-sample-gib", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline (+ graph) legs: no exact-sum member, no TS member, no -g CPU baseline")
    ap.add_argument("--no-selftest", action="store_true",
                    help="with peers (or under torchrun): skip the exchange's self-test in front of the timed work")
    ap.add_argument("--all-legs", action="store_true",
                    help="with more than one rank: also run the exact-sum and packet-scan members (default there: headline + graph)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="keep roofline.traffic at the constant of profiles/pmc_traffic.json instead of measuring it in this run "
                         "(two rocprofv3 --pmc passes over a child process behind the timed region: ~30 s)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-strong", action="store_true",
                    help="with more than one rank: leave the strong-scaling member out (the --gib stream cut into N shards, "
                         "beside the weak headline where every rank has --gib of its own)")
    ap.add_argument("--no-ts", action="store_true", help="leave the transport-stream member out of the plain invocation's line")
    ap.add_argument("--member-steps", type=int, default=20, help="steps of the exact / ts members of the plain invocation")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the end-to-end leg (the drop-in CLI on a file in /dev/shm -> stdout; PCIe-inclusive, reported "
                         "under \"e2e\", never as `value`)")
    args = ap.parse_args()
    if args.traffic_child:
        return traffic_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's N=1 command with --gpus N: no launcher around us, so be one (one rank per GPU, rank 0 prints the line)
        sys.exit(launch_ranks(args))

    # stdout carries exactly ONE line (the JSON): library chatter during set-up (e.g. the RCCL
    # version banner printed at communicator creation) is routed to stderr until then
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the papr product has no CPU path")
    if args.backend == "gloo":
        args.control = "gloo"
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but {torch.cuda.device_count()} GPU(s) are visible: one "
                         f"rank per GPU over RCCL (--backend gloo lets ranks share a GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ   # under torchrun even N=1 goes through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.control == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    if args.workload == "ts":
        line = run_ts(args, rank, world, local_rank, use_dist)
        if rank == 0:
            emit(line, args.full_json, real_stdout)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    pkg = ge.load_package()
    from dtv_utils_amd import exchange
    per_gpu = int(args.gib * (1 << 30)) // 8 // 8192 * 8192   # samples per rank, chunk aligned
    total = per_gpu * world
    shard = torch.empty(per_gpu * 8 + 65536, dtype=torch.uint8, device=device)   # HBM-resident shard
    gpu = pkg.PaprHip(local_rank)
    gpu.adopt(shard.data_ptr(), per_gpu, base_index=rank * per_gpu, keepalive=shard)
    gpu.generate(pkg.SynthSpec.spike(total, envelope=args.signal), rank * per_gpu, per_gpu)
    if use_dist:
        # a rank that waits longer than this for its peers cancels the exchange and the run ends with an error instead of
        # hanging until the driver's own limit (include/papr_exchange.h)
        os.environ.setdefault("PAPR_XCH_TIMEOUT_S", "120")
    if not use_dist:
        xch = exchange.Exchange.single()
    elif args.backend == "nccl":
        xch = exchange.Exchange.rccl(gpu, rank, world)      # ncclCommInitRank inside libpaprhip; collectives on its stream
    else:
        xch = exchange.Exchange.over_torch()                # the same C exchange code over gloo
    if use_dist and not args.no_selftest:
        # every collective of the step once, on tiny buffers with predictable contents, before anything is timed: rank 0
        # prints one stderr line per collective with its microseconds; a failure names the collective and ends the run
        xch.selftest(gpu, verbose=True)
    if args.exact:
        gpu.set_exact(True)
    one_sweep = not (args.two_pass or (args.exact and args.exact_two_pass))
    env = dict(pkg=pkg, gpu=gpu, xch=xch, world=world, rank=rank, device=device, use_dist=use_dist, per_gpu=per_gpu,
               total=total, one_sweep=one_sweep)
    modes = ["default", "graph"] if args.mode == "both" else [args.mode]
    lines = [run_mode(args, m, env) for m in modes]
    GRAPH_KEYS = ("value", "unit", "ms_per_step", "config", "roofline", "kernels", "parity_in_run", "parity_golden",
                  "report_sha256")
    # Strong scaling beside the weak headline (BASELINE.json quotes the metric on "10 GiB cfile, 1/2/4/8 GPU", which reads
    # either way): the SAME --gib stream cut into `world` shards, every rank --gib / world of it.  Same shard memory, same
    # context, same exchange; for 10 GiB the report is checked against the reference's recorded stdout of the 10 GiB stream.
    strong_lines = None
    if world > 1 and not args.no_strong and not args.two_pass and not args.force_miss:
        whole = int(args.gib * (1 << 30)) // 8 // 8192 * 8192
        per_s = whole // world // 8192 * 8192
        if per_s >= 8192:
            total_s = per_s * world
            gpu.adopt(shard.data_ptr(), per_s, base_index=rank * per_s, keepalive=shard)
            gpu.generate(pkg.SynthSpec.spike(total_s, envelope=args.signal), rank * per_s, per_s)
            env_s = dict(env, per_gpu=per_s, total=total_s, scaling="strong", golden_world=1 if total_s == whole else 0)
            strong_lines = [run_mode(args, m, env_s) for m in modes]
    # The plain invocation (what the driver runs) carries every leg in its ONE line: configs[1] as the headline,
    # configs[2] under "graph", the same two tables with the reference's sequential sum reproduced (what bin/papr does
    # by default) under "exact", the transport-stream scan under "ts", the reference's own CPU time for both tables.
    # With more than one rank the plain invocation times the headline and configs[2] only: every member leg holds
    # collectives all ranks must enter behind its own pre-heat, minutes of extra work in front of the line (--all-legs).
    full = (args.mode == "both" and not args.exact and not args.two_pass and not args.force_miss and args.signal == "gauss"
            and not args.headline_only and (world == 1 or args.all_legs))
    exact_lines = None
    if full:
        gpu.set_exact(True)
        exact_args = argparse.Namespace(**{**vars(args), "exact": True, "steps": max(1, min(args.steps, args.member_steps)),
                                           "warmup": min(args.warmup, 2)})
        exact_lines = [run_mode(exact_args, m, env) for m in modes]
        gpu.set_exact(False)
    gpu.close()
    del shard
    torch.cuda.empty_cache()
    ts_line = ts_damaged_line = None
    if full and not args.no_ts:
        ts_args = argparse.Namespace(**{**vars(args), "steps": max(1, min(args.steps, args.member_steps)),
                                        "warmup": min(args.warmup, 2)})
        ts_line = run_ts(ts_args, rank, world, local_rank, use_dist)
        # ... and the scan's unfriendly case: one damaged spot per 1000 packets (no CPU member: the clean leg has it)
        ts_damaged_line = run_ts(argparse.Namespace(**{**vars(ts_args), "damage": 1e-3, "no_cpu_baseline": True}),
                                 rank, world, local_rank, use_dist)
    if rank == 0:
        line = lines[0]
        legs = {}
        if len(lines) > 1:   # configs[2] rides along: same shard, same code path, the 0.1 dB table
            line["graph"] = {k: lines[1][k] for k in GRAPH_KEYS}
            legs["graph"] = leg_summary(lines[1])
        if world > 1:
            # both readings of "1/2/4/8 GPU" in one line: `weak` repeats the headline's numbers, `strong` is the --gib stream
            # cut into `world` shards.  speedup_vs_n1 is the driver's to compute from its own N=1 run: left null.
            def scaling_member(ls):
                m = dict(leg_summary(ls[0]), scaling=ls[0]["scaling"], n_gpus=world,
                         gib_per_gpu=round(ls[0]["config"]["samples_per_gpu"] * 8 / (1 << 30), 6),
                         gib_total=round(ls[0]["config"]["samples_total"] * 8 / (1 << 30), 6),
                         parity_golden=ls[0]["parity_golden"], speedup_vs_n1=None)
                if len(ls) > 1:
                    g = leg_summary(ls[1])
                    m["graph"] = {k: g[k] for k in ("ms_per_step", "value", "kernel_ms", "frac", "parity_in_run")}
                return m
            line["weak"] = scaling_member(lines)
            if strong_lines:
                line["strong"] = scaling_member(strong_lines)
                line["strong_full"] = strong_lines
        if exact_lines:
            want_sum = None
            try:
                mf = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
                if abs(args.gib - 10.0) < 1e-12:
                    want_sum = mf.get(f"big_spike{10 * world}g", {}).get("oracle_sequential_sum_hex")
            except Exception:
                pass
            ex = {k: exact_lines[0][k] for k in GRAPH_KEYS + ("steps", "warmup")}
            ex["graph"] = {k: exact_lines[1][k] for k in GRAPH_KEYS}
            for leg in (ex, ex["graph"]):
                leg["sum_hex"] = leg["config"]["sum_hex"]
                # the reference's own accumulator (papr.c:104) for this stream, recorded in tests/golden/manifest.json
                leg["sum_is_the_reference_s"] = None if want_sum is None else leg["sum_hex"] == want_sum
            ex["what"] = ("the same step with the reference's sequential double sum (papr.c:104) reproduced bit for bit in "
                          "the same single read: bin/papr's default arithmetic")
            line["exact"] = ex
            legs["exact"] = dict(leg_summary(exact_lines[0]), sum_is_the_reference_s=ex["sum_is_the_reference_s"])
            legs["exact_graph"] = dict(leg_summary(exact_lines[1]), sum_is_the_reference_s=ex["graph"]["sum_is_the_reference_s"])
        for tag, tl in (("ts", ts_line), ("ts_damaged", ts_damaged_line)):
            if not tl:
                continue
            line[tag] = tl
            legs[tag] = {"ms_per_step": round(tl["ms_per_step"], 4), "value": round(tl["value"], 1), "unit": tl["unit"],
                         "kernel_ms": round(tl["roofline"]["kernel_ms"], 4), "frac": round(tl["roofline"]["frac"], 4),
                         "step_ms": mmm(tl["roofline"].get("step_ms")), "launches_per_scan": tl["config"]["launches_per_scan"],
                         "lines": tl["config"]["sync_error_lines"] + tl["config"]["discontinuity_lines"],
                         "damage": args.damage if tag == "ts" else 1e-3}
            cbt = tl.get("cpu_baseline")
            if isinstance(cbt, dict) and "value" in cbt:
                legs[tag]["cpu_Mpackets_s"] = round(cbt["value"], 2)
                legs[tag]["gpu_report_identical"] = cbt.get("gpu_report_identical")
        if not args.no_cpu_baseline:
            # the reference program on ONE host core (it is single-threaded), on rank 0 — with peers too: the other ranks
            # wait in the closing barrier (10-15 s)
            mode0 = modes[0]
            sample = args.cpu_sample_gib if args.cpu_sample_gib else (1.0 if mode0 == "graph" else 4.0)   # ~10-15 s of reference CPU time
            sample = min(sample, args.gib)
            try:
                cb = cpu_baseline(pkg, mode0, sample)
            except Exception as e:  # the baseline must never take the GPU number down with it
                cb = {"error": repr(e)}
            line["cpu_baseline"] = cb
            if full:   # the reference's -g loop is O(N x 301): a quarter of the sample keeps it at ~15 s
                try:
                    line["cpu_baseline_graph"] = cpu_baseline(pkg, "graph", min(args.cpu_sample_gib or 0.5, args.gib))
                except Exception as e:
                    line["cpu_baseline_graph"] = {"error": repr(e)}
        if world == 1 and not args.no_e2e and not args.no_cpu_baseline and args.signal == "gauss":
            try:
                line["e2e"] = e2e_block(pkg, args.gib)
            except Exception as e:
                line["e2e"] = {"error": repr(e)}
            for tag in ("default", "graph"):
                if isinstance(line["e2e"].get(tag), dict):
                    t = line["e2e"][tag]
                    legs["e2e_" + tag] = {"seconds": round(t["seconds"], 4), "runs": E2E_RUNS, "open_s": t.get("open_s"),
                                          "ingest_GBps": t.get("ingest_GBps"),
                                          "ingest_frac_of_h2d_ceiling": t.get("ingest_frac_of_h2d_ceiling"),
                                          "all_stdout_identical": t.get("all_stdout_identical")}
        if (world == 1 and not args.no_live_traffic and not args.two_pass and not args.force_miss and not args.exact_two_pass
                and args.mode == "both" and line["roofline"]["kernel"].startswith("papr_sweep")):
            # roofline.traffic measured in THIS run (outside the timed region): the constant from profiles/ stays only when
            # rocprofv3 is not on the box or its pass leaves nothing, and `traffic_source` says which it is
            t0 = time.perf_counter()
            lt = live_traffic(args, line["roofline"]["kernel"], per_gpu * 8)
            if lt:
                for leg, m in ((line, "default"), (line.get("graph"), "graph")):
                    if leg:
                        r = leg["roofline"]
                        r["traffic_profile_constant"] = r.get("traffic")
                        r["traffic"], r["traffic_source"], r["traffic_stale"] = lt[m]["traffic"], lt["source"], None
                        r["traffic_read"], r["traffic_write"] = lt[m]["read"], lt[m]["write"]
                        r["traffic_over_algorithmic"] = lt[m]["traffic"] / r["algorithmic_bytes_per_launch"]
                line["roofline"]["traffic_measured_s"] = round(time.perf_counter() - t0, 1)
                if "graph" in legs:
                    legs["graph"]["traffic"] = round(lt["graph"]["traffic"])
            else:
                line["roofline"]["traffic_source"] = ("NOT measured in this run (no rocprofv3 pass): " +
                                                      (line["roofline"].get("traffic_source") or "no figure"))[:200]
        if legs:
            line["roofline"]["legs"] = legs
        emit(line, args.full_json, real_stdout)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
