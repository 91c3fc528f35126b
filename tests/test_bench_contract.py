"""bench.py's driver contract: ONE JSON line on stdout with the agreed keys, also under
torch.distributed.run (RCCL initialised, both exchanges executed)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import FULL_TIER, ROOT

pytestmark = pytest.mark.gpu

LINE_LIMIT = 6000   # bytes: what the driver's record is known to keep whole (round 4's 22 KB line was not)

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "parity_in_run"}


def run_bench(cmd, tmp_path, env=None):
    """Run bench.py; returns (the compact stdout line parsed, the full record it wrote beside it).  The checks of
    the line itself are the driver's: ONE line, short enough to be kept, parsable out of the last 8000 characters of
    stdout + stderr, the headline repeated in its last 2000 characters."""
    full_path = os.path.join(str(tmp_path), "bench_full.json")
    p = subprocess.run(cmd + ["--full-json", full_path], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    line = lines[0]
    assert len(line.encode()) <= LINE_LIMIT, len(line)
    d = json.loads(line)
    assert json.loads(p.stdout[-8000:].splitlines()[-1]) == d
    tail = line[-2000:]
    assert '"headline":' in tail
    h = json.loads("{" + tail[tail.index('"headline":'):])["headline"]
    assert h["value"] == d["value"] and h["ms_per_step"] == d["ms_per_step"] and h["roofline_frac"] == d["roofline"]["frac"]
    assert list(d)[-1] == "headline"
    with open(full_path) as f:
        full = json.load(f)
    assert abs(full["value"] - d["value"]) < 0.06 and full["steps"] == d["steps"]
    return d, full


def check(d, steps, warmup):
    assert KEYS <= set(d)
    assert d["steps"] == steps and d["warmup"] == warmup and d["n_gpus"] == 1
    assert d["unit"] == "Msamples/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert d["value"] > 0 and d["ms_per_step"] > 0
    return d


def test_single_process_line_with_cpu_baseline(tmp_path):
    d, full = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gib", "0.5", "--steps", "4", "--warmup", "1",
                         "--cpu-sample-gib", "0.125"], tmp_path)
    check(d, 4, 1)
    for cb in (d["cpu_baseline"], d["cpu_baseline_graph"]):
        assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0 and cb["nproc"] >= 1
        assert cb["gpu_stdout_identical"] is True and "GiB" in cb["sample"]
    assert d["headline"]["cpu_baseline_value"] == d["cpu_baseline"]["value"]
    # no --mode: configs[1] is the headline, configs[2] (papr -g) rides along: whole in the full record, one line in `legs`
    assert d["config"]["mode"] == "default" and full["graph"]["config"]["mode"] == "graph"
    assert full["graph"]["config"]["levels"] > 5 * d["config"]["levels"] and full["graph"]["value"] > 0
    assert full["graph"]["roofline"]["kernel"] == d["roofline"]["kernel"]
    legs = d["roofline"]["legs"]
    assert {"graph", "exact", "exact_graph", "ts", "ts_damaged", "e2e_default", "e2e_graph"} <= set(legs), sorted(legs)
    for name in ("graph", "exact", "exact_graph", "ts", "ts_damaged"):
        assert legs[name]["frac"] == d["roofline"][name + "_frac"] > 0 and legs[name]["ms_per_step"] > 0
    assert legs["exact"]["kernel"] == "papr_sweep3_kernel" and len(legs["exact"]["step_ms"]) == 3
    # the pre-heat is disclosed, and what the first untimed steps on the idle GPU cost
    assert d["config"]["preheat_steps"] == 32 and len(d["config"]["cold_first_steps_ms"]) == 5
    assert "all" not in d["roofline"]["step_ms"] and len(full["roofline"]["step_ms"]["all"]) == 4
    # 0.5 GiB has no recorded reference stdout: parity_in_run must say so rather than claim anything
    assert d["parity_in_run"] is None and full["graph"]["parity_in_run"] is None


def test_the_drivers_command_prints_a_line_the_driver_can_keep(tmp_path):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` verbatim, at full size, every leg: round 4's line was 22 KB and the
    driver's record of it `parsed: null`.  run_bench() asserts the size, that the last 8000 characters of stdout parse, and
    that the headline sits in the last 2000."""
    d, full = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], tmp_path)
    check(d, 20, 5)
    assert d["parity_in_run"] is True and d["cpu_baseline"]["kind"] in ("reference", "port")
    legs = d["roofline"]["legs"]
    for name in ("graph", "exact", "exact_graph"):
        assert legs[name]["parity_in_run"] is True, name
    assert legs["exact"]["sum_is_the_reference_s"] is True and legs["exact_graph"]["sum_is_the_reference_s"] is True
    assert legs["ts"]["gpu_report_identical"] is True
    assert legs["e2e_default"]["all_stdout_identical"] is True and legs["e2e_graph"]["all_stdout_identical"] is True
    # --preheat 0: the warm-up steps alone, and the line says so
    d0, _ = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "5", "--preheat", "0",
                       "--headline-only", "--no-cpu-baseline"], tmp_path)
    assert d0["config"]["preheat_steps"] == 0 and len(d0["config"]["cold_first_steps_ms"]) == 5


def test_full_size_line_checks_itself_against_the_reference_stdout(tmp_path):
    """The driver's invocation (10 GiB, both tables): the line's own parity flags are computed from the recorded
    stdout of the reference program for the same stream and must be true in both modes."""
    d, full = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline",
                         "--no-ts"], tmp_path)
    check(d, 5, 1)
    assert d["parity_in_run"] is True and d["parity_golden"] == "big_spike10g.default.txt"
    assert full["graph"]["parity_in_run"] is True and full["graph"]["parity_golden"] == "big_spike10g.graph.txt"
    assert d["roofline"]["legs"]["graph"]["parity_in_run"] is True
    assert d["config"]["one_sweep"]["resolved"] == 5 and full["config"]["one_sweep"]["steps_resolved_from_the_sweep"] == 5


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("extra", [[], ["--mode", "graph"], ["--two-pass"], ["--mode", "graph", "--exact"]], ids=str)
def test_torchrun_single_rank_uses_rccl(extra, tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--gib", "0.5",
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", *extra]
    compact, d = run_bench(cmd, tmp_path)
    check(compact, 3, 1)
    assert compact["config"]["exchange"].startswith("RCCL")
    assert d["config"]["exact_sequential_sum"] == ("--exact" in extra)
    if "--two-pass" in extra:
        assert d["config"]["one_sweep"] is None and d["config"]["reads_of_the_shard_per_step"] == 2
        assert not d["roofline"]["kernel"].startswith("papr_sweep")
    else:   # the default AND --exact: one read of the shard per step, every step answered from the sweep
        if "--exact" in extra:   # ... the sequential sum too: only the tiles whose speculated binade was wrong are redone
            assert 0 <= d["config"]["one_sweep"]["exact_redo_tiles_per_step"] <= d["config"]["samples_per_gpu"] / 2048 / 20
        assert d["config"]["one_sweep"]["steps_resolved_from_the_sweep"] == 3
        assert d["config"]["reads_of_the_shard_per_step"] == 1
        assert d["roofline"]["kernel"] == ("papr_sweep3_kernel" if "--exact" in extra else "papr_sweep_kernel")
        assert 0 < d["config"]["one_sweep"]["stash_samples"] < d["config"]["samples_per_gpu"] // 8
        # every exchange of the step was a collective in the stream (exact-sum steps: the programs too)
        per_step = 4 if "--exact" in extra else 3
        assert d["exchange"]["in_stream_collectives"] >= per_step * 3, d["exchange"]
        assert d["exchange"]["stats"]["calls"] == 0 and d["exchange"]["exact"]["calls"] == 0, d["exchange"]


def _run_bench(nproc, gib, extra, env=None):
    """(the FULL record of the run; the compact line is checked on the way by run_bench)"""
    import tempfile
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")]
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
               "--backend", "gloo"]
    cmd += ["--gpus", str(nproc), "--gib", str(gib), "--steps", "2", "--warmup", "1", "--preheat", "2", "--no-cpu-baseline", *extra]
    with tempfile.TemporaryDirectory() as tmp:
        return run_bench(cmd, tmp, env)[1]


@pytest.mark.parametrize("mode", ["default", "graph"])
def test_sharded_run_equals_single_shard_run(mode):
    """The N>1 code path of bench.py (rank-dependent shard of ONE global stream, stats all-gather + ordered
    merge, chained exact-sum programs, count all-reduce) on a box with one GPU: 2 and 4 ranks share it over
    gloo, and must reproduce the 1-rank result over the same stream bit for bit (2, 4 and 8 ranks)."""
    extra = ["--mode", mode, "--exact"]
    one = _run_bench(1, 0.5, extra)
    for ranks in ((2, 4, 8) if FULL_TIER else (2, 8)):   # 8 = the rank count of BASELINE.json configs[3]
        many = _run_bench(ranks, 0.5 / ranks, extra)
        assert many["n_gpus"] == ranks and many["config"]["samples_total"] == one["config"]["samples_total"]
        for key in ("sum_hex", "papr_db", "levels", "counts_crc32"):
            assert many["config"][key] == one["config"][key], (ranks, key)
        assert many["config"]["exchange"].startswith("gloo")


@pytest.mark.parametrize("mode", ["default", "graph"])
def test_one_sweep_equals_two_pass_at_every_rank_count(mode):
    """bench.py's default one-read step (estimate exchange + sweep + stash recount) against --two-pass on the
    same stream, 1, 2 and 4 ranks: same table, same counts."""
    ref = _run_bench(1, 0.5, ["--mode", mode, "--two-pass"])
    assert ref["config"]["one_sweep"] is None
    for ranks in (1, 2, 4):
        got = _run_bench(ranks, 0.5 / ranks, ["--mode", mode])
        assert got["config"]["one_sweep"]["steps_resolved_from_the_sweep"] == 2, got["config"]["one_sweep"]
        for key in ("papr_db", "levels", "counts_crc32", "samples_total"):
            assert got["config"][key] == ref["config"][key], (ranks, key)
        assert abs(float.fromhex(got["config"]["sum_hex"]) - float.fromhex(ref["config"]["sum_hex"])) <= \
            1e-12 * float.fromhex(ref["config"]["sum_hex"])


@pytest.mark.parametrize("slot_kb", [None, "64"], ids=["programs cross in the stream", "a slot too small: host exchange"])
def test_exact_sum_sharded_step_as_one_sequence_of_launches_with_real_worlds(slot_kb):
    """The same for exact-sum steps (bin/papr's default arithmetic): the sum programs' classification starts from the
    merged records' sum in front of the shard, read from DEVICE memory, every rank's program crosses the exchange in the
    stream in a fixed-size slot and all of them are replayed in file order while the GPU does the recount.  The chained
    sum must be the 1-rank run's bit for bit at 2, 4 and 8 ranks, both tables, with no host exchange of programs — and, with
    a slot made too small on purpose (PAPR_XPROG_SLOT_KB), the same result through the host exchange every rank falls
    back to together."""
    env = {"PAPR_XCH_IN_STREAM": "2"}
    if slot_kb:
        env["PAPR_XPROG_SLOT_KB"] = slot_kb
    for mode in ("default", "graph"):
        one = _run_bench(1, 0.5, ["--mode", mode, "--exact"])
        for ranks in ((2, 4, 8) if FULL_TIER else (2, 8)):
            got = _run_bench(ranks, 0.5 / ranks, ["--mode", mode, "--exact"], env=env)
            assert got["config"]["exact_sequential_sum"] is True
            assert got["config"]["one_sweep"]["steps_resolved_from_the_sweep"] == 2, got["config"]["one_sweep"]
            for key in ("sum_hex", "papr_db", "levels", "counts_crc32", "samples_total"):
                assert got["config"][key] == one["config"][key], (ranks, key)
            assert got["exchange"]["stats"]["calls"] == 0, got["exchange"]
            if slot_kb:
                assert got["exchange"]["exact"]["calls"] == 2, got["exchange"]
            else:   # four collectives per step (estimate records, pass-1 records, programs, counters), no host exchange
                assert got["exchange"]["in_stream_collectives"] >= 4 * 2 and got["exchange"]["exact"]["calls"] == 0, got["exchange"]


@pytest.mark.parametrize("mode", ["default", "graph"])
def test_sharded_step_as_one_sequence_of_launches_with_real_worlds(mode):
    """What an N-GPU run over RCCL executes — estimate records, pass-1 records and counters crossing the exchange as
    collectives on DEVICE buffers between the kernels that produce and consume them (papr_est_record_kernel, the guess
    from all ranks' records, papr_record_merge_kernel, papr_xpack_kernel) — with worlds of 2, 4 and 8 ranks on the one
    GPU: PAPR_XCH_IN_STREAM=2 lets the gloo callbacks stand in for ncclAllGather / ncclAllReduce.  Table and counts must
    be the 1-rank two-pass run's, the sum the tree sum's to 1e-12, and every step must have taken that path."""
    ref = _run_bench(1, 0.5, ["--mode", mode, "--two-pass"])
    for ranks in ((2, 4, 8) if FULL_TIER else (2, 8)):
        got = _run_bench(ranks, 0.5 / ranks, ["--mode", mode], env={"PAPR_XCH_IN_STREAM": "2"})
        assert got["config"]["one_sweep"]["steps_resolved_from_the_sweep"] == 2, got["config"]["one_sweep"]
        for key in ("papr_db", "levels", "counts_crc32", "samples_total"):
            assert got["config"][key] == ref["config"][key], (ranks, key)
        assert abs(float.fromhex(got["config"]["sum_hex"]) - float.fromhex(ref["config"]["sum_hex"])) <= \
            1e-12 * float.fromhex(ref["config"]["sum_hex"])
        # three collectives per step, none of the host exchanges
        assert got["exchange"]["in_stream_collectives"] >= 3 * 2, got["exchange"]
        assert got["exchange"]["stats"]["calls"] == 0 and got["exchange"]["counts"]["calls"] == 0, got["exchange"]


@pytest.mark.parametrize("extra", [[], ["--exact"]], ids=["tree", "exact"])
def test_two_rank_bench_line_proves_itself_at_full_size(extra, tmp_path):
    """What a driver's 2-GPU run prints, on one GPU (two ranks over gloo share it): 2 x 10 GiB shards of the 20 GiB
    stream, `parity_in_run` = the run's report equals the REFERENCE's recorded stdout for that stream, both tables;
    with --exact also the reference's sequential sum over the two chained shards."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--steps", "2", "--warmup", "1", "--cpu-sample-gib", "0.25", *extra]
    compact, d = run_bench(cmd, tmp_path)
    # more than one rank: headline + configs[2] only, and rank 0 still times the reference on one host core
    assert set(compact["roofline"]["legs"]) == {"graph"} and compact["cpu_baseline"]["value"] > 0
    assert compact["cpu_baseline"]["gpu_stdout_identical"] is True
    assert d["n_gpus"] == 2 and d["config"]["samples_total"] == 2 * 1342177280
    assert d["parity_in_run"] is True and d["parity_golden"] == "big_spike20g.default.txt"
    assert d["graph"]["parity_in_run"] is True and d["graph"]["parity_golden"] == "big_spike20g.graph.txt"
    if extra:
        assert d["config"]["sum_hex"] == "0x1.aaaa011478022p+31" and d["config"]["exact_sequential_sum"]


def test_gpus_n_without_a_launcher_runs_n_ranks_by_itself(tmp_path):
    """`python3 bench.py --gpus 2 --backend gloo --steps 3 --warmup 1` verbatim — no torchrun around it: bench.py is its own
    launcher (one rank per GPU; here two ranks share the one GPU over gloo), rank 0 prints the ONE line, and that line
    carries both readings of BASELINE's "10 GiB cfile, 1/2/4/8 GPU": `weak` (every rank its own 10 GiB: the headline) and
    `strong` (the 10 GiB stream cut into two shards), each checked against the reference's recorded stdout."""
    d, full = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
                         "--warmup", "1"], tmp_path)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["samples_total"] == 2 * 1342177280 and d["parity_in_run"] is True
    assert d["parity_golden"] == "big_spike20g.default.txt" and d["cpu_baseline"]["value"] > 0
    w, s = d["weak"], d["strong"]
    assert w["scaling"] == "weak" and w["n_gpus"] == 2 and w["gib_per_gpu"] == 10 and w["gib_total"] == 20
    assert w["ms_per_step"] == round(d["ms_per_step"], 4) and w["frac"] == round(d["roofline"]["frac"], 4)
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["gib_per_gpu"] == 5 and s["gib_total"] == 10
    for m in (w, s):
        assert m["speedup_vs_n1"] is None and m["ms_per_step"] > 0 and 0 < m["frac"] < 1 and m["parity_in_run"] is True
        assert m["graph"]["parity_in_run"] is True and m["graph"]["ms_per_step"] > 0
    assert s["parity_golden"] == "big_spike10g.default.txt"
    # (two ranks on one GPU take turns: the strong member's step covers the same 10 GiB a 1-GPU step does)
    assert full["strong_full"][0]["config"]["samples_total"] == 1342177280 and full["strong_full"][0]["scaling"] == "strong"


def test_gpus_n_over_rccl_with_fewer_gpus_is_refused_at_once():
    """The same command over RCCL (the default backend) needs one GPU per rank: with fewer visible it says so in one
    stderr line and exits non-zero before a single rank is started — no hang, nothing on stdout."""
    import time
    import torch
    n = torch.cuda.device_count() + 1
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, cwd=ROOT, timeout=120)
    dt = time.perf_counter() - t0
    assert p.returncode not in (0, None) and p.stdout.strip() == ""
    lines = [l for l in p.stderr.splitlines() if l.startswith("bench.py:")]
    assert len(lines) == 1 and f"--gpus {n}" in lines[0] and "visible" in lines[0], p.stderr[-500:]
    assert dt < 10.0, dt
    # ... and under a launcher that started the ranks anyway, the rank without a GPU of its own says so
    env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(n - 1), LOCAL_RANK=str(n - 1), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=120)
    assert p.returncode != 0 and "GPU(s) are visible" in p.stderr and p.stdout.strip() == ""
