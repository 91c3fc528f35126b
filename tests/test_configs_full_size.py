"""BASELINE.json configs[3] and configs[4] at their own workloads, on the one-GPU box.

configs[3] — `papr -g` on 80 GiB sharded over 8 ranks: eight ranks share the one 288 GB GPU (10 GiB shard each, gloo
carrying the C exchange), and the line must prove itself against the reference's recorded stdout for the 80 GiB
stream (tests/golden/big_spike80g.*.txt: 47 minutes of reference CPU time), with --exact also the reference's
sequential sum over the eight chained shards.

configs[4] — a file larger than the HBM budget, streamed through double-buffered staging: a 40 GiB file (more than one
GPU's 32 GiB share of the 256 GiB config; the reference streams any size, papr.c:100-101, 142-144) goes through
`bin/papr [-g]` under a 4 GiB budget, as one shard and as eight, and stdout must be the reference's recording
(big_spike40g.*.txt) after ONE pass over the file.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _golden(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


@pytest.mark.parametrize("extra", [[], ["--exact"], ["--collectives-on-device-buffers"], ["--exact", "--collectives-on-device-buffers"]],
                         ids=["tree", "exact", "tree, exchanges between the kernels", "exact, exchanges between the kernels"])
def test_eight_rank_bench_line_proves_itself_at_full_size(extra, manifest, tmp_path):
    """BASELINE configs[3] on the one GPU: 8 ranks x 10 GiB over gloo.  The third form takes the way an 8-GPU run over RCCL
    takes — the three exchanges as collectives on device buffers inside the step's sequence of launches — with the gloo
    callbacks standing in for ncclAllGather / ncclAllReduce (PAPR_XCH_IN_STREAM=2)."""
    import torch
    env = dict(os.environ)
    if "--collectives-on-device-buffers" in extra:
        extra = [e for e in extra if e != "--collectives-on-device-buffers"]
        env["PAPR_XCH_IN_STREAM"] = "2"
    free, total = torch.cuda.mem_get_info(0)
    if free < 110 * (1 << 30):
        pytest.skip(f"8 x 10 GiB shards + stashes need ~100 GiB of HBM; {free >> 30} GiB free")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo",
           "--steps", "2", "--warmup", "1", "--preheat", "4", "--no-cpu-baseline", "--no-e2e",
           "--full-json", str(tmp_path / "full.json"), *extra]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    # what an 8-GPU driver run would keep: ONE short line whose headline is the 80 GiB job's ...
    line = [l for l in p.stdout.splitlines() if l.strip()][-1]
    compact = json.loads(line)
    assert len(line) <= 6000 and compact["n_gpus"] == 8 and compact["parity_in_run"] is True
    assert compact["roofline"]["legs"]["graph"]["parity_in_run"] is True and set(compact["roofline"]["legs"]) == {"graph"}
    # ... and the full record beside it
    d = json.load(open(tmp_path / "full.json"))
    assert d["n_gpus"] == 8 and d["config"]["samples_total"] == 8 * 1342177280 == manifest["big_spike80g"]["nsamples"]
    if env.get("PAPR_XCH_IN_STREAM") == "2":
        assert d["exchange"]["in_stream_collectives"] >= 6 and d["exchange"]["stats"]["calls"] == 0, d["exchange"]
        assert d["exchange"]["exact"]["calls"] == 0, d["exchange"]   # (exact-sum steps: the programs crossed in the stream too)
    assert d["parity_in_run"] is True and d["parity_golden"] == "big_spike80g.default.txt"
    assert d["graph"]["parity_in_run"] is True and d["graph"]["parity_golden"] == "big_spike80g.graph.txt"
    assert d["report_sha256"] == manifest["big_spike80g"]["default"]["sha256"]
    assert d["graph"]["report_sha256"] == manifest["big_spike80g"]["graph"]["sha256"]
    assert d["graph"]["config"]["levels"] == manifest["big_spike80g"]["graph"]["lines"] == 302
    assert d["config"]["one_sweep"]["steps_resolved_from_the_sweep"] == 2     # one read of every shard per step
    if extra:
        assert d["config"]["exact_sequential_sum"] is True
        assert d["config"]["sum_hex"] == manifest["big_spike80g"]["oracle_sequential_sum_hex"]
        assert d["graph"]["config"]["sum_hex"] == manifest["big_spike80g"]["oracle_sequential_sum_hex"]


def _big_file_dir(need):
    """Where a `need`-byte file can live on this box: a real disk first (the box's own; what configs[4] means), tmpfs
    only if the disk is too small."""
    cands = [os.environ.get("PAPR_TEST_DISK_DIR"), os.path.join(ROOT, "gpurun_out"), "/var/tmp", "/tmp", "/dev/shm"]
    for cand in cands:
        if not cand:
            continue
        try:
            os.makedirs(cand, exist_ok=True)
            if shutil.disk_usage(cand).free > need + (4 << 30):
                if cand == "/dev/shm":   # tmpfs is memory: leave room for the page cache of nothing else, and the staging
                    with open("/proc/meminfo") as f:
                        avail = next(int(l.split()[1]) for l in f if l.startswith("MemAvailable")) * 1024
                    if avail < need + (24 << 30):
                        continue
                return cand
        except OSError:
            continue
    return None


def _make_big_file(orc, path, n, tmp_dir):
    """The spike stream of n samples as a file, from the DEVICE generator (tools/mkcfile_gpu.py: a 256 GiB file took
    oracle/mkcfile's 64 processes 315 s), and — the recorded stdouts were made from oracle/mkcfile's files — three 1 MiB
    windows of it against oracle/mkcfile's own bytes for the same ranges."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mkcfile_gpu
    seconds = mkcfile_gpu.make(path, n)
    assert os.path.getsize(path) == n * 8
    win = 131072
    probe = os.path.join(tmp_dir, f"papr_probe_{os.getpid()}.cfile")
    try:
        for first in (0, (n // 2) // 8192 * 8192 + 8192 * 3, n - win):
            if os.path.exists(probe):
                os.unlink(probe)
            subprocess.check_call([orc.MKCFILE, probe, str(n), "--spike", "--part", str(first), str(win)])   # (a sparse file)
            with open(probe, "rb") as f, open(path, "rb") as g:
                f.seek(first * 8)
                g.seek(first * 8)
                assert f.read(win * 8) == g.read(win * 8), f"the device generator's bytes differ from oracle/mkcfile's at sample {first}"
    finally:
        if os.path.exists(probe):
            os.unlink(probe)
    return seconds


@pytest.fixture(scope="module")
def file40g(orc):
    n = 5368709120   # 40 GiB
    d = _big_file_dir(n * 8)
    if d is None:
        pytest.skip("no filesystem on this box has 44 GiB free")
    path = os.path.join(d, f"papr_big_spike40g_{os.getpid()}.cfile")
    try:
        _make_big_file(orc, path, n, "/tmp")
        yield path
    finally:
        if os.path.exists(path):
            os.unlink(path)


@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
@pytest.mark.parametrize("gpus", [1, 8], ids=["one-shard", "eight-shards"])
def test_40_gib_file_streams_once_under_a_4_gib_budget(pkg, file40g, manifest, gpus, graph):
    env = dict(os.environ, PAPR_STATS="1", PAPR_HBM_BUDGET_MB="4096", PAPR_GPUS=str(gpus))
    if gpus > 1:
        env["PAPR_OVERSUBSCRIBE"] = "1"
    p = subprocess.run([pkg.CLI_PATH] + (["-g"] if graph else []) + [file40g], capture_output=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    tag = "graph" if graph else "default"
    assert p.stdout == _golden(f"big_spike40g.{tag}.txt")
    assert len(p.stdout.splitlines()) == manifest["big_spike40g"][tag]["lines"]
    info = json.loads(p.stderr.decode().splitlines()[-1])
    assert info["samples"] == manifest["big_spike40g"]["nsamples"] and info["gpus"] == gpus
    assert info["exact_sum"] == 1                       # the CLI's default arithmetic: the reference's sequential sum
    assert info["gpu0_ingest"]["resident"] == 0         # 40 GiB (5 GiB per shard at eight) against a 4 GiB budget
    assert info["gpu0_ingest"]["file_passes"] == 1      # ... and still ONE pass over the file
    assert info["shards_swept"] == gpus and info["shards_resolved_from_sweep"] == gpus


def test_40_gib_through_a_fifo_is_the_files_pass_1_with_the_references_sum(pkg, file40g, manifest):
    """An input that cannot be rewound, at full size: the 40 GiB file through a FIFO into `bin/papr` (`cat file > fifo`).  The
    stream crosses one 256 MiB window of HBM 160 times; what is printed must be the reference's recording for the FILE with
    every percentage zero (papr.c:142-143: pass 2 finds a stream at its end) — the `average power` line included, i.e. the
    reference's sequential double sum over 5.4e9 samples, carried exactly from window to window."""
    fifo = os.path.join("/tmp", f"papr_fifo_{os.getpid()}")
    if os.path.exists(fifo):
        os.unlink(fifo)
    os.mkfifo(fifo)
    try:
        feeder = subprocess.Popen(f"exec cat '{file40g}' > '{fifo}'", shell=True)
        p = subprocess.run([pkg.CLI_PATH, fifo], capture_output=True, env=dict(os.environ, PAPR_STATS="1"), timeout=900)
        feeder.wait(timeout=60)
    finally:
        os.unlink(fifo)
    assert p.returncode == 0, p.stderr[-2000:]
    want = _golden("big_spike40g.default.txt").splitlines()
    got = p.stdout.splitlines()
    assert len(got) == len(want) == manifest["big_spike40g"]["default"]["lines"]
    for g, w in zip(got, want):
        if w.startswith(b"percentage above"):
            assert g == w.split(b"=")[0] + b"= 0.00000000", (g, w)
        else:
            assert g == w, (g, w)
    info = json.loads(p.stderr.decode().splitlines()[-1])
    assert info["samples"] == manifest["big_spike40g"]["nsamples"] and info["exact_sum"] == 1
    assert info["stream_windows"] == 160 and info["gpu0_ingest"]["resident"] == 0


# ---- configs[4] at its own size: 256 GiB, whenever the box can hold the file in /dev/shm ------------------------------------

@pytest.fixture(scope="module")
def file256g(orc):
    n = 34359738368   # 256 GiB = eight shards of 2^32 samples
    need = n * 8
    if os.environ.get("PAPR_TEST_256G", "1") == "0":
        pytest.skip("PAPR_TEST_256G=0")
    try:
        with open("/proc/meminfo") as f:
            avail = next(int(l.split()[1]) for l in f if l.startswith("MemAvailable")) * 1024
        free = shutil.disk_usage("/dev/shm").free
    except OSError:
        pytest.skip("no /dev/shm")
    if free < need + (16 << 30) or avail < need + (64 << 30):
        pytest.skip(f"a 256 GiB file in /dev/shm needs {(need >> 30) + 64} GiB of available memory: this box has "
                    f"{avail >> 30} GiB (and {free >> 30} GiB free in /dev/shm)")
    path = f"/dev/shm/papr_big_spike256g_{os.getpid()}.cfile"
    try:
        _make_big_file(orc, path, n, "/tmp")
        yield path
    finally:
        if os.path.exists(path):
            os.unlink(path)


@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
@pytest.mark.parametrize("gpus", [8, 1], ids=["eight-shards", "one-shard"])
def test_256_gib_file_streams_once_under_a_4_gib_budget(pkg, file256g, manifest, gpus, graph):
    """BASELINE configs[4] as it stands: 256 GiB, every shard 2^32 samples (beyond the budget and beyond 32-bit indices at
    once), streamed through the sweep once; stdout = what the reference program printed for this very file on the GPU box's
    host (tests/golden/big_spike256g.*.txt, recorded by tools/config4_full_size.sh)."""
    env = dict(os.environ, PAPR_STATS="1", PAPR_HBM_BUDGET_MB="4096", PAPR_GPUS=str(gpus), PAPR_OVERSUBSCRIBE="1")
    p = subprocess.run([pkg.CLI_PATH] + (["-g"] if graph else []) + [file256g], capture_output=True, env=env, timeout=1800)
    assert p.returncode == 0, p.stderr[-2000:]
    tag = "graph" if graph else "default"
    assert p.stdout == _golden(f"big_spike256g.{tag}.txt")
    assert hashlib.sha256(p.stdout).hexdigest() == manifest["big_spike256g"][tag]["sha256"]
    info = json.loads(p.stderr.decode().splitlines()[-1])
    assert info["samples"] == manifest["big_spike256g"]["nsamples"] == 34359738368 and info["gpus"] == gpus
    assert info["exact_sum"] == 1 and info["gpu0_ingest"]["resident"] == 0 and info["gpu0_ingest"]["file_passes"] == 1
    assert info["shards_swept"] == gpus and info["shards_resolved_from_sweep"] == gpus
