"""The N>1 path on CPU: 2, 4 and 8 processes over gloo run the exchange code of the C ABI (papr_exchange_* in
libpaprhip.so, include/papr_hip.h) — the very functions bench.py and a one-process-per-GPU C caller use over RCCL —
with torch.distributed's gloo collectives plugged in as the transport.  Per-shard records come from the oracle here
(no GPU in this tier); what is under test is the sharding rule, the all-gather + ordered merge (and the "sum of the
ranks in front"), the count all-reduce and the all-gather + chain of exact-sum programs."""
import ctypes as C
import os
import socket
import struct
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden_path


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _raw_program(floats: np.ndarray) -> bytes:
    """A valid exact-sum program (papr_exact_format.h) that carries a shard as its raw tail / raw tiles only:
    header + zero groups + the samples, added one by one by papr_exact_chain exactly as papr.c:104 adds them."""
    n = floats.size // 2
    assert n < 2048, "helper for small shards: everything travels as the < 1 tile tail"
    header = struct.pack("<IIQQQIIII", 0x31535850, 3, n, 0, 0, n, 0, 0, 0)
    return header + floats.astype(np.float32).tobytes()


def _worker(rank, world, port, name, graph, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import __graft_entry__ as ge
    pkg, orc = ge.load_package(), ge.load_oracle()
    from dtv_utils_amd import exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = exchange.Exchange.over_torch()
        assert (x.rank, x.world, x.transport) == (rank, world, "gloo")
        x.selftest(None, verbose=True)   # the host-level collectives once, with contents every rank can predict (no GPU here)
        floats = np.fromfile(golden_path(name), dtype=np.float32)
        n = floats.size // 2
        first, count = exchange.shard_range(n, rank, world, align=8192 if n >= 8192 * world else 2)
        shard = floats[2 * first:2 * (first + count)]
        r = orc.run_mem(shard, graph)
        local = pkg.Stats()
        pkg.lib().papr_stats_init(C.byref(local))
        local.sum, local.n = r["sum"], r["n"]
        for k in ("peak", "re_pos", "re_neg", "im_pos", "im_neg"):
            setattr(local, k, r[k])
            setattr(local, k + "_idx", r[k + "_idx"] + first if r[k] != 0 else 0)
        total, before, every = x.stats(local)
        assert len(every) == world and every[rank].to_bytes() == local.to_bytes()
        assert before == float(sum(e.sum for e in every[:rank]))          # (left-to-right, as the merge adds)
        mean, papr, table = pkg.levels(total, graph)
        counts = x.counts(orc.count_mem(shard, table).astype(np.uint64))
        text = pkg.format_report(total, mean, papr, counts, graph)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(text)
        # exact-sum programs of different sizes, gathered and chained in rank order: the sequential sum of the
        # concatenation (here: the first 1500 samples of the fixture cut into `world` pieces)
        m = min(n, 1500)
        a, b = exchange.shard_range(m, rank, world, align=2)
        seq = x.exact_sum(_raw_program(floats[2 * a:2 * (a + b)]))
        assert seq == orc.run_mem(floats[:2 * m], False)["sum"], "chained sum differs from the sequential sum"
        t = x.timing().as_dict()
        assert t["stats"]["calls"] == 1 and t["counts"]["calls"] == 1 and t["exact"]["calls"] == 1
        x.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name,graph", [(2, "g1m", False), (2, "g1m", True), (2, "ties", False), (2, "spike20k", True),
                                              (4, "g1m", True), (8, "g1m", False), (3, "ofdm_dvbt", True)])
def test_exchange_over_gloo_reproduces_reference_stdout(tmp_path, world, name, graph):
    mp.spawn(_worker, args=(world, _free_port(), name, graph, str(tmp_path)), nprocs=world, join=True)
    want = open(os.path.join(ROOT, "tests", "golden", f"{name}.{'graph' if graph else 'default'}.txt")).read()
    for rank in range(world):
        assert open(tmp_path / f"rank{rank}.txt").read() == want, f"rank {rank}"


def test_single_rank_exchange_is_the_identity(pkg):
    from dtv_utils_amd import exchange
    x = exchange.Exchange.single()
    local = pkg.Stats()
    pkg.lib().papr_stats_init(C.byref(local))
    local.sum, local.n, local.peak, local.peak_idx = 12.5, 10, 3.0, 7
    total, before, every = x.stats(local)
    assert total.to_bytes() == local.to_bytes() and before == 0.0 and len(every) == 1
    c = np.array([5, 4, 3], dtype=np.uint64)
    assert np.array_equal(x.counts(c), c)
    assert x.exact_sum(_raw_program(np.array([1, 0, 0, 2, 3, 4], dtype=np.float32))) == 1.0 + 4.0 + 25.0
    with pytest.raises(pkg.PaprError):
        x.exact_sum(b"not a program")
    x.close()


def test_library_exports_the_exchange_abi(pkg):
    from dtv_utils_amd import exchange
    L = exchange._lib()
    for name in exchange.ABI_SYMBOLS:
        assert hasattr(L, name), name


def test_shard_range_covers_axis_without_overlap(pkg):
    from dtv_utils_amd import exchange
    for n in (0, 1, 8191, 8192, 8193, 131072, 1342177280, 34359738368):
        for world in (1, 2, 3, 4, 8):
            pos = 0
            for r in range(world):
                first, count = exchange.shard_range(n, r, world)
                assert first == pos and (first % 8192 == 0 or count == 0)
                pos += count
            assert pos == n


def _hub_threads(n, body):
    """n threads of this process, one handle of papr_exchange_open_local each; returns what body(x) returned or raised."""
    import threading
    from dtv_utils_amd import exchange
    xs = exchange.Exchange.local(n)
    out = [None] * n

    def run(r):
        try:
            out[r] = body(xs[r])
        except Exception as e:   # noqa: BLE001 - handed to the test
            out[r] = e
    ts = [threading.Thread(target=run, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
        assert not t.is_alive(), "a thread is still waiting at the hub"
    for x in xs:
        x.close()
    return out


@pytest.mark.parametrize("world", [2, 4, 8])
def test_selftest_over_the_thread_hub(pkg, world):
    """papr_exchange_selftest on the in-process transport (what bin/papr's shard threads meet at): the host-level
    all-gather and all-reduce with contents every rank can predict for every other rank."""
    out = _hub_threads(world, lambda x: x.selftest(None, verbose=False) and "ok")
    assert out == ["ok"] * world, out


def test_a_rank_that_never_arrives_is_timed_out_not_waited_for(pkg, monkeypatch):
    """PAPR_XCH_TIMEOUT_S: three of four threads enter a collective, the fourth never does.  Without the timeout they
    would wait for ever; with it they give up after the limit, with PAPR_E_STATE and a message that says why — and the
    late thread finds the exchange cancelled instead of waiting in turn."""
    import time
    monkeypatch.setenv("PAPR_XCH_TIMEOUT_S", "0.5")

    def body(x):
        if x.rank == 3:
            time.sleep(2.0)
        t0 = time.perf_counter()
        try:
            x.counts(np.arange(4, dtype=np.uint64))
        except pkg.PaprError as e:
            return (round(time.perf_counter() - t0, 1), str(e))
        return "no error"
    out = _hub_threads(4, body)
    for r in range(3):
        assert isinstance(out[r], tuple) and 0.4 <= out[r][0] <= 1.5 and "cancelled" in out[r][1], out
    assert isinstance(out[3], tuple) and out[3][0] <= 0.2 and "cancelled" in out[3][1], out


def test_abort_releases_the_threads_at_the_hub(pkg):
    """papr_exchange_abort from one thread: the peers' pending collective returns PAPR_E_STATE (the members are
    walked under the hub's mutex, a handle being closed meanwhile is not touched)."""
    import time

    def body(x):
        if x.rank == 0:
            time.sleep(0.3)
            x.abort()
            return "aborted"
        try:
            x.counts(np.arange(4, dtype=np.uint64))
        except pkg.PaprError as e:
            return str(e)
        return "no error"
    out = _hub_threads(3, body)
    assert out[0] == "aborted" and all("cancelled" in o for o in out[1:]), out
