"""The N>1 path on CPU: two processes over gloo run the same exchange code
bench.py uses over RCCL (dtv-utils_amd/exchange.py).  Per-shard records come
from the oracle here (no GPU in this tier); what is under test is the
sharding rule, the all-gather + ordered merge and the count all-reduce."""
import ctypes as C
import os
import socket
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


def _worker(rank, world, port, name, graph, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import __graft_entry__ as ge
    pkg, orc = ge.load_package(), ge.load_oracle()
    from dtv_utils_amd import exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        floats = np.fromfile(golden_path(name), dtype=np.float32)
        n = floats.size // 2
        first, count = exchange.shard_range(n, rank, world, align=8192)
        shard = floats[2 * first:2 * (first + count)]
        r = orc.run_mem(shard, graph)
        local = pkg.Stats()
        pkg.lib().papr_stats_init(C.byref(local))
        local.sum, local.n = r["sum"], r["n"]
        for k in ("peak", "re_pos", "re_neg", "im_pos", "im_neg"):
            setattr(local, k, r[k])
            setattr(local, k + "_idx", r[k + "_idx"] + first if r[k] != 0 else 0)
        dev = torch.device("cpu")
        total = exchange.merged_stats(local, dev)
        mean, papr, table = pkg.levels(total, graph)
        counts = exchange.allreduce_counts(orc.count_mem(shard, table).astype(np.uint64), dev)
        blobs = exchange.Exchange(dev).allgather_bytes(bytes([rank + 1]) * (1000 * rank + 3))
        assert blobs == [bytes([r + 1]) * (1000 * r + 3) for r in range(world)]
        text = pkg.format_report(total, mean, papr, counts, graph)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(text)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,graph", [("g1m", False), ("g1m", True), ("ties", False), ("spike20k", True)])
def test_two_rank_exchange_reproduces_reference_stdout(tmp_path, name, graph):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, graph, str(tmp_path)), nprocs=world, join=True)
    want = open(os.path.join(ROOT, "tests", "golden", f"{name}.{'graph' if graph else 'default'}.txt")).read()
    for rank in range(world):
        assert open(tmp_path / f"rank{rank}.txt").read() == want, f"rank {rank}"


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
