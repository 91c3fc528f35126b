#!/usr/bin/env python3
"""The bench / test workload (the spike stream of include/papr_synth.h) as a FILE, made by the device-side generator:
oracle/mkcfile's bytes (tests/test_synth.py pins the two generators against each other) at the speed of the file system
instead of the host's hash rate — a 256 GiB file in tmpfs took oracle/mkcfile's 64 processes 315 s (64 KiB fwrites into
one inode), which no test tier has room for.

  python tools/mkcfile_gpu.py <path> <nsamples> [--block-mib 2048] [--writers 32]

One GPU block at a time: papr_hip_generate into a device buffer, one copy into one of two pinned host buffers, and
`writers` threads pwrite() 64 MiB pieces of it to their offsets while the next block is generated and copied.  Test / bench
infrastructure: nothing of the product imports it."""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(path: str, nsamples: int, block_mib: int = 2048, writers: int = 32, device: int = 0) -> float:
    """Write the spike stream of `nsamples` samples to `path`; returns the seconds it took."""
    import numpy as np
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    t0 = time.perf_counter()
    block = max(8192, (block_mib << 20) // 8 // 8192 * 8192)   # samples per block
    block = min(block, max(nsamples, 1))
    dev = torch.device("cuda", device)
    shard = torch.empty(block * 8 + 65536, dtype=torch.uint8, device=dev)
    pinned = [torch.empty(block * 8, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    spec = pkg.SynthSpec.spike(nsamples)
    piece = 64 << 20
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        os.ftruncate(fd, nsamples * 8)
        with pkg.PaprHip(device) as gpu, ThreadPoolExecutor(max_workers=writers) as pool:
            gpu.adopt(shard.data_ptr(), block, keepalive=shard)
            pending = [[], []]
            for k, first in enumerate(range(0, nsamples, block)):
                n = min(block, nsamples - first)
                b = k & 1
                for f in pending[b]:      # the writes that still read this pinned buffer
                    f.result()
                gpu.generate(spec, first, n)
                pinned[b][:n * 8].copy_(shard[:n * 8])
                torch.cuda.synchronize(dev)
                host = pinned[b].numpy()
                base = first * 8

                def put(off, host=host, base=base, end=n * 8):
                    view = memoryview(host[off:min(off + piece, end)])
                    done = 0
                    while done < len(view):
                        done += os.pwrite(fd, view[done:], base + off + done)
                pending[b] = [pool.submit(put, off) for off in range(0, n * 8, piece)]
            for fs in pending:
                for f in fs:
                    f.result()
    finally:
        os.close(fd)
    if os.path.getsize(path) != nsamples * 8:
        raise RuntimeError("short file")
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("nsamples", type=int)
    ap.add_argument("--block-mib", type=int, default=2048)
    ap.add_argument("--writers", type=int, default=32)
    a = ap.parse_args()
    s = make(a.path, a.nsamples, a.block_mib, a.writers)
    print(f"{a.path}: {a.nsamples * 8} bytes in {s:.1f} s = {a.nsamples * 8 / s / 1e9:.2f} GB/s")


if __name__ == "__main__":
    main()
