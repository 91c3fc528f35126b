"""The ingest's io_uring reader (dtv-utils_amd/csrc/papr_uring.h) without a GPU: every byte it delivers against pread,
buffered and O_DIRECT, ring smaller than the number of requests (backlog), sizes around the 4 KiB block and the chunk."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("uring") / "uring_harness")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "dtv-utils_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "uring_harness.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("direct", [0, 1])
@pytest.mark.parametrize("size", [0, 1, 4095, 4096, 4097, 1048576, 3 * 1048576 + 4096 + 123])
def test_uring_reader_equals_pread(harness, tmp_path, size, direct):
    path = tmp_path / "f.bin"
    path.write_bytes(np.random.default_rng(size + direct).integers(0, 256, size, dtype=np.uint8).tobytes())
    for piece, chunk in ((4096, 65536), (65536, 1048576), (1 << 20, 1 << 22)):
        p = subprocess.run([harness, str(path), str(direct), str(piece), str(chunk)], capture_output=True, text=True)
        if p.stdout.startswith("unsupported"):
            pytest.skip("io_uring_setup is not offered here (the ingest keeps its reader threads)")
        assert p.returncode == 0 and p.stdout.startswith("ok"), (p.stdout, p.stderr)
