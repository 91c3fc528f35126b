"""The CPU oracle (oracle/papr_oracle.c) pinned against the real reference:
committed golden stdout (recorded from /root/reference/papr.c by
tests/golden/make_golden.py) and, when oracle/_ref/papr is present, live
differential fuzzing against the reference binary."""
import os
import subprocess

import numpy as np
import pytest

from conftest import golden_names, golden_path, golden_text


@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
@pytest.mark.parametrize("name", golden_names())
def test_oracle_cli_matches_golden_stdout(orc, manifest, name, graph):
    args = (["-g"] if graph else []) + [golden_path(name)]
    rc, out, err = orc.cli(args)
    want = manifest[name]["graph" if graph else "default"]
    assert rc == want["rc"]
    assert err.decode() == want["stderr"]
    assert out == golden_text(name, graph)


def test_golden_vectors_from_survey(orc):
    """Hand-checkable known answers."""
    k8 = golden_text("k8", False).decode().splitlines()
    assert k8[0] == "Peak magnitude = 5.000000"
    assert k8[1] == "average power = 4.937500, peak power = 25.000000 @ 48"
    assert k8[3] == "Maximum PAPR = 7.044329"
    assert k8[4:12] == [f"percentage above {d} dB = 12.50000000" for d in range(8)]
    assert k8[-2] == "peak real positive @ 48, peak imaginary positive @ 49"
    assert k8[-1] == "peak real negative @ 40, peak imaginary negative @ 33"
    assert golden_text("k8", True).decode().splitlines() == ["12.50000000"] * 71
    tie = golden_text("tie", False).decode().splitlines()
    assert tie[1] == "average power = 1.016000, peak power = 5.000000 @ 24"
    assert tie[3] == "Maximum PAPR = 6.920763"
    assert tie[4:11] == [f"percentage above {d} dB = 20.00000030" for d in range(7)]
    one = golden_text("one", False).decode().splitlines()
    assert one[1] == "average power = 9.000000, peak power = 9.000000 @ 0"
    assert one[4] == "percentage above 0 dB = 0.00000000"
    r = orc.run_file(golden_path("k8"), False)
    assert (r["n"], r["peak"], r["peak_idx"], r["sum"]) == (8, 25.0, 6, 39.5)


def test_cli_grammar_and_errors(orc):
    """argv grammar, messages and exit codes of papr.c:53-98."""
    usage = b"usage: papr -g <infile>\nOptions:\n\tg = graph suitable output\n"
    for args in ([], ["a", "b"], ["a", "b", "c"]):
        rc, out, err = orc.cli(args)
        assert (rc, out, err) == (255, b"", usage)
    rc, out, err = orc.cli(["/nonexistent/file.cfile"])
    assert (rc, out, err) == (255, b"", b"Cannot open bitstream file </nonexistent/file.cfile>\n")
    rc, out, err = orc.cli(["-xGy", golden_path("k8")])
    assert rc == 0 and err == b"Unsupported Option: x\nUnsupported Option: y\n"
    assert out == golden_text("k8", True)
    rc, out, err = orc.cli(["-", golden_path("k8")])
    assert rc == 0 and out == golden_text("k8", False)


def test_run_mem_equals_run_file(orc):
    for name in ("g1m", "odd", "chunk1plus", "ties", "tiny"):
        data = np.fromfile(golden_path(name), dtype=np.uint8)
        floats = data[: data.size // 4 * 4].view(np.float32)
        if data.size % 4:
            continue  # stray bytes only exist in the file form
        for graph in (False, True):
            a, b = orc.run_mem(floats, graph), orc.run_file(golden_path(name), graph)
            for k in a:
                if isinstance(a[k], np.ndarray):
                    assert np.array_equal(a[k], b[k]), (name, k)
                else:
                    assert a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]), (name, k)


def test_count_mem_and_levels_from_agree_with_full_run(orc):
    floats = np.fromfile(golden_path("spike20k"), dtype=np.float32)
    for graph in (False, True):
        full = orc.run_mem(floats, graph)
        mean, papr, table = orc.levels_from(full["sum"], full["n"], full["peak"], graph)
        assert mean == full["mean"] and papr == full["papr"] and np.array_equal(table, full["level"])
        assert np.array_equal(orc.count_mem(floats, table), full["count"])


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "papr")),
                    reason="oracle/_ref/papr (the compiled reference) is not present")
def test_oracle_vs_reference_binary_fuzz(orc, tmp_path):
    """Differential fuzz against the real reference program: sizes around the
    64 KiB fread chunk, odd float counts, stray bytes, spikes, non-finite values."""
    rng = np.random.default_rng(20260928)
    sizes = [0, 1, 2, 3, 8191, 8192, 8193, 16383, 16384, 16385, 20000, 24577, 40000]
    cases = 0
    for n in sizes:
        for variant in range(3):
            extra = []
            if variant == 1:
                extra = ["--extra-floats", "1"]
            if variant == 2:
                extra = ["--extra-floats", "1", "--extra-bytes", str(int(rng.integers(1, 4)))]
            if n >= 16 and rng.random() < 0.5:
                extra += ["--spike"]
            if n >= 100 and rng.random() < 0.3:
                extra += ["--set", str(int(rng.integers(0, n))), rng.choice(["nan", "-nan", "inf", "7.5"]), "0.5"]
            extra += ["--seed", str(int(rng.integers(1, 2**31)))]
            if rng.random() < 0.3:
                extra += ["--scale", str(float(rng.choice([1e-9, 2.2e-3 / 65536, 1e-3, 37.0])))]
            path = str(tmp_path / f"f{n}_{variant}.cfile")
            subprocess.check_call([orc.MKCFILE, path, str(n), *extra])
            for mode in ([], ["-g"]):
                got = orc.cli(mode + [path])
                want = orc.cli(mode + [path], binary=orc.REF_CLI)
                assert got == want, (n, variant, extra, mode)
                cases += 1
            os.unlink(path)
    assert cases == len(sizes) * 3 * 2
