"""The measurement helpers under tools/ are run by hand on the GPU box: here only that they still parse (bash -n, py_compile)
and that tools/README.md names every one of them — a helper that rots is worse than none."""
import glob
import os
import py_compile
import subprocess

import pytest

from conftest import ROOT

TOOLS = os.path.join(ROOT, "tools")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(TOOLS, "*.sh"))), ids=os.path.basename)
def test_shell_tools_parse(path):
    subprocess.check_call(["bash", "-n", path])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(TOOLS, "*.py"))), ids=os.path.basename)
def test_python_tools_compile(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)


def test_readme_names_every_tool():
    text = open(os.path.join(TOOLS, "README.md")).read()
    missing = [os.path.basename(p) for p in sorted(glob.glob(os.path.join(TOOLS, "*")))
               if os.path.basename(p) != "README.md" and os.path.basename(p) not in text]
    assert not missing, missing


def test_bench_without_a_gpu_says_so_and_starts_nothing():
    """bench.py has no CPU path to fall back to (the product has none): without a GPU it says so in one line and exits
    non-zero — as a single process, and as the launcher of its own ranks (`--gpus N`: no rank is started, nothing hangs)."""
    import subprocess
    import sys
    try:
        import torch
        if torch.cuda.is_available():
            import pytest
            pytest.skip("this box has a GPU")
    except ImportError:
        pass
    for extra in ([], ["--gpus", "4"], ["--gpus", "2", "--backend", "gloo"]):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", *extra],
                           capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and p.stdout.strip() == "", (extra, p.stdout[-300:])
        assert "needs a GPU" in p.stderr, (extra, p.stderr[-300:])
