import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


# PAPR_TEST_FULL=1: the randomised suites at their full length and every rank count / fixture in the transports' matrices
# (what rounds 1-5 ran every time: +2 minutes on the GPU box).  The default tier keeps every TEST and trims the repetitions
# inside the soak-like ones: half of the random packet streams, rank counts 2 and 8 of (2, 4, 8), a third of the fixtures
# through RCCL at world size 1 (all of them go through the CLI in test_cli_reproduces_reference_stdout either way).
FULL_TIER = os.environ.get("PAPR_TEST_FULL", "0") not in ("", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Everything compiled (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    lib = os.path.join(ROOT, "dtv-utils_amd", "libpaprhip.so")
    cli = os.path.join(ROOT, "bin", "papr")
    orc = os.path.join(ROOT, "oracle", "libpapr_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(cli) and os.path.exists(orc)):
        ge.build()
    return ge


@pytest.fixture(scope="session")
def pkg(built):
    return built.load_package()


@pytest.fixture(scope="session")
def orc(built):
    return built.load_oracle()


@pytest.fixture(scope="session")
def manifest():
    import json
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))


def golden_names():
    import json
    m = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    return sorted(k for k in m if not k.startswith("big_"))


def golden_path(name):
    return os.path.join(GOLDEN, name + ".cfile")


def golden_text(name, graph):
    with open(os.path.join(GOLDEN, f"{name}.{'graph' if graph else 'default'}.txt"), "rb") as f:
        return f.read()
