import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_first():
    """On a GPU box torch's HIP runtime is brought up before the product library's first HIP call (the order bench.py and the
    sharded launcher use): a `-k` selection that reaches a torch test only after the library has run otherwise finds torch
    without a device ("No HIP GPUs are available"), whatever the library did before."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    return True


@pytest.fixture(scope="session")
def built():
    """Everything compiled (product for gfx950, oracle, CPU checker binary)."""
    import __graft_entry__ as g
    need = [os.path.join(ROOT, "pangene_amd", "lib", "libpangene_amd.so"), os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "tests", "_build", "libpangene_oraclehost.so")]
    if not all(os.path.exists(p) for p in need):
        g.build()
    return True


@pytest.fixture(scope="session")
def expected():
    with open(os.path.join(GOLD, "expected.json")) as f:
        return json.load(f)


def golden_files(name):
    d = os.path.join(GOLD, name)
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".paf.gz") or f.endswith(".paf"))


def all_cases():
    with open(os.path.join(GOLD, "expected.json")) as f:
        exp = json.load(f)
    return [(s, v) for s in sorted(exp) for v in sorted(exp[s])]
