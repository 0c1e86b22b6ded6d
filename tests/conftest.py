import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """GPU sessions: bring torch's HIP runtime up BEFORE libdbgk.so is loaded.  The torch wheel bundles
    its own libamdhip64 (ROCm 7.0) while libdbgk.so is linked against /opt/rocm's (7.2); loaded in this
    order the process ends up with ONE runtime, in the other order torch.cuda.is_available() turns False
    for the tests that need torch.distributed (test_gpu_multigpu.py).  bench.py does the same."""
    expr = session.config.getoption("markexpr", "") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import torch
            torch.cuda.is_available()
        except Exception:  # noqa: BLE001 -- the tests that need torch report it themselves
            pass


def golden_cases():
    out = []
    for name in sorted(os.listdir(GOLDEN)):
        cj = os.path.join(GOLDEN, name, "case.json")
        if os.path.exists(cj):
            with open(cj) as fh:
                out.append(json.load(fh))
    return out


def golden_case_ids():
    return [c["name"] for c in golden_cases()]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py as O
    O.lib()
    return O
