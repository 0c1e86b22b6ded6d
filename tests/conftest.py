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
