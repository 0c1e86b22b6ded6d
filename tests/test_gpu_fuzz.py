"""GPU (-m gpu): randomised cross-check of the PARTITION engine (single handle, 1-3 in-process shards,
records delivered whole or in ranges, tables from 2^26 to beyond 2^31 slots, trimmed / short / empty
reads, N runs, repeats, multi-batch pushes) against the DIRECT engine -- profiles/fuzz_engines.py with a
fixed seed.  Both engines are pinned to the oracle separately in test_gpu_parity.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [7, 8])
def test_partition_engine_equals_direct_engine_on_random_configurations(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "fuzz_engines.py"), "12", str(seed)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "12 configurations, 0 mismatches" in r.stdout
