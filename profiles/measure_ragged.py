#!/usr/bin/env python3
"""Level-1 kernel time on a batch whose reads are mostly 150 bases with a tenth trimmed to 100..149:
the ragged lane-per-chunk kernel against the flat kernel (DBGK_L1_FLAT=1).  Kernel times from the
library's HIP events; the reads come from host memory, PCIe is not in the kernel time."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dbg_assembly_amd import capi  # noqa: E402

n_reads = int(os.environ.get("N_READS", 4_000_000))
rng = np.random.default_rng(3)
lens = np.where(rng.random(n_reads) < 0.9, 150, rng.integers(100, 150, n_reads)).astype(np.uint64)
offs = np.zeros(n_reads + 1, dtype=np.uint64)
offs[1:] = np.cumsum(lens)
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)]
size = capi.find_next_prime_ref(600_000_000)
with capi.Graph(k=31, table_slots=size, engine=capi.ENGINE_PARTITION, expected_kmers=int((lens - 30).sum()),
                max_batch_bases=int(offs[-1]) + 4096) as g:
    best = None
    for rep in range(3):
        g.reset()
        g.reset_timings()
        g.push_reads(bases, offs)
        st = g.finalize()
        tm = g.timings()
        if best is None or tm.insert_ms < best[0]:
            best = (tm.insert_ms, int(tm.uniform_launches), int(tm.insert_launches))
    print(json.dumps({"flat_forced": bool(os.environ.get("DBGK_L1_FLAT")), "level1_ms": best[0], "lane_per_chunk_launches": best[1],
                      "launches": best[2], "kmers": int(st.stored_kmers), "nodes": int(st.count), "digest": g.digest()}))
