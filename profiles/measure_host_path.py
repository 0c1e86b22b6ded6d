#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer path (dbgk_push_reads: pageable host memory -> pinned
double buffers -> H2D -> kernels -> finalize) on the bench workload.  Reported in DESIGN.md; never
used as bench.py's `value`."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dbg_assembly_amd import capi  # noqa: E402

n_reads = int(os.environ.get("N_READS", 10_000_000))
P = capi.synth_params(50_000_000, 150, cfg=2)
size = capi.find_next_prime_ref(600_000_000)
out = {}
for engine, name in ((capi.ENGINE_DIRECT, "direct"), (capi.ENGINE_PARTITION, "partition")):
    with capi.Graph(k=31, table_slots=size, engine=engine, expected_kmers=n_reads * 150, max_batch_bases=256 << 20) as g:
        d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
        bases = d_bases.to_host(np.uint8, nb)
        offsets = d_off.to_host(np.uint64)
        d_bases.free()
        d_off.free()
        best = None
        for rep in range(3):
            g.reset()
            g.sync()
            t0 = time.perf_counter()
            g.push_reads(bases, offsets)
            st = g.finalize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[name] = {"seconds": best, "M_kmers_per_s": st.stored_kmers / best / 1e6, "GB_per_s_host_bytes": nb / best / 1e9,
                     "count": int(st.count)}
print(json.dumps(out))
