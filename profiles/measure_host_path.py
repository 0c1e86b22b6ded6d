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

# the same workload through the command line (host/DBGgraph.cpp: gz/plain parser -> batches -> dbgk_push_reads -> finalize -> host
# KmerSet): which engine runs behind build_debruijn_graph(), its device time, and the wall time with the parser in front
import subprocess
import tempfile
import ctypes as C
from oracle import oracle_py as O  # the generator that writes the reads file (test infrastructure; nothing of it is timed)
cli = os.path.join(ROOT, "dbg_assembly_amd", "bin", "debruijn_contig")
with tempfile.TemporaryDirectory() as tmp:
    fa = os.path.join(tmp, "reads.fa")
    PO = O.synth_params(50_000_000, 150, cfg=2)
    O.lib().orc_synth_write_file(C.byref(PO), 0, n_reads, os.fsencode(fa), 2, 0)
    lib = os.path.join(tmp, "reads.lib")
    open(lib, "w").write(fa + "\n")
    for engine, name in (("2", "cli_partition"), ("1", "cli_direct")):
        env = dict(os.environ, DBGK_ENGINE=engine, DBGK_TIMINGS="1")
        t0 = time.perf_counter()
        r = subprocess.run([cli, "-k", "31", "-f", "2", "-i", "0.6", "-t", "16", "-o", os.path.join(tmp, "out"), lib], env=env, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        line = [l for l in r.stderr.splitlines() if l.startswith("GPU phases")]
        count = [l for l in r.stderr.splitlines() if l.startswith("count:")]
        host = [l for l in r.stderr.splitlines() if l.startswith("Host phases")]
        out[name] = {"rc": r.returncode, "wall_seconds": dt, "M_kmers_per_s_wall": n_reads * 120 / dt / 1e6, "gpu_phases": line[-1] if line else None, "host_phases": host[-1] if host else None,
                     "count_line": count[-1] if count else None}
print(json.dumps(out))
