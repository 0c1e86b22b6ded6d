#!/usr/bin/env python3
"""(round 5: DBGK_L1_PER_CU and friends need a library built with -DDBGK_EXPERIMENTS, profiles/tools/build_variant.sh)
Experiment (round 2, result in profiles/r02_exp_l1_beside_l2.txt): does the level-1 kernel of one batch run BESIDE level 2 of another?  Two handles on one GPU, same
workload (cfg2): handle A only pushes (level 1, asynchronous), handle B has its records in place and finalizes
(level 2, then the region build; DBGK_OVERLAP_CHUNKS=1 so that level 2 comes first).  Wall time of both together
against each alone.  Run with and without DBGK_L2_DIRECT=1 (the unstaged level-2 kernel needs no LDS to speak of and
fits on a CU next to level 1's 151 KiB; the staged one does not)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dbg_assembly_amd import capi  # noqa: E402

n_reads = int(os.environ.get("N_READS", 10_000_000))
P = capi.synth_params(50_000_000, 150, cfg=2)
size = capi.find_next_prime_ref(600_000_000)
mk = lambda: capi.Graph(k=31, table_slots=size, engine=capi.ENGINE_PARTITION, expected_kmers=n_reads * 120)
a, b = mk(), mk()
d_bases, d_off, nb = a.synth_reads_device(P, 0, n_reads)
res = {}


def timed(fn, reps=4):
    best = None
    for _ in range(reps):
        a.reset(); b.reset()
        b.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
        a.sync(); b.sync()
        t0 = time.perf_counter()
        fn()
        a.sync(); b.sync()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    return best


res["l1_alone_ms"] = timed(lambda: a.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb))
res["finalize_alone_ms"] = timed(lambda: b.finalize())


def both():
    a.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)   # returns after the stats round trip; level 1 is queued
    b.finalize()


res["together_ms"] = timed(both)
res["l2_direct"] = bool(os.environ.get("DBGK_L2_DIRECT"))
print(json.dumps(res))
