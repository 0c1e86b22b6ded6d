#!/usr/bin/env python3
"""BASELINE configs[2] at its STATED size on ONE MI355X: 200 M x 150 bp reads of a 1 Gb genome (24 G k-mer occurrences,
k = 31) into ONE table of 8.6 G slots (138 GB).  The records of the whole job (192 GB, twice) do not fit beside the
table, so the PARTITION engine streams: the record store holds STORE_KMERS occurrences, every time it is full the
records are merged into the table (incremental region build, dbgk_flush) -- the same mechanism that lets
build_debruijn_graph() take input of unknown size.  The reads are generated on the device in chunks of 25 M (what one
rank of an 8-GPU node would generate) outside the timed regions.
Check at this size (no CPU oracle can follow): the DIRECT engine -- pinned to the oracle at small sizes, a completely
different code path (global atomics, no records) -- builds the same table from the same reads; count, k-mer total,
order-independent node digest and the link-depth histogram must agree.
    python profiles/measure_cfg3_full.py            (N_CHUNKS=8 READS_PER_CHUNK=25000000 STORE_KMERS=4000000000)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dbg_assembly_amd import capi  # noqa: E402

n_chunks = int(os.environ.get("N_CHUNKS", 8))
per = int(os.environ.get("READS_PER_CHUNK", 25_000_000))
store = int(os.environ.get("STORE_KMERS", 4_000_000_000))
genome = int(os.environ.get("GENOME", 125_000_000 * n_chunks))
slots = int(os.environ.get("SLOTS", 1_075_000_000 * n_chunks))
k, kpr = 31, 120
P = capi.synth_params(genome, 150, cfg=3)
size = capi.find_next_prime_ref(slots)
res = {"reads": n_chunks * per, "kmers": n_chunks * per * kpr, "genome": genome, "table_slots": size, "store_kmers": store}


def run(engine, expected):
    with capi.Graph(k=k, table_slots=size, engine=engine, expected_kmers=expected) as g:
        busy = 0.0
        for c in range(n_chunks):
            d_bases, d_off, nb = g.synth_reads_device(P, c * per, per)   # untimed: input generation
            g.sync()
            t0 = time.perf_counter()
            g.push_reads_device(d_bases.ptr, d_off.ptr, per, nb)
            g.sync()
            busy += time.perf_counter() - t0
            d_bases.free()
            d_off.free()
            print("  engine %d chunk %d/%d done, %.3f s so far" % (engine, c + 1, n_chunks, busy), file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        st = g.finalize()
        g.sync()
        busy += time.perf_counter() - t0
        tm = g.timings()
        out = {"seconds": busy, "G_kmers_per_s": n_chunks * per * kpr / busy / 1e9, "count": int(st.count), "total_kmers": int(st.total_kmers),
               "digest": g.digest(), "depth_stat_sum": int(sum(g.link_stats(2).depth_stat)),
               "device_ms": {"mark": tm.mark_ms, "level1": tm.insert_ms, "level2": tm.partition_ms, "build": tm.build_ms, "fixup": tm.fixup_ms,
                             "launches": int(tm.insert_launches)}}
        return out


res["partition_streamed"] = run(capi.ENGINE_PARTITION, store)
print(json.dumps(res), file=sys.stderr, flush=True)
res["direct"] = run(capi.ENGINE_DIRECT, 0)
res["equal"] = all(res["partition_streamed"][x] == res["direct"][x] for x in ("count", "total_kmers", "digest", "depth_stat_sum"))
print(json.dumps(res))
sys.exit(0 if res["equal"] else 1)
