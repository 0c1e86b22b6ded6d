#!/usr/bin/env python3
"""N-rank slot-range flow at the geometry of a whole node on ONE GPU, through the C++ communicator
(dbgk_comm_*: N sharded handles in one process, the exchange done with device copies in pieces overlapped
with the builds).  Checks what an 8-GPU node would compute -- the global table geometry (BASELINE cfg3: ONE
table of ~2^33 slots, 64-bit hash/size division, level-2 fan-out 4096), bucket capacities, hand-overs --
against the single-handle build of the same reads.  Not a timing.
    WORLD=8 READS_PER_RANK=3000000 GENOME=120000000 SLOTS=8600000000   (cfg3 geometry, reads scaled to one GPU's HBM)
    WORLD=4 READS_PER_RANK=10000000 GENOME=200000000 SLOTS=2400000000  (cfg2 per GPU)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dbg_assembly_amd import capi  # noqa: E402

world = int(os.environ.get("WORLD", 8))
n_reads = int(os.environ.get("READS_PER_RANK", 3_000_000))
genome = int(os.environ.get("GENOME", 120_000_000))
slots = int(os.environ.get("SLOTS", 8_600_000_000))
k, kpr = 31, 120
P = capi.synth_params(genome, 150, cfg=3)
size = capi.find_next_prime_ref(slots)

# the reads of every rank, generated on the device and brought to the host once (the communicator takes host batches)
parts = []
with capi.Graph(k=k, table_slots=1009, engine=capi.ENGINE_DIRECT) as tmp:
    for r in range(world):
        d_bases, d_off, nb = tmp.synth_reads_device(P, r * n_reads, n_reads)
        parts.append((d_bases.to_host(np.uint8, nb).copy(), d_off.to_host(np.uint64).copy()))
        d_bases.free()
        d_off.free()

with capi.Comm(k=k, table_slots=size, devices=[0] * world, expected_kmers=n_reads * kpr, max_batch_bases=512 << 20) as c:
    for bases, offsets in parts:   # one push per rank: round robin puts rank r's reads on shard r
        c.push_reads(bases, offsets)
    st = c.finalize()
    res = {"world": world, "reads_per_rank": n_reads, "genome": genome, "global_slots": size, "count": int(st.count),
           "kmers": int(st.total_kmers), "digest": c.digest(), "depth_sum": int(sum(c.link_stats(2).depth_stat))}

# the same reads through ONE handle (any table size gives the same node multiset)
one = capi.find_next_prime_ref(min(4_000_000_000, max(600_000_000, 3 * res["count"])))
with capi.Graph(k=k, table_slots=one, engine=capi.ENGINE_PARTITION, expected_kmers=n_reads * kpr * world, max_batch_bases=512 << 20) as g:
    for bases, offsets in parts:
        g.push_reads(bases, offsets)
    st = g.finalize()
    res["single_handle"] = {"slots": one, "count": int(st.count), "kmers": int(st.total_kmers), "digest": g.digest(),
                            "depth_sum": int(sum(g.link_stats(2).depth_stat))}
res["equal"] = all(res[x] == res["single_handle"][x] for x in ("count", "kmers", "digest", "depth_sum"))
print(json.dumps(res))
sys.exit(0 if res["equal"] else 1)
