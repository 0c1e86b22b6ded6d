#!/usr/bin/env python3
"""N-rank slot-range flow at bench geometry on ONE GPU: N sharded handles in one process, the all-to-all
done with device copies, the records delivered in 8 ranges (dbgk_shard_plan / dbgk_shard_build_range).
Checks what an 8-GPU node would compute -- table geometry (global table of ~2^32 slots at N = 8, the
wide-divisor level-1 kernel), bucket capacities, hand-overs -- against the single-handle build of the
same reads.  Not a timing.   WORLD=4 READS_PER_RANK=10000000  |  WORLD=8 READS_PER_RANK=2000000"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dbg_assembly_amd import capi  # noqa: E402

world = int(os.environ.get("WORLD", 4))
n_reads = int(os.environ.get("READS_PER_RANK", 10_000_000))
k, kpr = 31, 120
P = capi.synth_params(50_000_000 * world, 150, cfg=2)
per_gpu_slots = min(600_000_000, (2 ** 32 - 2 ** 22) // world)     # bench.py's sizing
size = capi.find_next_prime_ref(per_gpu_slots * world)
graphs = [capi.Graph(k=k, table_slots=size, engine=capi.ENGINE_PARTITION, expected_kmers=n_reads * kpr,
                     shard_count=world, shard_index=r) for r in range(world)]
for r, g in enumerate(graphs):
    d_bases, d_off, nb = g.synth_reads_device(P, r * n_reads, n_reads)
    g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
    g.sync()
    d_bases.free()
    d_off.free()
infos = [g.shard_info() for g in graphs]
c0 = graphs[0]
for s in range(world):
    for d in range(world):
        c0.memcpy_d2d(infos[d].d_recv_cnt + s * infos[d].cnt_chunk_bytes, infos[s].d_send_cnt + d * infos[s].cnt_chunk_bytes,
                      infos[s].cnt_chunk_bytes)
c0.sync()
for g in graphs:
    g.shard_plan()
B = infos[0].buckets_per_rank
per = -(-B // 8)
for j0 in range(0, B, per):
    j1 = min(j0 + per, B)
    for s in range(world):
        for d in range(world):
            c0.memcpy_d2d(infos[d].d_recv + s * infos[d].chunk_bytes + j0 * infos[d].bucket_bytes,
                          infos[s].d_send + d * infos[s].chunk_bytes + j0 * infos[s].bucket_bytes, (j1 - j0) * infos[s].bucket_bytes)
    c0.sync()
    for g, info in zip(graphs, infos):
        a, b = min(j0, info.own_buckets), min(j1, info.own_buckets)
        if b > a:
            g.shard_build_range(a, b)
stats = []
for g in graphs:
    g.shard_mark_exchanged()
    stats.append(g.finalize())
ovf = [g.shard_overflow() for g in graphs]
out = [g.shard_outgoing() for g in graphs]
for (p, n) in ovf:
    for g in graphs:
        if n:
            g.shard_merge(p, n, is_triple=True)
for s, (p, n) in enumerate(out):
    if n:
        graphs[(s + 1) % world].shard_merge(p, n, from_previous_shard=True)
for s in range(1, world):
    if stats[s].polyA_l_link or stats[s].polyA_r_link:
        graphs[0].add_polyA(stats[s].polyA_l_link, stats[s].polyA_r_link)
final = [g.refresh_stats() for g in graphs]
digest = sum(g.digest() for g in graphs) % (1 << 64)
count = sum(int(f.count) for f in final)
res = {"world": world, "reads_per_rank": n_reads, "global_slots": size, "buckets_per_rank": B, "count": count,
       "kmers": sum(int(s.total_kmers) for s in stats), "overflow_observations": sum(n for _, n in ovf),
       "handed_over_nodes": sum(n for _, n in out), "digest": digest}
for g in graphs:
    g.close()
# the same reads through ONE handle (a table of the per-GPU size is enough for the node multiset)
one = capi.find_next_prime_ref(min(4_000_000_000, max(600_000_000, 3 * count)))
with capi.Graph(k=k, table_slots=one, engine=capi.ENGINE_PARTITION, expected_kmers=n_reads * kpr * world) as g:
    for r in range(world):
        d_bases, d_off, nb = g.synth_reads_device(P, r * n_reads, n_reads)
        g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
        g.sync()
        d_bases.free()
        d_off.free()
    st = g.finalize()
    res["single_handle"] = {"count": int(st.count), "kmers": int(st.total_kmers), "digest": g.digest()}
res["equal"] = (res["count"], res["kmers"], res["digest"]) == tuple(res["single_handle"][x] for x in ("count", "kmers", "digest"))
print(json.dumps(res))
sys.exit(0 if res["equal"] else 1)
