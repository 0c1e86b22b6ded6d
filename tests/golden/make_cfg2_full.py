#!/usr/bin/env python3
"""Golden record of the FULL BASELINE cfg2 workload (10 M x 150 bp synthetic reads, k = 31): node count,
k-mer total, order-independent digest and DepthStat of the graph, computed by the CPU oracle
(oracle/dbg_oracle.c, pinned to the real reference by the other fixtures in this directory) --
tests/test_gpu_parity.py and bench.py compare the GPU result at full size with it.
    python tests/golden/make_cfg2_full.py [threads]      (CPU only; ~5 min, ~10 GB of RAM)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

N_READS, GENOME, K = 10_000_000, 50_000_000, 31


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    L = O.lib()
    P = O.synth_params(GENOME, 150, cfg=2)
    t0 = time.time()
    p = O.GraphParams(K, 250, threads, 0.4, 10, 0.7, 10000)   # 400 M slots: no enlarge on the way
    g = L.orc_graph_create(C.byref(p))
    step = 1_000_000   # the reads are generated piece by piece (one 'file' each)
    for first in range(0, N_READS, step):
        bases, offsets = O.synth_reads(P, first, step)
        L.orc_graph_add_file_mem(g, bases.ctypes.data, offsets.ctypes.data, step)
        print("reads %d  %.0f s" % (first + step, time.time() - t0), flush=True)
    L.orc_graph_finish(g)
    ks = L.orc_graph_kmerset(g).contents
    table = np.ctypeslib.as_array(C.cast(ks.array, C.POINTER(C.c_uint8)), shape=(ks.size * 16,)).view(O.NODE_DTYPE)
    flags = np.ctypeslib.as_array(C.cast(ks.nul_flag, C.POINTER(C.c_uint8)), shape=(ks.size // 8 + 1,))
    digest, depth = 0, np.zeros(256, dtype=np.int64)
    chunk = 1 << 26
    for lo in range(0, ks.size, chunk):   # occupied slots piece by piece (the digest is a sum, DepthStat a histogram)
        hi = min(lo + chunk, ks.size)
        occ = np.unpackbits(flags[lo // 8:(hi + 7) // 8])[:hi - lo].astype(bool)
        nodes = np.ascontiguousarray(table[lo:hi][occ])
        digest = (digest + O.nodes_digest(nodes)) % (1 << 64)
        depth += np.array(O.link_stats(nodes, 2).depth_stat, dtype=np.int64)
    out = {"workload": "BASELINE cfg2: synth_params(genome_len=50000000, read_len=150, cfg=2), reads [0, 10000000), k=31",
           "n_reads": N_READS, "genome_len": GENOME, "k": K, "total_reads": int(L.orc_graph_total_reads(g)),
           "total_kmers": int(L.orc_graph_total_kmers(g)), "count": int(ks.count), "digest": int(digest),
           "depth_stat": [int(x) for x in depth], "made_by": "oracle/dbg_oracle.c, %d threads" % threads}
    L.orc_graph_destroy(g)
    with open(os.path.join(ROOT, "tests", "golden", "cfg2_full.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("count %d digest %d in %.0f s" % (out["count"], out["digest"], time.time() - t0))


if __name__ == "__main__":
    main()
