#!/usr/bin/env python3
"""Golden record of the FULL BASELINE cfg2 workload (10 M x 150 bp synthetic reads, k = 31): node count,
k-mer total, order-independent digest and DepthStat of the graph, computed by the CPU oracle
(oracle/dbg_oracle.c, pinned to the real reference by the other fixtures in this directory) --
tests/test_gpu_parity.py and bench.py compare the GPU result at full size with it.
    python tests/golden/make_cfg2_full.py [threads]      (CPU only; ~5 min, ~10 GB of RAM)
    python tests/golden/make_cfg2_full.py --ref [threads]
        runs the REAL reference (oracle/_ref/ref_dbg build -S, compiled in place from /root/reference by oracle/Makefile)
        on the same 10 M reads written as one-line FASTA, asserts reads / k-mer total / count / digest / DepthStat equal
        the committed record and stamps it "confirmed_by": the full-size record is then pinned to the reference itself.  Since
        round 5 the reference also hashes its canonical dump -- the non-null slots' 16-byte KmerNodes sorted by kmer, SHA-256
        (ref_dbg -H) -- and the record keeps it as "sorted_sha256": what dbgk_export_sorted returns at full size is compared
        byte for byte (tests/test_gpu_parity.py, bench.py).
    python tests/golden/make_cfg2_full.py --ref-cfg3 [threads]
        the same from the REAL reference for a cfg3-seeded input small enough for it (the first 5 M reads of bench.py's
        cfg3 share: synth cfg = 3, 125 Mb genome) -> tests/golden/cfg3_pin.json: cfg3's generator gets an oracle pin too."""
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


def run_reference(P, n_reads, threads, init):
    """ref_dbg build -S -H on reads [0, n_reads) of generator P written as one-line FASTA -> its JSON line"""
    import subprocess
    import tempfile
    assert O.have_ref(), "oracle/_ref/ref_dbg is missing: make -C oracle ref (needs /root/reference)"
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        fa = os.path.join(tmp, "reads.fa")
        O.lib().orc_synth_write_file(C.byref(P), 0, n_reads, os.fsencode(fa), 2, 0)
        libf = os.path.join(tmp, "reads.lib")
        with open(libf, "w") as fh:
            fh.write(fa + "\n")
        print("reads written (%.1f GB) %.0f s" % (os.path.getsize(fa) / 1e9, time.time() - t0), flush=True)
        cmd = [O.REF_BIN, "build", "-k", str(K), "-r", "250", "-f", "2", "-t", str(threads), "-i", str(init), "-l", "0.7",
               "-e", "10", "-b", "10000", "-S", "-H", "-q", libf]
        out = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def pin_cfg3(threads):
    n_reads, genome = 5_000_000, 125_000_000
    t0 = time.time()
    js = run_reference(O.synth_params(genome, 150, cfg=3), n_reads, threads, 0.5)
    assert js["nonnull_slots"] == js["count"] == js["sorted_records"]
    out = {"workload": "cfg3's generator at a size the reference takes: synth_params(genome_len=%d, read_len=150, cfg=3), reads [0, %d), k=%d" % (genome, n_reads, K),
           "n_reads": n_reads, "genome_len": genome, "k": K, "total_reads": js["reads"], "total_kmers": js["kmers"], "count": js["count"],
           "digest": js["digest"], "depth_stat": js["depth_stat"], "sorted_sha256": js["sorted_sha256"],
           "made_by": "ref_dbg (the real reference, DBG_contig/{seqKmer,kmerSet,DBGgraph,gzstream}.cpp compiled in place) build -k %d -t %d -i 0.5 "
                      "-b 10000 -S -H on the reads as one-line FASTA; its wall %.0f s" % (K, threads, js["wall_s"])}
    with open(os.path.join(ROOT, "tests", "golden", "cfg3_pin.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("cfg3 pin: count %d digest %d sha256 %s in %.0f s" % (out["count"], out["digest"], out["sorted_sha256"], time.time() - t0))


def confirm_with_reference(threads):
    path = os.path.join(ROOT, "tests", "golden", "cfg2_full.json")
    with open(path) as fh:
        gold = json.load(fh)
    t0 = time.time()
    js = run_reference(O.synth_params(GENOME, 150, cfg=2), N_READS, threads, 0.4)
    got = (js["reads"], js["kmers"], js["count"], js["nonnull_slots"], js["digest"], js["depth_stat"])
    want = (gold["total_reads"], gold["total_kmers"], gold["count"], gold["count"], gold["digest"], gold["depth_stat"])
    assert got == want, "the real reference disagrees with tests/golden/cfg2_full.json: %r vs %r" % (got[:5], want[:5])
    assert js["sorted_records"] == gold["count"]
    gold["sorted_sha256"] = js["sorted_sha256"]   # SHA-256 of the reference's canonical dump as packed 16-byte KmerNodes, sorted by kmer
    gold["confirmed_by"] = ("ref_dbg (the real reference, DBG_contig/{seqKmer,kmerSet,DBGgraph,gzstream}.cpp compiled in place) "
                            "build -k 31 -t %d -i 0.4 -b 10000 -S -H on the same 10 M reads as one-line FASTA: reads, k-mer total, "
                            "count, digest and DepthStat identical, sorted_sha256 taken from it; its wall %.0f s" % (threads, js["wall_s"]))
    with open(path, "w") as fh:
        json.dump(gold, fh, indent=1)
    print("confirmed by the real reference in %.0f s (its build: %.0f s)" % (time.time() - t0, js["wall_s"]))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--ref":
        return confirm_with_reference(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    if len(sys.argv) > 1 and sys.argv[1] == "--ref-cfg3":
        return pin_cfg3(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
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
