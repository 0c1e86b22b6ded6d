#!/usr/bin/env python3
"""Randomised cross-check of the two forms of the WIDE engine on the GPU: for many random configurations (k up to 63,
read-length mix, N rate, repeats, table sizes from 2^26 up to beyond 2^31 slots, batching, flushes, record stores that
are ample / too small / absurdly small) a handle that goes through 16-byte records (expected_kmers > 0,
dbgk_wide_partition.h) must produce the same node multiset (count, k-mer totals, order-independent digest) as one that
uses the fused-atomic kernels (expected_kmers = 0).  No oracle involved: both sides are device code; the CPU
restatement pins each of them in tests/test_wide.py.
    python profiles/fuzz_wide.py [n_configs] [seed] [only_this_config]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dbg_assembly_amd import capi  # noqa: E402
from fuzz_engines import make_reads, pack  # noqa: E402


def push(g, reads, rng, may_flush):
    cuts = sorted(rng.sample(range(len(reads)), min(len(reads), rng.randint(0, 3))))
    for a, b in zip([0] + cuts, cuts + [len(reads)]):
        if b > a:
            g.push_reads(*pack(reads[a:b]))
            if may_flush and rng.random() < 0.3:
                g.flush()


def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    bad = 0
    for c in range(n_cfg):
        if only >= 0 and c != only:
            continue
        rng = random.Random(seed * 100019 + c)
        k = rng.choice([63, 63, 62, 47, 33, 32, 31, 17, 5, 1])
        L = rng.choice([150, 100, 64, 250, 400])
        max_read_len = rng.choice([250, 250, 100, 1000000])
        shape = rng.random()
        reads = make_reads(rng, rng.randint(1, 5000), rng.randint(max(L, 80), 60000), L, rng.choice([0.0, 0.003, 0.05]), rng.random() < 0.3,
                           uniform=shape < 0.35, nearly=0.35 <= shape < 0.6)
        if rng.random() < 0.3:   # keys whose low word is 0 (k > 32) and the key-0 node
            reads += [b"C" + b"A" * rng.randint(40, 130), b"A" * rng.randint(1, 200), b"GT" + b"A" * 90 + b"C"] * rng.randint(1, 30)
        with capi.Graph(k=k, table_slots=capi.find_next_prime_ref(3_000_000), engine=capi.ENGINE_WIDE, max_read_len=max_read_len) as g:
            push(g, reads, rng, False)
            st = g.finalize()
            want = (int(st.count), int(st.total_kmers), int(st.stored_kmers), g.digest())
        size = capi.find_next_prime_ref(rng.choice([1 << 26, 70_000_000, 100_000_007, 300_000_000, 2_200_000_000]))
        expected = rng.choice([max(1, want[2]), 3 * want[2] + 1, max(1, want[2] // 3), 1])
        with capi.Graph(k=k, table_slots=size, engine=capi.ENGINE_WIDE, max_read_len=max_read_len, expected_kmers=expected,
                        max_batch_bases=rng.choice([0, 1 << 16, 1 << 20])) as g:
            assert g.store_room()[1] == expected, "the record path was not selected"
            push(g, reads, rng, True)
            st = g.finalize()
            got = (int(st.count), int(st.total_kmers), int(st.stored_kmers), g.digest())
            g.reset()
            push(g, reads, rng, True)
            st = g.finalize()
            again = (int(st.count), int(st.total_kmers), int(st.stored_kmers), g.digest())
        ok = got == want and again == want
        print("config %3d  k %2d  L %3d  r %7d  reads %5d  slots %10d  expected %8d  nodes %8d  %s" %
              (c, k, L, max_read_len, len(reads), size, expected, want[0], "ok" if ok else "MISMATCH %r %r %r" % (want, got, again)), flush=True)
        bad += 0 if ok else 1
    print("%d configurations, %d mismatches" % (n_cfg, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
