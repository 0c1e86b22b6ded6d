#!/usr/bin/env python3
"""Randomised cross-check of the two k-mer frequency paths on the GPU: for random k (12..16), read mixes (repeats, poly-A / poly-T,
N, lower case, ragged and equal lengths), batchings and store sizes (ample, exact, a fraction of the input -> flushes, tiny ->
overflow lists) the PARTITION form (k >= 13: 64-KiB table blocks in LDS; k = 12: LDS hash table per region) must leave the same
4^k-byte table, distinct count and k-mer total as the atomic kernel.  No oracle involved: both sides are device code; the oracle
pins both separately in tests/test_kfreq.py.
    python profiles/fuzz_kfreq.py [n_configs] [seed]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dbg_assembly_amd import capi  # noqa: E402

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def make_reads(rng, n, G, L, uniform):
    g = "".join(rng.choice("ACGT") for _ in range(G))
    if rng.random() < 0.5:
        unit = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 30)))
        g = g[:G // 2] + unit * (G // (2 * len(unit)) + 1)
    out = []
    for _ in range(n):
        ln = L if (uniform or rng.random() < 0.7) else rng.randint(0, L + 100)
        ln = min(ln, len(g))
        s = rng.randint(0, len(g) - ln)
        r = list(g[s:s + ln])
        if rng.random() < 0.5:
            r = [COMP[c] for c in reversed(r)]
        for j in range(len(r)):
            x = rng.random()
            if x < 0.01:
                r[j] = rng.choice("ACGT")
            elif x < 0.012:
                r[j] = rng.choice("Nn")
            elif x < 0.02:
                r[j] = r[j].lower()
        out.append("".join(r).encode())
    ln = L if uniform else rng.randint(20, L)
    out += [b"A" * ln] * rng.randint(0, 300) + [b"T" * ln] * rng.randint(0, 40)
    rng.shuffle(out)
    return out


def pack(reads):
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    return np.frombuffer(b"".join(reads), dtype=np.uint8).copy(), offsets


def table_of(k, expected, bases, offsets, pieces, max_batch, packed=False):
    if packed:   # the same reads through the 2-bit entry (dbgk_push_reads_packed)
        words, _ = capi.pack_bases(bases)
    with capi.Graph(k=k, table_slots=0, engine=capi.ENGINE_KFREQ, max_read_len=1000000, expected_kmers=expected, max_batch_bases=max_batch) as g:
        n = len(offsets) - 1
        cuts = [n * i // pieces for i in range(pieces + 1)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a and packed and pieces == 1:
                g.push_reads_packed(words, offsets)
            elif b > a:
                g.push_reads(bases[int(offsets[a]):int(offsets[b])], offsets[a:b + 1] - offsets[a])
        st = g.finalize()
        return g.kfreq_counts(), int(st.count), int(st.stored_kmers)


def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    bad = 0
    for c in range(n_cfg):
        k = rng.choice([12, 13, 13, 14, 14, 15, 16])
        L = rng.choice([60, 100, 150, 151, 250])
        uniform = rng.random() < 0.5
        n = rng.randint(500, 6000)
        reads = make_reads(rng, n, rng.randint(2000, 200000), L, uniform)
        bases, offsets = pack(reads)
        total = sum(max(0, len(r) - k + 1) for r in reads)
        expected = max(1, int(total * rng.choice([4.0, 1.0, 0.4, 0.15, 0.01])))
        pieces = rng.randint(1, 4)
        max_batch = rng.choice([1 << 16, 1 << 20, 1 << 26])
        # a third of the configurations through the LINEAR level-1 forms (what tables of 4^18 bytes take: 1024 level-1 buckets), a
        # third through the wave-per-bucket forms whatever the geometry, the rest as the library decides; half of the one-piece
        # configurations as 2-bit words
        lin = rng.choice(["1", "0", None])
        packed = pieces == 1 and rng.random() < 0.5
        os.environ.pop("DBGK_TEST_HOOKS", None)
        want = table_of(k, 0, bases, offsets, 1, 1 << 26)
        if lin is not None:
            os.environ["DBGK_TEST_HOOKS"] = "l1_linear=%s" % lin
        got = table_of(k, expected, bases, offsets, pieces, max_batch, packed)
        os.environ.pop("DBGK_TEST_HOOKS", None)
        ok = got[1:] == want[1:] and np.array_equal(got[0], want[0])
        print("cfg %3d k=%2d L=%3d %s reads=%5d kmers=%8d expected=%9d pieces=%d batch=2^%d linear=%s %s distinct=%8d max=%3d  %s"
              % (c, k, L, "uniform" if uniform else "ragged ", len(reads), total, expected, pieces, max_batch.bit_length() - 1, lin, "packed" if packed else "ascii ",
                 want[1], int(want[0].max()),
                 "ok" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
    print("%d configurations, %d mismatches" % (n_cfg, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
