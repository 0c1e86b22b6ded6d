"""The 128-bit key path against an INDEPENDENT checker (tests/wide_checker.py: strings and Python ints, nothing shared
with include/dbgk_wide.h).  PARITY UNPINNED above k = 32 -- the reference stops at 31 -- so the chain is:

  checker at k <= 32  == the real reference's dumps (tests/golden)                      [CPU]
  oracle/wide_oracle.cpp == checker at k in {33, 47, 62, 63}                            [CPU]
  WIDE engine (atomic kernels AND the 16-byte-record path) == checker at those k        [GPU]

with inputs that cover trimming at -r, poly-A / poly-T (key 0), keys whose low word is 0, more than 255 repeats
(saturation), lower case, N, reads shorter than k, reads of exactly k bases."""
import random

import numpy as np
import pytest

import wide_checker as W
from conftest import golden_cases
from helpers import case_reads, dump_sha256

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
SMALL_GOLDENS = [c for c in golden_cases() if "synth" not in c and c["name"] != "enlarge_cap_e1"]


def _reads(rng, n, G=4000, L=150):
    g = "".join(rng.choice("ACGT") for _ in range(G))
    out = []
    for _ in range(n):
        ln = L if rng.random() < 0.6 else rng.randint(0, L + 150)
        s = rng.randint(0, G - min(ln, G))
        r = list(g[s:s + ln])
        if rng.random() < 0.5:
            r = [COMP[c] for c in reversed(r)]
        for j in range(len(r)):
            x = rng.random()
            if x < 0.01:
                r[j] = rng.choice("ACGT")
            elif x < 0.014:
                r[j] = rng.choice("Nn")
        r = "".join(r)
        out.append((r.lower() if rng.random() < 0.1 else r).encode())
    unit = b"ACGTTGCATGCAAGCTTAGCTAGGATCCGATCGATTACGATACGTTGCATGCAAGCTTAGCTAGGATC"
    out += [b"A" * 200, b"T" * 177, b"", b"ACGT", b"C" + b"A" * 120, b"GT" + b"A" * 90 + b"C", b"T" * 64 + b"G", unit[:40] * 4] * 3
    out += [unit] * 300                        # one 67-mer 300 times: every counter it touches saturates
    out += [b"G" + b"A" * 70 + b"T"] * 2       # keys with 32+ trailing A: low word 0, high word not
    out += [unit[:33], unit[:47], unit[:62], unit[:63], unit[:64]]   # reads of exactly k bases (one window, no neighbour)
    rng.shuffle(out)
    return out


@pytest.mark.parametrize("case", SMALL_GOLDENS, ids=[c["name"] for c in SMALL_GOLDENS])
def test_independent_checker_at_k_le_32_equals_the_real_references_dump(oracle, case):
    p, ref = case["params"], case["ref"]
    reads = []
    for bases, offsets in case_reads(case, oracle):
        reads += W.split_reads(bases, offsets)
    nodes, total = W.build(reads, p["k"], p["max_read_len"])
    got = W.as_sorted_nodes(nodes)
    assert total == ref["kmers"] and len(got) == ref["count"]
    assert not got["kmer_hi"].any()
    narrow = np.zeros(len(got), dtype=oracle.NODE_DTYPE)
    narrow["kmer"], narrow["l_link"], narrow["r_link"] = got["kmer_lo"], got["l_link"], got["r_link"]
    assert dump_sha256(narrow, ref["reads"], ref["kmers"], ref["count"]) == case["dump_sha256"]


@pytest.mark.parametrize("k,r", [(33, 250), (47, 100), (62, 64), (63, 250), (63, 63), (63, 120), (32, 250), (31, 100)])
def test_wide_restatement_equals_independent_checker_PARITY_UNPINNED_above_k32(oracle, k, r):
    rng = random.Random(7000 + 100 * k + r)
    reads = _reads(rng, 250)
    bases, offsets = oracle.pack_reads(reads)
    got, total = oracle.wide_build(bases, offsets, k, r)
    nodes, want_total = W.build(reads, k, r)
    want = W.as_sorted_nodes(nodes)
    assert total == want_total
    assert np.array_equal(got, want)
    if k > 32:   # the inputs reach what they are meant to reach
        assert want["kmer_hi"].any() and ((want["kmer_lo"] == 0) & (want["kmer_hi"] != 0)).any()
        if r > k:   # (with -r == k every read is one window without neighbours: all counters stay 0)
            links = np.concatenate([want["l_link"], want["r_link"]])
            assert max(int(((links >> s) & 0xFF).max()) for s in (0, 8, 16, 24)) == 255   # saturation reached
    assert want[0]["kmer_hi"] == 0 and want[0]["kmer_lo"] == 0 and (r == k or want[0]["l_link"] or want[0]["r_link"])   # poly-A / poly-T seen


@pytest.fixture(scope="module")
def capi():
    from dbg_assembly_amd import capi as c
    assert c.lib().dbgk_device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback exists)"
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("k,r", [(33, 250), (47, 100), (62, 64), (63, 250), (63, 63), (63, 120)])
@pytest.mark.parametrize("path", ["atomic", "records"])
def test_wide_engine_equals_independent_checker_PARITY_UNPINNED(capi, oracle, k, r, path):
    rng = random.Random(9000 + 100 * k + r)
    reads = _reads(rng, 250)
    bases, offsets = oracle.pack_reads(reads)
    nodes, want_total = W.build(reads, k, r)
    want = W.as_sorted_nodes(nodes)
    slots, expected = (capi.find_next_prime_ref(3 * len(want)), 0) if path == "atomic" else (capi.find_next_prime_ref(1 << 26), len(bases))
    with capi.Graph(k=k, table_slots=slots, max_read_len=r, engine=capi.ENGINE_WIDE, expected_kmers=expected, max_batch_bases=1 << 15) as g:
        assert (g.store_room()[1] > 0) == (path == "records")
        g.push_reads(bases, offsets)
        st = g.finalize()
        assert (int(st.total_reads), int(st.total_kmers), int(st.count)) == (len(reads), want_total, len(want))
        assert np.array_equal(g.wide_export_sorted(), want.astype(capi.NODE32_DTYPE))
