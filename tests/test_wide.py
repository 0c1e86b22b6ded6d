"""WIDE key path: k-mers of up to 63 bases, 128-bit keys, 32-byte nodes (BASELINE cfg5; include/dbgk_wide.h).

The reference stops at k = 31, so for k > 32 there is nothing to pin against: those tests say PARITY UNPINNED
in their names and compare the HIP kernels with this build's own CPU restatement (oracle/wide_oracle.cpp), which
shares the rule definitions (include/dbgk_wide.h) but not the extraction code.  What anchors the path is the
k <= 32 half: there every wide rule reduces to the reference's 64-bit one, and

  CPU: the wide restatement == the pinned oracle (dbg_oracle.c, itself pinned to the real reference) on every
       golden fixture, node for node, and == the real reference's dump hash;
  GPU: the WIDE engine == the golden dumps of the real reference at k <= 32, and its digest / host-table layout
       coincide with the 64-bit engines' (same hash, same slot).
"""
import random

import numpy as np
import pytest

from conftest import golden_cases, golden_case_ids
from helpers import case_reads, dump_sha256

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def as_narrow(nodes32, oracle):
    """wide nodes whose high word is 0 -> the 16-byte node type"""
    assert not nodes32["kmer_hi"].any()
    out = np.zeros(len(nodes32), dtype=oracle.NODE_DTYPE)
    out["kmer"], out["l_link"], out["r_link"] = nodes32["kmer_lo"], nodes32["l_link"], nodes32["r_link"]
    return out


@pytest.mark.parametrize("case", golden_cases(), ids=golden_case_ids())
def test_wide_restatement_at_k_le_32_equals_pinned_oracle_and_reference_dump(oracle, case):
    p, ref = case["params"], case["ref"]
    if case["name"] == "enlarge_cap_e1":
        pytest.skip("the reference drops reads at the -e cap: a property of the block loop, not of the key path")
    files = case_reads(case, oracle)
    bases = np.concatenate([b for b, _ in files]) if files else np.zeros(0, np.uint8)
    offs = [np.zeros(1, np.uint64)]
    for _, o in files:
        offs.append(o[1:] + offs[-1][-1])
    offsets = np.concatenate(offs)
    nodes32, total = oracle.wide_build(bases, offsets, p["k"], p["max_read_len"])
    narrow = as_narrow(nodes32, oracle)
    want = oracle.build_graph(files_mem=files, k=p["k"], max_read_len=p["max_read_len"], init_hash_size=max(p["init_hash_size"], 0.001))
    assert total == want.total_kmers == ref["kmers"]
    assert np.array_equal(narrow, want.nodes)
    assert dump_sha256(narrow, ref["reads"], ref["kmers"], ref["count"]) == case["dump_sha256"]   # the real reference's dump
    assert oracle.wide_digest(nodes32) == oracle.nodes_digest(want.nodes)


def test_wide_core_known_answers(oracle):
    """KATs of the reference (tests/golden/kat.txt: seq2bit / get_rev_com_kbit) through the 128-bit code path, and
    the 63-mer analogues by construction (reverse complement of a string, computed independently in Python)"""
    rng = random.Random(63)
    for k in (1, 2, 31, 32, 33, 47, 62, 63):
        for _ in range(50):
            s = "".join(rng.choice("ACGT") for _ in range(k))
            rc = "".join(COMP[c] for c in reversed(s))
            val = lambda t: sum("ACGT".index(c) << (2 * (len(t) - 1 - i)) for i, c in enumerate(t))
            canon = min(val(s), val(rc))
            nodes, total = oracle.wide_build(np.frombuffer(s.encode(), np.uint8), np.array([0, k], np.uint64), k, 250)
            assert total == 1
            key = [(int(n["kmer_hi"]) << 64) | int(n["kmer_lo"]) for n in nodes if n["kmer_hi"] or n["kmer_lo"]]
            assert key == ([canon] if canon else [])
    line = [l.split("\t") for l in open(__file__.replace("test_wide.py", "golden/kat.txt")) if l.startswith("seq2bit\tACGTACGTACGTACGTACGTACGTACGTACG\t")][0]
    fwd, rc = int(line[2]), int(line[4])
    nodes, _ = oracle.wide_build(np.frombuffer(b"ACGTACGTACGTACGTACGTACGTACGTACG", np.uint8), np.array([0, 31], np.uint64), 31, 250)
    assert int(nodes[1]["kmer_lo"]) == min(fwd, rc)


def _reads(rng, n, G=6000, L=150):
    g = "".join(rng.choice("ACGT") for _ in range(G))
    out = []
    for _ in range(n):
        ln = L if rng.random() < 0.7 else rng.randint(0, L + 150)
        s = rng.randint(0, G - min(ln, G))
        r = list(g[s:s + ln])
        if rng.random() < 0.5:
            r = [COMP[c] for c in reversed(r)]
        for j in range(len(r)):
            x = rng.random()
            if x < 0.01:
                r[j] = rng.choice("ACGT")
            elif x < 0.013:
                r[j] = "N"
        r = "".join(r)
        out.append((r.lower() if rng.random() < 0.1 else r).encode())
    # poly-A / poly-T (key 0), runs of A after other bases (keys whose low word is 0 at k > 32), short and empty reads
    out += [b"A" * 200, b"T" * 177, b"", b"ACGT", b"C" + b"A" * 120, b"GT" + b"A" * 90 + b"C", b"T" * 64 + b"G", b"ACGTTGCATGCAAGCTTAGCTAGGATCCGATCGATTACGAT" * 4] * 3
    out += [b"ACGTTGCATGCAAGCTTAGCTAGGATCCGATCGATTACGATACGTTGCATGCAAGCTTAGCTAGGATC"] * 300   # saturation
    return out


@pytest.fixture(scope="module")
def capi():
    from dbg_assembly_amd import capi as c
    assert c.lib().dbgk_device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback exists)"
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in golden_cases() if c["name"] != "enlarge_cap_e1"],
                         ids=[c["name"] for c in golden_cases() if c["name"] != "enlarge_cap_e1"])
def test_wide_engine_at_k_le_32_equals_the_reference_goldens(capi, oracle, case):
    p, ref = case["params"], case["ref"]
    files = case_reads(case, oracle)
    with capi.Graph(k=p["k"], table_slots=ref["size"], max_read_len=p["max_read_len"], engine=capi.ENGINE_WIDE) as g:
        for bases, offsets in files:
            g.push_reads(bases, offsets)
        st = g.finalize()
        assert (st.total_reads, st.total_kmers, st.count) == (ref["reads"], ref["kmers"], ref["count"])
        narrow = as_narrow(g.wide_export_sorted(), oracle)
        assert dump_sha256(narrow, st.total_reads, st.total_kmers, st.count) == case["dump_sha256"]
        assert g.digest() == oracle.nodes_digest(narrow)   # the 64-bit engines' digest
        array, flags = g.wide_export_host_table()
        assert oracle.wide_check_host_table(array, flags, ref["size"], st.count) == 0
        # with every high word 0 the layout is a valid REFERENCE table (slot = hash_code(kmer) % size)
        assert oracle.check_host_table(as_narrow(array, oracle), flags, ref["size"], st.count) == 0
        assert list(g.link_stats(2).depth_stat) == list(oracle.link_stats(narrow, 2).depth_stat)


@pytest.mark.gpu
@pytest.mark.parametrize("k,r", [(63, 250), (33, 250), (47, 100), (62, 64), (63, 63), (32, 250), (17, 100), (1, 250)])
def test_wide_engine_equals_cpu_restatement_PARITY_UNPINNED_above_k32(capi, oracle, k, r):
    rng = random.Random(k * 1000 + r)
    reads = _reads(rng, 1500)
    rng.shuffle(reads)
    bases, offsets = oracle.pack_reads(reads)
    want, total = oracle.wide_build(bases, offsets, k, r)
    size = capi.find_next_prime_ref(max(2 * len(want), 1000))
    with capi.Graph(k=k, table_slots=size, max_read_len=r, engine=capi.ENGINE_WIDE, max_batch_bases=1 << 16) as g:
        half = len(reads) // 2
        g.push_reads(bases[:int(offsets[half])], offsets[:half + 1])
        g.push_reads(bases[int(offsets[half]):], offsets[half:] - offsets[half])
        st = g.finalize()
        assert (int(st.total_reads), int(st.total_kmers), int(st.count)) == (len(reads), total, len(want))
        got = g.wide_export_sorted()
        assert np.array_equal(got, want)
        assert g.digest() == oracle.wide_digest(want)
        array, flags = g.wide_export_host_table()
        assert oracle.wide_check_host_table(array, flags, size, st.count) == 0
        occ = np.unpackbits(flags)[:size].astype(bool)
        assert np.array_equal(np.sort(array[occ], order=["kmer_hi", "kmer_lo"]), want)
        if k > 32:
            assert want["kmer_hi"].any() and ((want["kmer_lo"] == 0) & (want["kmer_hi"] != 0)).any()  # the side-table keys are exercised
        # the handle is reusable; 64-bit-only entry points refuse it
        with pytest.raises(capi.DbgkError):
            g.export_sorted()
        g.reset()
        g.push_reads(bases, offsets)
        assert g.finalize().count == len(want)
        assert g.digest() == oracle.wide_digest(want)


@pytest.mark.gpu
def test_wide_engine_cfg5_shaped_sample_PARITY_UNPINNED(capi, oracle):
    """BASELINE cfg5's shape at a size the CPU restatement finishes in seconds: 150-base reads at 0.1 % substitutions,
    k = 63 (88 windows per read); device-generated reads == the oracle's generator"""
    n_reads, G = 60000, 300000
    P, PO = capi.synth_params(G, 150, sub_rate=0.001, cfg=5), oracle.synth_params(G, 150, sub_rate=0.001, cfg=5)
    bases, offsets = oracle.synth_reads(PO, 0, n_reads)
    want, total = oracle.wide_build(bases, offsets, 63, 250)
    size = capi.find_next_prime_ref(3 * len(want))
    with capi.Graph(k=63, table_slots=size, engine=capi.ENGINE_WIDE) as g:
        d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
        assert np.array_equal(d_bases.to_host(np.uint8, nb), bases)
        g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
        st = g.finalize()
        assert (int(st.total_kmers), int(st.stored_kmers), int(st.count)) == (total, 88 * n_reads, len(want))
        assert np.array_equal(g.wide_export_sorted(), want)
        d_bases.free()
        d_off.free()


@pytest.mark.gpu
@pytest.mark.parametrize("k,n_parts", [(63, 3), (47, 2), (31, 4)])
def test_wide_nodes_shipped_to_their_owners_PARITY_UNPINNED_above_k32(capi, oracle, k, n_parts):
    """several GPUs with 128-bit keys (here: handles on the one GPU of the box): reads shard by record, every handle
    builds the graph of its share, ships each node to its owner (hash128 >> 32) % n and the owners add the counters up
    (dbgk_wide_partition_export / dbgk_wide_merge_nodes); the union of the owners' tables == the graph of all reads"""
    rng = random.Random(k * 10 + n_parts)
    reads = _reads(rng, 2500)
    rng.shuffle(reads)
    bases, offsets = oracle.pack_reads(reads)
    want, total = oracle.wide_build(bases, offsets, k, 250)
    size = capi.find_next_prime_ref(3 * len(want))
    mk = lambda: capi.Graph(k=k, table_slots=size, engine=capi.ENGINE_WIDE)
    sources, owners = [mk() for _ in range(n_parts)], [mk() for _ in range(n_parts)]
    try:
        for i, g in enumerate(sources):
            g.push_reads(*oracle.pack_reads(reads[i::n_parts]))
            g.finalize()
        for g in sources:
            counts = g.wide_partition_export(n_parts)
            assert int(counts.sum()) == g.stats.count
            buf = g.malloc(int(counts.sum()) * 32)
            assert np.array_equal(g.wide_partition_export(n_parts, buf.ptr, int(counts.sum())), counts)
            first = 0
            for p in range(n_parts):
                owners[p].wide_merge_nodes(buf.ptr + first * 32, int(counts[p]))
                first += int(counts[p])
            for o in owners:
                o.sync()
            buf.free()
        got, key0_seen = [], 0
        for p, o in enumerate(owners):
            st = o.finalize()
            nodes = o.wide_export_sorted()
            if p:   # every handle reports a key-0 node; the real one was merged onto owner 0
                assert nodes[0]["kmer_hi"] == 0 and nodes[0]["kmer_lo"] == 0 and nodes[0]["l_link"] == 0 and nodes[0]["r_link"] == 0
                nodes = nodes[1:]
            got.append(nodes)
        got = np.sort(np.concatenate(got), order=["kmer_hi", "kmer_lo"])
        assert np.array_equal(got, want)
        assert sum(o.digest() for o in owners[:1]) % (1 << 64) != 0
    finally:
        for g in sources + owners:
            g.close()


# ---- WIDE through radix-partitioned records (dbgk_wide_partition.h): expected_kmers > 0, table of >= 2^26 slots --------

M64 = (1 << 64) - 1


def _hash_code(k):
    """kmerSet.h:105-116 on Python integers"""
    k = (k + (~(k << 32) & M64)) & M64
    k ^= k >> 22
    k = (k + (~(k << 13) & M64)) & M64
    k ^= k >> 8
    k = (k + (k << 3)) & M64
    k ^= k >> 15
    k = (k + (~(k << 27) & M64)) & M64
    k ^= k >> 31
    return k


def _hash_code_inverse(h):
    inv = lambda a: pow(a & M64, -1, 1 << 64)
    h ^= h >> 31; h ^= h >> 62
    h = ((h + 1) * inv(1 - (1 << 27))) & M64
    h ^= h >> 15; h ^= h >> 30; h ^= h >> 60
    h = (h * inv(9)) & M64
    h ^= h >> 8; h ^= h >> 16; h ^= h >> 32
    h = ((h + 1) * inv(1 - (1 << 13))) & M64
    h ^= h >> 22; h ^= h >> 44
    h = ((h + 1) * inv(1 - (1 << 32))) & M64
    return h


def _kmer_string(value, k):
    return "".join("ACGT"[(value >> (2 * (k - 1 - i))) & 3] for i in range(k))


def _is_canonical(s):
    return s <= "".join(COMP[c] for c in reversed(s))


def test_python_hash_code_inverse():
    for x in (0, 1, 0x0123456789ABCDEF, M64, 488296166657017542):
        assert _hash_code_inverse(_hash_code(x)) == x


def _crafted_reads(size, rng):
    """reads of exactly k bases (one window each, no neighbours) whose keys stress the region build:
    k = 63: pairs of DIFFERENT keys with the SAME hash128 -- (hi1, lo1) and (hi2, lo1 ^ hash_code(hi1) ^ hash_code(hi2)) --
            so that the second half of their records is identical and only the high word tells them apart;
    k = 32: many keys whose home slots are the last two of one 2048-slot region (they probe past the region's end:
            the spill list) and of the table's last, partial region (they wrap around to slot 0)"""
    pairs = []
    while len(pairs) < 40:
        hi1, hi2, lo1 = rng.getrandbits(62) | 1, rng.getrandbits(62) | 1, rng.getrandbits(64) | 1
        lo2 = lo1 ^ _hash_code(hi1) ^ _hash_code(hi2)
        a, b = _kmer_string((hi1 << 64) | lo1, 63), _kmer_string((hi2 << 64) | lo2, 63)
        if hi1 != hi2 and lo2 != 0 and _is_canonical(a) and _is_canonical(b):
            pairs += [a, b]
    tails = []
    for home in [5 * 2048 + 2046, 5 * 2048 + 2047, size - 2, size - 1]:
        found = 0
        q = rng.getrandbits(20)
        while found < 70:
            q += 1
            s = _kmer_string(_hash_code_inverse(q * size + home), 32)
            if _is_canonical(s) and "A" * 16 != s[16:]:
                tails.append(s)
                found += 1
    return [p.encode() for p in pairs], [t.encode() for t in tails]


@pytest.mark.gpu
@pytest.mark.parametrize("k,r,store", [(63, 250, "ample"), (33, 250, "ample"), (47, 100, "small"), (32, 250, "ample"), (17, 100, "heavy"), (63, 63, "heavy")])
def test_wide_records_path_equals_cpu_restatement_PARITY_UNPINNED_above_k32(capi, oracle, k, r, store):
    """WIDE handle with a known input size: 16-byte records, two scatter levels, 2048-slot regions built in LDS.
    ample: everything becomes records; small: the store takes the first batches, the table is built when it is full and
    the rest joins through the atomic kernels; heavy: 3000 copies of one read -- far more records than a final bucket
    holds (its capacity follows the MEAN fill), so most of them overflow into the observation list"""
    rng = random.Random(k * 77 + r)
    reads = _reads(rng, 2500)
    if store == "heavy":
        reads += [b"GATTACAGATTACACCGGTTAACCGGTTTTGACGTCAGCATGCATGCATCGATCGATCGGCTAGCTAGGCT"] * 3000
    rng.shuffle(reads)
    bases, offsets = oracle.pack_reads(reads)
    want, total = oracle.wide_build(bases, offsets, k, r)
    windows = sum(max(0, min(len(x), r) - k + 1) for x in reads)
    size = capi.find_next_prime_ref((1 << 26) + 12345)
    expected = {"ample": windows + 1000, "small": windows // 3, "heavy": windows + 1000}[store]
    with capi.Graph(k=k, table_slots=size, max_read_len=r, engine=capi.ENGINE_WIDE, max_batch_bases=1 << 16, expected_kmers=expected) as g:
        assert g.store_room() == (0, expected)   # the record path is in use
        g.push_reads(bases, offsets)
        pending, cap = g.store_room()
        assert (cap == 0) == (store == "small")  # small: already built, the handle has gone over to the atomic kernels
        st = g.finalize()
        assert (int(st.total_reads), int(st.total_kmers), int(st.count)) == (len(reads), total, len(want))
        assert np.array_equal(g.wide_export_sorted(), want)
        assert g.digest() == oracle.wide_digest(want)
        if k == 63 and store == "ample":
            array, flags = g.wide_export_host_table()
            assert oracle.wide_check_host_table(array, flags, size, st.count) == 0
        g.reset()   # reusable: records again
        assert g.store_room() == (0, expected)
        half = len(reads) // 2
        g.push_reads(bases[:int(offsets[half])], offsets[:half + 1])
        g.push_reads(bases[int(offsets[half]):], offsets[half:] - offsets[half])
        assert g.finalize().count == len(want)
        assert g.digest() == oracle.wide_digest(want)


@pytest.mark.gpu
def test_wide_records_path_crafted_keys_PARITY_UNPINNED(capi, oracle):
    size = capi.find_next_prime_ref((1 << 26) + 999)
    rng = random.Random(4242)
    pairs, tails = _crafted_reads(size, rng)
    for k, reads in ((63, pairs * 3 + _reads(rng, 300)), (32, tails * 2 + _reads(rng, 300))):
        bases, offsets = oracle.pack_reads(reads)
        want, total = oracle.wide_build(bases, offsets, k, 250)
        with capi.Graph(k=k, table_slots=size, engine=capi.ENGINE_WIDE, expected_kmers=200000) as g:
            assert g.store_room()[1] == 200000
            g.push_reads(bases, offsets)
            st = g.finalize()
            assert int(st.count) == len(want)
            assert np.array_equal(g.wide_export_sorted(), want)
            array, flags = g.wide_export_host_table()
            assert oracle.wide_check_host_table(array, flags, size, st.count) == 0
            if k == 32:   # the crafted keys really crowd the end of their regions: some sit outside them now
                occ = np.unpackbits(flags)[:size].astype(bool)
                assert occ[6 * 2048:6 * 2048 + 8].any() and occ[:8].any()


@pytest.mark.gpu
def test_wide_records_path_cfg5_shaped_sample_PARITY_UNPINNED(capi, oracle):
    n_reads, G = 60000, 300000
    P, PO = capi.synth_params(G, 150, sub_rate=0.001, cfg=5), oracle.synth_params(G, 150, sub_rate=0.001, cfg=5)
    bases, offsets = oracle.synth_reads(PO, 0, n_reads)
    want, total = oracle.wide_build(bases, offsets, 63, 250)
    size = capi.find_next_prime_ref(1 << 27)
    with capi.Graph(k=63, table_slots=size, engine=capi.ENGINE_WIDE, expected_kmers=n_reads * 150) as g:
        d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
        g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
        assert g.store_room()[0] > 0
        st = g.finalize()
        assert (int(st.total_kmers), int(st.stored_kmers), int(st.count)) == (total, 88 * n_reads, len(want))
        assert np.array_equal(g.wide_export_sorted(), want)
        d_bases.free()
        d_off.free()


@pytest.mark.gpu
@pytest.mark.parametrize("k,L,plain", [(63, 150, 0), (63, 150, 1), (63, 64, 0), (63, 63, 0), (47, 100, 0), (33, 250, 0), (33, 250, 1), (32, 150, 0), (17, 150, 0),
                                       (5, 36, 0)])
def test_wide_records_path_equal_length_reads_PARITY_UNPINNED_above_k32(capi, oracle, monkeypatch, k, L, plain):
    """batches of equal-length reads take k_wide_scatter_l1_uniform (lanes mapped to chunks of 8 valid windows, bases
    funnelled out of an LDS-packed byte range) unless k is so small against L that the flat kernel wastes nothing; its pipelined
    form (copy-out of a tile inside the next tile's positions) where a tile's packed words fit beside the stage buffer -- (63, 150),
    (33, 250), (32, 150) -- and the plain form elsewhere or when the hook asks for it"""
    if plain:
        from helpers import set_hooks
        set_hooks(monkeypatch, wide_l1_plain=1)
    rng = random.Random(k * 1000 + L)
    G = 20000
    g = "".join(rng.choice("ACGT") for _ in range(G))
    reads = []
    for _ in range(3000):
        s = rng.randint(0, G - L)
        r = list(g[s:s + L])
        if rng.random() < 0.5:
            r = [COMP[c] for c in reversed(r)]
        for j in range(L):
            x = rng.random()
            if x < 0.01:
                r[j] = rng.choice("ACGTNn")
            elif x < 0.02:
                r[j] = r[j].lower()
        reads.append("".join(r).encode())
    reads += [b"A" * L] * 50 + [b"T" * L] * 20 + [(b"C" + b"A" * 200)[:L]] * 7 + [(b"ACGTTGCATGCAAGCTTAGCTAGGATCCGATCGATTACGAT" * 8)[:L]] * 300
    rng.shuffle(reads)
    bases, offsets = oracle.pack_reads(reads)
    want, total = oracle.wide_build(bases, offsets, k, 250)
    size = capi.find_next_prime_ref((1 << 26) + 777)
    with capi.Graph(k=k, table_slots=size, engine=capi.ENGINE_WIDE, expected_kmers=len(bases), max_batch_bases=200000) as g:
        for a in range(0, len(reads), 1000):   # several batches, each of equal-length reads
            b = min(a + 1000, len(reads))
            g.push_reads(bases[int(offsets[a]):int(offsets[b])], offsets[a:b + 1] - offsets[a])
        st = g.finalize()
        assert (int(st.total_reads), int(st.total_kmers), int(st.count)) == (len(reads), total, len(want))
        assert np.array_equal(g.wide_export_sorted(), want)
        assert g.digest() == oracle.wide_digest(want)


@pytest.mark.gpu
def test_wide_full_size_cfg5_share_records_equal_atomics_PARITY_UNPINNED(capi):
    """BASELINE cfg5 at the size bench.py runs it (one GPU's eighth: 75 M x 150 bp, k = 63, 6.6 G k-mers, 1.6 G slots):
    no CPU restatement can follow here, so the check is that the two forms of the engine -- fused atomics and
    partitioned records, each equal to the restatement at small sizes -- arrive at the same node multiset (count,
    order-independent digest, link-depth histogram), and that the record path repeats it on a reused handle"""
    n_reads, G, slots = 75_000_000, 375_000_000, 1_600_000_000
    P = capi.synth_params(G, 150, sub_rate=0.001, cfg=5)
    size = capi.find_next_prime_ref(slots)
    got = {}
    for name, expected in (("atomic", 0), ("records", n_reads * 88)):
        with capi.Graph(k=63, table_slots=size, engine=capi.ENGINE_WIDE, expected_kmers=expected) as g:
            d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
            for rep in range(2 if name == "records" else 1):
                g.reset()
                g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
                st = g.finalize()
                res = (int(st.count), int(st.stored_kmers), int(st.total_kmers), g.digest(), [int(x) for x in g.link_stats(2).depth_stat])
                assert got.setdefault(name, res) == res
            d_bases.free()
            d_off.free()
    assert got["atomic"][1] == n_reads * 88
    assert got["atomic"] == got["records"]
