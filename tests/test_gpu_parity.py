"""GPU (-m gpu): the HIP path behind the C ABI against the oracle, the golden fixtures made by the
real reference, and size-independent properties.  Bit-exact everywhere (integer work)."""
import random

import numpy as np
import pytest

from conftest import golden_cases, golden_case_ids
from helpers import case_reads, dump_sha256, push_with_reference_schedule, set_hooks

pytestmark = pytest.mark.gpu

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


@pytest.fixture(scope="module")
def capi():
    from dbg_assembly_amd import capi as c
    assert c.lib().dbgk_device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback exists)"
    return c


def rand_reads(rng, n, G=5000, L=150, var_len=True):
    g = "".join(rng.choice("ACGT") for _ in range(G))
    out = []
    for _ in range(n):
        ln = L
        if var_len and rng.random() < 0.3:
            ln = rng.randint(0, L + 150)
        s = rng.randint(0, G - ln)
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
    out += [b"A" * 200, b"T" * 77, b"", b"ACGT", b"a" * 31, b"C" * 64]
    return out


@pytest.mark.parametrize("k,r", [(31, 250), (17, 100), (32, 250), (1, 250), (16, 120), (4, 33), (21, 21)])
def test_extract_kernel_equals_oracle(capi, oracle, k, r):
    rng = random.Random(k * 1000 + r)
    reads = rand_reads(rng, 400)
    bases, offsets = oracle.pack_reads(reads)
    with capi.Graph(k=k, table_slots=1009, max_read_len=r) as g:
        kmer, left, right, valid = g.extract_kmers(bases, offsets)
    for i, seq in enumerate(reads):
        s = int(offsets[i])
        km, lb, rb = oracle.parse_read(seq, k, r)
        n = len(km)
        assert valid[s:s + len(seq)].sum() == n, (i, seq)
        assert valid[s:s + n].all()
        assert np.array_equal(kmer[s:s + n], km), i
        assert np.array_equal(left[s:s + n], lb), i
        assert np.array_equal(right[s:s + n], rb), i


def _build(capi, files, k, r, size, engine=0, batch_bases=0):
    g = capi.Graph(k=k, table_slots=size, max_read_len=r, engine=engine, max_batch_bases=batch_bases)
    for bases, offsets in files:
        g.push_reads(bases, offsets)
    st = g.finalize()
    return g, st


@pytest.mark.parametrize("case", golden_cases(), ids=golden_case_ids())
def test_golden_cases(capi, oracle, case):
    """Same inputs as the real reference saw -> identical canonical dump, totals and a valid
    host-layout table."""
    p, ref = case["params"], case["ref"]
    files = case_reads(case, oracle)
    if case["name"] == "enlarge_cap_e1":  # the reference drops the rest of a file at the -e cap (DBGgraph.cpp:346-350)
        g = capi.Graph(k=p["k"], table_slots=ref["size"], max_read_len=p["max_read_len"])
        assert push_with_reference_schedule(g, files, p, capi) == ref["size"]
        st = g.finalize()
    else:
        g, st = _build(capi, files, p["k"], p["max_read_len"], ref["size"])
    try:
        assert (st.total_reads, st.total_kmers, st.count) == (ref["reads"], ref["kmers"], ref["count"])
        nodes = g.export_sorted()
        assert dump_sha256(nodes, st.total_reads, st.total_kmers, st.count) == case["dump_sha256"]
        # host table at the reference's size: valid probe layout, same node multiset
        array, flags = g.export_host_table()
        assert oracle.check_host_table(array, flags, ref["size"], st.count) == 0
        # and at a different size (device-side re-seat)
        other = capi.find_next_prime_ref(2 * ref["size"])
        array2, flags2 = g.export_host_table(other)
        assert oracle.check_host_table(array2, flags2, other, st.count) == 0
        occ = np.unpackbits(flags2)[:other].astype(bool)
        got = np.sort(array2[occ], order="kmer")
        assert np.array_equal(got, nodes)
        assert g.digest() == oracle.nodes_digest(nodes)
    finally:
        g.close()


@pytest.mark.parametrize("k", [31, 17])
def test_synthetic_200k_reads_vs_oracle(capi, oracle, k):
    """cfg1-sized job: device-generated reads == host generator; table == oracle; stats == oracle"""
    n_reads, G = 200000, 1000000
    P, PO = capi.synth_params(G, 150, cfg=1), oracle.synth_params(G, 150, cfg=1)
    bases, offsets = oracle.synth_reads(PO, 0, n_reads)
    size = capi.find_next_prime_ref(20000000)
    with capi.Graph(k=k, table_slots=size) as g:
        d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
        assert np.array_equal(d_bases.to_host(np.uint8, nb), bases)
        assert np.array_equal(d_off.to_host(np.uint64), offsets)
        g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
        st = g.finalize()
        nodes = g.export_sorted()
        ls = g.link_stats(2)
        dig = g.digest()
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=k, init_hash_size=0.02, threads=1)
    assert (st.total_reads, st.total_kmers, st.stored_kmers, st.count) == \
           (ref.total_reads, ref.total_kmers, ref.total_kmers, ref.count)
    assert np.array_equal(nodes, ref.nodes)
    assert dig == oracle.nodes_digest(ref.nodes)
    rs = oracle.link_stats(ref.nodes, 2)
    assert list(ls.depth_stat) == list(rs.depth_stat)
    assert (ls.total_nodes, ls.deleted_lowfreq, ls.linear_nodes, ls.tip_nodes, ls.branch_nodes) == \
           (rs.total_nodes, rs.deleted_lowfreq, rs.linear_nodes, rs.tip_nodes, rs.branch_nodes)


def test_batching_and_order_invariance(capi, oracle):
    """host pushes in many small batches (forces the double-buffered staging to wrap) and in
    reversed read order give the same node multiset"""
    rng = random.Random(7)
    reads = rand_reads(rng, 3000, G=20000)
    b1, o1 = oracle.pack_reads(reads)
    b2, o2 = oracle.pack_reads(reads[::-1])
    size = capi.find_next_prime_ref(1000000)
    g1, s1 = _build(capi, [(b1, o1)], 31, 250, size)
    g2, s2 = _build(capi, [(b2, o2)], 31, 250, size, batch_bases=4096)
    g3 = capi.Graph(k=31, table_slots=size, max_batch_bases=1 << 16)
    for lo in range(0, len(reads), 500):
        bb, oo = oracle.pack_reads(reads[lo:lo + 500])
        g3.push_reads(bb, oo)
    s3 = g3.finalize()
    try:
        n1 = g1.export_sorted()
        assert np.array_equal(n1, g2.export_sorted()) and np.array_equal(n1, g3.export_sorted())
        assert s1.count == s2.count == s3.count and s1.total_kmers == s2.total_kmers == s3.total_kmers
        ref = oracle.build_graph(files_mem=[(b1, o1)], k=31, init_hash_size=0.001)
        assert np.array_equal(n1, ref.nodes)
    finally:
        g1.close(), g2.close(), g3.close()


def test_shard_merge_equals_whole(capi, oracle):
    """multi-GPU building blocks on one GPU: two shards, partition by owner, merge == whole job
    (saturating add is exact under any split)"""
    rng = random.Random(11)
    reads = rand_reads(rng, 2000, G=8000) + [b"A" * 150] * 400 + [reads_ for reads_ in [b"ACGTTGCA" * 20] * 300]
    rng.shuffle(reads)
    size = capi.find_next_prime_ref(600000)
    whole, sw = _build(capi, [oracle.pack_reads(reads)], 31, 250, size)
    a, sa = _build(capi, [oracle.pack_reads(reads[:1300])], 31, 250, size)
    b, sb = _build(capi, [oracle.pack_reads(reads[1300:])], 31, 250, size)
    try:
        n_parts = 3
        cb = b.partition_counts(n_parts)
        assert int(cb.sum()) == sb.count
        buf = b.malloc(int(cb.sum()) * 16)
        b.partition_export(n_parts, buf.ptr, int(cb.sum()))
        nodes_b = buf.to_host(capi.NODE_DTYPE)
        # every node sits in its owner's range
        h = np.array([oracle.lib().orc_hash_code(int(x)) for x in nodes_b["kmer"][:2000]], dtype=np.uint64)
        bounds = np.concatenate([[0], np.cumsum(cb)]).astype(np.int64)
        part_of = np.searchsorted(bounds, np.arange(2000), side="right") - 1
        own = ((h >> np.uint64(32)) % np.uint64(n_parts)).astype(np.int64)
        own[nodes_b["kmer"][:2000] == 0] = 0
        assert np.array_equal(own, part_of[:len(own)])
        # merge all parts of b into a -> equals the whole job
        a.merge_nodes(buf.ptr, int(cb.sum()))
        sm = a.refresh_stats()
        assert sm.count == sw.count
        assert np.array_equal(a.export_sorted(), whole.export_sorted())
        assert a.digest() == whole.digest()
        buf.free()
    finally:
        whole.close(), a.close(), b.close()


def test_table_full_is_an_error_not_a_hang(capi, oracle):
    rng = random.Random(3)
    reads = rand_reads(rng, 300, var_len=False)
    g = capi.Graph(k=31, table_slots=1009)
    g.push_reads(*oracle.pack_reads(reads))
    with pytest.raises(capi.DbgkError) as ei:
        g.finalize()
    assert ei.value.status == capi.ERR_TABLE_FULL
    g.close()


def test_state_errors(capi, oracle):
    g = capi.Graph(k=31, table_slots=100003)
    with pytest.raises(capi.DbgkError):
        g.stats = capi.Stats()
        g.export_host_table()
    g.push_reads(*oracle.pack_reads([b"ACGT" * 40]))
    g.finalize()
    with pytest.raises(capi.DbgkError):
        g.push_reads(*oracle.pack_reads([b"ACGT" * 40]))
    g.reset()
    g.push_reads(*oracle.pack_reads([]))
    st = g.finalize()
    assert st.count == 1 and st.total_reads == 0  # only the key-0 node (DBGgraph.cpp:418)
    nodes = g.export_sorted()
    assert nodes.tolist() == [(0, 0, 0)]
    g.close()


def test_full_size_properties_cfg2_sample(capi):
    """BASELINE cfg2-shaped input at 1/5 size (2 M reads, 240 M k-mers): properties that need no
    oracle -- shard+merge == whole, re-run determinism of the digest, stored_kmers = 120/read."""
    n_reads, G = 2000000, 10000000
    P = capi.synth_params(G, 150, cfg=2)
    size = capi.find_next_prime_ref(150000000)
    with capi.Graph(k=31, table_slots=size) as g, capi.Graph(k=31, table_slots=size) as g2:
        d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
        g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
        st = g.finalize()
        assert st.stored_kmers == 120 * n_reads == st.total_kmers
        dig = g.digest()
        ls = g.link_stats(2)
        assert sum(ls.depth_stat) == 8 * st.count
        # two halves on a second handle + merge
        half = n_reads // 2
        e_bases, e_off, eb = g2.synth_reads_device(P, 0, half)
        g2.push_reads_device(e_bases.ptr, e_off.ptr, half, eb)
        g2.finalize()
        g.reset()
        f_bases, f_off, fb = g.synth_reads_device(P, half, n_reads - half)
        g.push_reads_device(f_bases.ptr, f_off.ptr, n_reads - half, fb)
        sb = g.finalize()
        cnt = g.partition_counts(1)
        buf = g.malloc(int(cnt[0]) * 16)
        g.partition_export(1, buf.ptr, int(cnt[0]))
        g2.merge_nodes(buf.ptr, int(cnt[0]))
        sm = g2.refresh_stats()
        assert sm.count == st.count and sb.count < st.count
        assert g2.digest() == dig
        for b in (d_bases, d_off, e_bases, e_off, f_bases, f_off, buf):
            b.free()


# ------------------------------------------------------------------------------------------------
# PARTITION engine (records -> radix partition by final slot range -> LDS-built regions)
# ------------------------------------------------------------------------------------------------
PART_SLOTS = 70000000  # the engine needs >= 2^26 slots (8-byte records)


@pytest.mark.parametrize("case", golden_cases(), ids=golden_case_ids())
def test_partition_engine_golden_cases(capi, oracle, case):
    p, ref = case["params"], case["ref"]
    files = case_reads(case, oracle)
    size = capi.find_next_prime_ref(PART_SLOTS)
    n_bases = sum(int(o[-1]) for _, o in files)
    g = capi.Graph(k=p["k"], table_slots=size, max_read_len=p["max_read_len"], engine=capi.ENGINE_PARTITION,
                   expected_kmers=max(n_bases, 1))
    try:
        if case["name"] == "enlarge_cap_e1":  # block by block with a flush after each: the reference's -e cap schedule
            assert push_with_reference_schedule(g, files, p, capi) == ref["size"]
        else:
            for bases, offsets in files:
                g.push_reads(bases, offsets)
        st = g.finalize()
        assert (st.total_reads, st.total_kmers, st.count) == (ref["reads"], ref["kmers"], ref["count"])
        nodes = g.export_sorted()
        assert dump_sha256(nodes, st.total_reads, st.total_kmers, st.count) == case["dump_sha256"]
        array, flags = g.export_host_table()
        assert oracle.check_host_table(array, flags, size, st.count) == 0  # valid linear-probing layout
    finally:
        g.close()


def test_partition_engine_equals_direct_and_oracle(capi, oracle):
    n_reads, G = 200000, 1000000
    P, PO = capi.synth_params(G, 150, cfg=1), oracle.synth_params(G, 150, cfg=1)
    bases, offsets = oracle.synth_reads(PO, 0, n_reads)
    size = capi.find_next_prime_ref(PART_SLOTS)
    with capi.Graph(k=31, table_slots=size, engine=capi.ENGINE_PARTITION, expected_kmers=n_reads * 150) as g:
        d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
        # two pushes (records accumulate), then one finalize
        half = n_reads // 2
        g.push_reads_device(d_bases.ptr, d_off.ptr, half, half * 150)
        h_off = np.arange(n_reads - half + 1, dtype=np.uint64) * np.uint64(150)
        d_off2 = g.malloc(h_off.nbytes)
        d_off2.from_host(h_off)
        g.push_reads_device(d_bases.ptr + half * 150 - (half * 150) % 16, d_off2.ptr, 0, 0)  # empty push is harmless
        tail = g.malloc((n_reads - half) * 150 + 64)
        tail.from_host(bases[half * 150:])
        g.push_reads_device(tail.ptr, d_off2.ptr, n_reads - half, (n_reads - half) * 150)
        st = g.finalize()
        nodes = g.export_sorted()
        dig = g.digest()
        ls = g.link_stats(2)
        array, flags = g.export_host_table()
        assert oracle.check_host_table(array, flags, size, st.count) == 0
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, init_hash_size=0.02, threads=1)
    assert (st.total_reads, st.total_kmers, st.count) == (ref.total_reads, ref.total_kmers, ref.count)
    assert np.array_equal(nodes, ref.nodes)
    assert dig == oracle.nodes_digest(ref.nodes)
    assert list(ls.depth_stat) == list(oracle.link_stats(ref.nodes, 2).depth_stat)


@pytest.mark.parametrize("store,explicit_flush", [(50000, False), (1 << 22, True), (7, False)])
def test_partition_engine_streams_input_of_unknown_size(capi, oracle, store, explicit_flush):
    """Record store much smaller than the input (expected_kmers = store): pushes flush on their own when the
    store is full -- regions that already hold nodes are loaded back into LDS, the new records merged in and
    the region written out again.  Explicit dbgk_flush between pushes, exact counts in between, a device-side
    resize half way (new bucket geometry), heavy poly-A / repeat content (key-0 side node, saturation)."""
    rng = random.Random(store)
    reads = rand_reads(rng, 6000, G=40000) + [b"A" * 150] * 300 + [b"ACGTTGCATGCAAGCTTAGCTAGGATCCGATCGATTACGAT" * 3] * 400
    rng.shuffle(reads)
    bases, offsets = oracle.pack_reads(reads)
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, init_hash_size=0.001)
    size = capi.find_next_prime_ref(PART_SLOTS)
    with capi.Graph(k=31, table_slots=size, engine=capi.ENGINE_PARTITION, expected_kmers=store, max_batch_bases=1 << 16) as g:
        n = len(reads)
        cuts = [0, n // 5, n // 2, n // 2 + 1, (4 * n) // 5, n]
        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            lo, hi = int(offsets[a]), int(offsets[b])
            g.push_reads(bases[lo:hi], offsets[a:b + 1] - offsets[a])
            if explicit_flush:
                g.flush()
                assert g.store_room()[0] == 0
                part = oracle.build_graph(files_mem=[(bases[:hi], offsets[:b + 1])], k=31, init_hash_size=0.001)
                assert g.refresh_stats().count == part.count  # exact after a flush
            if i == 2:
                g.resize_table(capi.find_next_prime_ref(2 * PART_SLOTS + 12345))
        st = g.finalize()
        nodes = g.export_sorted()
        array, flags = g.export_host_table()
        assert oracle.check_host_table(array, flags, g.table_slots, st.count) == 0
    assert (st.total_reads, st.total_kmers, st.count) == (ref.total_reads, ref.total_kmers, ref.count)
    assert np.array_equal(nodes, ref.nodes)


@pytest.mark.parametrize("pieces", [2, 6, 40])
@pytest.mark.parametrize("store_pieces", [0, 3])
def test_early_level2_rounds_between_the_batches_equal_oracle(capi, oracle, monkeypatch, pieces, store_pieces):
    """EARLY level 2 (dbgk.hip early_l2): every push first scatters what the batches before it left in the level-1 buckets.  The input
    comes in 2 / 6 / 40 host batches with a round in front of every batch (hook early_l2_min=1), once with all records resident
    and once through a store that holds three batches (flush rounds in between: the incremental build, the counters of taken
    records reset with the store); mixed-length reads, poly-A, a heavy repeat.  Nodes, counts and host layout against the oracle;
    the same input with the rounds switched off (hook early_l2=0) gives the same digest."""
    rng = random.Random(1000 + pieces)
    reads = rand_reads(rng, 9000, G=60000) + [b"A" * 150] * 200 + [b"ACGTTGCATGCAAGCTTAGCTAGGATCCGATCGATTACGAT" * 3] * 300
    rng.shuffle(reads)
    bases, offsets = oracle.pack_reads(reads)
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, init_hash_size=0.001)
    size = capi.find_next_prime_ref(PART_SLOTS)
    n = len(reads)
    per = (n + pieces - 1) // pieces
    total = int(offsets[-1])
    store = total if store_pieces == 0 else store_pieces * (total // pieces) + 4096
    digests = []
    for early in ("1", "0"):
        set_hooks(monkeypatch, early_l2=early, early_l2_min=1)
        with capi.Graph(k=31, table_slots=size, engine=capi.ENGINE_PARTITION, expected_kmers=store, max_batch_bases=1 << 20) as g:
            for a in range(0, n, per):
                b = min(n, a + per)
                lo, hi = int(offsets[a]), int(offsets[b])
                g.push_reads(bases[lo:hi], offsets[a:b + 1] - offsets[a])
            st = g.finalize()
            tm = g.timings()
            nodes = g.export_sorted()
            digests.append(g.digest())
            array, flags = g.export_host_table()
            assert oracle.check_host_table(array, flags, g.table_slots, st.count) == 0
        assert (st.total_reads, st.total_kmers, st.count) == (ref.total_reads, ref.total_kmers, ref.count)
        assert np.array_equal(nodes, ref.nodes)
        if early == "1" and store_pieces == 0:
            assert tm.partition_launches >= pieces, "no level-2 round ran between the batches"   # one per batch after the first + the last at finalize
    assert digests[0] == digests[1] == oracle.nodes_digest(ref.nodes)


def test_partition_engine_streaming_device_pushes_equal_one_shot(capi):
    """cfg2-shaped reads pushed from device memory in 8 pieces through a store that holds 3 of them:
    digest and counts equal the one-shot build (records of all pieces resident)."""
    n_reads, G, pieces = 800000, 4000000, 8
    P = capi.synth_params(G, 150, cfg=2)
    size = capi.find_next_prime_ref(100000000)
    res = []
    for store in (n_reads * 150, 3 * (n_reads // pieces) * 150):
        with capi.Graph(k=31, table_slots=size, engine=capi.ENGINE_PARTITION, expected_kmers=store) as g:
            bufs = []
            per = n_reads // pieces
            for i in range(pieces):
                d_bases, d_off, nb = g.synth_reads_device(P, i * per, per)
                g.push_reads_device(d_bases.ptr, d_off.ptr, per, nb)
                bufs += [d_bases, d_off]
            st = g.finalize()
            res.append((int(st.count), int(st.stored_kmers), int(st.total_kmers), g.digest(), list(g.link_stats(2).depth_stat)))
            for b in bufs:
                b.free()
    assert res[0] == res[1]


@pytest.mark.parametrize("n_reads", [2_000_000, 2_800_000])
def test_streamed_build_at_high_load_equals_direct_engine(capi, n_reads):
    """the incremental region build where it is stressed: the SMALLEST table the engine takes (2^26 slots) filled to
    ~0.67 / ~0.87 by cfg2-shaped reads in four to six flush rounds -- regions near full, nodes spilling past their
    region's end (merged through the global path, then FOREIGN blockers in the next region's image on every later
    round), long probe chains.  Count, totals, digest and DepthStat must equal the DIRECT engine's (oracle-pinned)."""
    G = 10_000_000
    P = capi.synth_params(G, 150, cfg=2)
    size = capi.find_next_prime_ref(1 << 26)
    res = []
    for engine, store in ((capi.ENGINE_DIRECT, 0), (capi.ENGINE_PARTITION, 60_000_000)):
        with capi.Graph(k=31, table_slots=size, engine=engine, expected_kmers=store) as g:
            pieces, per = 10, n_reads // 10
            bufs = []
            for i in range(pieces):
                d_bases, d_off, nb = g.synth_reads_device(P, i * per, per)
                g.push_reads_device(d_bases.ptr, d_off.ptr, per, nb)
                bufs += [d_bases, d_off]
            st = g.finalize()
            res.append((int(st.count), int(st.total_kmers), int(st.stored_kmers), g.digest(), list(g.link_stats(2).depth_stat)))
            load = st.count / size
            for b in bufs:
                b.free()
    assert res[0] == res[1]
    assert load > (0.6 if n_reads < 2_500_000 else 0.8)


def test_partition_engine_bucket_overflow_goes_through_direct_path(capi, oracle):
    """expected_kmers far too small + one heavily repeated read: bucket capacities overflow and
    the excess records take the global-atomic path; the result must not change"""
    rng = random.Random(21)
    reads = rand_reads(rng, 1500, G=6000) + [b"ACGTTGCATGCAAGCTTAGCTAGGATCCGATCGATTACGAT" * 3] * 3000
    bases, offsets = oracle.pack_reads(reads)
    size = capi.find_next_prime_ref(PART_SLOTS)
    with capi.Graph(k=31, table_slots=size, engine=capi.ENGINE_PARTITION, expected_kmers=1) as g:
        g.push_reads(bases, offsets)
        st = g.finalize()
        nodes = g.export_sorted()
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, init_hash_size=0.001)
    assert st.count == ref.count and np.array_equal(nodes, ref.nodes)


def test_full_size_direct_vs_partition_digest(capi):
    """2 M reads of the cfg2 workload: two independent GPU implementations must agree bit for bit"""
    n_reads, G = 2000000, 10000000
    P = capi.synth_params(G, 150, cfg=2)
    size = capi.find_next_prime_ref(150000000)
    res = []
    for engine in (capi.ENGINE_DIRECT, capi.ENGINE_PARTITION):
        with capi.Graph(k=31, table_slots=size, engine=engine, expected_kmers=n_reads * 150) as g:
            d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
            g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
            st = g.finalize()
            res.append((st.count, st.stored_kmers, g.digest(), list(g.link_stats(2).depth_stat)))
            d_bases.free()
            d_off.free()
    assert res[0] == res[1]


@pytest.mark.parametrize("L,k", [(100, 31), (130, 31), (110, 17)])
def test_ragged_level1_tiles_that_start_in_a_short_reads_empty_tail(capi, oracle, L, k, monkeypatch):
    """RAGGED form of the lane-per-chunk level-1 kernel with a lane count per read (Q) that does not divide
    the 1024 lanes of a tile, and some reads much shorter than C * (Q - 1): a tile can then start in the
    empty tail lanes of a short read, and its byte range must start at the NEXT read (round-1 advisor
    finding: the range started past it and live lanes decoded stale LDS words).  ~93 % full-length reads
    keep the batch inside the ragged gate; checked against the oracle and the DIRECT engine."""
    rng = np.random.default_rng(L * 100 + k)
    n_reads, G = 200000, 400000
    genome = rng.integers(0, 4, G, dtype=np.uint8)
    lens = np.full(n_reads, L, dtype=np.int64)
    short = rng.random(n_reads) < 0.07
    lens[short] = rng.integers(k, 2 * k, int(short.sum()))
    starts = rng.integers(0, G - L, n_reads)
    offsets = np.zeros(n_reads + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    idx = np.repeat(starts - offsets[:-1].astype(np.int64), lens) + np.arange(int(offsets[-1]))
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[genome[idx]].copy()
    size = capi.find_next_prime_ref(PART_SLOTS)
    res = []
    # (since round 4 such a batch takes the prefix form by default -- exact lane counts per read; the hook l1_prefix=0 keeps the ragged form,
    # which stays in the library: both are run)
    for engine, prefix in ((capi.ENGINE_DIRECT, None), (capi.ENGINE_PARTITION, "0"), (capi.ENGINE_PARTITION, "1")):
        if prefix is not None:
            set_hooks(monkeypatch, l1_prefix=prefix)
        with capi.Graph(k=k, table_slots=size, engine=engine, expected_kmers=int(offsets[-1]), max_batch_bases=int(offsets[-1]) + 4096) as g:
            g.push_reads(bases, offsets)   # ONE batch: the tiles of interest need many reads in a row
            st = g.finalize()
            if prefix == "0":
                assert g.timings().uniform_launches == 1, "the batch was expected to take the ragged lane-per-chunk kernel"
            if prefix == "1":
                assert g.timings().prefix_launches == 1, "the batch was expected to take the prefix form"
            res.append((int(st.count), int(st.stored_kmers), g.digest()))
            nodes = g.export_sorted()
    assert res[0] == res[1] == res[2]
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=k, init_hash_size=0.02, threads=1)
    assert res[1][0] == ref.count and np.array_equal(nodes, ref.nodes)


def test_full_size_cfg2_bench_workload_equals_the_cpu_oracle(capi):
    """The FULL BASELINE cfg2 workload with bench.py's exact handle parameters (10 M x 150 bp reads, k = 31,
    600 000 001-slot table, expected_kmers = 120 per read): node count, k-mer totals, order-independent
    digest of the node multiset and DepthStat of both engines == tests/golden/cfg2_full.json, which the CPU
    oracle computed at full size (tests/golden/make_cfg2_full.py)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg2_full.json")
    gold = json.load(open(path))
    n_reads, G = gold["n_reads"], gold["genome_len"]
    P = capi.synth_params(G, 150, cfg=2)
    size = capi.find_next_prime_ref(600000000)
    for engine in (capi.ENGINE_PARTITION, capi.ENGINE_DIRECT):
        with capi.Graph(k=gold["k"], table_slots=size, max_read_len=250, engine=engine,
                        expected_kmers=n_reads * 120 if engine == capi.ENGINE_PARTITION else 0) as g:
            d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
            g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
            st = g.finalize()
            if engine == capi.ENGINE_PARTITION:
                assert g.timings().uniform_launches == 1  # the kernel the bench times
            got = (int(st.total_reads), int(st.total_kmers), int(st.stored_kmers), int(st.count), g.digest(),
                   [int(x) for x in g.link_stats(2).depth_stat])
            d_bases.free()
            d_off.free()
            sha = _sorted_sha256(g)
        assert got == (gold["total_reads"], gold["total_kmers"], gold["total_kmers"], gold["count"], gold["digest"], gold["depth_stat"]), engine
        # "bit-exact kmerSet", literally: the canonical dump -- every node as the reference's 16-byte KmerNode, sorted by kmer -- hashed
        # by the REAL reference at full size (ref_dbg -H, make_cfg2_full.py --ref) against the bytes dbgk_export_sorted returns
        assert sha == gold["sorted_sha256"], engine


def _sorted_sha256(g):
    import hashlib
    nodes = g.export_sorted()
    assert nodes.dtype.itemsize == 16
    h = hashlib.sha256()
    view = nodes.view(np.uint8).reshape(-1)
    for lo in range(0, len(view), 1 << 28):
        h.update(view[lo:lo + (1 << 28)].data)
    return h.hexdigest()


def test_cfg3_generator_5m_reads_equal_the_real_reference(capi):
    """cfg3's read generator (synth cfg = 3) pinned to the reference itself at a size it takes: the first 5 M reads of bench.py's cfg3
    share (125 Mb genome), k = 31 -- counts, digest, DepthStat and the SHA-256 of the sorted 16-byte-node dump as the REAL reference
    produced them (tests/golden/cfg3_pin.json, make_cfg2_full.py --ref-cfg3), both engines.  (The full cfg3 share stays an
    engine-against-engine check, tests/test_gpu_cfg3.py: no CPU takes 25 M reads in a test.)"""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg3_pin.json")))
    n_reads = gold["n_reads"]
    P = capi.synth_params(gold["genome_len"], 150, cfg=3)
    size = capi.find_next_prime_ref(600000000)
    for engine in (capi.ENGINE_PARTITION, capi.ENGINE_DIRECT):
        with capi.Graph(k=gold["k"], table_slots=size, max_read_len=250, engine=engine,
                        expected_kmers=n_reads * 120 if engine == capi.ENGINE_PARTITION else 0) as g:
            d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
            g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
            st = g.finalize()
            got = (int(st.total_reads), int(st.total_kmers), int(st.count), g.digest(), [int(x) for x in g.link_stats(2).depth_stat])
            d_bases.free()
            d_off.free()
            sha = _sorted_sha256(g)
        assert got == (gold["total_reads"], gold["total_kmers"], gold["count"], gold["digest"], gold["depth_stat"]), engine
        assert sha == gold["sorted_sha256"], engine


@pytest.mark.parametrize("slots,force", [(4_290_000_000, None), (600_000_000, "1"), (4_290_000_000, "0")])
def test_full_size_cfg2_reads_into_many_level1_buckets(capi, monkeypatch, slots, force):
    """What every rank of an 8-GPU job runs: cfg2's reads partitioned by the level-1 buckets of a 4.29 G-slot table
    (n1 = 1023), where the equal-length level-1 kernel takes its LINEAR form (8 windows per lane, bucket tags, linear
    copy-out).  The node multiset does not depend on the table size: count, digest and DepthStat == the full-size oracle
    record; the same with the form forced on the bench's own table and forced off on the big one"""
    import json
    import os
    if force is not None:
        set_hooks(monkeypatch, l1_linear=force)
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg2_full.json")))
    n_reads = gold["n_reads"]
    P = capi.synth_params(gold["genome_len"], 150, cfg=2)
    with capi.Graph(k=gold["k"], table_slots=capi.find_next_prime_ref(slots), max_read_len=250, engine=capi.ENGINE_PARTITION,
                    expected_kmers=n_reads * 120) as g:
        d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
        g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
        st = g.finalize()
        assert g.timings().uniform_launches == 1
        got = (int(st.total_reads), int(st.total_kmers), int(st.count), g.digest(), [int(x) for x in g.link_stats(2).depth_stat])
        d_bases.free()
        d_off.free()
    assert got == (gold["total_reads"], gold["total_kmers"], gold["count"], gold["digest"], gold["depth_stat"])


# ------------------------------------------------------------------------------------------------
# sharded table: N handles on ONE GPU stand in for N ranks; the all-to-all is done with in-process
# device copies.  Validates slot-range ownership end to end on real hardware.
# ------------------------------------------------------------------------------------------------
def _sharded_build(capi, oracle, reads, n_shards, expected_per_shard, k=31, slots=PART_SLOTS, want_tables=True, ranged=0):
    size = capi.find_next_prime_ref(slots)
    graphs = [capi.Graph(k=k, table_slots=size, engine=capi.ENGINE_PARTITION, expected_kmers=expected_per_shard,
                         shard_count=n_shards, shard_index=i) for i in range(n_shards)]
    try:
        for i, g in enumerate(graphs):
            g.push_reads(*oracle.pack_reads(reads[i::n_shards]))  # reads shard by record
            g.sync()
        infos = [g.shard_info() for g in graphs]
        assert infos[0].slot_lo == 0 and infos[-1].slot_hi == size
        for a, b in zip(infos[:-1], infos[1:]):
            assert a.slot_hi == b.slot_lo
        for s in range(n_shards):       # the all-to-all: chunk d of rank s -> slot s of rank d's inbox
            for d in range(n_shards):
                if not ranged:
                    graphs[0].memcpy_d2d(infos[d].d_recv + s * infos[d].chunk_bytes, infos[s].d_send + d * infos[s].chunk_bytes,
                                         infos[s].chunk_bytes)
                graphs[0].memcpy_d2d(infos[d].d_recv_cnt + s * infos[d].cnt_chunk_bytes,
                                     infos[s].d_send_cnt + d * infos[s].cnt_chunk_bytes, infos[s].cnt_chunk_bytes)
        if ranged:                      # the records travel in `ranged` pieces; every piece is built as soon as it is there
            B = infos[0].buckets_per_rank
            per = -(-B // ranged)
            for g in graphs:
                g.sync()
                g.shard_plan()
            for j0 in range(0, B, per):
                j1 = min(j0 + per, B)
                for s in range(n_shards):
                    for d in range(n_shards):
                        off_d, off_s = s * infos[d].chunk_bytes + j0 * infos[d].bucket_bytes, d * infos[s].chunk_bytes + j0 * infos[s].bucket_bytes
                        graphs[0].memcpy_d2d(infos[d].d_recv + off_d, infos[s].d_send + off_s, (j1 - j0) * infos[s].bucket_bytes)
                graphs[0].sync()
                for g, info in zip(graphs, infos):
                    own0, own1 = min(j0, info.own_buckets), min(j1, info.own_buckets)
                    if own1 > own0:
                        g.shard_build_range(own0, own1)
                        with pytest.raises(capi.DbgkError):
                            g.shard_build_range(own0, own1)  # a range is consumed once, in order
                        g.sync()                             # legal between ranges: waits for both of the handle's streams
        stats = []
        for g in graphs:
            g.shard_mark_exchanged()
            stats.append(g.finalize())
        ovf = [g.shard_overflow() for g in graphs]
        out = [g.shard_outgoing() for g in graphs]
        for (p, n) in ovf:              # overflow observations are offered to every shard, which keeps its own
            for g in graphs:
                if n:
                    g.shard_merge(p, n, is_triple=True)
        for src in graphs:              # ... and, where a rank's overflow list ran full, its side table of aggregated surplus
            p, n = src.shard_heavy()
            for g in graphs:
                if n:
                    g.shard_merge(p, n)
        for s, (p, n) in enumerate(out):  # nodes that ran off the end of shard s continue in shard s+1
            if n:
                graphs[(s + 1) % n_shards].shard_merge(p, n, from_previous_shard=True)
        for s in range(1, n_shards):    # key-0 node lives on shard 0
            if stats[s].polyA_l_link or stats[s].polyA_r_link:
                graphs[0].add_polyA(stats[s].polyA_l_link, stats[s].polyA_r_link)
        final = [g.refresh_stats() for g in graphs]
        for g, (p, n) in zip(graphs, out):
            assert g.shard_outgoing()[1] == n  # nothing handed over twice
        nodes = np.concatenate([g.export_sorted() for g in graphs])
        tables = [g.export_host_table(g.stats.table_slots) for g in graphs] if want_tables else None
        return final, stats, nodes, infos, tables, (sum(n for _, n in ovf), sum(n for _, n in out)), size
    finally:
        for g in graphs:
            g.close()


@pytest.mark.parametrize("n_shards,ranged", [(2, 0), (3, 0), (2, 4), (3, 64)])
def test_sharded_table_equals_oracle(capi, oracle, n_shards, ranged):
    rng = random.Random(31 + n_shards)
    reads = rand_reads(rng, 4000, G=30000) + [b"A" * 150] * 300 + [b"T" * 99] * 50
    rng.shuffle(reads)
    final, stats, nodes, infos, tables, (n_ovf, n_out), size = _sharded_build(capi, oracle, reads, n_shards, 600000, ranged=ranged)
    ref = oracle.build_graph(files_mem=[oracle.pack_reads(reads)], k=31, init_hash_size=0.002)
    assert sum(int(s.count) for s in final) == ref.count
    assert sum(int(s.total_kmers) for s in stats) == ref.total_kmers
    assert np.array_equal(np.sort(nodes, order="kmer"), ref.nodes)
    # every shard holds exactly the keys whose home slot lies in its range (or was handed over from the shard before)
    L = oracle.lib()
    for info, (array, flags) in zip(infos, tables):
        occ = np.flatnonzero(array["kmer"] != 0)
        for i in occ[:300]:
            home = L.orc_hash_code(int(array["kmer"][i])) % size
            assert (info.slot_lo <= home < info.slot_hi and home <= info.slot_lo + i) or home >= infos[info.rank - 1].slot_lo
    # concatenated shards form one valid linear-probing table of the global size (key-0 node aside)
    whole = np.concatenate([t[0] for t in tables])
    assert len(whole) == size
    fl = np.packbits((whole["kmer"] != 0).astype(np.uint8))
    fl = np.concatenate([fl, np.zeros(size // 8 + 1 - len(fl), np.uint8)])
    assert oracle.check_host_table(whole, fl, size, ref.count - 1) == 0


def test_heavy_hitters_beyond_the_overflow_list(capi, oracle):
    """a tandem repeat puts millions of occurrences on a handful of k-mers: their final buckets overflow, the
    overflow list (1/16 of the input + 1 M) fills up, and the rest is aggregated in the side table"""
    rng = random.Random(99)
    unit = "ACGGTCA"
    reads = [(unit * 30)[rng.randint(0, 6):][:150].encode() for _ in range(30000)] + rand_reads(rng, 2000, G=20000)
    rng.shuffle(reads)
    ref = oracle.build_graph(files_mem=[oracle.pack_reads(reads)], k=31, init_hash_size=0.002)
    with capi.Graph(k=31, table_slots=PART_SLOTS, engine=capi.ENGINE_PARTITION, expected_kmers=ref.total_kmers) as g:
        g.push_reads(*oracle.pack_reads(reads))
        st = g.finalize()
        assert st.count == ref.count and st.total_kmers == ref.total_kmers
        assert np.array_equal(g.export_sorted(), ref.nodes)


def test_tables_of_2_31_slots_and_more_use_the_wide_divisor_path(capi, oracle):
    """the 8-GPU bench builds ONE table of ~2^32 slots: above 2^31 slots the level-1 kernel divides with the
    two-step 2-by-1 form instead of the multiply-high shortcut; single handle and two shards"""
    rng = random.Random(2031)
    reads = rand_reads(rng, 6000, G=40000) + [b"A" * 150] * 100
    ref = oracle.build_graph(files_mem=[oracle.pack_reads(reads)], k=31, init_hash_size=0.002)
    size = capi.find_next_prime_ref(2_200_000_000)
    assert size >= 1 << 31
    with capi.Graph(k=31, table_slots=size, engine=capi.ENGINE_PARTITION, expected_kmers=1_000_000) as g:
        g.push_reads(*oracle.pack_reads(reads))
        st = g.finalize()
        assert st.count == ref.count and st.total_kmers == ref.total_kmers
        assert np.array_equal(g.export_sorted(), ref.nodes)
    final, stats, nodes, infos, _, _, size2 = _sharded_build(capi, oracle, reads, 2, 600000, slots=2_200_000_000, want_tables=False)
    assert size2 == size and infos[1].slot_hi == size
    assert sum(int(s.count) for s in final) == ref.count
    assert np.array_equal(np.sort(nodes, order="kmer"), ref.nodes)


@pytest.mark.parametrize("slots,n_shards", [(4_400_000_000, 0), (8_700_000_000, 0), (8_700_000_000, 2), (4_400_000_000, 3)])
def test_tables_of_2_32_slots_and_more(capi, oracle, slots, n_shards):
    """BASELINE cfg3 needs one table of ~2^33 slots (5.6 G nodes); the reference's KmerSet.size is a
    uint64_t that doubles up to 2^10 times (kmerSet.h:90, DBGgraph.cpp:337-343).  Above 2^32 slots hash / size is a
    64-bit multiply-high, slot indices need more than 32 bits and level 2 fans out to 2048 / 4096 final buckets
    per level-1 bucket (up to 2^34 slots).  Single handle (70 and 139 GB tables: level-2 fan-out 2048 and 4096) and two shards of one 2^33-slot
    table on one GPU; streaming flush in between (the incremental region build at the wide geometry)."""
    rng = random.Random(slots % 1000)
    reads = rand_reads(rng, 6000, G=40000) + [b"A" * 150] * 100
    bases, offsets = oracle.pack_reads(reads)
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, init_hash_size=0.002)
    size = capi.find_next_prime_ref(slots)
    assert size >= 1 << 32
    if n_shards == 0:
        with capi.Graph(k=31, table_slots=size, engine=capi.ENGINE_PARTITION, expected_kmers=1_000_000) as g:
            half = len(reads) // 2
            g.push_reads(bases[:int(offsets[half])], offsets[:half + 1])
            g.flush()
            g.push_reads(bases[int(offsets[half]):], offsets[half:] - offsets[half])
            st = g.finalize()
            assert st.count == ref.count and st.total_kmers == ref.total_kmers
            assert np.array_equal(g.export_sorted(), ref.nodes)
            assert g.digest() == oracle.nodes_digest(ref.nodes)
    else:
        final, stats, nodes, infos, _, _, size2 = _sharded_build(capi, oracle, reads, n_shards, 600000, slots=slots, want_tables=False)
        assert size2 == size and infos[-1].slot_hi == size
        assert sum(int(s.count) for s in final) == ref.count
        assert np.array_equal(np.sort(nodes, order="kmer"), ref.nodes)


def test_sharded_table_with_overflow_and_heavy_repeats(capi, oracle):
    """tiny bucket capacities force overflow observations; a heavily repeated read saturates counters
    across shards"""
    rng = random.Random(77)
    reads = rand_reads(rng, 1500, G=6000) + [b"ACGTTGCATGCAAGCTTAGCTAGGATCCGATCGATTACGAT" * 3] * 2500
    rng.shuffle(reads)
    final, stats, nodes, infos, tables, (n_ovf, n_out), size = _sharded_build(capi, oracle, reads, 2, 1)
    ref = oracle.build_graph(files_mem=[oracle.pack_reads(reads)], k=31, init_hash_size=0.001)
    assert n_ovf > 0
    assert sum(int(s.count) for s in final) == ref.count
    assert np.array_equal(np.sort(nodes, order="kmer"), ref.nodes)


def test_sharded_heavy_hitters_beyond_the_overflow_list(capi, oracle):
    """the heavy-hitter side table on SHARDED handles (round 1: none, such input ended in DBGK_ERR_CAPACITY): a tandem
    repeat overflows its final buckets, the overflow lists fill up (1 M + 1/16 of the input), the surplus is aggregated
    in every rank's side table, which is offered to all ranks after the build -- through the hand-written protocol of
    _sharded_build and through the C++ communicator"""
    rng = random.Random(98)
    unit = "ACGGTCA"
    reads = [(unit * 30)[rng.randint(0, 6):][:150].encode() for _ in range(40000)] + rand_reads(rng, 2000, G=20000)
    rng.shuffle(reads)
    bases, offsets = oracle.pack_reads(reads)
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, init_hash_size=0.002)
    final, stats, nodes, infos, _, (n_ovf, n_out), size = _sharded_build(capi, oracle, reads, 2, ref.total_kmers // 2, want_tables=False)
    assert n_ovf >= 2 * (1 << 20)   # both overflow lists ran full
    assert sum(int(s.count) for s in final) == ref.count
    assert np.array_equal(np.sort(nodes, order="kmer"), ref.nodes)
    with capi.Comm(k=31, table_slots=capi.find_next_prime_ref(PART_SLOTS), devices=[0, 0, 0], expected_kmers=ref.total_kmers // 3) as c:
        n = len(reads)
        for a, b in zip(range(0, n, n // 6 + 1), list(range(n // 6 + 1, n, n // 6 + 1)) + [n]):
            c.push_reads(bases[int(offsets[a]):int(offsets[b])], offsets[a:b + 1] - offsets[a])
        st = c.finalize()
        assert st.count == ref.count and c.digest() == oracle.nodes_digest(ref.nodes)


# ------------------------------------------------------------------------------------------------
# dbgk_comm_*: the C++ host layer's multi-GPU path (N sharded handles inside one process, peer copies).
# One GPU here, so the shards share the device; the protocol is the same.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_shards,store,pieces", [(2, 1 << 22, 8), (3, 40000, 3), (4, 1 << 22, 1)])
def test_comm_in_process_shards_equal_oracle(capi, oracle, n_shards, store, pieces, monkeypatch):
    """reads dealt round robin to the shards, records exchanged in pieces, every shard builds its slot range;
    a small store forces several flush rounds (exchange + incremental build each); explicit flush + exact
    count in between; export at the global size (shards side by side) and at another size (host re-seat)"""
    monkeypatch.setenv("DBGK_COMM_PIECES", str(pieces))
    rng = random.Random(n_shards * 7 + pieces)
    reads = rand_reads(rng, 5000, G=30000) + [b"A" * 150] * 200 + [b"ACGTTGCATGCAAGCTTAGCTAGGATCCGATCGATTACGAT" * 3] * 300
    rng.shuffle(reads)
    bases, offsets = oracle.pack_reads(reads)
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, init_hash_size=0.001)
    size = capi.find_next_prime_ref(PART_SLOTS)
    with capi.Comm(k=31, table_slots=size, devices=[0] * n_shards, expected_kmers=store, max_batch_bases=1 << 16) as c:
        n = len(reads)
        cuts = list(range(0, n, n // 9)) + [n]
        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            lo, hi = int(offsets[a]), int(offsets[b])
            c.push_reads(bases[lo:hi], offsets[a:b + 1] - offsets[a])
            if i == 4:
                c.flush()
                part = oracle.build_graph(files_mem=[(bases[:hi], offsets[:b + 1])], k=31, init_hash_size=0.001)
                assert c.refresh_stats().count == part.count
        st = c.finalize()
        assert (st.total_reads, st.total_kmers, st.count) == (ref.total_reads, ref.total_kmers, ref.count)
        assert c.digest() == oracle.nodes_digest(ref.nodes)
        assert list(c.link_stats(2).depth_stat) == list(oracle.link_stats(ref.nodes, 2).depth_stat)
        for host_size in (size, capi.find_next_prime_ref(3 * ref.count)):
            array, flags = c.export_host_table(host_size)
            assert oracle.check_host_table(array, flags, host_size, st.count) == 0
            occ = np.unpackbits(flags)[:host_size].astype(bool)
            assert np.array_equal(np.sort(array[occ], order="kmer"), ref.nodes)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in golden_cases() if c["name"] in ("mixed150_k31", "saturate_k31", "polyA_k31", "even_k16", "synth_20k_k31")],
                         ids=lambda c: c["name"])
@pytest.mark.parametrize("engine,cutoff", [(2, 2), (1, 0), (2, 7)])
def test_export_with_the_consumers_first_pass_on_the_device_PARITY_UNPINNED(capi, oracle, case, engine, cutoff):
    """dbgk_export_host_table_links: the table AND what calculate_kmer_links (contig.cpp:107-181) makes of it -- KmerLink record
    per slot, del_flag bitmap, tip / branch slot lists in ascending order, DepthStat -- against the restatement of those lines
    applied to the exported table itself, slot for slot (parity unpinned: contig.cpp needs Boost, absent here)"""
    p = case["params"]
    files = case_reads(case, oracle)
    host_size = capi.find_next_prime_ref(max(3 * case["ref"]["count"], 1000))   # != the device table: the export re-seats the nodes
    with capi.Graph(k=p["k"], table_slots=capi.find_next_prime_ref(70_000_000 if engine == 2 else 2 * case["ref"]["count"] + 1000),
                    max_read_len=p["max_read_len"], engine=engine, expected_kmers=case["ref"]["kmers"] + 1000 if engine == 2 else 0) as g:
        for bases, offsets in files:
            g.push_reads(bases, offsets)
        st = g.finalize()
        array, flags, klink, dele, tips, branches, ls = g.export_host_table_links(cutoff, host_size)
        assert oracle.check_host_table(array, flags, host_size, st.count) == 0
        plain, plain_flags = g.export_host_table(host_size)   # (re-seated anew by atomics: a valid layout of its own, same nodes)
        assert oracle.check_host_table(plain, plain_flags, host_size, st.count) == 0
        assert np.array_equal(np.sort(plain[plain["kmer"] != 0], order="kmer"), np.sort(array[array["kmer"] != 0], order="kmer"))
    want_rec, want_del, want_tips, want_branches = oracle.kmer_links(array, flags, cutoff)
    assert np.array_equal(klink, want_rec)
    assert np.array_equal(dele, want_del)
    assert np.array_equal(tips, want_tips) and np.array_equal(branches, want_branches)
    assert np.all(np.diff(tips.astype(np.int64)) > 0) and np.all(np.diff(branches.astype(np.int64)) > 0)
    occ = np.unpackbits(flags)[:host_size].astype(bool)
    ref = oracle.link_stats(array[occ], cutoff)
    assert list(ls.depth_stat) == list(ref.depth_stat)
    assert (ls.total_nodes, ls.deleted_lowfreq, ls.linear_nodes, ls.tip_nodes, ls.branch_nodes) == \
           (ref.total_nodes, ref.deleted_lowfreq, ref.linear_nodes, len(tips), len(branches))
    assert ls.tip_nodes == ref.tip_nodes and ls.branch_nodes == ref.branch_nodes


@pytest.mark.gpu
@pytest.mark.parametrize("n_reads,size_hint", [(40_000, 17_000_000), (3, 17_000_000), (400_000, 60_000_000), (200_000, 25_165_857), (60_000, 7_000_000)])
def test_host_table_through_the_occupied_nodes_only_equals_the_plain_copy(capi, oracle, monkeypatch, n_reads, size_hint):
    """Large host tables leave the device as the stream of their occupied nodes + the occupancy bits and are laid out at their slots
    by host threads (d2h_compact, dbgk.hip); the hook export_full=1 copies every slot as before.  Same image on the device -> the two
    must agree byte for byte, in buffers that held garbage before (every slot and every flag byte is written).  Sizes that are no
    multiple of 64 or 4096, a nearly empty and a well filled table, with and without the link pass."""
    set_hooks(monkeypatch, export_compact_min=0)
    monkeypatch.setenv("DBGK_EXPORT_THREADS", "5")
    P = oracle.synth_params(300_000, 150, cfg=1)
    bases, offsets = oracle.synth_reads(P, 0, n_reads)
    size = capi.find_next_prime_ref(size_hint) if size_hint != 25_165_857 else size_hint   # (an odd non-prime works as well)
    lib = capi.lib()

    def export(g, links):
        array = np.full(size * 16, 0xAB, dtype=np.uint8).view(capi.NODE_DTYPE)
        flags = np.full(size // 8 + 1, 0xCD, dtype=np.uint8)
        assert len(array) == size
        if not links:
            assert lib.dbgk_export_host_table(g._h, size, array.ctypes.data, flags.ctypes.data) == 0
            return array, flags
        return g.export_host_table_links(2)[:4]

    with capi.Graph(k=31, table_slots=size, max_read_len=150, engine=1) as g:
        g.push_reads(bases, offsets)
        st = g.finalize()
        for links in (False, True):
            set_hooks(monkeypatch, export_full=None)
            got = export(g, links)
            set_hooks(monkeypatch, export_full=1)
            want = export(g, links)
            for a, b in zip(got, want):
                assert np.array_equal(a, b)
        assert oracle.check_host_table(got[0], got[1], size, st.count) == 0
        assert int(np.unpackbits(got[1])[:size].sum()) == st.count


@pytest.mark.gpu
@pytest.mark.parametrize("n_shards,staging", [(2, False), (3, True)])
def test_comm_export_with_the_consumers_first_pass_PARITY_UNPINNED(capi, oracle, n_shards, staging, monkeypatch):
    """dbgk_comm_export_host_table_links: the table of a communicator AND calculate_kmer_links' first pass for it (contig.cpp:107-181),
    slot numbers those of the ONE host table: against the restatement applied to the exported table itself, slot for slot, and the
    node multiset against the oracle (parity of the link records unpinned: contig.cpp needs Boost)"""
    if staging:
        monkeypatch.setenv("DBGK_COMM_HOST_STAGING", "1")
        set_hooks(monkeypatch, export_compact_min=0)   # the small host table too leaves as occupied nodes + bits (d2h_compact)
    rng = random.Random(40 + n_shards)
    reads = rand_reads(rng, 3000, G=20000) + [b"A" * 150] * 300 + [b"T" * 90] * 40
    rng.shuffle(reads)
    bases, offsets = oracle.pack_reads(reads)
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, init_hash_size=0.001)
    size = capi.find_next_prime_ref(PART_SLOTS)
    for host_size in (capi.find_next_prime_ref(3 * ref.count),) + ((size,) if n_shards == 2 else ()):   # (the 67 M-slot host table once)
        with capi.Comm(k=31, table_slots=size, devices=[0] * n_shards, expected_kmers=len(bases), max_batch_bases=1 << 18) as c:
            c.push_reads(bases, offsets)
            st = c.finalize()
            assert st.count == ref.count
            array, flags, klink, dele, tips, branches, ls = c.export_host_table_links(host_size, st.count, cutoff=2)
        assert oracle.check_host_table(array, flags, host_size, st.count) == 0
        occ = np.unpackbits(flags)[:host_size].astype(bool)
        assert np.array_equal(np.sort(array[occ], order="kmer"), ref.nodes)
        want_rec, want_del, want_tips, want_branches = oracle.kmer_links(array, flags, 2)
        assert np.array_equal(klink, want_rec) and np.array_equal(dele, want_del)
        assert np.array_equal(tips, want_tips) and np.array_equal(branches, want_branches)
        want = oracle.link_stats(array[occ], 2)
        assert list(ls.depth_stat) == list(want.depth_stat) and (ls.total_nodes, ls.tip_nodes, ls.branch_nodes) == (want.total_nodes, len(tips), len(branches))


@pytest.mark.gpu
@pytest.mark.parametrize("n_shards,staging", [(2, False), (3, True)])
def test_comm_resize_across_shards_and_host_staged_copies(capi, oracle, n_shards, staging, monkeypatch):
    """dbgk_comm_resize: the table of a communicator enlarged mid-stream (enlarge_kmerset_parallel, kmerSet.cpp:132-189, for a
    table that lives on several GPUs): every node re-seated into the shard that owns its new home slot, totals and key-0 links
    carried over, the rest of the input joining afterwards.  With DBGK_COMM_HOST_STAGING every copy between members goes through
    pinned host memory -- the path taken when two GPUs refuse peer access."""
    if staging:
        monkeypatch.setenv("DBGK_COMM_HOST_STAGING", "1")
    rng = random.Random(90 + n_shards)
    reads = rand_reads(rng, 4000, G=25000) + [b"A" * 150] * 100 + [b"T" * 90] * 40
    rng.shuffle(reads)
    bases, offsets = oracle.pack_reads(reads)
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, init_hash_size=0.001)
    size = capi.find_next_prime_ref(PART_SLOTS)
    bigger = capi.find_next_prime_ref(2 * PART_SLOTS + 12345)
    with capi.Comm(k=31, table_slots=size, devices=[0] * n_shards, expected_kmers=200000, max_batch_bases=1 << 16) as c:
        n = len(reads)
        cuts = list(range(0, n, n // 6)) + [n]
        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            lo, hi = int(offsets[a]), int(offsets[b])
            c.push_reads(bases[lo:hi], offsets[a:b + 1] - offsets[a])
            if i == 2:
                part = oracle.build_graph(files_mem=[(bases[:hi], offsets[:b + 1])], k=31, init_hash_size=0.001)
                c.resize(bigger)
                st_mid = c.refresh_stats()
                assert (st_mid.count, st_mid.total_kmers, st_mid.table_slots) == (part.count, part.total_kmers, bigger)
        st = c.finalize()
        assert (st.total_reads, st.total_kmers, st.count) == (ref.total_reads, ref.total_kmers, ref.count)
        assert c.digest() == oracle.nodes_digest(ref.nodes)
        array, flags = c.export_host_table(bigger)
        assert oracle.check_host_table(array, flags, bigger, st.count) == 0
        occ = np.unpackbits(flags)[:bigger].astype(bool)
        assert np.array_equal(np.sort(array[occ], order="kmer"), ref.nodes)


def test_comm_resize_before_anything_was_built(capi, oracle):
    """dbgk_comm_resize right after dbgk_comm_create, and again after pushes that were never flushed: a shard that has not been
    through a region build holds no table (uninitialised memory) -- nothing may be carried over from it (round-3 advisor finding:
    the scan re-seated garbage nodes).  The job afterwards is the oracle's; the reads go in 2-bit packed."""
    rng = random.Random(12)
    reads = rand_reads(rng, 1500, G=20000)
    bases, offsets = oracle.pack_reads(reads)
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, init_hash_size=0.001)
    size, bigger, biggest = (capi.find_next_prime_ref(PART_SLOTS + d) for d in (0, 40_000_000, 90_000_000))
    with capi.Comm(k=31, table_slots=size, devices=[0, 0, 0], expected_kmers=400000, max_batch_bases=1 << 16) as c:
        c.resize(bigger)                                  # nothing pushed at all
        st0 = c.refresh_stats()
        assert (int(st0.count), int(st0.total_kmers), int(st0.table_slots)) == (1, 0, bigger)
        half = len(reads) // 2
        lo = int(offsets[half])
        words, other = capi.pack_bases(bases)
        c.push_reads_packed(words, offsets[:half + 1], other)
        c.resize(biggest)                                 # (flushes what was pushed, then carries those nodes over)
        c.push_reads(bases[lo:], offsets[half:] - offsets[half])
        st = c.finalize()
        assert (st.total_reads, st.total_kmers, st.count) == (ref.total_reads, ref.total_kmers, ref.count)
        assert c.digest() == oracle.nodes_digest(ref.nodes)


def test_push_reads_from_page_locked_memory_skips_the_staging_copy(capi, oracle):
    """dbgk_push_reads from a pinned caller buffer (torch pinned tensor == hipHostMalloc): host-to-device copies come straight
    out of the caller's memory, in several batches; same graph as from pageable memory and as the oracle; the buffer may be
    overwritten as soon as the call returns"""
    import torch
    PO = oracle.synth_params(300000, 150, cfg=2)
    bases, offsets = oracle.synth_reads(PO, 0, 20000)
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, init_hash_size=0.01)
    pinned = torch.empty(len(bases), dtype=torch.uint8).pin_memory()
    view = pinned.numpy()
    for engine, slots, expected in ((capi.ENGINE_DIRECT, 10000019, 0), (capi.ENGINE_PARTITION, capi.find_next_prime_ref(1 << 26), 20000 * 120)):
        with capi.Graph(k=31, table_slots=slots, engine=engine, expected_kmers=expected, max_batch_bases=1 << 19) as g:
            view[:] = bases
            g.push_reads(view, offsets)
            view[:] = 65      # every copy out of the buffer has run when the call returns
            st = g.finalize()
            assert (int(st.count), int(st.total_kmers)) == (ref.count, ref.total_kmers)
            assert np.array_equal(g.export_sorted(), ref.nodes)


@pytest.mark.parametrize("k,L,plain", [(31, 150, 0), (23, 150, 0), (31, 94, 0), (17, 136, 0), (32, 151, 0), (31, 151, 0), (25, 140, 0), (31, 145, 0),
                                       (21, 151, 0), (31, 100, 0), (19, 250, 0), (15, 100, 0), (31, 150, 1), (21, 151, 1)])
def test_equal_length_reads_in_regular_tiles_equal_oracle(capi, oracle, monkeypatch, k, L, plain):
    """The regular-tile form of the equal-length level-1 kernel (every tile = 1024 / Q whole reads from a 16-byte boundary,
    Q a power of two, reads that fill their lanes exactly, k >= 17: rolls on 32-bit halves, no per-position validity test)
    on the inputs its short cuts could get wrong: poly-A / poly-T reads and long A / T runs inside reads (key 0 is kept
    aside), a read's first and last window (no left / right neighbour), lower case and N, both strands; 5 whole tiles
    and a remainder that takes the general form.  (k, L) cover 15 and 16 windows per lane, 4 / 8 lanes per read, k = 17
    (the entering complement base lands on bit 0 of the high word) and k = 32 (mask of all ones); the last four are reads that do
    NOT fill their lanes (151 bases at k = 31: 121 windows = 7 x 16 + 9; 140 at k = 25: 7 x 15 + 11; 145 at k = 31: 7 x 15 + 10;
    151 at k = 21: 8 x 16 + 3 ... in 16 lanes it is not regular: 9 x 15 - 4), the form whose positions test their validity.
    Lane counts that are no power of two (nine lanes at (21, 151), five at (31, 100), fifteen at (19, 250)) take the general
    equal-length form, pipelined from k = 17 on ((15, 100): its plain tile loop); plain = 1 forces the plain loop everywhere."""
    if plain:
        set_hooks(monkeypatch, l1_plain=1)
    rng = random.Random(31000 + 100 * k + L)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    genome = "".join(rng.choice("ACGT") for _ in range(6000))
    genome = genome[:2000] + "A" * (k + 5) + genome[2000:4000] + "T" * (2 * k) + genome[4000:]
    reads = []
    lanes_per_read = {(31, 150): 8, (23, 150): 8, (31, 94): 4, (17, 136): 8, (32, 151): 8, (31, 151): 8, (25, 140): 8, (31, 145): 8, (21, 151): 9, (31, 100): 5, (19, 250): 15, (15, 100): 6}[(k, L)]
    n = 5 * (1024 // lanes_per_read) + 37
    for i in range(n):
        x = rng.random()
        if x < 0.03:
            r = "A" * L
        elif x < 0.05:
            r = "T" * L
        else:
            s = rng.randint(0, len(genome) - L)
            r = genome[s:s + L]
            if rng.random() < 0.5:
                r = "".join(comp[c] for c in reversed(r))
            r = list(r)
            for j in range(L):
                y = rng.random()
                if y < 0.004:
                    r[j] = rng.choice("ACGT")
                elif y < 0.006:
                    r[j] = rng.choice("Nn")
            r = "".join(r)
            if rng.random() < 0.1:
                r = r.lower()
        reads.append(r.encode())
    bases, offsets = oracle.pack_reads(reads)
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=k, max_read_len=250, init_hash_size=0.001)
    with capi.Graph(k=k, table_slots=capi.find_next_prime_ref(1 << 26), max_read_len=250, engine=capi.ENGINE_PARTITION,
                    expected_kmers=len(bases)) as g:
        g.push_reads(bases, offsets)
        st = g.finalize()
        assert g.timings().uniform_launches >= 1
        assert (int(st.count), int(st.total_kmers), int(st.total_reads)) == (ref.count, ref.total_kmers, ref.total_reads)
        nodes = g.export_sorted()
        assert np.array_equal(nodes, ref.nodes)
        assert nodes[0]["kmer"] == 0 and (nodes[0]["l_link"] or nodes[0]["r_link"])   # poly-A / poly-T seen
