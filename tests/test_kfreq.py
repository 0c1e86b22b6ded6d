"""correct_error k-mer frequency table.

CPU: the restated loaders (oracle_py.kfreq_load_*) against the REAL reference loaders
(oracle/_ref/ref_kfreq1, ref_kfreq8) on random tables -- pins the file format.
GPU: KFREQ engine counts == oracle counts; the `kmerfreq` tool's files, read back through the
loaders, give the expected high-frequency bit table."""
import os
import random
import subprocess

import numpy as np
import pytest

from helpers import set_hooks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "dbg_assembly_amd", "bin", "kmerfreq")
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


@pytest.mark.parametrize("k", [11, 12])
def test_restated_loaders_equal_reference_loaders(oracle, tmp_path, k):
    if not oracle.have_ref_kfreq():
        pytest.skip("oracle/_ref loaders not built (no /root/reference here)")
    rng = np.random.default_rng(k)
    total = 4 ** k
    # 1-bit file: random sparse bitmap, includes non-canonical bits and palindromes
    bits = (rng.random(total) < 0.01).astype(np.uint8)
    path1 = str(tmp_path / "one.cz")
    oracle.kfreq_write_cz(path1, np.packbits(bits).tobytes(), oracle.KFREQ_BLOCK_KMERS // 8)
    got_ref, js = oracle.ref_kfreq_load(path1, k, one_bit=True, threads_or_cutoff=3)
    got_py, hif = oracle.kfreq_load_1bit(path1, k)
    assert np.array_equal(got_ref, got_py) and js["total"] == total and js["hifreq"] == hif
    # 8-bit file: random counts
    counts = rng.integers(0, 40, total).astype(np.uint8) * (rng.random(total) < 0.05)
    counts[rng.integers(0, total, 50)] = 255
    path8 = str(tmp_path / "eight.cz")
    oracle.kfreq_write_cz(path8, counts.astype(np.uint8).tobytes(), oracle.KFREQ_BLOCK_KMERS)
    for cutoff in (0, 10):
        got_ref, js = oracle.ref_kfreq_load(path8, k, one_bit=False, threads_or_cutoff=cutoff)
        got_py, n_total, n_effect = oracle.kfreq_load_8bit(path8, k, cutoff)
        assert np.array_equal(got_ref, got_py)
        assert (js["kmers"], js["effect"]) == (n_total, n_effect)


def _reads(rng, n, G=4000, L=100):
    g = "".join(rng.choice("ACGT") for _ in range(G))
    out = []
    for _ in range(n):
        ln = L if rng.random() < 0.8 else rng.randint(0, L + 30)
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
        out.append("".join(r).encode())
    return out + [b"A" * 90] * 300 + [b"ACGTTGCA" * 12] * 20 + [b"", b"ACG"]


@pytest.mark.gpu
@pytest.mark.parametrize("k,expected", [(9, 0), (13, 0), (13, 400000), (11, 1), (12, 400000), (14, 350000), (13, 1), (14, 3000),
                                        (13, 100000), (14, 60000), (12, 100000)])   # (the last three: a store a third / a fifth of the input -- flushes, blocks loaded back)
def test_kfreq_engine_counts_equal_oracle(oracle, k, expected):
    """expected == 0: atomics on the direct-addressed byte table; expected > 0 (the input size is known): occurrences go
    through the PARTITION engine -- k >= 13: partitioned by (permuted) 64-KiB block of the table and added up in an LDS image of
    the block; k < 13: partitioned by hash and aggregated per key in an LDS hash table -- same table at the end; expected = 1 or
    far too small under-sizes every bucket, so most occurrences take the overflow path"""
    from dbg_assembly_amd import capi
    rng = random.Random(k)
    reads = _reads(rng, 3000)
    bases, offsets = oracle.pack_reads(reads)
    want = oracle.kfreq_expected_counts([(bases, offsets)], k)
    with capi.Graph(k=k, table_slots=0, engine=capi.ENGINE_KFREQ, max_read_len=1000000, expected_kmers=expected) as g:
        g.push_reads(bases[:int(offsets[1500])], offsets[:1501])
        g.push_reads(bases[int(offsets[1500]):], offsets[1500:] - offsets[1500])
        st = g.finalize()
        got = g.kfreq_counts()
        assert np.array_equal(got, want)
        assert st.count == int((want > 0).sum()) and want.max() == 255
        assert st.stored_kmers == sum(max(0, len(r) - k + 1) for r in reads)
        for cutoff in (0, 1, 10, 254):
            bits = np.unpackbits(g.kfreq_bits(cutoff))
            assert np.array_equal(bits, (want > cutoff).astype(np.uint8))
        # table-only entry points are refused on a KFREQ handle
        with pytest.raises(capi.DbgkError):
            g.export_sorted()
        # the handle is reusable
        g.reset()
        g.push_reads(bases, offsets)
        g.finalize()
        assert np.array_equal(g.kfreq_counts(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["equal", "ragged", "mixed"])
@pytest.mark.parametrize("k", [13, 15])
def test_kfreq_direct_blocks_through_every_level1_form(oracle, monkeypatch, shape, k):
    """Direct blocks move 32-bit level-1 records (round 4): written by the wave-per-bucket copy-out, by the linear copy-out
    (hook l1_linear=1: what a table of 4^18 bytes with its 1024 level-1 buckets takes) and read back by level 2.  Reads of one
    length (the level-1 instantiation compiled for this case), mostly full length (ragged tiles), any length (lane prefix / flat
    kernel), as ASCII and as 2-bit words: the same 4^k-byte table as the oracle's."""
    from dbg_assembly_amd import capi
    rng = random.Random(100 * k + len(shape))
    genome = "".join(rng.choice("ACGT") for _ in range(30000))

    def read(ln):
        a = rng.randint(0, len(genome) - ln)
        r = genome[a:a + ln]
        return "".join(rng.choice("acgtN") if rng.random() < 0.01 else ch for ch in r).encode()

    if shape == "equal":
        reads = [read(150) for _ in range(4000)]
    elif shape == "ragged":
        reads = [read(150 if rng.random() < 0.9 else rng.randint(100, 150)) for _ in range(4000)]
    else:
        reads = [read(rng.choice([0, 5, k - 1, k, 40, 90, 150, 151, 400])) for _ in range(5000)]
    reads += [b"A" * 150] * 200 + [b"T" * 150] * 30 if shape == "equal" else [b"A" * 90] * 200
    rng.shuffle(reads)
    bases, offsets = oracle.pack_reads(reads)
    want = oracle.kfreq_expected_counts([(bases, offsets)], k)
    words, other = capi.pack_bases(bases)
    assert other == 0
    expected = sum(max(0, len(r) - k + 1) for r in reads)
    for lin in ("0", "1"):
        set_hooks(monkeypatch, l1_linear=lin)
        for packed in (False, True):
            with capi.Graph(k=k, table_slots=0, engine=capi.ENGINE_KFREQ, max_read_len=1000000, expected_kmers=expected) as g:
                if packed:
                    g.push_reads_packed(words, offsets)
                else:
                    g.push_reads(bases, offsets)
                st = g.finalize()
                assert np.array_equal(g.kfreq_counts(), want), (lin, packed)
                assert st.stored_kmers == expected


@pytest.mark.gpu
@pytest.mark.parametrize("k,fmt", [(12, 2), (13, 1)])
def test_kmerfreq_tool_files_load_like_the_reference(oracle, tmp_path, k, fmt):
    assert os.path.exists(TOOL), "kmerfreq not built"
    rng = random.Random(100 + k)
    reads = _reads(rng, 2500)
    f1, f2 = str(tmp_path / "a.txt"), str(tmp_path / "b.txt.gz")
    oracle.write_reads_file(f1, reads[:1400], fmt=fmt)
    oracle.write_reads_file(f2, reads[1400:], fmt=fmt, gz=True)
    lib = str(tmp_path / "reads.lib")
    open(lib, "w").write(f1 + "\n" + f2 + "\n")
    want = oracle.kfreq_expected_counts([oracle.pack_reads(reads)], k)
    idx = np.arange(4 ** k, dtype=np.uint64)
    rc = oracle.revcomp_values(idx, k).astype(np.int64)
    for bits_fmt, cutoff in ((1, 1), (8, 10)):
        prefix = str(tmp_path / ("out%d" % bits_fmt))
        extra = ["-e", "30000"] if bits_fmt == 8 else ["-a"]  # 8-bit run: partitioned counting through a small store (many rounds); 1-bit: atomics
        r = subprocess.run([TOOL, "-k", str(k), "-f", str(fmt), "-b", str(bits_fmt), "-m", str(cutoff), "-t", "4", "-o", prefix] + extra + [lib],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        cz = prefix + ".kmer.freq.cz"
        n_blocks = len(open(cz + ".len").read().split())
        assert n_blocks == max(1, 4 ** k // oracle.KFREQ_BLOCK_KMERS)
        hi = want > cutoff
        expect = hi.copy()
        expect[rc[hi]] = True  # both loaders end with v and rc(v) marked
        if bits_fmt == 1:
            got, hif = oracle.kfreq_load_1bit(cz, k)
            assert hif == int(hi.sum())  # only canonical k-mers were marked in the file
        else:
            got, n_total, n_effect = oracle.kfreq_load_8bit(cz, k, cutoff)
            assert n_effect == int((want > 0).sum()) and n_total == int(want.astype(np.int64).sum())
        assert np.array_equal(np.unpackbits(got).astype(bool), expect)
        if oracle.have_ref_kfreq():  # and through the real reference loader (the binary travels with the repo)
            got_ref, _ = oracle.ref_kfreq_load(cz, k, one_bit=(bits_fmt == 1), threads_or_cutoff=(3 if bits_fmt == 1 else cutoff))
            assert np.array_equal(got_ref, got)


# ---- several GPUs: partial tables combined by the per-byte saturating add --------------------

@pytest.mark.gpu
def test_kfreq_merge_counts_of_two_partial_tables(oracle):
    """dbgk_kfreq_merge_counts: min(255, a + b) per counter, the rule kfreq_reduce (multigpu.py) and dbgk_comm_* use"""
    from dbg_assembly_amd import capi
    k = 11
    reads = _reads(random.Random(77), 3000)
    want = oracle.kfreq_expected_counts([oracle.pack_reads(reads)], k)
    halves = [oracle.pack_reads(reads[0::2]), oracle.pack_reads(reads[1::2])]
    parts = [oracle.kfreq_expected_counts([h], k) for h in halves]
    assert ((parts[0].astype(np.int32) + parts[1]) > 255).any() and parts[0].max() == 255  # saturation in the sum AND in one part
    with capi.Graph(k=k, table_slots=0, engine=capi.ENGINE_KFREQ, max_read_len=1000000) as a, \
            capi.Graph(k=k, table_slots=0, engine=capi.ENGINE_KFREQ, max_read_len=1000000, expected_kmers=200000) as b:
        for g, (bases, offsets) in zip((a, b), halves):
            g.push_reads(bases, offsets)
            g.finalize()
        ptr, n = b.kfreq_device_counts()
        assert n == 4 ** k
        lo = 4 ** k // 3 & ~15
        a.kfreq_merge_counts(ptr + lo, lo, n - lo)  # in two pieces: [lo, n) then [0, lo)
        a.kfreq_merge_counts(ptr, 0, lo)
        assert np.array_equal(a.kfreq_counts(), want)
        assert a.refresh_stats().count == int((want > 0).sum())
        with pytest.raises(capi.DbgkError):
            a.kfreq_merge_counts(ptr + 8, 8, 16)  # alignment is part of the contract


@pytest.mark.gpu
@pytest.mark.parametrize("k,n_shards,expected,stage_mb", [(9, 2, 0, 0), (12, 3, 150000, 1), (13, 4, 20000, 1)])
def test_kfreq_communicator_equals_oracle(oracle, k, n_shards, expected, stage_mb, monkeypatch):
    """dbgk_comm_* with DBGK_ENGINE_KFREQ: n whole tables on (here) one GPU, reduced by range at finalize; staging
    buffers of 1 MiB make the copy / add pipeline go through many chunks (4^13 = 64 MiB of counters)"""
    from dbg_assembly_amd import capi
    if stage_mb:
        set_hooks(monkeypatch, comm_kfreq_stage_mb=stage_mb)
    reads = _reads(random.Random(k * 31 + n_shards), 4000)
    want = oracle.kfreq_expected_counts([oracle.pack_reads(reads)], k)
    with capi.Comm(k, 0, [0] * n_shards, max_read_len=1000000, expected_kmers=expected, engine=capi.ENGINE_KFREQ) as c:
        step = 450
        for i in range(0, len(reads), step):
            c.push_reads(*oracle.pack_reads(reads[i:i + step]))
        st = c.finalize()
        assert st.count == int((want > 0).sum())
        assert st.total_reads == len(reads) and st.stored_kmers == sum(max(0, len(r) - k + 1) for r in reads)
        assert np.array_equal(c.kfreq_counts(), want)
        first, n = 4 ** k // 5 + 3, 4 ** k // 2 + 11   # a range across owners, unaligned
        assert np.array_equal(c.kfreq_counts(first, n), want[first:first + n])
        for cutoff in (0, 1, 254):
            assert np.array_equal(np.unpackbits(c.kfreq_bits(cutoff)), (want > cutoff).astype(np.uint8))
        fb, nb = 4 ** k // 8 // 3 + 1, 4 ** k // 8 // 2
        assert np.array_equal(c.kfreq_bits(1, fb, nb), np.packbits(want > 1)[fb:fb + nb])
        with pytest.raises(capi.DbgkError):
            c.digest()  # graph-only entry points are refused


@pytest.mark.gpu
def test_kmerfreq_tool_on_several_gpu_shards(oracle, tmp_path):
    """`DBGK_GPU_LIST=0,0,0 kmerfreq`: the same files as one GPU writes"""
    k = 12
    reads = _reads(random.Random(5), 2500)
    f1 = str(tmp_path / "a.fa")
    oracle.write_reads_file(f1, reads, fmt=2)
    lib = str(tmp_path / "reads.lib")
    open(lib, "w").write(f1 + "\n")
    outs = {}
    for name, env in (("one", {}), ("three", {"DBGK_GPU_LIST": "0,0,0", "DBGK_BATCH_BYTES": "30000"})):
        prefix = str(tmp_path / name)
        r = subprocess.run([TOOL, "-k", str(k), "-f", "2", "-b", "8", "-t", "4", "-o", prefix, lib], capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-1500:]
        assert ("counting on 3 GPUs" in r.stderr) == (name == "three")
        outs[name] = (open(prefix + ".kmer.freq.cz", "rb").read(), open(prefix + ".kmer.freq.cz.len").read(), r.stderr.strip().splitlines()[-2])
    assert outs["one"] == outs["three"]
    want = oracle.kfreq_expected_counts([oracle.pack_reads(reads)], k)
    got, n_total, n_effect = oracle.kfreq_load_8bit(str(tmp_path / "three.kmer.freq.cz"), k, 0)
    assert n_effect == int((want > 0).sum()) and n_total == int(want.astype(np.int64).sum())


# ---- BASELINE cfg4 at its stated size: k = 17, 4^17 = 17 179 869 184 counters (16 GiB) --------------------
# Counting semantics are THIS BUILD'S (the reference's producer, `kmerfreq`, is not in its repository:
# "counting parity unpinned"): every window of every read, N as A, canonical = min(forward,
# reverse complement), saturating byte.  What is checked: the device table against a SPARSE restatement built
# from the pinned extraction (oracle.parse_read), and the files against the reference's REAL loaders.
K17 = 17


def _sparse_counts(oracle, bases, offsets, k):
    """-> (sorted distinct canonical k-mers, min(255, occurrences)) from the oracle's per-read extraction"""
    raw = np.ascontiguousarray(bases, dtype=np.uint8).tobytes()
    parts = []
    for i in range(len(offsets) - 1):
        km, _, _ = oracle.parse_read(raw[int(offsets[i]):int(offsets[i + 1])], k, 1000)   # (reads are 150 bases: never trimmed)
        parts.append(km)
    allk = np.concatenate(parts)
    uniq, cnt = np.unique(allk, return_counts=True)
    return uniq.astype(np.uint64), np.minimum(cnt, 255).astype(np.uint8), len(allk)


@pytest.fixture(scope="module")
def cfg4_sample(oracle):
    n_reads = 120000   # first reads of the cfg2/cfg4 generator workload: 16.1 M windows at k = 17
    PO = oracle.synth_params(50_000_000, 150, cfg=2)
    bases, offsets = oracle.synth_reads(PO, 0, n_reads)
    uniq, cnt, total = _sparse_counts(oracle, bases, offsets, K17)
    return n_reads, bases, offsets, uniq, cnt, total


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["atomics", "partitioned", "partitioned_small_store"])
def test_kfreq_engine_k17_16GiB_table_equals_sparse_restatement(oracle, cfg4_sample, mode):
    from dbg_assembly_amd import capi
    n_reads, bases, offsets, uniq, cnt, total = cfg4_sample
    expected = {"atomics": 0, "partitioned": total, "partitioned_small_store": total // 5}[mode]
    P = capi.synth_params(50_000_000, 150, cfg=2)
    with capi.Graph(k=K17, table_slots=0, engine=capi.ENGINE_KFREQ, max_read_len=1000000, expected_kmers=expected) as g:
        pieces = 4   # several device pushes: the small store flushes in between
        per = n_reads // pieces
        bufs = []
        for i in range(pieces):
            d_bases, d_off, nb = g.synth_reads_device(P, i * per, per)
            if i == 0:
                assert np.array_equal(d_bases.to_host(np.uint8, nb), bases[:nb])
            g.push_reads_device(d_bases.ptr, d_off.ptr, per, nb)
            bufs += [d_bases, d_off]
        st = g.finalize()
        assert (int(st.count), int(st.stored_kmers)) == (len(uniq), total)
        span = 1 << 30   # the 16 GiB table is read back and checked in 1 GiB slices (indices above 2^32 included)
        for first in range(0, 4 ** K17, span):
            got = g.kfreq_counts(first, span)
            lo, hi = np.searchsorted(uniq, [first, first + span])
            assert int(np.count_nonzero(got)) == hi - lo, first
            assert np.array_equal(got[(uniq[lo:hi] - np.uint64(first)).astype(np.int64)], cnt[lo:hi]), first
        # the bit table of the 1-bit format, last eighth (byte offsets beyond 2^31)
        fb = 7 * (4 ** K17 // 8) // 8
        bits = g.kfreq_bits(1, fb, 4 ** K17 // 8 - fb)
        lo = np.searchsorted(uniq, fb * 8)
        want_pos = uniq[lo:][cnt[lo:] > 1] - np.uint64(fb * 8)
        assert int(np.bitwise_count(bits).sum()) == len(want_pos)
        assert np.all(bits[(want_pos >> np.uint64(3)).astype(np.int64)] & (np.uint8(128) >> (want_pos & np.uint64(7)).astype(np.uint8)))
        for b in bufs:
            b.free()


@pytest.mark.gpu
@pytest.mark.parametrize("bits_fmt,cutoff,K17", [(1, 1, 17), (8, 2, 15)])
def test_kmerfreq_tool_k17_2048_blocks_through_the_reference_loaders(oracle, cfg4_sample, tmp_path, bits_fmt, cutoff, K17):
    """`kmerfreq -k 17`: 2048 blocks like test/01.clean_correct/clean_reads.lib.kmer.freq.cz.len, loaded by the
    reference's own loaders (correct_error/main_parallel_senior.cpp:334-408, main.cpp:161-220) to exactly the
    bit table the sparse restatement predicts (v and rc(v) set for every k-mer above the cutoff).  The 8-bit format -- whose
    serial loader needs 40 s for the 16 GiB of k = 17 -- runs at k = 15 (1 GiB, 128 blocks; k = 12 / 13 in the test below)."""
    if not oracle.have_ref_kfreq():
        pytest.skip("oracle/_ref loaders did not travel")
    n_reads, bases, offsets, uniq, cnt, total = cfg4_sample
    fa = str(tmp_path / "reads.fa")
    n_used = 40000
    raw = np.ascontiguousarray(bases[:int(offsets[n_used])]).tobytes()
    with open(fa, "wb") as fh:
        for i in range(n_used):
            fh.write(b">r\n" + raw[int(offsets[i]):int(offsets[i + 1])] + b"\n")
    lib = str(tmp_path / "reads.lib")
    open(lib, "w").write(fa + "\n")
    uniq, cnt, total = _sparse_counts(oracle, bases[:int(offsets[n_used])], offsets[:n_used + 1], K17)
    prefix = str(tmp_path / "out")
    r = subprocess.run([TOOL, "-k", str(K17), "-f", "2", "-b", str(bits_fmt), "-m", str(cutoff), "-t", "16", "-o", prefix, lib],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    cz = prefix + ".kmer.freq.cz"
    assert len(open(cz + ".len").read().split()) == (4 ** K17 // (8 << 20) if bits_fmt == 8 else 4 ** K17 // 8 // (1 << 20))
    got, js = oracle.ref_kfreq_load(cz, K17, one_bit=(bits_fmt == 1), threads_or_cutoff=(8 if bits_fmt == 1 else cutoff), timeout=900)
    hi = uniq[cnt > cutoff]
    want_pos = np.unique(np.concatenate([hi, oracle.revcomp_values(hi, K17)]))
    assert len(got) == 4 ** K17 // 8
    assert int(np.bitwise_count(got).sum()) == len(want_pos)
    assert np.all(got[(want_pos >> np.uint64(3)).astype(np.int64)] & (np.uint8(128) >> (want_pos & np.uint64(7)).astype(np.uint8)))
    if bits_fmt == 1:
        assert js["hifreq"] == len(hi)   # the file marks canonical k-mers only; the loader mirrors them
    else:
        assert (js["kmers"], js["effect"]) == (int(cnt.astype(np.int64).sum()), len(uniq))
