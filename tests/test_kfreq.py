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
@pytest.mark.parametrize("k,expected", [(9, 0), (13, 0), (13, 400000), (11, 1)])
def test_kfreq_engine_counts_equal_oracle(oracle, k, expected):
    """expected == 0: atomics on the direct-addressed byte table; expected > 0 (the input size is known):
    occurrences partitioned by hash and aggregated per key in LDS (PARTITION engine), same table at the
    end; expected = 1 under-sizes every bucket, so most occurrences take the overflow path"""
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
        extra = ["-e", "300000"] if bits_fmt == 8 else []  # the 8-bit run counts through the partitioned engine
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
