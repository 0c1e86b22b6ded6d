"""link_scaffold seed index: chop_contig_to_kmerset
(link_scaffold/map_func.cpp:119-173) over add_kmerset (link_scaffold/kmerSet.cpp:168-210).

CPU: the restatement (oracle_py.seed_index) against the golden dumps written by the REAL reference
code (tests/golden/seed_*.{fa,dump}, made by make_seed_golden.py) and, where oracle/_ref/ref_seed
exists, against live runs on random scaffolds.
GPU: the SEEDIDX engine through the C ABI against the same goldens and the restatement."""
import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN

GOLD = sorted(glob.glob(os.path.join(GOLDEN, "seed_*.fa")))
COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def read_fasta(path):
    seqs, cur = [], None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if cur is not None:
                seqs.append("".join(cur))
            cur = []
        elif line:
            cur.append(line)
    if cur is not None:
        seqs.append("".join(cur))
    return seqs


def gold_k(path):
    return int(os.path.basename(path).split("_")[1][1:])


def scaffolds(seed, n_contigs, mean_len, k, n_gap_rate=0.002):
    """random scaffolds: shared repeats, reverse-strand copies, lower case, N gaps between blocks >= k"""
    rng = np.random.default_rng(seed)
    genome = "".join(rng.choice(list("ACGT"), size=mean_len * 3))
    out = []
    for _ in range(n_contigs):
        parts = []
        for _ in range(int(rng.integers(1, 5))):
            a = int(rng.integers(0, len(genome) - k - 1))
            blk = genome[a:a + int(rng.integers(k, mean_len))]
            if rng.random() < 0.3:
                blk = blk[::-1].translate(COMP)
            if rng.random() < 0.2:
                blk = blk.lower()
            parts.append(blk)
            parts.append("N" * int(rng.integers(1, 40)))
        out.append("".join(parts[:-1]) if rng.random() < 0.5 else "".join(parts))
    return out


def assert_same(a, b):
    assert len(a) == len(b)
    for f in ("kmer", "id", "pos", "freq", "direct"):
        assert np.array_equal(a[f], b[f]), f


# ------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("fa", GOLD, ids=[os.path.basename(p)[:-3] for p in GOLD])
def test_restatement_equals_reference_golden(oracle, fa):
    meta, want = oracle.parse_seed_dump(fa[:-3] + ".dump")
    got = oracle.seed_index(read_fasta(fa), gold_k(fa))
    assert meta["count"] == len(want) == len(got)
    assert_same(got, want)


def test_golden_cases_cover_the_edges(oracle):
    assert len(GOLD) >= 5
    _, pal = oracle.parse_seed_dump(os.path.join(GOLDEN, "seed_k16_even_pal.dump"))
    rc = oracle.revcomp_values(pal["kmer"], 16)
    assert np.any((rc == pal["kmer"]) & (pal["direct"] == 0))          # palindromes are stored with direct = 0
    _, pa = oracle.parse_seed_dump(os.path.join(GOLDEN, "seed_k17_polyA_pal.dump"))
    assert pa["kmer"][0] == 0 and pa["freq"][0] == 0 and pa["direct"][0] == 1   # key 0 is an ordinary key
    _, sc = oracle.parse_seed_dump(os.path.join(GOLDEN, "seed_k31_scaffolds.dump"))
    assert 0 < sc["freq"].sum() < len(sc) and set(sc["id"].tolist()) == {0, 1, 2}


@pytest.mark.parametrize("k,seed", [(31, 1), (21, 2), (12, 3), (32, 4)])
def test_restatement_equals_reference_live(oracle, tmp_path, k, seed):
    if not oracle.have_ref_seed():
        pytest.skip("oracle/_ref/ref_seed not built (no /root/reference here)")
    contigs = scaffolds(seed, 40, 3000, k)
    fa = str(tmp_path / "c.fa")
    oracle.write_contig_fasta(fa, contigs, width=60 if seed % 2 else 0)
    # a small initial table makes the reference enlarge several times on the way
    meta, want = oracle.ref_seed(fa, k, str(tmp_path / "d.txt"), hash_size=1000 if seed == 2 else 0)
    got = oracle.seed_index(contigs, k)
    assert meta["count"] == len(got)
    assert_same(got, want)


# ------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def capi():
    from dbg_assembly_amd import capi as c
    assert c.lib().dbgk_device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback exists)"
    return c


def gpu_seed_index(capi, contigs, k, pushes=1, table_slots=None):
    total = sum(len(c) for c in contigs)
    slots = table_slots or capi.find_next_prime_ref(max(3 * total, 1000))   # map_pair.cpp:122: hash_size = 3 x total length
    g = capi.Graph(k, max_read_len=250, table_slots=slots, engine=capi.ENGINE_SEEDIDX, max_batch_bases=max(total, 1 << 16))
    per = (len(contigs) + pushes - 1) // pushes
    for i in range(0, len(contigs), per):
        part = contigs[i:i + per]
        bases = np.frombuffer("".join(part).encode(), dtype=np.uint8)
        offs = np.zeros(len(part) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(c) for c in part])
        g.push_reads(bases, offs)
    st = g.finalize()
    return g, st


@pytest.mark.gpu
@pytest.mark.parametrize("fa", GOLD, ids=[os.path.basename(p)[:-3] for p in GOLD])
def test_gpu_seed_index_equals_reference_golden(capi, oracle, fa):
    meta, want = oracle.parse_seed_dump(fa[:-3] + ".dump")
    contigs = read_fasta(fa)
    for pushes in (1, 3):
        g, st = gpu_seed_index(capi, contigs, gold_k(fa), pushes=pushes)
        assert st.count == meta["count"] and st.total_reads == len(contigs)
        got = g.seed_export_sorted()
        assert_same(oracle.seed_unpack(got["kmer"], got["payload"]), want)
        g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k,n_contigs,mean_len", [(31, 300, 20000), (17, 50, 5000), (32, 20, 3000), (13, 400, 4000)])
def test_gpu_seed_index_equals_restatement(capi, oracle, k, n_contigs, mean_len):
    contigs = scaffolds(100 + k, n_contigs, mean_len, k)
    contigs[3] = ""                                  # an empty sequence keeps its index
    contigs[5] = "ACGT"                              # shorter than k
    want = oracle.seed_index(contigs, k)
    g, st = gpu_seed_index(capi, contigs, k, pushes=4)
    assert st.count == len(want)
    got = g.seed_export_sorted()
    assert_same(oracle.seed_unpack(got["kmer"], got["payload"]), want)
    # host table in the reference's layout rules: every key reachable from its home slot over filled slots
    size = capi.find_next_prime_ref(int(len(want) / 0.5) + 100)
    array, flags = g.seed_export_host_table(size)
    filled = np.unpackbits(flags)[:size].astype(bool)
    assert filled.sum() == len(want)
    tab = np.sort(array[filled], order="kmer")
    assert np.array_equal(tab["kmer"], want["kmer"]) and np.array_equal(tab["payload"], oracle.seed_payload(want))
    assert not array[~filled]["kmer"].any() and not array[~filled]["payload"].any()
    home = np.array([oracle.lib().orc_hash_code(int(x)) % size for x in array["kmer"][filled][:2000]], dtype=np.int64)
    idx = np.nonzero(filled)[0][:2000]
    for h, i in zip(home.tolist(), idx.tolist()):   # exist_kmerset's walk (link_scaffold/kmerSet.cpp:216-238)
        j = h
        while j != i:
            assert filled[j]
            j = 0 if j + 1 == size else j + 1
    g.close()


@pytest.mark.gpu
def test_gpu_seed_index_small_table_reports_full_and_grows(capi, oracle):
    contigs = scaffolds(7, 30, 2000, 21)
    want = oracle.seed_index(contigs, 21)
    g, st = gpu_seed_index(capi, contigs, 21, table_slots=capi.find_next_prime_ref(int(len(want) * 1.3)))
    assert st.count == len(want)
    got = g.seed_export_sorted()
    assert_same(oracle.seed_unpack(got["kmer"], got["payload"]), want)
    g.close()
    with pytest.raises(capi.DbgkError):
        gpu_seed_index(capi, contigs, 21, table_slots=capi.find_next_prime_ref(len(want) // 2))


@pytest.mark.gpu
def test_graph_exports_refuse_seed_handles(capi):
    g, _ = gpu_seed_index(capi, ["ACGTACGTTGCATGCATGCAAAGGCTT" * 3], 15)
    with pytest.raises(capi.DbgkError):
        g.export_sorted()
    g.close()
