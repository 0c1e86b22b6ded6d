"""GPU (-m gpu): reads of MIXED lengths -- the input debruijn_contig really gets (quality-trimmed reads; the reference's own recorded
run has a mean length of 243 of 250, test/02.build_contig/Ecoli_corrected_reads.contig.log:437-438).  The bench workload cfg2t
(dbg_assembly_amd/workloads.py: 30 % of cfg2's reads trimmed to 60..149 bases) on a 200 k-read sample against the oracle, both
engines, both forms of the input; and the CPU side of the workload generator."""
import numpy as np
import pytest

from helpers import set_hooks

from dbg_assembly_amd import workloads


def test_trimmed_workload_is_a_function_of_the_read_index():
    a = workloads.trimmed_lengths(0, 100000)
    b = workloads.trimmed_lengths(40000, 1000)
    assert np.array_equal(a[40000:41000], b)
    short = a[a < 150]
    assert 0.28 < len(short) / len(a) < 0.32 and short.min() >= 60 and short.max() <= 149
    bases = np.frombuffer(b"ACGTACGTAC" * 15 * 3, dtype=np.uint8)
    out, off = workloads.trim_reads(bases, 150, np.array([150, 60, 149], dtype=np.uint64))
    assert list(off) == [0, 150, 210, 359] and bytes(out[150:210]) == bytes(bases[150:210]) and bytes(out[210:]) == bytes(bases[300:449])


@pytest.mark.gpu
def test_cfg2t_sample_equals_oracle(oracle):
    from dbg_assembly_amd import capi
    n_reads = 200000
    PO = oracle.synth_params(50_000_000, 150, cfg=2)
    fixed, _ = oracle.synth_reads(PO, 0, n_reads)
    bases, offsets = workloads.trim_reads(fixed, 150, workloads.trimmed_lengths(0, n_reads))
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=31, max_read_len=250, init_hash_size=0.05, threads=4)
    want = ref.nodes.astype(capi.NODE_DTYPE)
    words, other = capi.pack_bases(bases)
    assert other == 0
    for name, engine, slots, expected in (("direct", capi.ENGINE_DIRECT, 40000003, 0), ("partition", capi.ENGINE_PARTITION, capi.find_next_prime_ref(70000000), len(bases))):
        for packed in (False, True):
            with capi.Graph(k=31, table_slots=slots, max_read_len=250, engine=engine, expected_kmers=expected) as g:
                if packed:
                    g.push_reads_packed(words, offsets, other)
                else:
                    g.push_reads(bases, offsets)
                st = g.finalize()
                assert (st.total_reads, st.total_kmers, st.stored_kmers, st.count) == (ref.total_reads, ref.total_kmers, ref.total_kmers, ref.count)
                assert np.array_equal(g.export_sorted(), want), (name, packed)


def _shape_reads(rng, shape):
    import random
    g = "".join(rng.choice("ACGT") for _ in range(6000))

    def piece(ln):
        s0 = rng.randint(0, len(g) - ln) if ln <= len(g) else 0
        r = list((g * (ln // len(g) + 2))[s0:s0 + ln])
        for j in range(len(r)):
            if rng.random() < 0.01:
                r[j] = rng.choice("ACGTNacgt")
        return "".join(r).encode()
    if shape == "mixed":            # anything from empty to 300 bases
        return [piece(rng.randint(0, 300)) for _ in range(3000)] + [b"", b"A" * 200, b"T" * 31, b"ACGT"]
    if shape == "trimmed":          # cfg2t's profile
        return [piece(150 if rng.random() < 0.7 else rng.randint(60, 149)) for _ in range(4000)]
    if shape == "tiny":             # one or two windows per read: a tile's byte range does not fit the LDS image (global-memory path)
        return [piece(rng.choice([31, 32, 31, 33])) for _ in range(5000)]
    if shape == "gaps":             # thousands of reads without a window between the others (ranges with holes)
        out = []
        for _ in range(300):
            out.append(piece(rng.randint(100, 250)))
            out += [piece(rng.randint(0, 30)) for _ in range(rng.randint(0, 400))]
        return out
    if shape == "long":             # reads far beyond -r 100: trimmed to 100 bases, their tails skipped; and a few very long ones
        return [piece(rng.choice([80, 100, 101, 250, 1000, 5000])) for _ in range(600)]
    raise ValueError(shape)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,k,r", [("mixed", 31, 250), ("trimmed", 31, 250), ("tiny", 31, 250), ("gaps", 31, 250), ("long", 31, 100), ("long", 17, 5000),
                                       ("mixed", 32, 250), ("mixed", 17, 120), ("mixed", 5, 250)])
def test_prefix_level1_kernel_equals_oracle(oracle, monkeypatch, shape, k, r):
    """k_extract_scatter_prefix (every read exactly the lanes its windows need): reads of any lengths, reads without a window,
    reads trimmed at -r, byte ranges that do not fit the LDS image -- against the oracle; ASCII and 2-bit input; hook l1_prefix=1
    also sends the batches of the ragged form through it, =0 is the flat kernel (the A side of the comparison)"""
    import random
    from dbg_assembly_amd import capi
    rng = random.Random(len(shape) * 100 + k * 7 + r)
    reads = _shape_reads(rng, shape)
    bases, offsets = oracle.pack_reads(reads)
    ref = oracle.build_graph(files_mem=[(bases, offsets)], k=k, max_read_len=r, init_hash_size=0.001, threads=1)
    want = ref.nodes.astype(capi.NODE_DTYPE)
    words, other = capi.pack_bases(bases)
    size = capi.find_next_prime_ref(70000000)
    # (plain = 1: the tile loop with its phases one after the other -- what k < 17 and tables of more than 960 level-1 buckets take --
    # instead of the pipelined one)
    for env, packed, batch, plain in (("1", False, 0, None), ("1", True, 0, None), ("1", True, 1 << 15, None), ("0", False, 0, None), ("1", True, 0, 1)):
        set_hooks(monkeypatch, l1_prefix=env, l1_plain=plain)
        with capi.Graph(k=k, table_slots=size, max_read_len=r, engine=capi.ENGINE_PARTITION, expected_kmers=max(len(bases), 1), max_batch_bases=batch) as g:
            if packed:
                g.push_reads_packed(words, offsets, other)
            else:
                g.push_reads(bases, offsets)
            st = g.finalize()
            tm = g.timings()
            assert (st.total_reads, st.total_kmers, st.count) == (ref.total_reads, ref.total_kmers, ref.count), (env, packed, batch)
            assert np.array_equal(g.export_sorted(), want), (shape, env, packed, batch)
            if env == "1" and batch == 0:
                assert tm.prefix_launches >= 1, "the batch did not take the prefix form"
            if env == "0":
                assert tm.prefix_launches == 0
