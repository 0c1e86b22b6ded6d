"""BASELINE configs[2] (cfg3: 200 M x 150 bp reads of a 1 Gb genome, k = 31, ONE table of 8.6 G slots over 8 GPUs) at what one
GPU can hold -- the workload generator `synth cfg = 3`, the geometry of the stated job, no CPU oracle at these sizes:

  * one GPU's share exactly as `bench.py --config cfg3` runs it (25 M reads of a 125 Mb genome, 1.075 G slots): the
    PARTITION engine == the DIRECT engine (count, k-mer total, node digest, link-depth histogram);
  * the 8-rank geometry: 8 slot-range shards of the 8.6 G-slot table (64-bit hash / size division, level-2 fan-out 4096:
    the three-level partition) on ONE GPU through the C++ communicator, reduced reads per rank so that 8 shards of 17 GB
    fit one card, == the single-handle build of the same reads.
The N = 8 run on hardware is the driver's (SCALE); these pin everything but the wires."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def capi():
    from dbg_assembly_amd import capi as c
    assert c.lib().dbgk_device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback exists)"
    return c


def _summary(g, st):
    return (int(st.count), int(st.total_kmers), int(st.stored_kmers), g.digest(), [int(x) for x in g.link_stats(2).depth_stat])


@pytest.mark.gpu
def test_cfg3_share_of_one_gpu_partition_equals_direct(capi):
    n_reads, genome, slots, k = 25_000_000, 125_000_000, 1_075_000_000, 31
    size = capi.find_next_prime_ref(slots)
    P = capi.synth_params(genome, 150, cfg=3)
    got = {}
    for name, engine, expected in (("partition", capi.ENGINE_PARTITION, n_reads * 120), ("direct", capi.ENGINE_DIRECT, 0)):
        with capi.Graph(k=k, table_slots=size, max_read_len=250, engine=engine, expected_kmers=expected) as g:
            d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
            g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
            st = g.finalize()
            got[name] = _summary(g, st)
            d_bases.free()
            d_off.free()
    assert got["partition"][2] == n_reads * 120
    assert got["partition"] == got["direct"]
    assert 0.25 * size < got["partition"][0] < 0.7 * size   # the load the stated job puts on its table


@pytest.mark.gpu
def test_cfg3_eight_rank_geometry_on_one_gpu_equals_single_handle(capi):
    world, n_reads, genome, slots, k = 8, 2_000_000, 80_000_000, 8_600_000_000, 31
    size = capi.find_next_prime_ref(slots)
    assert size >= 1 << 33
    P = capi.synth_params(genome, 150, cfg=3)
    parts = []
    with capi.Graph(k=k, table_slots=1009, engine=capi.ENGINE_DIRECT) as tmp:
        for r in range(world):
            d_bases, d_off, nb = tmp.synth_reads_device(P, r * n_reads, n_reads)
            parts.append((d_bases.to_host(np.uint8, nb).copy(), d_off.to_host(np.uint64).copy()))
            d_bases.free()
            d_off.free()
    with capi.Comm(k=k, table_slots=size, devices=[0] * world, expected_kmers=n_reads * 120, max_batch_bases=512 << 20) as c:
        for bases, offsets in parts:   # one push per rank: round robin puts rank r's reads on shard r
            c.push_reads(bases, offsets)
        st = c.finalize()
        sharded = (int(st.count), int(st.total_kmers), c.digest(), [int(x) for x in c.link_stats(2).depth_stat])
    one = capi.find_next_prime_ref(600_000_000)
    with capi.Graph(k=k, table_slots=one, engine=capi.ENGINE_PARTITION, expected_kmers=n_reads * 120 * world, max_batch_bases=512 << 20) as g:
        for bases, offsets in parts:
            g.push_reads(bases, offsets)
        st = g.finalize()
        single = (int(st.count), int(st.total_kmers), g.digest(), [int(x) for x in g.link_stats(2).depth_stat])
    assert sharded == single
    assert single[1] == world * n_reads * 120
