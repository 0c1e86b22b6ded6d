"""Level 1 alone on cfg2's reads for tables of several sizes (= numbers of level-1 buckets), the linear form against the wave-per-bucket
form (pipelined for equal-length reads): python3 profiles/tools/l1_forms_by_buckets.py  -- insert_ms of the library's phase timers"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dbg_assembly_amd import capi

n_reads, k, L = 10_000_000, 31, 150
P = capi.synth_params(50_000_000, L, sub_rate=0.005, cfg=2)
for slots in (600_000_000, 1_200_000_000, 2_400_000_000, 4_290_000_000):
    size = capi.find_next_prime_ref(slots)
    for lin in ("0", "1"):
        os.environ["DBGK_TEST_HOOKS"] = "l1_linear=" + lin
        os.environ["DBGK_TIMINGS"] = "1"
        g = capi.Graph(k=k, table_slots=size, max_read_len=250, device=0, engine=capi.ENGINE_PARTITION, expected_kmers=n_reads * (L - k + 1))
        d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
        d_packed = g.pack_bases_device(d_bases.ptr, nb)
        d_bases.free(); d_off.free()
        g.sync()
        best = None
        for it in range(3):
            g.reset(); g.sync()
            g.push_reads_packed_uniform_device(d_packed.ptr, n_reads, L)
            st = g.finalize(); g.sync()
            tm = g.timings()
            best = tm.insert_ms if best is None else min(best, tm.insert_ms)
        print("slots %11d  n1 %4d  l1_linear=%s  level 1 %.2f ms  (count %d)" % (size, (size + (1 << 22) - 1) >> 22, lin, best, int(st.count)), flush=True)
        g.close()
