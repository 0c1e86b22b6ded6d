"""The first-H2D region of cfg2 alone (first host-to-device copy of the packed read blocks -> final table), a few
repetitions, for a rocprofv3 --kernel-trace --memory-copy-trace timeline:  python3 profiles/tools/h2d_region.py [reps] [batch_bases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from dbg_assembly_amd import capi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n_reads, k = 10_000_000, 31
P = capi.synth_params(50_000_000, 150, sub_rate=0.005, cfg=2)
size = capi.find_next_prime_ref(600_000_000)
g = capi.Graph(k=k, table_slots=size, max_read_len=250, device=0, engine=capi.ENGINE_PARTITION, expected_kmers=n_reads * 120, max_batch_bases=batch)
d_bases, d_off, nb = g.synth_reads_device(P, 0, n_reads)
d_packed = g.pack_bases_device(d_bases.ptr, nb)
g.sync()
words = d_packed.to_host(np.uint32, ((nb + 15) // 16) * 4)
p_words = torch.empty(len(words), dtype=torch.int32).pin_memory()
p_words.numpy()[:] = words.view(np.int32)
times = []
for it in range(reps + 1):
    g.sync()
    t1 = time.perf_counter()
    g.reset()
    g.push_reads_packed_uniform(p_words.data_ptr(), n_reads, 150, 0)
    st = g.finalize()
    g.sync()
    if it:
        times.append((time.perf_counter() - t1) * 1e3)
print("count", int(st.count), "region ms", " ".join("%.3f" % t for t in times))
