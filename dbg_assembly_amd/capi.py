"""ctypes binding of the C ABI in include/dbgk.h (dbg_assembly_amd/lib/libdbgk.so).

Plumbing only: every compute call goes straight into the HIP library.  There is no Python or CPU
fallback -- loading fails loudly if the library has not been built, and dbgk_create fails if no
gfx950 device is present.
"""
import ctypes as C
import os

import numpy as np

# ONE HIP runtime per process.  torch ships its own libamdhip64 / libhsa-runtime64; if libdbgk.so (linked against /opt/rocm) is
# loaded FIRST, both copies end up in the process and torch's finds no device ("No HIP GPUs are available").  With torch's loaded
# first the library binds to that copy (same soname) and everything -- zero-copy tensors over the library's buffers, pinned
# tensors recognised by dbgk_push_reads -- shares one runtime (profiles/ubench/hip_runtime_order.py shows both orders).  A C++
# caller never sees this; it only concerns Python processes that use both.
try:
    import torch  # noqa: F401  (plumbing: load order only)
except ImportError:
    pass

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DBGK_LIB") or os.path.join(HERE, "lib", "libdbgk.so")  # DBGK_LIB: A/B runs of two builds in one session

NODE_DTYPE = np.dtype([("kmer", "<u8"), ("l_link", "<u4"), ("r_link", "<u4")])
NODE32_DTYPE = np.dtype([("kmer_hi", "<u8"), ("kmer_lo", "<u8"), ("l_link", "<u4"), ("r_link", "<u4"), ("reserved", "<u8")])  # dbgk_node32

OK, ERR_ARG, ERR_HIP, ERR_TABLE_FULL, ERR_STATE, ERR_NOMEM, ERR_CAPACITY = 0, -1, -2, -3, -4, -5, -6
ENGINE_AUTO, ENGINE_DIRECT, ENGINE_PARTITION, ENGINE_KFREQ, ENGINE_SEEDIDX, ENGINE_WIDE = 0, 1, 2, 3, 4, 5
FLAG_TRACK_FIRST_SEEN = 1
FLAG_PREALLOC_STAGING = 2


class SynthParams(C.Structure):
    _fields_ = [("genome_len", C.c_uint64), ("read_len", C.c_uint32), ("sub_thr", C.c_uint32),
                ("n_thr", C.c_uint32), ("reserved", C.c_uint32), ("genome_seed", C.c_uint64),
                ("read_seed", C.c_uint64), ("err_seed", C.c_uint64)]


def synth_params(genome_len, read_len=150, sub_rate=0.005, n_rate=0.0001, cfg=2):
    """SURVEY.md section 8(d) seeds: genome 0xD8B6A55E0000+cfg, reads 0x5EED0000+cfg."""
    return SynthParams(genome_len, read_len, int(round(sub_rate * 2 ** 32)), int(round(n_rate * 2 ** 24)), 0,
                       0xD8B6A55E0000 + cfg, 0x5EED0000 + cfg, 0xE4404000 + cfg)


class Config(C.Structure):
    _fields_ = [("kmer_size", C.c_int32), ("max_read_len", C.c_int32), ("table_slots", C.c_uint64),
                ("device_id", C.c_int32), ("engine", C.c_int32), ("max_batch_bases", C.c_uint64),
                ("expected_kmers", C.c_uint64), ("shard_count", C.c_uint32), ("shard_index", C.c_uint32),
                ("flags", C.c_uint64), ("n_passes", C.c_uint64), ("reserved", C.c_uint64 * 1)]


class ShardInfo(C.Structure):
    _fields_ = [("n_ranks", C.c_uint32), ("rank", C.c_uint32), ("table_slots_global", C.c_uint64),
                ("slot_lo", C.c_uint64), ("slot_hi", C.c_uint64), ("d_send", C.c_void_p), ("d_recv", C.c_void_p),
                ("chunk_bytes", C.c_uint64), ("d_send_cnt", C.c_void_p), ("d_recv_cnt", C.c_void_p),
                ("cnt_chunk_bytes", C.c_uint64), ("buckets_per_rank", C.c_uint32), ("own_buckets", C.c_uint32),
                ("bucket_bytes", C.c_uint64), ("cnt_bucket_bytes", C.c_uint64)]


class PlanInfo(C.Structure):
    _fields_ = [("table_slots", C.c_uint64), ("r", C.c_uint32), ("level1_buckets", C.c_uint32), ("final_per_level1", C.c_uint32),
                ("three_level", C.c_uint32), ("buckets_per_rank", C.c_uint32), ("own_buckets", C.c_uint32), ("first_bucket", C.c_uint32),
                ("reserved", C.c_uint32), ("slot_lo", C.c_uint64), ("slot_hi", C.c_uint64), ("records_per_level1_bucket", C.c_uint64),
                ("records_per_final_bucket", C.c_uint64), ("table_bytes", C.c_uint64), ("level1_store_bytes", C.c_uint64),
                ("inbox_bytes", C.c_uint64), ("final_store_bytes", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("total_reads", C.c_uint64), ("total_kmers", C.c_uint64), ("stored_kmers", C.c_uint64),
                ("count", C.c_uint64), ("count_conflict", C.c_uint64), ("table_slots", C.c_uint64),
                ("polyA_l_link", C.c_uint32), ("polyA_r_link", C.c_uint32), ("other_bytes", C.c_uint64)]


class LinkStats(C.Structure):
    _fields_ = [("depth_stat", C.c_int64 * 256), ("total_nodes", C.c_int64), ("deleted_lowfreq", C.c_int64),
                ("linear_nodes", C.c_int64), ("tip_nodes", C.c_int64), ("branch_nodes", C.c_int64)]


class Timings(C.Structure):
    _fields_ = [("mark_ms", C.c_float), ("insert_ms", C.c_float), ("partition_ms", C.c_float),
                ("build_ms", C.c_float), ("fixup_ms", C.c_float), ("finalize_ms", C.c_float),
                ("insert_launches", C.c_uint64), ("l2_build_wall_ms", C.c_float), ("partition_launches", C.c_uint32),
                ("uniform_launches", C.c_uint32), ("prefix_launches", C.c_uint32), ("reserved", C.c_uint64 * 1)]


class DbgkError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        L = lib()
        msg = "%s: %s" % (what, L.dbgk_strerror(status).decode())
        if status == ERR_HIP:
            msg += " [%s]" % L.dbgk_last_error().decode()
        super().__init__(msg)


# every symbol include/dbgk.h declares: (name, restype, argtypes)
_u64, _vp, _i = C.c_uint64, C.c_void_p, C.c_int
SYMBOLS = [
    ("dbgk_create", _i, [C.POINTER(Config), C.POINTER(_vp)]),
    ("dbgk_destroy", _i, [_vp]),
    ("dbgk_reset", _i, [_vp]),
    ("dbgk_push_reads", _i, [_vp, _vp, _vp, _u64]),
    ("dbgk_push_acquire", _i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_u64), C.POINTER(_u64)]),
    ("dbgk_push_commit", _i, [_vp, _u64]),
    ("dbgk_push_reads_device", _i, [_vp, _vp, _vp, _u64, _u64]),
    ("dbgk_pack_bases", _i, [_vp, _u64, _vp, _u64, C.POINTER(_u64)]),
    ("dbgk_pack_reads", _i, [_vp, _u64, _vp, _u64, C.POINTER(_u64)]),
    ("dbgk_unpack_bases", _i, [_vp, _u64, _u64, _vp]),
    ("dbgk_push_reads_packed", _i, [_vp, _vp, _vp, _u64, _u64]),
    ("dbgk_push_commit_packed", _i, [_vp, _u64, _u64]),
    ("dbgk_push_reads_packed_device", _i, [_vp, _vp, _vp, _u64, _u64]),
    ("dbgk_pack_bases_device", _i, [_vp, _vp, _u64, _vp]),
    ("dbgk_push_reads_packed_uniform", _i, [_vp, _vp, _u64, C.c_uint32, _u64]),
    ("dbgk_push_reads_packed_uniform_device", _i, [_vp, _vp, _u64, C.c_uint32]),
    ("dbgk_finalize", _i, [_vp, C.POINTER(Stats)]),
    ("dbgk_sync", _i, [_vp]),
    ("dbgk_resize_table", _i, [_vp, _u64]),
    ("dbgk_flush", _i, [_vp]),
    ("dbgk_store_room", _i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    ("dbgk_copy_nodes_peer", _i, [_vp, _vp, _vp, _vp, _u64]),
    ("dbgk_export_host_table", _i, [_vp, _u64, _vp, _vp]),
    ("dbgk_export_host_table_links", _i, [_vp, _u64, _vp, _vp, C.c_int32, _vp, _vp, _vp, _u64, C.POINTER(_u64), _vp, _u64, C.POINTER(_u64), _vp]),
    ("dbgk_export_sorted", _i, [_vp, _vp, _u64, C.POINTER(_u64)]),
    ("dbgk_export_first_seen_order", _i, [_vp, _vp, _vp, _u64, C.POINTER(_u64)]),
    ("dbgk_digest", _i, [_vp, C.POINTER(_u64)]),
    ("dbgk_link_stats_device", _i, [_vp, C.c_int32, C.POINTER(LinkStats)]),
    ("dbgk_wide_export_sorted", _i, [_vp, _vp, _u64, C.POINTER(_u64)]),
    ("dbgk_wide_export_host_table", _i, [_vp, _u64, _vp, _vp]),
    ("dbgk_wide_partition_export", _i, [_vp, C.c_uint32, _vp, _u64, _vp]),
    ("dbgk_wide_merge_nodes", _i, [_vp, _vp, _u64]),
    ("dbgk_wide_pass_info", _i, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("dbgk_wide_begin_pass", _i, [_vp, C.c_uint32]),
    ("dbgk_wide_end_pass", _i, [_vp]),
    ("dbgk_shard_side_export", _i, [_vp, C.POINTER(_vp), C.POINTER(_u64)]),
    ("dbgk_shard_side_clear", _i, [_vp]),
    ("dbgk_seed_export_sorted", _i, [_vp, _vp, _u64, C.POINTER(_u64)]),
    ("dbgk_seed_export_host_table", _i, [_vp, _u64, _vp, _vp]),
    ("dbgk_kfreq_export_counts", _i, [_vp, _u64, _u64, _vp]),
    ("dbgk_kfreq_export_bits", _i, [_vp, C.c_uint32, _u64, _u64, _vp]),
    ("dbgk_kfreq_merge_counts", _i, [_vp, _vp, _u64, _u64]),
    ("dbgk_kfreq_device_counts", _i, [_vp, C.POINTER(_vp), C.POINTER(_u64)]),
    ("dbgk_extract_kmers", _i, [_vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    ("dbgk_partition_counts", _i, [_vp, C.c_uint32, _vp]),
    ("dbgk_partition_export", _i, [_vp, C.c_uint32, _vp, _u64]),
    ("dbgk_merge_nodes", _i, [_vp, _vp, _u64]),
    ("dbgk_refresh_stats", _i, [_vp, C.POINTER(Stats)]),
    ("dbgk_plan_partition", _i, [_u64, _u64, C.c_uint32, C.c_uint32, C.POINTER(PlanInfo)]),
    ("dbgk_shard_buffers", _i, [_vp, C.POINTER(ShardInfo)]),
    ("dbgk_shard_mark_exchanged", _i, [_vp]),
    ("dbgk_shard_plan", _i, [_vp]),
    ("dbgk_shard_build_range", _i, [_vp, C.c_uint32, C.c_uint32]),
    ("dbgk_shard_outgoing", _i, [_vp, C.POINTER(_vp), C.POINTER(_u64)]),
    ("dbgk_shard_overflow", _i, [_vp, C.POINTER(_vp), C.POINTER(_u64)]),
    ("dbgk_shard_heavy", _i, [_vp, C.POINTER(_vp), C.POINTER(_u64)]),
    ("dbgk_shard_merge", _i, [_vp, _vp, _u64, _i, _i]),
    ("dbgk_add_polyA", _i, [_vp, C.c_uint32, C.c_uint32]),
    ("dbgk_memcpy_d2d", _i, [_vp, _vp, _vp, C.c_size_t]),
    ("dbgk_comm_create", _i, [C.POINTER(Config), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(_vp)]),
    ("dbgk_comm_destroy", _i, [_vp]),
    ("dbgk_comm_size", C.c_uint32, [_vp]),
    ("dbgk_comm_handle", _vp, [_vp, C.c_uint32]),
    ("dbgk_comm_push_reads", _i, [_vp, _vp, _vp, _u64]),
    ("dbgk_comm_push_reads_packed", _i, [_vp, _vp, _vp, _u64, _u64]),
    ("dbgk_comm_flush", _i, [_vp]),
    ("dbgk_comm_refresh_stats", _i, [_vp, C.POINTER(Stats)]),
    ("dbgk_comm_finalize", _i, [_vp, C.POINTER(Stats)]),
    ("dbgk_comm_digest", _i, [_vp, C.POINTER(_u64)]),
    ("dbgk_comm_link_stats", _i, [_vp, C.c_int32, C.POINTER(LinkStats)]),
    ("dbgk_comm_export_host_table", _i, [_vp, _u64, _vp, _vp]),
    ("dbgk_comm_resize", _i, [_vp, _u64]),
    ("dbgk_comm_export_host_table_links", _i, [_vp, _u64, _vp, _vp, C.c_int32, _vp, _vp, _vp, _u64, C.POINTER(_u64), _vp, _u64, C.POINTER(_u64), _vp]),
    ("dbgk_comm_wide_export_sorted", _i, [_vp, _vp, _u64, C.POINTER(_u64)]),
    ("dbgk_comm_wide_export_host_table", _i, [_vp, _u64, _vp, _vp]),
    ("dbgk_comm_kfreq_export_counts", _i, [_vp, _u64, _u64, _vp]),
    ("dbgk_comm_kfreq_export_bits", _i, [_vp, C.c_uint32, _u64, _u64, _vp]),
    ("dbgk_synth_reads_device", _i, [_vp, C.POINTER(SynthParams), _u64, _u64, _vp, _vp]),
    ("dbgk_device_malloc", _i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    ("dbgk_device_free", _i, [_vp, _vp]),
    ("dbgk_memcpy_d2h", _i, [_vp, _vp, _vp, C.c_size_t]),
    ("dbgk_memcpy_h2d", _i, [_vp, _vp, _vp, C.c_size_t]),
    ("dbgk_get_timings", _i, [_vp, C.POINTER(Timings)]),
    ("dbgk_reset_timings", _i, [_vp]),
    ("dbgk_stream", _vp, [_vp]),
    ("dbgk_measure_copy_bandwidth", _i, [_vp, C.c_size_t, _i, C.POINTER(C.c_double)]),
    ("dbgk_measure_copy_bandwidth2", _i, [_vp, C.c_size_t, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("dbgk_measure_gather_bandwidth", _i, [_vp, C.c_size_t, _u64, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("dbgk_device_count", _i, []),
    ("dbgk_abi_version", _i, []),
    ("dbgk_strerror", C.c_char_p, [_i]),
    ("dbgk_last_error", C.c_char_p, []),
]

_lib = None


def lib():
    """Load libdbgk.so (raises if it has not been built: no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C dbg_assembly_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _chk(status, what):
    if status != OK:
        raise DbgkError(status, what)


def pack_bases(bases, out=None, first_base=0):
    """ASCII bases -> 2-bit words (dbgk_pack_bases, host): -> (uint32 words, number of bytes outside ACGTNacgtn).  With `out`
    the bases are packed into that buffer from base position first_base on (boundary words are OR-ed into)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    if out is None:
        out = np.zeros((first_base + len(bases) + 15) // 16, dtype=np.uint32)
    other = C.c_uint64(0)
    _chk(lib().dbgk_pack_bases(bases.ctypes.data, len(bases), out.ctypes.data, first_base, C.byref(other)), "dbgk_pack_bases")
    return out, other.value


class ReadRef(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("len", C.c_uint32)]


def pack_reads(reads, out, first_base=0):
    """a list of bytes objects packed back to back into `out` from base position first_base on (dbgk_pack_reads) -> other bytes"""
    refs = (ReadRef * len(reads))()
    keep = [np.frombuffer(r, dtype=np.uint8) if len(r) else np.zeros(1, np.uint8) for r in reads]
    for i, (r, k) in enumerate(zip(reads, keep)):
        refs[i].seq, refs[i].len = k.ctypes.data, len(r)
    other = C.c_uint64(0)
    _chk(lib().dbgk_pack_reads(refs, len(reads), out.ctypes.data, first_base, C.byref(other)), "dbgk_pack_reads")
    return other.value


def unpack_bases(packed, n_bases, first_base=0):
    packed = np.ascontiguousarray(packed, dtype=np.uint32)
    out = np.empty(n_bases, dtype=np.uint8)
    _chk(lib().dbgk_unpack_bases(packed.ctypes.data, first_base, n_bases, out.ctypes.data), "dbgk_unpack_bases")
    return out


class DeviceBuffer:
    """Raw device allocation owned by a Graph handle."""

    def __init__(self, graph, nbytes):
        self.graph = graph
        self.nbytes = nbytes
        p = C.c_void_p()
        _chk(lib().dbgk_device_malloc(graph._h, nbytes, C.byref(p)), "dbgk_device_malloc")
        self.ptr = p.value

    def free(self):
        if self.ptr:
            lib().dbgk_device_free(self.graph._h, self.ptr)
            self.ptr = None

    def to_host(self, dtype=np.uint8, nbytes=None):
        n = self.nbytes if nbytes is None else nbytes
        out = np.empty(n, dtype=np.uint8)
        _chk(lib().dbgk_memcpy_d2h(self.graph._h, out.ctypes.data, self.ptr, n), "dbgk_memcpy_d2h")
        return out.view(dtype)

    def from_host(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _chk(lib().dbgk_memcpy_h2d(self.graph._h, self.ptr, arr.ctypes.data, arr.nbytes), "dbgk_memcpy_h2d")


class Graph:
    """One GPU-resident k-mer graph under construction (thin wrapper over a dbgk_handle)."""

    def __init__(self, k, table_slots, max_read_len=250, device=0, engine=ENGINE_AUTO, max_batch_bases=0,
                 expected_kmers=0, shard_count=0, shard_index=0, flags=0, n_passes=0):
        self._h = None
        cfg = Config(k, max_read_len, table_slots, device, engine, max_batch_bases, expected_kmers,
                     shard_count, shard_index, flags, n_passes)
        h = C.c_void_p()
        _chk(lib().dbgk_create(C.byref(cfg), C.byref(h)), "dbgk_create")
        self._h = h
        self.k = k
        self.table_slots = table_slots
        self.stats = None

    def close(self):
        if self._h:
            lib().dbgk_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- hot path
    def push_reads(self, bases, offsets):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _chk(lib().dbgk_push_reads(self._h, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1), "dbgk_push_reads")

    def push_reads_zero_copy(self, bases, offsets):
        """the batch written straight into the handle's pinned staging buffers (dbgk_push_acquire / dbgk_push_commit), in pieces
        that fit them"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n, r0 = len(offsets) - 1, 0
        while r0 < n:
            pb, po, cb, cr = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
            _chk(lib().dbgk_push_acquire(self._h, C.byref(pb), C.byref(po), C.byref(cb), C.byref(cr)), "dbgk_push_acquire")
            base0 = int(offsets[r0])
            r1 = int(np.searchsorted(offsets, base0 + cb.value, side="right")) - 1
            r1 = min(max(r1, r0 + 1), r0 + cr.value, n)
            nb = int(offsets[r1]) - base0
            assert nb <= cb.value, "a read larger than the staging buffer"
            C.memmove(pb.value, bases.ctypes.data + base0, nb)
            rel = (offsets[r0:r1 + 1] - offsets[r0]).astype(np.uint64)
            C.memmove(po.value, rel.ctypes.data, rel.nbytes)
            _chk(lib().dbgk_push_commit(self._h, r1 - r0), "dbgk_push_commit")
            r0 = r1

    def push_reads_device(self, d_bases, d_offsets, n_reads, n_bases):
        _chk(lib().dbgk_push_reads_device(self._h, d_bases, d_offsets, n_reads, n_bases), "dbgk_push_reads_device")

    # ---- the same batches, 2 bits per base (include/dbgk.h "2-bit packed reads")
    def push_reads_packed(self, packed, offsets, other_bytes=0):
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _chk(lib().dbgk_push_reads_packed(self._h, packed.ctypes.data, offsets.ctypes.data, len(offsets) - 1, other_bytes), "dbgk_push_reads_packed")

    def push_reads_packed_ptr(self, packed_ptr, offsets_ptr, n_reads, other_bytes=0):
        """host pointers (e.g. a pinned torch tensor's data_ptr)"""
        _chk(lib().dbgk_push_reads_packed(self._h, packed_ptr, offsets_ptr, n_reads, other_bytes), "dbgk_push_reads_packed")

    def push_reads_packed_zero_copy(self, bases, offsets):
        """ASCII reads packed straight into the handle's pinned staging buffers (dbgk_push_acquire / dbgk_pack_bases /
        dbgk_push_commit_packed), in pieces that fit them -- what a reader thread of the host layer does"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n, r0 = len(offsets) - 1, 0
        while r0 < n:
            pb, po, cb, cr = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
            _chk(lib().dbgk_push_acquire(self._h, C.byref(pb), C.byref(po), C.byref(cb), C.byref(cr)), "dbgk_push_acquire")
            base0 = int(offsets[r0])
            r1 = int(np.searchsorted(offsets, base0 + cb.value, side="right")) - 1
            r1 = min(max(r1, r0 + 1), r0 + cr.value, n)
            nb = int(offsets[r1]) - base0
            assert nb <= cb.value, "a read larger than the staging buffer"
            C.memset(pb.value, 0, ((nb + 15) // 16) * 4)
            other = C.c_uint64(0)
            _chk(lib().dbgk_pack_bases(bases.ctypes.data + base0, nb, pb.value, 0, C.byref(other)), "dbgk_pack_bases")
            rel = (offsets[r0:r1 + 1] - offsets[r0]).astype(np.uint64)
            C.memmove(po.value, rel.ctypes.data, rel.nbytes)
            _chk(lib().dbgk_push_commit_packed(self._h, r1 - r0, other.value), "dbgk_push_commit_packed")
            r0 = r1

    def push_reads_packed_uniform(self, packed, n_reads, read_len, other_bytes=0):
        """n_reads reads of read_len bases each, packed back to back (host array or host pointer): no offsets"""
        ptr = packed if isinstance(packed, int) else np.ascontiguousarray(packed, dtype=np.uint32).ctypes.data
        _chk(lib().dbgk_push_reads_packed_uniform(self._h, ptr, n_reads, read_len, other_bytes), "dbgk_push_reads_packed_uniform")

    def push_reads_packed_uniform_device(self, d_packed, n_reads, read_len):
        _chk(lib().dbgk_push_reads_packed_uniform_device(self._h, d_packed, n_reads, read_len), "dbgk_push_reads_packed_uniform_device")

    def push_reads_packed_device(self, d_packed, d_offsets, n_reads, n_bases):
        _chk(lib().dbgk_push_reads_packed_device(self._h, d_packed, d_offsets, n_reads, n_bases), "dbgk_push_reads_packed_device")

    def pack_bases_device(self, d_bases, n_bases):
        """ASCII bases in device memory -> a DeviceBuffer of 2-bit words (dbgk_pack_bases_device)"""
        buf = DeviceBuffer(self, ((n_bases + 15) // 16) * 4 + 64)
        _chk(lib().dbgk_pack_bases_device(self._h, d_bases, n_bases, buf.ptr), "dbgk_pack_bases_device")
        return buf

    def finalize(self):
        st = Stats()
        _chk(lib().dbgk_finalize(self._h, C.byref(st)), "dbgk_finalize")
        self.stats = st
        return st

    def sync(self):
        _chk(lib().dbgk_sync(self._h), "dbgk_sync")

    def reset(self):
        _chk(lib().dbgk_reset(self._h), "dbgk_reset")
        self.stats = None

    def flush(self):
        _chk(lib().dbgk_flush(self._h), "dbgk_flush")

    def store_room(self):
        a, b = C.c_uint64(), C.c_uint64()
        _chk(lib().dbgk_store_room(self._h, C.byref(a), C.byref(b)), "dbgk_store_room")
        return a.value, b.value

    def resize_table(self, new_slots):
        _chk(lib().dbgk_resize_table(self._h, new_slots), "dbgk_resize_table")
        self.table_slots = new_slots

    # ---- results
    def export_sorted(self):
        n = self.stats.count
        out = np.zeros(n, dtype=NODE_DTYPE)
        got = C.c_uint64()
        _chk(lib().dbgk_export_sorted(self._h, out.ctypes.data, n, C.byref(got)), "dbgk_export_sorted")
        assert got.value == n, (got.value, n)
        return out

    def export_host_table(self, host_size=None):
        size = self.table_slots if host_size is None else host_size
        array = np.zeros(size, dtype=NODE_DTYPE)
        flags = np.zeros(size // 8 + 1, dtype=np.uint8)
        _chk(lib().dbgk_export_host_table(self._h, size, array.ctypes.data, flags.ctypes.data), "dbgk_export_host_table")
        return array, flags

    def export_host_table_links(self, cutoff=2, host_size=None):
        """dbgk_export_host_table + the consumer's whole first pass (calculate_kmer_links, contig.cpp:107-181) for that table:
        -> (array, nul_flag, klink u16[size], del_flag, tip slots, branch slots, LinkStats)"""
        size = self.table_slots if host_size is None else host_size
        array = np.zeros(size, dtype=NODE_DTYPE)
        flags = np.zeros(size // 8 + 1, dtype=np.uint8)
        klink = np.zeros(size, dtype=np.uint16)
        dele = np.zeros(size // 8 + 1, dtype=np.uint8)
        cap = int(self.stats.count)
        tips, branches = np.zeros(max(cap, 1), dtype=np.uint64), np.zeros(max(cap, 1), dtype=np.uint64)
        nt, nb = C.c_uint64(), C.c_uint64()
        st = LinkStats()
        _chk(lib().dbgk_export_host_table_links(self._h, size, array.ctypes.data, flags.ctypes.data, cutoff, klink.ctypes.data, dele.ctypes.data,
                                                tips.ctypes.data, cap, C.byref(nt), branches.ctypes.data, cap, C.byref(nb), C.byref(st)),
             "dbgk_export_host_table_links")
        return array, flags, klink, dele, tips[:nt.value], branches[:nb.value], st

    def export_first_seen_order(self):
        n = self.stats.count - 1
        nodes = np.zeros(max(n, 1), dtype=NODE_DTYPE)
        pos = np.zeros(max(n, 1), dtype=np.uint64)
        got = C.c_uint64()
        _chk(lib().dbgk_export_first_seen_order(self._h, nodes.ctypes.data, pos.ctypes.data, n, C.byref(got)),
             "dbgk_export_first_seen_order")
        return nodes[:got.value], pos[:got.value]

    def digest(self):
        d = C.c_uint64()
        _chk(lib().dbgk_digest(self._h, C.byref(d)), "dbgk_digest")
        return d.value

    def link_stats(self, cutoff=2):
        st = LinkStats()
        _chk(lib().dbgk_link_stats_device(self._h, cutoff, C.byref(st)), "dbgk_link_stats_device")
        return st

    def extract_kmers(self, bases, offsets):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nb = int(offsets[-1])
        kmer = np.zeros(nb, np.uint64)
        left = np.zeros(nb, np.uint8)
        right = np.zeros(nb, np.uint8)
        valid = np.zeros(nb, np.uint8)
        _chk(lib().dbgk_extract_kmers(self._h, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                      kmer.ctypes.data, left.ctypes.data, right.ctypes.data, valid.ctypes.data),
             "dbgk_extract_kmers")
        return kmer, left, right, valid

    # ---- multi-GPU building blocks
    def partition_counts(self, n_parts):
        counts = np.zeros(n_parts, np.uint64)
        _chk(lib().dbgk_partition_counts(self._h, n_parts, counts.ctypes.data), "dbgk_partition_counts")
        return counts

    def partition_export(self, n_parts, d_nodes, capacity):
        _chk(lib().dbgk_partition_export(self._h, n_parts, d_nodes, capacity), "dbgk_partition_export")

    def merge_nodes(self, d_nodes, n):
        _chk(lib().dbgk_merge_nodes(self._h, d_nodes, n), "dbgk_merge_nodes")

    def refresh_stats(self):
        st = Stats()
        _chk(lib().dbgk_refresh_stats(self._h, C.byref(st)), "dbgk_refresh_stats")
        self.stats = st
        return st

    # ---- WIDE engine (128-bit keys)
    def wide_export_sorted(self):
        n = int(self.stats.count)
        out = np.zeros(n, dtype=NODE32_DTYPE)
        got = C.c_uint64()
        _chk(lib().dbgk_wide_export_sorted(self._h, out.ctypes.data, n, C.byref(got)), "dbgk_wide_export_sorted")
        assert got.value == n, (got.value, n)
        return out

    def wide_export_host_table(self, size=None):
        """size: the handle's slots (a shard: the slots of its range, default the whole table)"""
        size = self.table_slots if size is None else int(size)
        array = np.zeros(size, dtype=NODE32_DTYPE)
        flags = np.zeros(size // 8 + 1, dtype=np.uint8)
        _chk(lib().dbgk_wide_export_host_table(self._h, size, array.ctypes.data, flags.ctypes.data), "dbgk_wide_export_host_table")
        return array, flags

    def wide_partition_export(self, n_parts, d_nodes=None, capacity=0):
        counts = np.zeros(n_parts, np.uint64)
        _chk(lib().dbgk_wide_partition_export(self._h, n_parts, d_nodes, capacity, counts.ctypes.data), "dbgk_wide_partition_export")
        return counts

    def wide_merge_nodes(self, d_nodes, n):
        _chk(lib().dbgk_wide_merge_nodes(self._h, d_nodes, n), "dbgk_wide_merge_nodes")

    # ---- SEEDIDX engine
    SEED_DTYPE = np.dtype([("kmer", "<u8"), ("payload", "<u8")])  # payload = {id:32, pos:30, freq:1, direct:1}

    def seed_export_sorted(self):
        n = int(self.stats.count)
        out = np.zeros(max(n, 1), dtype=self.SEED_DTYPE)
        got = C.c_uint64()
        _chk(lib().dbgk_seed_export_sorted(self._h, out.ctypes.data, n, C.byref(got)), "dbgk_seed_export_sorted")
        return out[:got.value]

    def seed_export_host_table(self, host_size):
        array = np.zeros(host_size, dtype=self.SEED_DTYPE)
        flags = np.zeros(host_size // 8 + 1, dtype=np.uint8)
        _chk(lib().dbgk_seed_export_host_table(self._h, host_size, array.ctypes.data, flags.ctypes.data), "dbgk_seed_export_host_table")
        return array, flags

    # ---- KFREQ engine
    def kfreq_counts(self, first=0, n=None):
        n = (4 ** self.k - first) if n is None else n
        out = np.zeros(n, np.uint8)
        _chk(lib().dbgk_kfreq_export_counts(self._h, first, n, out.ctypes.data), "dbgk_kfreq_export_counts")
        return out

    def kfreq_bits(self, cutoff, first_byte=0, n_bytes=None):
        n_bytes = (4 ** self.k // 8 - first_byte) if n_bytes is None else n_bytes
        out = np.zeros(n_bytes, np.uint8)
        _chk(lib().dbgk_kfreq_export_bits(self._h, cutoff, first_byte, n_bytes, out.ctypes.data), "dbgk_kfreq_export_bits")
        return out

    def kfreq_device_counts(self):
        """(device address, number of counters) of a finalized KFREQ handle"""
        ptr, n = C.c_void_p(), _u64()
        _chk(lib().dbgk_kfreq_device_counts(self._h, C.byref(ptr), C.byref(n)), "dbgk_kfreq_device_counts")
        return int(ptr.value), int(n.value)

    def kfreq_merge_counts(self, d_counts, first, n):
        """counts[first, first + n) += n counters in device memory of this GPU (saturating)"""
        _chk(lib().dbgk_kfreq_merge_counts(self._h, C.c_void_p(int(d_counts)), first, n), "dbgk_kfreq_merge_counts")

    # ---- sharded table (slot-range ownership)
    def shard_info(self):
        info = ShardInfo()
        _chk(lib().dbgk_shard_buffers(self._h, C.byref(info)), "dbgk_shard_buffers")
        return info

    def shard_mark_exchanged(self):
        _chk(lib().dbgk_shard_mark_exchanged(self._h), "dbgk_shard_mark_exchanged")

    def shard_plan(self):
        _chk(lib().dbgk_shard_plan(self._h), "dbgk_shard_plan")

    def shard_build_range(self, j0, j1):
        _chk(lib().dbgk_shard_build_range(self._h, j0, j1), "dbgk_shard_build_range")

    def shard_outgoing(self):
        p, n = C.c_void_p(), C.c_uint64()
        _chk(lib().dbgk_shard_outgoing(self._h, C.byref(p), C.byref(n)), "dbgk_shard_outgoing")
        return p.value, n.value

    def shard_overflow(self):
        p, n = C.c_void_p(), C.c_uint64()
        _chk(lib().dbgk_shard_overflow(self._h, C.byref(p), C.byref(n)), "dbgk_shard_overflow")
        return p.value, n.value

    def shard_heavy(self):
        p, n = C.c_void_p(), C.c_uint64()
        _chk(lib().dbgk_shard_heavy(self._h, C.byref(p), C.byref(n)), "dbgk_shard_heavy")
        return p.value, n.value

    def shard_merge(self, d_nodes, n, is_triple=False, from_previous_shard=False):
        _chk(lib().dbgk_shard_merge(self._h, d_nodes, n, int(is_triple), int(from_previous_shard)), "dbgk_shard_merge")

    # ---- WIDE through records: passes over the input, the side table of a shard
    def wide_pass_info(self):
        a, b = C.c_uint32(), C.c_uint32()
        _chk(lib().dbgk_wide_pass_info(self._h, C.byref(a), C.byref(b)), "dbgk_wide_pass_info")
        return a.value, b.value

    def wide_begin_pass(self, p):
        _chk(lib().dbgk_wide_begin_pass(self._h, p), "dbgk_wide_begin_pass")

    def wide_end_pass(self):
        _chk(lib().dbgk_wide_end_pass(self._h), "dbgk_wide_end_pass")

    def shard_side_export(self):
        p, n = C.c_void_p(), C.c_uint64()
        _chk(lib().dbgk_shard_side_export(self._h, C.byref(p), C.byref(n)), "dbgk_shard_side_export")
        return p.value, n.value

    def shard_side_clear(self):
        _chk(lib().dbgk_shard_side_clear(self._h), "dbgk_shard_side_clear")

    def add_polyA(self, l_link, r_link):
        _chk(lib().dbgk_add_polyA(self._h, l_link, r_link), "dbgk_add_polyA")

    def memcpy_d2d(self, dst, src, nbytes):
        _chk(lib().dbgk_memcpy_d2d(self._h, dst, src, nbytes), "dbgk_memcpy_d2d")

    # ---- utilities
    def malloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def synth_reads_device(self, params, first, n_reads):
        """-> (DeviceBuffer bases, DeviceBuffer offsets, n_bases)"""
        nb = n_reads * params.read_len
        d_bases = self.malloc(nb + 64)
        d_off = self.malloc((n_reads + 1) * 8)
        _chk(lib().dbgk_synth_reads_device(self._h, C.byref(params), first, n_reads, d_bases.ptr, d_off.ptr),
             "dbgk_synth_reads_device")
        return d_bases, d_off, nb

    def timings(self):
        t = Timings()
        _chk(lib().dbgk_get_timings(self._h, C.byref(t)), "dbgk_get_timings")
        return t

    def reset_timings(self):
        _chk(lib().dbgk_reset_timings(self._h), "dbgk_reset_timings")

    def copy_bandwidth(self, nbytes=1 << 30, iters=10):
        g = C.c_double()
        _chk(lib().dbgk_measure_copy_bandwidth(self._h, nbytes, iters, C.byref(g)), "dbgk_measure_copy_bandwidth")
        return g.value

    def copy_bandwidth_detail(self, nbytes=1 << 30, iters=10):
        """(best of the library's copy kernels and the runtime's memcpy, the runtime's hipMemcpyDtoD alone), GB/s read + written"""
        g, m = C.c_double(), C.c_double()
        _chk(lib().dbgk_measure_copy_bandwidth2(self._h, nbytes, iters, C.byref(g), C.byref(m)), "dbgk_measure_copy_bandwidth2")
        return g.value, m.value

    def gather_bandwidth(self, nbytes=16 << 30, n_accesses=1 << 30):
        """random 64-byte sectors of an nbytes buffer: (GB/s, G sectors/s)"""
        g, a = C.c_double(), C.c_double()
        _chk(lib().dbgk_measure_gather_bandwidth(self._h, nbytes, n_accesses, C.byref(g), C.byref(a)), "dbgk_measure_gather_bandwidth")
        return g.value, a.value


class Comm:
    """N sharded handles of one table inside this process (dbgk_comm_*): the C++ host layer's multi-GPU path."""

    def __init__(self, k, table_slots, devices, max_read_len=250, expected_kmers=0, max_batch_bases=0, engine=ENGINE_PARTITION):
        self._c = None
        cfg = Config(k, max_read_len, table_slots, 0, engine, max_batch_bases, expected_kmers, 0, 0, 0)
        dev = (C.c_int32 * len(devices))(*devices)
        c = C.c_void_p()
        _chk(lib().dbgk_comm_create(C.byref(cfg), dev, len(devices), C.byref(c)), "dbgk_comm_create")
        self._c = c
        self.k = k
        self.table_slots = table_slots
        self.stats = None

    def close(self):
        if self._c:
            lib().dbgk_comm_destroy(self._c)
            self._c = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def push_reads(self, bases, offsets):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _chk(lib().dbgk_comm_push_reads(self._c, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1), "dbgk_comm_push_reads")

    def export_host_table_links(self, host_size, count, cutoff=2):
        """dbgk_comm_export_host_table_links -> (array, nul_flag, klink u16[size], del_flag, tip slots, branch slots, LinkStats)"""
        array = np.zeros(host_size, dtype=NODE_DTYPE)
        flags = np.zeros(host_size // 8 + 1, dtype=np.uint8)
        klink = np.zeros(host_size, dtype=np.uint16)
        dele = np.zeros(host_size // 8 + 1, dtype=np.uint8)
        cap = int(count)
        tips, branches = np.zeros(max(cap, 1), dtype=np.uint64), np.zeros(max(cap, 1), dtype=np.uint64)
        nt, nb = C.c_uint64(), C.c_uint64()
        st = LinkStats()
        _chk(lib().dbgk_comm_export_host_table_links(self._c, host_size, array.ctypes.data, flags.ctypes.data, cutoff, klink.ctypes.data, dele.ctypes.data,
                                                     tips.ctypes.data, cap, C.byref(nt), branches.ctypes.data, cap, C.byref(nb), C.byref(st)),
             "dbgk_comm_export_host_table_links")
        return array, flags, klink, dele, tips[:nt.value], branches[:nb.value], st

    def push_reads_packed(self, packed, offsets, other_bytes=0):
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _chk(lib().dbgk_comm_push_reads_packed(self._c, packed.ctypes.data, offsets.ctypes.data, len(offsets) - 1, other_bytes), "dbgk_comm_push_reads_packed")

    def flush(self):
        _chk(lib().dbgk_comm_flush(self._c), "dbgk_comm_flush")

    def resize(self, new_slots):
        _chk(lib().dbgk_comm_resize(self._c, new_slots), "dbgk_comm_resize")
        self.table_slots = new_slots

    def refresh_stats(self):
        st = Stats()
        _chk(lib().dbgk_comm_refresh_stats(self._c, C.byref(st)), "dbgk_comm_refresh_stats")
        return st

    def finalize(self):
        st = Stats()
        _chk(lib().dbgk_comm_finalize(self._c, C.byref(st)), "dbgk_comm_finalize")
        self.stats = st
        return st

    def digest(self):
        d = C.c_uint64()
        _chk(lib().dbgk_comm_digest(self._c, C.byref(d)), "dbgk_comm_digest")
        return d.value

    def link_stats(self, cutoff=2):
        st = LinkStats()
        _chk(lib().dbgk_comm_link_stats(self._c, cutoff, C.byref(st)), "dbgk_comm_link_stats")
        return st

    def export_host_table(self, host_size=None):
        size = self.table_slots if host_size is None else host_size
        array = np.zeros(size, dtype=NODE_DTYPE)
        flags = np.zeros(size // 8 + 1, dtype=np.uint8)
        _chk(lib().dbgk_comm_export_host_table(self._c, size, array.ctypes.data, flags.ctypes.data), "dbgk_comm_export_host_table")
        return array, flags

    # ---- a communicator of WIDE handles (engine=ENGINE_WIDE, k <= 63)
    def wide_export_sorted(self):
        n = int(self.stats.count)
        out = np.zeros(n, dtype=NODE32_DTYPE)
        got = C.c_uint64()
        _chk(lib().dbgk_comm_wide_export_sorted(self._c, out.ctypes.data, n, C.byref(got)), "dbgk_comm_wide_export_sorted")
        return out[:got.value]

    def wide_export_host_table(self):
        size = self.table_slots
        array = np.zeros(size, dtype=NODE32_DTYPE)
        flags = np.zeros(size // 8 + 1, dtype=np.uint8)
        _chk(lib().dbgk_comm_wide_export_host_table(self._c, size, array.ctypes.data, flags.ctypes.data), "dbgk_comm_wide_export_host_table")
        return array, flags

    # ---- a communicator of frequency tables (engine=ENGINE_KFREQ)
    def kfreq_counts(self, first=0, n=None):
        n = 4 ** self.k - first if n is None else n
        out = np.empty(n, dtype=np.uint8)
        _chk(lib().dbgk_comm_kfreq_export_counts(self._c, first, n, out.ctypes.data), "dbgk_comm_kfreq_export_counts")
        return out

    def kfreq_bits(self, cutoff, first_byte=0, n_bytes=None):
        n_bytes = 4 ** self.k // 8 - first_byte if n_bytes is None else n_bytes
        out = np.empty(n_bytes, dtype=np.uint8)
        _chk(lib().dbgk_comm_kfreq_export_bits(self._c, cutoff, first_byte, n_bytes, out.ctypes.data), "dbgk_comm_kfreq_export_bits")
        return out


# ---- host helpers mirrored from the reference (kmerSet.cpp:72-95), needed to size tables ---------

def is_prime_ref(num):
    """kmerSet.cpp:72-81, including its float-sqrt / strict-< quirk (9, 25, 49 ... pass)."""
    if num < 4:
        return True
    if num % 2 == 0:
        return False
    bound = int(np.sqrt(np.float32(num)))
    i = 3
    while i < bound:
        if num % i == 0:
            return False
        i += 2
    return True


def find_next_prime_ref(num):
    """kmerSet.cpp:85-95"""
    if num % 2 == 0:
        num += 1
    while not is_prime_ref(num):
        num += 2
    return num


def plan_partition(table_slots, expected_kmers, shard_count=0, shard_index=0):
    """geometry of a PARTITION handle with these parameters, computed on the host (no device is touched): PlanInfo"""
    info = PlanInfo()
    _chk(lib().dbgk_plan_partition(table_slots, expected_kmers, shard_count, shard_index, C.byref(info)), "dbgk_plan_partition")
    return info
