"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/_build/liboracle.so (the CPU restatement,
oracle/dbg_oracle.c) and a subprocess wrapper around oracle/_ref/ref_dbg (the real reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")
REF_BIN = os.path.join(HERE, "_ref", "ref_dbg")

NODE_DTYPE = np.dtype([("kmer", "<u8"), ("l_link", "<u4"), ("r_link", "<u4")])


class SynthParams(C.Structure):
    """include/dbgk_synth.h: dbgk_synth_params"""
    _fields_ = [("genome_len", C.c_uint64), ("read_len", C.c_uint32), ("sub_thr", C.c_uint32),
                ("n_thr", C.c_uint32), ("reserved", C.c_uint32), ("genome_seed", C.c_uint64),
                ("read_seed", C.c_uint64), ("err_seed", C.c_uint64)]


def synth_params(genome_len, read_len=150, sub_rate=0.005, n_rate=0.0001, cfg=2):
    """Seeds follow SURVEY.md section 8(d): genome 0xD8B6A55E0000+cfg, reads 0x5EED0000+cfg."""
    return SynthParams(genome_len, read_len, int(round(sub_rate * 2 ** 32)), int(round(n_rate * 2 ** 24)), 0,
                       0xD8B6A55E0000 + cfg, 0x5EED0000 + cfg, 0xE4404000 + cfg)


class KmerSetStruct(C.Structure):
    _fields_ = [("e_size", C.c_uint32), ("size", C.c_uint64), ("count", C.c_uint64),
                ("count_conflict", C.c_uint64), ("max", C.c_uint64), ("load_factor", C.c_float),
                ("iter_ptr", C.c_uint64), ("array", C.c_void_p), ("nul_flag", C.c_void_p),
                ("del_flag", C.c_void_p)]


class GraphParams(C.Structure):
    _fields_ = [("kmer_size", C.c_int), ("max_read_len", C.c_int), ("thread_num", C.c_int),
                ("init_hash_size", C.c_double), ("max_double_hash_times", C.c_uint64),
                ("hash_load_factor", C.c_float), ("buffer_num", C.c_int)]


class LinkStats(C.Structure):
    _fields_ = [("depth_stat", C.c_int64 * 256), ("total_nodes", C.c_int64), ("deleted_lowfreq", C.c_int64),
                ("linear_nodes", C.c_int64), ("tip_nodes", C.c_int64), ("branch_nodes", C.c_int64)]


_lib = None


def build(force=False):
    """make port (+ ref when /root/reference is present)"""
    subprocess.run(["make", "-s", "-C", HERE, "port"] + (["-B"] if force else []), check=True)
    subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    u64, vp = C.c_uint64, C.c_void_p
    L.orc_seq2bit.restype = u64
    L.orc_seq2bit.argtypes = [C.c_char_p, C.c_int]
    L.orc_bit2seq.argtypes = [u64, C.c_int, C.c_char_p]
    L.orc_count_other_bytes.restype = u64
    L.orc_count_other_bytes.argtypes = [C.c_char_p, u64]
    L.orc_rev_com_kbit.restype = u64
    L.orc_rev_com_kbit.argtypes = [u64, C.c_int]
    L.orc_pow_integer.restype = u64
    L.orc_pow_integer.argtypes = [C.c_int, C.c_int]
    L.orc_hash_code.restype = u64
    L.orc_hash_code.argtypes = [u64]
    L.orc_is_prime.argtypes = [u64]
    L.orc_find_next_prime.restype = u64
    L.orc_find_next_prime.argtypes = [u64]
    L.orc_get_next_kmer_depth.restype = C.c_uint8
    L.orc_get_next_kmer_depth.argtypes = [C.c_uint32, C.c_uint8]
    L.orc_kmerset_init.restype = C.POINTER(KmerSetStruct)
    L.orc_kmerset_init.argtypes = [u64, C.c_float]
    L.orc_kmerset_enlarge.argtypes = [C.POINTER(KmerSetStruct), u64]
    L.orc_kmerset_add_node.argtypes = [C.POINTER(KmerSetStruct), vp]
    L.orc_kmerset_exist.restype = u64
    L.orc_kmerset_exist.argtypes = [C.POINTER(KmerSetStruct), u64]
    L.orc_kmerset_free.argtypes = [C.POINTER(KmerSetStruct)]
    L.orc_kmerset_dump_sorted.restype = u64
    L.orc_kmerset_dump_sorted.argtypes = [C.POINTER(KmerSetStruct), vp]
    L.orc_nodes_digest.restype = u64
    L.orc_nodes_digest.argtypes = [vp, u64]
    L.orc_check_host_table.argtypes = [vp, vp, u64, u64]
    L.orc_parse_read.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.orc_graph_default_params.argtypes = [C.POINTER(GraphParams)]
    L.orc_graph_create.restype = vp
    L.orc_graph_create.argtypes = [C.POINTER(GraphParams)]
    L.orc_graph_add_file_mem.argtypes = [vp, vp, vp, u64]
    L.orc_graph_add_file.argtypes = [vp, C.c_char_p, C.c_int]
    L.orc_graph_finish.argtypes = [vp]
    L.orc_graph_kmerset.restype = C.POINTER(KmerSetStruct)
    L.orc_graph_kmerset.argtypes = [vp]
    for f in ("orc_graph_total_reads", "orc_graph_total_kmers", "orc_graph_double_times"):
        getattr(L, f).restype = u64
        getattr(L, f).argtypes = [vp]
    L.orc_graph_destroy.argtypes = [vp]
    L.orc_calc_link_stats.argtypes = [vp, u64, C.c_int, C.POINTER(LinkStats)]
    L.orc_read_sequences.restype = C.c_int64
    L.orc_read_sequences.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(vp)]
    L.orc_synth_fill.argtypes = [C.POINTER(SynthParams), u64, u64, vp]
    L.orc_synth_write_file.argtypes = [C.POINTER(SynthParams), u64, u64, C.c_char_p, C.c_int, C.c_int]
    _lib = L
    return L


# ------------------------------------------------------------------------------------------------
# convenience layer
# ------------------------------------------------------------------------------------------------

def pack_reads(seqs):
    """list of bytes -> (bases uint8[sum], offsets uint64[n+1])"""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.uint64, count=len(seqs))
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if len(seqs) else np.zeros(0, np.uint8)
    return bases, offsets


def count_other_bytes(bases):
    """bytes that are none of ACGTNacgtn: the reference reads out of bounds on them, this build reads them as 'A' and counts them"""
    b = np.ascontiguousarray(bases, dtype=np.uint8).tobytes()
    return int(lib().orc_count_other_bytes(b, len(b)))


def synth_reads(params, first, n_reads):
    """(bases uint8[n*L], offsets uint64[n+1]) from the counter-based generator"""
    L = params.read_len
    out = np.empty(n_reads * L, dtype=np.uint8)
    lib().orc_synth_fill(C.byref(params), first, n_reads, out.ctypes.data)
    offsets = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L))
    return out, offsets


def parse_read(seq, k, max_read_len=250):
    n_max = max(max_read_len - k + 1, 1)
    km = np.zeros(n_max, np.uint64)
    lb = np.zeros(n_max, np.uint8)
    rb = np.zeros(n_max, np.uint8)
    n = lib().orc_parse_read(seq, len(seq), k, max_read_len, km.ctypes.data, lb.ctypes.data, rb.ctypes.data)
    return km[:n].copy(), lb[:n].copy(), rb[:n].copy()


class GraphResult:
    def __init__(self, nodes, count, size, total_reads, total_kmers, double_times, conflict, max_cutoff):
        self.nodes = nodes  # sorted structured array (kmer, l_link, r_link)
        self.count = count
        self.size = size
        self.total_reads = total_reads
        self.total_kmers = total_kmers
        self.double_times = double_times
        self.conflict = conflict
        self.max = max_cutoff


def build_graph(files_mem=None, files=None, k=31, max_read_len=250, threads=1, init_hash_size=0.001,
                load_factor=0.7, max_double=10, buffer_num=10000, fmt=2, want_table=False):
    """Run the restated build_debruijn_graph.  files_mem: list of (bases, offsets) pairs, one per
    'file'; files: list of paths.  Returns GraphResult (canonical dump sorted by kmer)."""
    L = lib()
    p = GraphParams(k, max_read_len, threads, init_hash_size, max_double, load_factor, buffer_num)
    g = L.orc_graph_create(C.byref(p))
    if not g:
        raise MemoryError("orc_graph_create")
    try:
        for bases, offsets in (files_mem or []):
            bases = np.ascontiguousarray(bases, dtype=np.uint8)
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            L.orc_graph_add_file_mem(g, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1)
        for path in (files or []):
            L.orc_graph_add_file(g, os.fsencode(path), fmt)
        L.orc_graph_finish(g)
        ks = L.orc_graph_kmerset(g).contents
        nodes = np.zeros(ks.count, dtype=NODE_DTYPE)
        n = L.orc_kmerset_dump_sorted(L.orc_graph_kmerset(g), nodes.ctypes.data)
        assert n == ks.count, (n, ks.count)
        res = GraphResult(nodes, ks.count, ks.size, L.orc_graph_total_reads(g), L.orc_graph_total_kmers(g),
                          L.orc_graph_double_times(g), ks.count_conflict, ks.max)
        if want_table:
            res.table = np.ctypeslib.as_array(C.cast(ks.array, C.POINTER(C.c_uint8)),
                                              shape=(ks.size * 16,)).view(NODE_DTYPE).copy()
            res.nul_flag = np.ctypeslib.as_array(C.cast(ks.nul_flag, C.POINTER(C.c_uint8)),
                                                 shape=(ks.size // 8 + 1,)).copy()
        return res
    finally:
        L.orc_graph_destroy(g)


def nodes_digest(nodes):
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    return lib().orc_nodes_digest(nodes.ctypes.data, len(nodes))


def check_host_table(array, nul_flag, size, expect_count):
    array = np.ascontiguousarray(array, dtype=NODE_DTYPE)
    nul_flag = np.ascontiguousarray(nul_flag, dtype=np.uint8)
    return lib().orc_check_host_table(array.ctypes.data, nul_flag.ctypes.data, size, expect_count)


def link_stats(nodes, cutoff=2):
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    st = LinkStats()
    lib().orc_calc_link_stats(nodes.ctypes.data, len(nodes), cutoff, C.byref(st))
    return st


def kmer_links(array, nul_flag, cutoff=2):
    """TEST INFRASTRUCTURE: restatement of calculate_kmer_links (DBG_contig/contig.cpp:107-181) over a host table: per slot the
    2-byte KmerLink record (contig.h:31-42, as a uint16: l_link_num | l_link_base << 2 | r_link_num << 4 | r_link_base << 6 |
    linear << 8), the del_flag bitmap (MSB first), tip_nodes and branch_nodes in slot order.  PARITY UNPINNED: contig.cpp cannot
    be compiled here (Boost headers absent), this follows its text line by line."""
    array = np.ascontiguousarray(array, dtype=NODE_DTYPE)
    size = len(array)
    occ = np.unpackbits(np.ascontiguousarray(nul_flag, dtype=np.uint8))[:size].astype(bool)   # ! is_entity_null (:121)
    rec = np.zeros(size, dtype=np.uint16)
    nums = []
    for side, word in enumerate((array["l_link"].astype(np.int64), array["r_link"].astype(np.int64))):
        depth = np.stack([(word >> (24 - 8 * j)) & 0xFF for j in range(4)], axis=1)               # get_next_kmer_depth (:129,:147)
        above = depth > cutoff                                                                     # :132,:150
        num = np.minimum(above.sum(axis=1), 3)                                                     # :133-135 (2-bit field, stops at 3)
        masked = np.where(above, depth, 0)
        base = np.where(above.any(axis=1), masked.argmax(axis=1), 0)                               # :136-139 strict <: the first maximum
        rec |= ((num | (base << 2)) << (4 * side)).astype(np.uint16)
        nums.append(num)
    ln, rn = nums
    rec |= (((ln == 1) & (rn == 1)).astype(np.uint16) << 8)                                        # :170-173
    rec[~occ] = 0
    dele = occ & (ln == 0) & (rn == 0)                                                             # :165-168
    del_flag = np.packbits(np.concatenate([dele, np.zeros((-size) % 8 + 8, dtype=bool)]))[:size // 8 + 1]
    tips = np.nonzero(occ & (ln + rn == 1))[0].astype(np.uint64)                                   # :174-176
    branches = np.nonzero(occ & ((ln > 1) | (rn > 1)))[0].astype(np.uint64)                        # :177-179
    return rec, del_flag, tips, branches


def read_sequences(path, fmt):
    b, o = C.c_void_p(), C.c_void_p()
    n = lib().orc_read_sequences(os.fsencode(path), fmt, C.byref(b), C.byref(o))
    if n < 0:
        raise OSError(path)
    offsets = np.ctypeslib.as_array(C.cast(o, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
    total = int(offsets[-1])
    bases = (np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint8)), shape=(max(total, 1),))[:total].copy())
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(b)
    libc.free(o)
    return bases, offsets


def write_reads_file(path, seqs, fmt=2, gz=False):
    """one-line FASTA (fmt 2) / FASTQ (fmt 1) writer for test inputs"""
    import gzip
    op = gzip.open if gz else open
    with op(path, "wb") as fh:
        for i, s in enumerate(seqs):
            if fmt == 1:
                fh.write(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")
            else:
                fh.write(b">r%d\n" % i + s + b"\n")


# ------------------------------------------------------------------------------------------------
# the real reference (only where oracle/_ref/ref_dbg exists)
# ------------------------------------------------------------------------------------------------

def have_ref():
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def parse_dump(path):
    """text dump 'kmer<TAB>l(%08x)<TAB>r(%08x)' (header line starts with '#') -> (header dict, nodes)"""
    hdr = {}
    rows = []
    with open(path) as fh:
        for line in fh:
            if line.startswith("#"):
                t = line[1:].split()
                hdr = {t[i]: int(t[i + 1]) for i in range(0, len(t), 2)}
                continue
            a, b, c = line.split()
            rows.append((int(a), int(b, 16), int(c, 16)))
    return hdr, np.array(rows, dtype=NODE_DTYPE) if rows else np.zeros(0, NODE_DTYPE)


def write_dump(path, nodes, total_reads, total_kmers, count):
    with open(path, "w") as fh:
        fh.write("#reads %d kmers %d count %d\n" % (total_reads, total_kmers, count))
        for r in nodes:
            fh.write("%d\t%08x\t%08x\n" % (int(r["kmer"]), int(r["l_link"]), int(r["r_link"])))


def read_table_image(path):
    """raw table image written by ref_driver.cpp -T / DBGK_DUMP_TABLE: (size, count, array, nul_flag)"""
    raw = open(path, "rb").read()
    size, count = np.frombuffer(raw[:16], dtype=np.uint64)
    size, count = int(size), int(count)
    array = np.frombuffer(raw[16:16 + size * 16], dtype=NODE_DTYPE)
    flags = np.frombuffer(raw[16 + size * 16:16 + size * 16 + size // 8 + 1], dtype=np.uint8)
    return size, count, array, flags


def ref_build(lib_file, k=31, max_read_len=250, threads=1, init_hash_size=0.001, load_factor=0.7,
              max_double=10, buffer_num=10000, fmt=2, dump=None, timeout=600, image=None):
    """Run the real reference (oracle/_ref/ref_dbg build ...).  Returns its JSON summary."""
    cmd = [REF_BIN, "build", "-k", str(k), "-r", str(max_read_len), "-f", str(fmt), "-t", str(threads),
           "-i", repr(float(init_hash_size)), "-l", repr(float(load_factor)), "-e", str(max_double),
           "-b", str(buffer_num), "-q"]
    if dump:
        cmd += ["-d", dump]
    if image:
        cmd += ["-T", image]
    cmd.append(lib_file)
    out = subprocess.run(cmd, check=True, capture_output=True, timeout=timeout, text=True).stdout
    return json.loads(out.strip().splitlines()[-1])


# ------------------------------------------------------------------------------------------------
# correct_error k-mer frequency table: restated LOADERS, expected counts
# ------------------------------------------------------------------------------------------------
KFREQ_BLOCK_KMERS = 8 * 1024 * 1024  # SrcBlockSize (correct_error/main.cpp:48, main_parallel_senior.cpp:71)


def revcomp_values(v, k):
    """vectorised get_rev_com_kbit (correct_error/seqKmer.cpp:88-97) on a uint64 array"""
    x = ~np.asarray(v, dtype=np.uint64)
    for sh, m in ((2, 0x3333333333333333), (4, 0x0F0F0F0F0F0F0F0F), (8, 0x00FF00FF00FF00FF),
                  (16, 0x0000FFFF0000FFFF), (32, 0x00000000FFFFFFFF)):
        m = np.uint64(m)
        x = ((x & m) << np.uint64(sh)) | ((x & ~m) >> np.uint64(sh))
    return x >> np.uint64(64 - 2 * k)


def kfreq_expected_counts(files_mem, k, max_read_len=1000000):
    """min(255, occurrences) of every canonical k-mer, indexed by k-mer value (4^k entries); the
    extraction is the graph path's (orc_parse_read: N is A, canonical = min, tie forward)."""
    acc = np.zeros(4 ** k, dtype=np.int64)
    for bases, offsets in files_mem:
        raw = np.ascontiguousarray(bases, dtype=np.uint8).tobytes()
        for i in range(len(offsets) - 1):
            seq = raw[int(offsets[i]):int(offsets[i + 1])]
            km, _, _ = parse_read(seq, k, max_read_len)
            if len(km):
                np.add.at(acc, km.astype(np.int64), 1)
    return np.minimum(acc, 255).astype(np.uint8)


def _read_cz_blocks(path):
    import zlib
    lens = [int(x) for x in open(path + ".len").read().split()]
    data = open(path, "rb").read()
    out, pos = [], 0
    for n in lens:
        out.append(zlib.decompress(data[pos:pos + n]))
        pos += n
    return out


def kfreq_load_1bit(path, k):
    """make_kmerFreq_1bit_table_from_1BitGz_pthread (correct_error/main_parallel_senior.cpp:334-408):
    block b of the file is the bit table from byte b * 1 MiB; then every set bit i with
    i <= rc(i) sets bit rc(i) (thread_setrevcompkmer :310-329).  Returns (bits uint8[4^k/8], hifreq)."""
    total = 4 ** k
    table = np.zeros(total // 8, dtype=np.uint8)
    for b, blk in enumerate(_read_cz_blocks(path)):
        start = b * (KFREQ_BLOCK_KMERS // 8)
        table[start:start + len(blk)] = np.frombuffer(blk, dtype=np.uint8)
    idx = np.flatnonzero(np.unpackbits(table)).astype(np.uint64)
    rc = revcomp_values(idx, k)
    keep = idx <= rc
    bits = np.unpackbits(table)
    bits[rc[keep].astype(np.int64)] = 1
    return np.packbits(bits), int(keep.sum())


def kfreq_load_8bit(path, k, low_freq_cutoff):
    """make_kmerFreq_1bit_table_from_8BitGz (correct_error/main.cpp:161-220): counts > cutoff set the
    bit of idx and of rc(idx).  Returns (bits, num_total_kmers, num_effect_kmers)."""
    total = 4 ** k
    counts = np.zeros(total, dtype=np.uint8)
    for b, blk in enumerate(_read_cz_blocks(path)):
        start = b * KFREQ_BLOCK_KMERS
        counts[start:start + len(blk)] = np.frombuffer(blk, dtype=np.uint8)
    hi = np.flatnonzero(counts > low_freq_cutoff).astype(np.uint64)
    bits = np.zeros(total, dtype=np.uint8)
    bits[hi.astype(np.int64)] = 1
    bits[revcomp_values(hi, k).astype(np.int64)] = 1
    return np.packbits(bits), int(counts.astype(np.int64).sum()), int((counts > 0).sum())


def kfreq_write_cz(path, raw_bytes, block_bytes):
    """test-side writer of the .cz / .cz.len container (blocks zlib-compressed independently)"""
    import zlib
    raw = bytes(raw_bytes)
    with open(path, "wb") as fz, open(path + ".len", "w") as fl:
        for pos in range(0, len(raw), block_bytes):
            c = zlib.compress(raw[pos:pos + block_bytes])
            fz.write(c)
            fl.write("%d\n" % len(c))


def have_ref_kfreq():
    return all(os.access(os.path.join(HERE, "_ref", b), os.X_OK) for b in ("ref_kfreq1", "ref_kfreq8"))


def ref_kfreq_load(path, k, one_bit=True, threads_or_cutoff=4, timeout=600):
    """run the REAL reference loader on a .cz file -> (bits uint8[4^k/8], json stats)"""
    import tempfile
    exe = os.path.join(HERE, "_ref", "ref_kfreq1" if one_bit else "ref_kfreq8")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "bits.bin")
        r = subprocess.run([exe, path, str(k), str(threads_or_cutoff), out], check=True, capture_output=True,
                           text=True, timeout=timeout)
        return np.fromfile(out, dtype=np.uint8), json.loads(r.stdout.strip().splitlines()[-1])


# ------------------------------------------------------------------------------------------------
# link_scaffold seed index: restatement of chop_contig_to_kmerset
# (link_scaffold/map_func.cpp:119-173) over add_kmerset (link_scaffold/kmerSet.cpp:168-210).
# Pinned against the real code through oracle/_ref/ref_seed (tests/test_seedidx.py).
# ------------------------------------------------------------------------------------------------
REF_SEED = os.path.join(HERE, "_ref", "ref_seed")
SEED_DTYPE = np.dtype([("kmer", "<u8"), ("id", "<u4"), ("pos", "<u4"), ("freq", "u1"), ("direct", "u1")])
_CODE = np.zeros(256, dtype=np.uint64)  # alphabet[] (link_scaffold/seqKmer.cpp:8-24): A a N n -> 0, C c 1, G g 2, T t 3
for _ch, _v in (("C", 1), ("c", 1), ("G", 2), ("g", 2), ("T", 3), ("t", 3)):
    _CODE[ord(_ch)] = _v


def seed_index(contigs, k):
    """Every k-mer of every contig; windows never span an upper-case 'N' (scaffold_to_contig,
    map_func.cpp:303-324); canonical by strict '<' (map_func.cpp:160-166); a key keeps the
    (id, pos, direct) of its first occurrence in (contig, position) order and freq = 1 only while it
    was added once (kmerSet.cpp:179-201).  Blocks shorter than k hold no k-mer (the reference reads
    out of bounds for blocks shorter than k - 1; not exercised).  Returns nodes sorted by kmer."""
    keys, ids, poss, dirs = [], [], [], []
    mask = np.uint64((1 << (2 * k)) - 1) if k < 32 else np.uint64(0xFFFFFFFFFFFFFFFF)
    for cid, seq in enumerate(contigs):
        raw = np.frombuffer(seq.encode() if isinstance(seq, str) else seq, dtype=np.uint8)
        n = raw.size - k + 1
        if n <= 0:
            continue
        code = _CODE[raw]
        is_n = (raw == ord("N")).astype(np.int64)
        csum = np.concatenate([[0], np.cumsum(is_n)])
        ok = (csum[k:] - csum[:-k]) == 0  # no 'N' inside the window
        fwd = np.zeros(n, dtype=np.uint64)
        rc = np.zeros(n, dtype=np.uint64)
        for j in range(k):
            c = code[j:j + n]
            fwd = (fwd << np.uint64(2)) | c
            rc |= (np.uint64(3) - c) << np.uint64(2 * j)
        fwd &= mask
        direct = fwd < rc
        key = np.where(direct, fwd, rc)
        pos = np.nonzero(ok)[0]
        keys.append(key[pos]); ids.append(np.full(pos.size, cid, dtype=np.uint32))
        poss.append(pos.astype(np.uint32)); dirs.append(direct[pos].astype(np.uint8))
    if not keys:
        return np.zeros(0, dtype=SEED_DTYPE)
    key = np.concatenate(keys); cid = np.concatenate(ids); pos = np.concatenate(poss); direct = np.concatenate(dirs)
    uniq, first, counts = np.unique(key, return_index=True, return_counts=True)  # first = first occurrence in input order
    out = np.zeros(uniq.size, dtype=SEED_DTYPE)
    out["kmer"] = uniq
    out["id"] = cid[first]
    out["pos"] = pos[first]
    out["direct"] = direct[first]
    out["freq"] = (counts == 1).astype(np.uint8)
    return out


def seed_payload(nodes):
    """the reference's bit-field word {id:32, pos:30, freq:1, direct:1} (link_scaffold/kmerSet.h:54-61)"""
    return (nodes["id"].astype(np.uint64) | (nodes["pos"].astype(np.uint64) << np.uint64(32))
            | (nodes["freq"].astype(np.uint64) << np.uint64(62)) | (nodes["direct"].astype(np.uint64) << np.uint64(63)))


def seed_unpack(kmer, payload):
    out = np.zeros(len(kmer), dtype=SEED_DTYPE)
    payload = np.asarray(payload, dtype=np.uint64)
    out["kmer"] = kmer
    out["id"] = (payload & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    out["pos"] = ((payload >> np.uint64(32)) & np.uint64(0x3FFFFFFF)).astype(np.uint32)
    out["freq"] = ((payload >> np.uint64(62)) & np.uint64(1)).astype(np.uint8)
    out["direct"] = (payload >> np.uint64(63)).astype(np.uint8)
    return out


def write_contig_fasta(path, contigs, width=0):
    with open(path, "w") as f:
        for i, s in enumerate(contigs):
            f.write(">ctg%d len=%d\n" % (i, len(s)))
            if width:
                for j in range(0, len(s), width):
                    f.write(s[j:j + width] + "\n")
            else:
                f.write(s + "\n")


def parse_seed_dump(path):
    with open(path) as f:
        head = f.readline().split()
        meta = {"contigs": int(head[1]), "size": int(head[3]), "count": int(head[5])}
        rows = [tuple(int(x) for x in line.split()) for line in f]
    return meta, np.array(rows, dtype=SEED_DTYPE) if rows else np.zeros(0, dtype=SEED_DTYPE)


def have_ref_seed():
    return os.path.exists(REF_SEED)


def ref_seed(fasta, k, dump, hash_size=0, load_factor=0.5, timeout=600):
    subprocess.run([REF_SEED, fasta, str(k), str(hash_size), str(load_factor), dump], check=True, timeout=timeout,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return parse_seed_dump(dump)


# ------------------------------------------------------------------------------------------------
# WIDE key path (k <= 63, 128-bit keys): oracle/wide_oracle.cpp.  PARITY UNPINNED for k > 32 (the reference
# stops at k = 31); anchored by equality with the pinned oracle above at k <= 31 (tests/test_wide.py).
# ------------------------------------------------------------------------------------------------
NODE32_DTYPE = np.dtype([("kmer_hi", "<u8"), ("kmer_lo", "<u8"), ("l_link", "<u4"), ("r_link", "<u4"), ("reserved", "<u8")])
WIDE_LIB_PATH = os.path.join(HERE, "_build", "liboracle_wide.so")
_wide = None


def wide_lib():
    global _wide
    if _wide is None:
        if not os.path.exists(WIDE_LIB_PATH):
            build(force=True)
        L = C.CDLL(WIDE_LIB_PATH)
        L.orcw_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint64)]
        L.orcw_free.argtypes = [C.c_void_p]
        L.orcw_digest.restype = C.c_uint64
        L.orcw_digest.argtypes = [C.c_void_p, C.c_uint64]
        L.orcw_check_host_table.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        _wide = L
    return _wide


def wide_build(bases, offsets, k, max_read_len=250):
    """-> (nodes sorted by (kmer_hi, kmer_lo) with the key-0 node first, Kmer_total_num)"""
    L = wide_lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    p, n, tot = C.c_void_p(), C.c_uint64(), C.c_uint64()
    rc = L.orcw_build(bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, k, max_read_len, C.byref(p), C.byref(n), C.byref(tot))
    assert rc == 0, rc
    try:
        nodes = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value * 32,)).view(NODE32_DTYPE).copy()
    finally:
        L.orcw_free(p)
    return nodes, tot.value


def wide_digest(nodes):
    nodes = np.ascontiguousarray(nodes, dtype=NODE32_DTYPE)
    return wide_lib().orcw_digest(nodes.ctypes.data, len(nodes))


def wide_check_host_table(array, nul_flag, size, expect_count):
    array = np.ascontiguousarray(array, dtype=NODE32_DTYPE)
    nul_flag = np.ascontiguousarray(nul_flag, dtype=np.uint8)
    return wide_lib().orcw_check_host_table(array.ctypes.data, nul_flag.ctypes.data, size, expect_count)
