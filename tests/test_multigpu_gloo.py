"""CPU, world_size 2, gloo: the collective plumbing of dbg_assembly_amd/multigpu.py (bucket-count
all-reduce, all-to-all of aggregated nodes, owner merge, scalar all-reduce) driven by an
oracle-backed engine.  The product engine (HipEngine) is exercised on the GPU box; this test
covers sharding, owner function, split sizes and the key-0 bookkeeping for N > 1."""
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

NODE = np.dtype([("kmer", "<u8"), ("l_link", "<u4"), ("r_link", "<u4")])


def _sat_merge(nodes):
    """aggregate duplicate keys with per-byte saturating add (numpy restatement, test-only)"""
    if len(nodes) == 0:
        return nodes
    nodes = np.sort(nodes, order="kmer")
    keys, start = np.unique(nodes["kmer"], return_index=True)
    b = nodes.view(np.uint8).reshape(-1, 16)[:, 8:].astype(np.uint32)
    sums = np.minimum(np.add.reduceat(b, start, axis=0), 255).astype(np.uint8)
    out = np.zeros(len(keys), NODE)
    out["kmer"] = keys
    out.view(np.uint8).reshape(-1, 16)[:, 8:] = sums
    return out


class OracleEngine:
    """test double with the HipEngine interface, computing with the oracle on numpy/torch-CPU"""
    device = "cpu"

    def __init__(self, O, bases, offsets, k):
        self.O = O
        self.res = O.build_graph(files_mem=[(bases, offsets)], k=k, init_hash_size=0.001, threads=1)
        self.nodes = self.res.nodes.astype(NODE)

        class S:
            pass
        self.stats = S()
        self.stats.total_reads, self.stats.total_kmers = self.res.total_reads, self.res.total_kmers
        self.stats.stored_kmers, self.stats.count = self.res.total_kmers, self.res.count

    def local_stats(self):
        return self.stats

    def _owner(self, n_parts):
        L = self.O.lib()
        own = np.array([(L.orc_hash_code(int(k)) >> 32) % n_parts for k in self.nodes["kmer"]], dtype=np.int64)
        own[self.nodes["kmer"] == 0] = 0
        return own

    def partition_counts(self, n_parts):
        return np.bincount(self._owner(n_parts), minlength=n_parts).astype(np.int64)

    def partition_export(self, n_parts, total):
        order = np.argsort(self._owner(n_parts), kind="stable")
        return torch.from_numpy(self.nodes[order].view(np.uint8).copy())

    def new_buffer(self, n_nodes):
        return torch.zeros(max(n_nodes, 1) * 16, dtype=torch.uint8)

    def reset_table(self):
        self.nodes = np.zeros(0, NODE)

    def merge(self, buf, n_nodes):
        got = buf.numpy()[:n_nodes * 16].view(NODE)
        self.nodes = _sat_merge(np.concatenate([self.nodes, got]))

    def finish(self):
        class S:
            pass
        s = S()
        has0 = bool(len(self.nodes) and self.nodes["kmer"][0] == 0)
        s.count = len(self.nodes) + (0 if has0 else 1)  # every handle reports a key-0 node
        return s

    def sync(self):
        pass


def _worker(rank, world, port, reads, k, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as O
    from dbg_assembly_amd.multigpu import exchange_and_merge
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = reads[rank::world]  # shard by record
    eng = OracleEngine(O, *O.pack_reads(mine), k)
    out = exchange_and_merge(eng)
    q.put((rank, out, eng.nodes.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_and_merge_gloo(oracle, world):
    rng = random.Random(5)
    g = "".join(rng.choice("ACGT") for _ in range(3000))
    reads = []
    for _ in range(600):
        s = rng.randint(0, 3000 - 100)
        reads.append(g[s:s + 100].encode())
    reads += [b"A" * 100] * 300 + [b"T" * 50] * 10  # key-0 node with saturating links on several ranks
    rng.shuffle(reads)
    k = 21
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, reads, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = oracle.build_graph(files_mem=[oracle.pack_reads(reads)], k=k, init_hash_size=0.001)
    merged = np.concatenate([np.frombuffer(b, dtype=NODE) for _, _, b in results])
    merged = np.sort(merged, order="kmer")
    assert np.array_equal(merged, whole.nodes.astype(NODE))  # disjoint owners, exact saturating merge
    for rank, out, b in results:
        assert out["count"] == whole.count
        assert out["total_reads"] == whole.total_reads and out["total_kmers"] == whole.total_kmers
        owned = np.frombuffer(b, dtype=NODE)
        L = oracle.lib()
        for key in owned["kmer"][:200]:
            assert (0 if key == 0 else (L.orc_hash_code(int(key)) >> 32) % world) == rank


# ------------------------------------------------------------------------------------------------
# slot-range ownership flow (multigpu.sharded_finalize) with a host-memory stand-in for capi.Graph
# ------------------------------------------------------------------------------------------------
import ctypes as C


class FakeShardGraph:
    """capi.Graph look-alike on host memory: 'records' are 16-byte observations {kmer, lb | rb << 8}
    bucketed by slot-range owner; finalize aggregates what arrived with numpy.  Only the interface
    sharded_finalize uses is provided."""

    B = 5          # level-1 buckets per rank
    CAP = 1 << 13  # records per bucket

    def __init__(self, O, reads, k, world, rank, size):
        self.O, self.world, self.rank, self.size = O, world, rank, size
        L = O.lib()
        span = -(-size // world)
        sub = -(-span // self.B)
        self.slot_lo, self.slot_hi = rank * span, min(size, (rank + 1) * span)
        self.send = np.zeros((world, self.B, self.CAP), NODE)
        self.send_cnt = np.zeros((world, self.B), np.uint32)
        self.recv = np.zeros((world, self.B, self.CAP), NODE)
        self.recv_cnt = np.zeros((world, self.B), np.uint32)
        self.calls, self.built = [], []
        self.polyA = np.zeros(8, np.int64)
        self.total_reads, self.total_kmers = len(reads), 0
        for seq in reads:
            km, lb, rb = O.parse_read(seq, k)
            self.total_kmers += len(km)
            for key, l, r in zip(km.tolist(), lb.tolist(), rb.tolist()):
                if key == 0:
                    if l != 4:
                        self.polyA[l] += 1
                    if r != 4:
                        self.polyA[4 + r] += 1
                    continue
                slot = L.orc_hash_code(key) % size
                d = slot // span
                j = (slot - d * span) // sub
                i = int(self.send_cnt[d, j])
                self.send[d, j, i] = (key, l | (r << 8), 0)
                self.send_cnt[d, j] = i + 1
        self.nodes = np.zeros(0, NODE)
        self.extra = []

    def shard_info(self):
        class I:
            pass
        i = I()
        i.n_ranks, i.rank = self.world, self.rank
        i.chunk_bytes, i.cnt_chunk_bytes = self.B * self.CAP * 16, self.B * 4
        i.buckets_per_rank, i.own_buckets = self.B, self.B
        i.bucket_bytes, i.cnt_bucket_bytes = self.CAP * 16, 4
        i.d_send, i.d_recv = self.send.ctypes.data, self.recv.ctypes.data
        i.d_send_cnt, i.d_recv_cnt = self.send_cnt.ctypes.data, self.recv_cnt.ctypes.data
        return i

    def sync(self):
        pass

    def shard_mark_exchanged(self):
        self.exchanged = True

    def shard_plan(self):
        self.calls.append("plan")

    def _take(self, j0, j1):
        for j in range(j0, j1):
            self.built.append(np.concatenate([self.recv[s, j, :int(self.recv_cnt[s, j])] for s in range(self.world)]))

    def shard_build_range(self, j0, j1):
        assert self.calls and self.calls[0] == "plan" and j0 == sum(b - a for _, a, b in self.calls[1:])  # in order, once
        self.calls.append(("range", j0, j1))
        self._take(j0, j1)  # what has arrived by now is what gets built: a late transfer would lose records

    def _aggregate(self, triples):
        if len(triples) == 0:
            return np.zeros(0, NODE)
        keys, inv = np.unique(triples["kmer"], return_inverse=True)
        cnt = np.zeros((len(keys), 8), np.int64)
        code = triples["l_link"]
        lb, rb = code & 0xFF, (code >> 8) & 0xFF
        for side, b in ((0, lb), (4, rb)):
            ok = b != 4
            np.add.at(cnt, (inv[ok], side + b[ok].astype(np.int64)), 1)
        return keys, np.minimum(cnt, 255)

    def finalize(self):
        assert self.exchanged
        done = sum(b - a for _, a, b in self.calls[1:])
        self._take(done, self.B)  # whatever the caller did not build by ranges
        got = np.concatenate(self.built)
        keys, cnt = self._aggregate(got)
        self.keys, self.cnt = keys, cnt

        class S:
            pass
        st = S()
        st.total_reads, st.total_kmers, st.stored_kmers = self.total_reads, self.total_kmers, self.total_kmers
        pa = np.minimum(self.polyA, 255)
        st.polyA_l_link = int(pa[0] << 24 | pa[1] << 16 | pa[2] << 8 | pa[3])
        st.polyA_r_link = int(pa[4] << 24 | pa[5] << 16 | pa[6] << 8 | pa[7])
        self.pa_links = [st.polyA_l_link, st.polyA_r_link]
        return st

    def shard_outgoing(self):
        return 0, 0

    def shard_overflow(self):
        return 0, 0

    def shard_heavy(self):
        return 0, 0

    def add_polyA(self, l, r):
        for j, v in enumerate((l, r)):
            a = np.array([(self.pa_links[j] >> s) & 0xFF for s in (24, 16, 8, 0)]) + np.array([(v >> s) & 0xFF for s in (24, 16, 8, 0)])
            a = np.minimum(a, 255)
            self.pa_links[j] = int(a[0] << 24 | a[1] << 16 | a[2] << 8 | a[3])

    def refresh_stats(self):
        class S:
            pass
        s = S()
        s.count = len(self.keys) + (1 if self.rank == 0 else 0)
        return s

    def result_nodes(self):
        out = np.zeros(len(self.keys) + (1 if self.rank == 0 else 0), NODE)
        z = 1 if self.rank == 0 else 0
        if z:
            out[0] = (0, self.pa_links[0], self.pa_links[1])
        out["kmer"][z:] = self.keys
        c = self.cnt
        out["l_link"][z:] = c[:, 0] << 24 | c[:, 1] << 16 | c[:, 2] << 8 | c[:, 3]
        out["r_link"][z:] = c[:, 4] << 24 | c[:, 5] << 16 | c[:, 6] << 8 | c[:, 7]
        return out


def _wrap_host(ptr, nbytes, device):
    return torch.frombuffer((C.c_uint8 * int(nbytes)).from_address(int(ptr)), dtype=torch.uint8)


def _shard_worker(rank, world, port, reads, k, size, q, chunks):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as O
    from dbg_assembly_amd.multigpu import sharded_finalize
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = FakeShardGraph(O, reads[rank::world], k, world, rank, size)
    out = sharded_finalize(g, "cpu", wrap=_wrap_host, exchange_chunks=chunks, verify_exchange=True)
    out["ranged_calls"] = len(g.calls)
    q.put((rank, out, g.result_nodes().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks", [(2, 1), (2, 3), (3, 8)])
def test_sharded_finalize_gloo(oracle, world, chunks):
    rng = random.Random(9)
    g = "".join(rng.choice("ACGT") for _ in range(2500))
    reads = []
    for _ in range(500):
        s = rng.randint(0, 2500 - 90)
        reads.append(g[s:s + 90].encode())
    reads += [b"A" * 90] * 280 + [b"T" * 40] * 7
    rng.shuffle(reads)
    k, size = 21, 1000003
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, reads, k, size, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = oracle.build_graph(files_mem=[oracle.pack_reads(reads)], k=k, init_hash_size=0.001)
    merged = np.sort(np.concatenate([np.frombuffer(b, dtype=NODE) for _, _, b in results]), order="kmer")
    assert np.array_equal(merged, whole.nodes.astype(NODE))
    for rank, out, _ in results:
        assert (out["count"], out["total_reads"], out["total_kmers"]) == (whole.count, whole.total_reads, whole.total_kmers)
        assert (out["ranged_calls"] > 1) == (chunks > 1)  # the pieces were built as they arrived


# ---- kfreq_reduce: partial k-mer frequency tables -> one table, ranges owned by rank ----------

class FakeKfreqGraph:
    """a finalized KFREQ handle on the host: the oracle's counts of this rank's reads"""

    def __init__(self, O, reads, k):
        self.table = np.ascontiguousarray(O.kfreq_expected_counts([O.pack_reads(reads)], k))
        self.merges = 0

    def kfreq_device_counts(self):
        return self.table.ctypes.data, self.table.size

    def kfreq_merge_counts(self, ptr, first, n):
        other = np.frombuffer((C.c_uint8 * n).from_address(ptr), dtype=np.uint8)
        self.table[first:first + n] = np.minimum(self.table[first:first + n].astype(np.int32) + other, 255)
        self.merges += 1

    def sync(self):
        pass


def _kfreq_worker(rank, world, port, reads, k, chunk, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as O
    from dbg_assembly_amd.multigpu import kfreq_reduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = FakeKfreqGraph(O, reads[rank::world], k)
    lo, hi = kfreq_reduce(g, "cpu", wrap=_wrap_host, chunk_bytes=chunk)
    q.put((rank, lo, hi, g.table[lo:hi].tobytes(), g.merges))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunk", [(2, 256 << 20), (3, 4096), (2, 1024)])
def test_kfreq_reduce_gloo(oracle, world, chunk):
    rng = random.Random(world * 7 + chunk)
    g = "".join(rng.choice("ACGT") for _ in range(3000))
    reads = []
    for _ in range(900):
        s = rng.randint(0, 3000 - 80)
        reads.append(g[s:s + 80].encode())
    reads += [b"A" * 80] * 70 + [b"ACGTTGCA" * 10] * 40 + [b"AC"]  # counters that saturate only in the sum
    rng.shuffle(reads)
    k = 8
    want = oracle.kfreq_expected_counts([oracle.pack_reads(reads)], k)
    assert want.max() == 255
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_kfreq_worker, args=(r, world, port, reads, k, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][1] == 0 and results[-1][2] == 4 ** k
    for a, b in zip(results, results[1:]):
        assert a[2] == b[1]  # the ranges tile the table
    got = np.concatenate([np.frombuffer(r[3], dtype=np.uint8) for r in results])
    assert np.array_equal(got, want)
    if chunk < 4 ** k // world:
        assert all(r[4] > world - 1 for r in results)  # several rounds of chunks


# ------------------------------------------------------------------------------------------------
# the slot-range flow for 128-bit keys (multigpu.wide_sharded_build): passes over the input, 32-byte list entries, the
# side table of keys with a zero low word gathered onto rank 0 -- with a host-memory stand-in whose arithmetic is the
# independent checker's (tests/wide_checker.py: strings and Python ints)
# ------------------------------------------------------------------------------------------------
NODE32 = np.dtype([("kmer_hi", "<u8"), ("kmer_lo", "<u8"), ("l_link", "<u4"), ("r_link", "<u4"), ("reserved", "<u8")])


def _py_hash_code(k):
    M = (1 << 64) - 1
    k = (k + (~(k << 32) & M)) & M
    k ^= k >> 22
    k = (k + (~(k << 13) & M)) & M
    k ^= k >> 8
    k = (k + (k << 3)) & M
    k ^= k >> 15
    k = (k + (~(k << 27) & M)) & M
    k ^= k >> 31
    return k


class FakeWideShardGraph:
    """capi.Graph look-alike for a sharded WIDE handle on host memory.  'Records' are 32-byte observations {hi, lo, lb, rb};
    pass p of P keeps own-bucket indices [p * Bp, (p + 1) * Bp) of every rank.  Only what wide_sharded_build uses."""
    B, P, CAP = 6, 2, 1 << 13

    def __init__(self, W, reads, k, r, world, rank, size):
        self.W, self.world, self.rank, self.size = W, world, rank, size
        self.Bp = -(-self.B // self.P)
        self.span = -(-size // world)
        self.sub = -(-self.span // self.B)
        self.obs, self.total_kmers = W.observations(reads, k, r)
        self.total_reads = len(reads)
        self.done = 0
        self.open = False
        self.agg = {}          # key -> [[4], [4]] raw counts of my slot range
        self.side = {}         # keys with a zero low word (incl. key 0): raw counts, saturated at export
        self.pushed = 0
        self.send = np.zeros((world, self.Bp, self.CAP), NODE32)
        self.send_cnt = np.zeros((world, self.Bp), np.uint32)
        self.recv = np.zeros((world, self.Bp, self.CAP), NODE32)
        self.recv_cnt = np.zeros((world, self.Bp), np.uint32)

    def wide_pass_info(self):
        return self.P, self.done

    def wide_begin_pass(self, p):
        assert p == self.done and not self.open
        self.open, self.pass_, self.calls, self.taken = True, p, [], 0
        self.send_cnt[:] = 0
        self.recv_cnt[:] = 0

    def push_all(self):
        """what push_all(g) of the caller does: every observation of this rank's reads, once per pass"""
        self.pushed += 1
        M = (1 << 64)
        for key, lb, rb in self.obs:
            hi, lo = key // M, key % M
            if lo == 0:   # side table / key 0: never a record, counted in pass 0 only
                if self.pass_ == 0:
                    c = self.side.setdefault(key, [[0] * 4, [0] * 4])
                    if lb is not None:
                        c[0][lb] += 1
                    if rb is not None:
                        c[1][rb] += 1
                continue
            h = _py_hash_code(lo ^ _py_hash_code(hi)) if hi else _py_hash_code(lo)
            slot = h % self.size
            d = slot // self.span
            j = (slot - d * self.span) // self.sub
            jj = j - self.pass_ * self.Bp
            if not 0 <= jj < self.Bp:
                continue
            i = int(self.send_cnt[d, jj])
            self.send[d, jj, i] = (hi, lo, 4 if lb is None else lb, 4 if rb is None else rb, 0)
            self.send_cnt[d, jj] = i + 1

    def shard_info(self):
        class I:
            pass
        i = I()
        i.n_ranks, i.rank = self.world, self.rank
        i.chunk_bytes, i.cnt_chunk_bytes = self.Bp * self.CAP * 32, self.Bp * 4
        i.buckets_per_rank = self.Bp
        i.own_buckets = max(0, min(self.Bp, self.B - self.pass_ * self.Bp))
        i.bucket_bytes, i.cnt_bucket_bytes = self.CAP * 32, 4
        i.d_send, i.d_recv = self.send.ctypes.data, self.recv.ctypes.data
        i.d_send_cnt, i.d_recv_cnt = self.send_cnt.ctypes.data, self.recv_cnt.ctypes.data
        return i

    def sync(self):
        pass

    def shard_plan(self):
        self.calls.append("plan")

    def _take(self, j0, j1):
        for jj in range(j0, j1):
            for s in range(self.world):
                for rec in self.recv[s, jj, :int(self.recv_cnt[s, jj])]:
                    key = (int(rec["kmer_hi"]) << 64) | int(rec["kmer_lo"])
                    c = self.agg.setdefault(key, [[0] * 4, [0] * 4])
                    if rec["l_link"] != 4:
                        c[0][int(rec["l_link"])] += 1
                    if rec["r_link"] != 4:
                        c[1][int(rec["r_link"])] += 1
        self.taken = j1

    def shard_build_range(self, j0, j1):
        assert self.calls and self.calls[0] == "plan" and j0 == self.taken   # in order, once
        self.calls.append(("range", j0, j1))
        self._take(j0, j1)   # what has arrived by now is what gets built: a late transfer would lose records

    def shard_mark_exchanged(self):
        self.exchanged = True

    def wide_end_pass(self):
        assert self.open and self.exchanged
        self._take(self.taken, self.shard_info().own_buckets)
        self.open, self.exchanged = False, False
        self.done += 1

    def finalize(self):
        assert self.done == self.P and not self.open

        class S:
            pass
        st = S()
        st.total_reads, st.total_kmers, st.stored_kmers = self.total_reads, self.total_kmers, len(self.obs)
        return st

    def shard_overflow(self):
        return 0, 0

    def shard_outgoing(self):
        return 0, 0

    def shard_heavy(self):
        return 0, 0

    def _node(self, key, c):
        sat = [[min(255, x) for x in side] for side in c]
        return (key >> 64, key & ((1 << 64) - 1), self.W.link_word(sat[0]), self.W.link_word(sat[1]), 0)

    def shard_side_export(self):
        keys = [0] + sorted(k for k in self.side if k)
        self._side_buf = np.zeros(len(keys), NODE32)
        for i, key in enumerate(keys):
            self._side_buf[i] = self._node(key, self.side.get(key, [[0] * 4, [0] * 4]))
        return self._side_buf.ctypes.data, len(keys)

    def wide_merge_nodes(self, ptr, n):
        got = np.frombuffer((C.c_uint8 * (int(n) * 32)).from_address(int(ptr)), dtype=NODE32)
        for nd in got:
            key = (int(nd["kmer_hi"]) << 64) | int(nd["kmer_lo"])
            assert nd["kmer_lo"] == 0
            c = self.side.setdefault(key, [[0] * 4, [0] * 4])
            mine = self._node(key, c)   # saturated first, then added per byte, saturating: exact for any split
            for side, word, add in ((0, mine[2], int(nd["l_link"])), (1, mine[3], int(nd["r_link"]))):
                c[side] = [min(255, ((word >> s) & 0xFF) + ((add >> s) & 0xFF)) for s in (24, 16, 8, 0)]

    def shard_side_clear(self):
        self.side = {}

    def refresh_stats(self):
        class S:
            pass
        s = S()
        s.count = len(self.agg) + len([k for k in self.side if k]) + (1 if self.rank == 0 else 0)
        return s

    def result_nodes(self):
        rows = [self._node(k, c) for k, c in self.agg.items()] + [self._node(k, c) for k, c in self.side.items() if k]
        if self.rank == 0:
            rows.append(self._node(0, self.side.get(0, [[0] * 4, [0] * 4])))
        out = np.zeros(len(rows), NODE32)
        for i, row in enumerate(rows):
            out[i] = row
        return out


def _wide_worker(rank, world, port, reads, k, r, size, q, chunks):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import wide_checker as W
    from dbg_assembly_amd.multigpu import wide_sharded_build
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = FakeWideShardGraph(W, reads[rank::world], k, r, world, rank, size)
    out = wide_sharded_build(g, "cpu", lambda h: h.push_all(), wrap=_wrap_host, exchange_chunks=chunks, verify_exchange=True)
    out["pushes"] = g.pushed
    q.put((rank, out, g.result_nodes().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks", [(2, 1), (3, 3)])
def test_wide_sharded_build_gloo_PARITY_UNPINNED_above_k32(world, chunks):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import wide_checker as W
    rng = random.Random(63)
    g = "".join(rng.choice("ACGT") for _ in range(2500))
    reads = []
    for _ in range(150):
        s = rng.randint(0, 2500 - 130)
        reads.append(g[s:s + rng.randint(60, 130)].encode())
    reads += [b"A" * 100] * 300 + [b"T" * 70] * 10 + [b"G" + b"A" * 80 + b"C"] * 3 + [b"ACGTTGCA" * 12] * 270   # key 0, zero low words, saturation
    rng.shuffle(reads)
    k, r, size = 63, 120, 1000003
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wide_worker, args=(rk, world, port, reads, k, r, size, q, chunks)) for rk in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nodes, total = W.build(reads, k, r)
    want = W.as_sorted_nodes(nodes)
    merged = np.concatenate([np.frombuffer(b, dtype=NODE32) for _, _, b in results])
    merged = np.sort(merged, order=["kmer_hi", "kmer_lo"])
    assert np.array_equal(merged, want)   # disjoint slot ranges, the side table once, exact saturating merge
    assert ((want["kmer_lo"] == 0) & (want["kmer_hi"] != 0)).any()          # keys of the side table took part
    assert max(int(((want["l_link"] >> s) & 0xFF).max()) for s in (0, 8, 16, 24)) == 255   # and saturating counters
    for rank, out, _ in results:
        assert out["count"] == len(want) and out["total_kmers"] == total and out["total_reads"] == len(reads)
        assert out["passes"] == FakeWideShardGraph.P and out["pushes"] == FakeWideShardGraph.P   # the input is read once per pass


def _corrupt_worker(rank, world, port, reads, k, size, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as O
    import dbg_assembly_amd.multigpu as M
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = FakeShardGraph(O, reads[rank::world], k, world, rank, size)
    real = M._exchange_range

    def truncated(send, recv, info, j0, j1, world_, rank_, group, name="level-1 exchange"):   # rank 1 "receives" only part of what rank 0 sent it
        real(send, recv, info, j0, j1, world_, rank_, group, name)
        if rank_ == 1:
            lo = 0 * info.chunk_bytes + j0 * info.bucket_bytes
            recv[lo + 16:lo + 64] = 0
    M._exchange_range = truncated
    try:
        M.sharded_finalize(g, "cpu", wrap=_wrap_host, exchange_chunks=1, verify_exchange=True)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_a_truncated_exchange_fails_loudly_on_every_rank():
    """multi-GPU hardening: what a rank received is checksummed against what its peers sent; a transfer that lost bytes stops ALL
    ranks with an error instead of building a wrong table (or leaving the other ranks waiting)"""
    rng = random.Random(11)
    g = "".join(rng.choice("ACGT") for _ in range(3000))
    reads = [g[s:s + 100].encode() for s in (rng.randint(0, 2900) for _ in range(400))]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_corrupt_worker, args=(r, world, port, reads, 21, 100003, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "differ from what was sent" in results[1] and "[0]" in results[1]
    assert "differ from what was sent" in results[0] and "seen by another rank" in results[0]


# ---- the ring hand-off protocol and its failure agreement (multigpu._hand_offs) --------------------------------------------
class _RingGraph:
    """a scripted shard: node v settles in shard v % world and passes through every other one (it is appended to that shard's
    outgoing list again), the way a node passes through a shard that is full behind its first slots; `fail_on_merge`: the
    first merge on this rank raises; `never_settle`: every node passes through every shard (a full table)"""

    def __init__(self, rank, world, start, fail_on_merge=False, never_settle=False):
        self.rank, self.world = rank, world
        self.out = np.zeros(64, NODE)
        self.n_out = 0
        self.kept = []
        self.fail, self.never = fail_on_merge, never_settle
        for v in start:
            self._leave(v)

    def _leave(self, v):
        self.out[self.n_out] = (v, 7, 9)
        self.n_out += 1

    def shard_overflow(self):
        return 0, 0

    def shard_heavy(self):
        return 0, 0

    def shard_outgoing(self):
        return self.out.ctypes.data, self.n_out

    def shard_merge(self, ptr, n, is_triple=False, from_previous_shard=False):
        assert from_previous_shard and not is_triple
        if self.fail:
            raise ValueError("scripted failure on rank %d" % self.rank)
        got = np.frombuffer((C.c_uint8 * (int(n) * 16)).from_address(int(ptr)), dtype=NODE)
        for v in got["kmer"].tolist():
            if not self.never and v % self.world == self.rank:
                self.kept.append(v)
            else:
                self._leave(v)

    def sync(self):
        pass


def _ring_worker(rank, world, port, scenario, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dbg_assembly_amd.multigpu import _hand_offs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start = [rank * 100 + i for i in range(1, 6)]
    g = _RingGraph(rank, world, start, fail_on_merge=(scenario == "fail" and rank == 1), never_settle=(scenario == "full"))
    try:
        n_ovf, handed = _hand_offs(g, "cpu", None, _wrap_host, 16)
        q.put((rank, "ok", handed, sorted(g.kept)))
    except RuntimeError as e:
        q.put((rank, "raised", str(e), []))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["settle", "fail", "full"])
def test_ring_hand_off_repeats_until_nothing_moves_and_all_ranks_fail_together(scenario):
    """the hand-off of nodes that ran off the end of a shard: repeated until an all-reduce says nothing moved (a node may pass
    through several shards); a failure on ONE rank, or nodes that never settle (full table), make EVERY rank raise -- nobody
    is left waiting in a collective"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if scenario == "settle":
        assert all(r[1] == "ok" for r in res)
        everything = sorted(v for r in range(world) for v in [r * 100 + i for i in range(1, 6)])
        assert sorted(v for r in res for v in r[3]) == everything
        assert all(v % world == r[0] for r in res for v in r[3])
        # node v starts as "left shard s": it is delivered to s + 1, s + 2, ... until it reaches shard v % world
        hops = sum(((v % world - (s + 1)) % world) + 1 for s in range(world) for v in [s * 100 + i for i in range(1, 6)])
        assert all(r[2] == hops for r in res)
    else:
        assert all(r[1] == "raised" for r in res), res
        if scenario == "fail":
            assert sum("this rank" in r[2] for r in res) == 1 and sum("another rank" in r[2] for r in res) == 2
        else:
            assert all("the table is full" in r[2] for r in res)


# ---- first-contact hardening: a collective that cannot complete ends the process with the stage's name ----------------------
def _diverge_worker(rank, world, port, q):
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dbg_assembly_amd import multigpu as M
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["DBGK_COLLECTIVE_TIMEOUT_S"] = "3"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.ones(4, dtype=torch.int64)
    M.all_reduce(t, op=dist.ReduceOp.SUM, name="bucket-count all-reduce")   # everyone takes part: completes, is logged
    q.put((rank, "first", M.stage_summary()))
    if rank == 0:
        M.all_reduce(t, op=dist.ReduceOp.SUM, name="exchange piece 3 of 8")   # rank 1 never joins this one
        q.put((rank, "returned", None))
    else:
        time.sleep(30)   # "took another branch": the watchdog of rank 0 must end rank 0 long before this


def test_a_collective_that_cannot_complete_ends_the_process_with_the_stage_name(capfd):
    """multigpu.stage: every collective runs under a wall-clock limit (DBGK_COLLECTIVE_TIMEOUT_S); when the ranks diverge, the
    rank that waits leaves with exit code 17 and names the stage on stderr instead of hanging the job; completed stages are
    logged with their bytes and peers (bench.py's rccl.stages)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_diverge_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    first = dict((r, s) for r, _, s in (q.get(timeout=60) for _ in range(2)))
    for r in (0, 1):
        st = first[r]["bucket-count all-reduce"]
        assert st["calls"] == 1 and st["bytes_out"] == 32 and st["peers"] == 1 and st["ms"] >= 0.0
    procs[0].join(timeout=40)
    assert procs[0].exitcode == 17, procs[0].exitcode
    procs[1].terminate()
    procs[1].join(timeout=20)
    assert q.empty(), "the all-reduce nobody else joined must not return"
    err = capfd.readouterr().err
    assert "stage 'exchange piece 3 of 8' has not completed" in err and "rank 0" in err


def test_bench_plans_its_geometry_for_2_4_and_8_ranks_without_a_gpu():
    """`bench.py --gpus N` dry run of the argument / geometry planning for N = 2, 4, 8 (no GPU call): the table sizes, per-rank
    shares and level-1 bucket ownership the sharded flow would use -- what the first run on a real node starts from"""
    import subprocess
    import sys
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for config in ("cfg2", "cfg3"):
        for n in (2, 4, 8):
            out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--config", config, "--plan-only"],
                                 check=True, capture_output=True, text=True).stdout
            plan = json.loads(out.strip().splitlines()[-1])
            assert plan["world"] == n and plan["config"] == config
            assert plan["table_slots_global"] < 2 ** 34 and plan["table_slots_global"] >= 2 ** 26
            assert plan["level1_buckets"] <= 1024 and plan["buckets_per_rank"] * n >= plan["level1_buckets"]
            assert sum(plan["own_buckets"]) == plan["level1_buckets"] and all(b > 0 for b in plan["own_buckets"])
            assert plan["records_per_rank"] == plan["reads_per_gpu"] * 120
            # what a rank holds must fit one MI355X (288 GB): its table shard, the level-1 store it fills and the inbox it receives
            assert plan["bytes_per_rank"]["total"] < 288e9
            if config == "cfg2":
                assert plan["table_slots_global"] < 2 ** 32, "cfg2 keeps the global table below 2^32 slots (level-2 fan-out 1024)"
