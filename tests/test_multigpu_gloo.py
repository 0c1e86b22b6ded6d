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
