"""Multi-GPU merge of per-rank k-mer graphs.

Reads shard by record: every rank builds a table from its own reads with the single-GPU path.
Keys are then OWNED by hash, owner(key) = (hash_code(key) >> 32) % world -- the device analogue of
the reference's `kmer % threadNum` ownership (DBG_contig/DBGgraph.cpp:148) -- and the per-rank
tables are merged with three collectives over RCCL/xGMI (torch.distributed, backend "nccl"):

  1. all-reduce (sum) of the world x world matrix of per-owner node counts, each rank filling its
     own row  ("bucket-count all-reduce": every rank learns all send/receive sizes),
  2. all-to-all of the locally aggregated 16-byte nodes, grouped by owner,
  3. all-reduce (sum) of the scalar totals (Total_reads_num, Kmer_total_num, node count).

The owner folds what it receives into a fresh table with per-byte saturating adds, which is exact
for any split of the input (min(255, min(255,a)+min(255,b)) == min(255,a+b)).

The engine argument abstracts the compute so the collective plumbing can be exercised on CPU with
the gloo backend (tests/test_multigpu_gloo.py injects an oracle-backed engine); the product engine
is HipEngine below and has no fallback.
"""
import numpy as np
import torch
import torch.distributed as dist

NODE_BYTES = 16


class HipEngine:
    """Adapter over a capi.Graph: node buffers are torch uint8 CUDA tensors (plumbing only)."""

    def __init__(self, graph, device):
        self.g = graph
        self.device = device

    def local_stats(self):
        return self.g.stats

    def partition_counts(self, n_parts):
        return self.g.partition_counts(n_parts).astype(np.int64)

    def partition_export(self, n_parts, total_nodes):
        buf = torch.empty(max(int(total_nodes), 1) * NODE_BYTES, dtype=torch.uint8, device=self.device)
        self.g.partition_export(n_parts, buf.data_ptr(), int(total_nodes))
        return buf

    def new_buffer(self, n_nodes):
        return torch.empty(max(int(n_nodes), 1) * NODE_BYTES, dtype=torch.uint8, device=self.device)

    def reset_table(self):
        self.g.reset()

    def merge(self, buf, n_nodes):
        self.g.merge_nodes(buf.data_ptr(), int(n_nodes))

    def finish(self):
        return self.g.finalize()  # sync + counters; marks the owned table exportable

    def sync(self):
        self.g.sync()


def exchange_and_merge(engine, group=None):
    """Run collectives 1-3 for an engine whose local table is finalized.  Returns a dict with the
    global totals and this rank's owned node count.  After the call the engine's table holds
    exactly the keys this rank owns, fully merged."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    st = engine.local_stats()
    local_reads, local_kmers, local_stored = int(st.total_reads), int(st.total_kmers), int(st.stored_kmers)

    # 1. bucket-count all-reduce
    counts = engine.partition_counts(world)  # nodes of MY table owned by each rank
    dev = getattr(engine, "device", "cpu")
    mat = torch.zeros((world, world), dtype=torch.int64, device=dev)
    mat[rank] = torch.as_tensor(counts, dtype=torch.int64, device=dev)
    dist.all_reduce(mat, op=dist.ReduceOp.SUM, group=group)
    mat_h = mat.cpu().numpy()
    send_counts = mat_h[rank]           # what I send to each owner
    recv_counts = mat_h[:, rank]        # what each rank sends to me

    # 2. all-to-all of the aggregated nodes
    send = engine.partition_export(world, int(send_counts.sum()))
    recv = engine.new_buffer(int(recv_counts.sum()))
    engine.sync()
    n_send, n_recv = int(send_counts.sum()), int(recv_counts.sum())
    dist.all_to_all_single(recv[:n_recv * NODE_BYTES], send[:n_send * NODE_BYTES],
                           output_split_sizes=[int(c) * NODE_BYTES for c in recv_counts],
                           input_split_sizes=[int(c) * NODE_BYTES for c in send_counts], group=group)
    if recv.is_cuda:
        torch.cuda.current_stream().synchronize()

    # owner-side merge into a fresh table
    engine.reset_table()
    engine.merge(recv, n_recv)
    owned = engine.finish()

    # 3. scalar totals.  Every rank's owned count includes one key-0 node (rank 0 owns the real
    # one); the global graph has exactly one, so subtract the world-1 placeholders.
    tot = torch.tensor([local_reads, local_kmers, local_stored, int(owned.count)], dtype=torch.int64, device=dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    tot = tot.cpu().numpy()
    return {"total_reads": int(tot[0]), "total_kmers": int(tot[1]), "stored_kmers": int(tot[2]),
            "count": int(tot[3]) - (world - 1), "owned_count": int(owned.count),
            "sent_nodes": n_send, "recv_nodes": n_recv}
