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


class WideHipEngine(HipEngine):
    """the same flow for a WIDE handle (k <= 63): 32-byte nodes {kmer_hi, kmer_lo, l_link, r_link, reserved},
    owner of a node = (hash128 >> 32) % n (dbgk_wide_partition_export / dbgk_wide_merge_nodes)"""
    node_bytes = 32

    def partition_counts(self, n_parts):
        return self.g.wide_partition_export(n_parts).astype(np.int64)

    def partition_export(self, n_parts, total_nodes):
        buf = torch.empty(max(int(total_nodes), 1) * self.node_bytes, dtype=torch.uint8, device=self.device)
        self.g.wide_partition_export(n_parts, buf.data_ptr(), int(total_nodes))
        return buf

    def new_buffer(self, n_nodes):
        return torch.empty(max(int(n_nodes), 1) * self.node_bytes, dtype=torch.uint8, device=self.device)

    def merge(self, buf, n_nodes):
        self.g.wide_merge_nodes(buf.data_ptr(), int(n_nodes))


# One message per peer and transfer is bounded: a single 3.5 GB all_to_all_single (the node buffers of a
# bench-size table) was observed to move only its first 1.78 GB on this RCCL build and report success
# (profiles/ history, round 1).  Every bulk transfer below goes through point-to-point sends/receives of
# at most MAX_MESSAGE_BYTES; all peers' pieces of one round travel as one batch.
MAX_MESSAGE_BYTES = 512 << 20


# ---- transport ------------------------------------------------------------------------------------------------
# The product transport is RCCL (backend "nccl"): every collective below runs on device memory over xGMI.  A process
# group WITHOUT a device transport (gloo) can still drive real handles: device tensors are then staged through host
# copies around each collective.  That is how several ranks are rehearsed on ONE GPU (bench.py --backend gloo
# --one-gpu, tests/test_gpu_multigpu.py: RCCL refuses two ranks on one device) and what a node whose RCCL cannot do
# peer transfers falls back to.  Only the transport changes: every kernel of the flow still runs on the GPU.
def _host_staged(t, group):
    return bool(t.is_cuda) and dist.get_backend(group) != "nccl"


# ---- every collective is a named STAGE with a wall-clock limit --------------------------------------------------
# The first run on a real multi-GPU node is where a divergence between the ranks shows (one rank raises or takes another
# branch, the others wait in a collective for ever).  Every wrapper below therefore runs inside `stage(name, ...)`: a
# watchdog thread ends the PROCESS with the stage's name, the rank and the elapsed time when a stage has not completed within
# DBGK_COLLECTIVE_TIMEOUT_S seconds (default 120; a torchrun job then goes down as a whole instead of hanging), and the
# completed stages are logged -- milliseconds (device events for RCCL, wall clock for host-staged transports), bytes out / in,
# peers -- so that one SCALE line can explain its own efficiency (bench.py "rccl.stages").
import os
import threading
import time


class _Stages:
    def __init__(self):
        self.lock = threading.Lock()
        self.open = {}
        self.log = []
        self.next_id = 0
        self.thread = None

    def timeout_s(self):
        return float(os.environ.get("DBGK_COLLECTIVE_TIMEOUT_S", "120"))

    def _watch(self):
        while True:
            time.sleep(0.2)
            now = time.perf_counter()
            with self.lock:
                for sid, st in list(self.open.items()):
                    ev = st.get("end_event")
                    if ev is not None and ev.query():
                        self._close(sid, st["start_event"].elapsed_time(ev))
                    elif now - st["t0"] > st["limit"]:
                        self._die(st, now)

    def _die(self, st, now):
        try:
            rank = dist.get_rank() if dist.is_initialized() else -1
        except Exception:  # noqa: BLE001
            rank = -1
        os.write(2, ("dbgk multigpu: rank %d: stage '%s' has not completed after %.0f s (limit %.0f s, DBGK_COLLECTIVE_TIMEOUT_S): "
                     "the ranks have diverged or a peer is gone -- leaving\n" % (rank, st["name"], now - st["t0"], st["limit"])).encode())
        os._exit(17)

    def _close(self, sid, ms):
        st = self.open.pop(sid)
        self.log.append({"stage": st["name"], "ms": float(ms), "bytes_out": int(st["bytes_out"]), "bytes_in": int(st["bytes_in"]), "peers": int(st["peers"])})

    def begin(self, name, bytes_out, bytes_in, peers, on_gpu):
        with self.lock:
            if self.thread is None:
                self.thread = threading.Thread(target=self._watch, daemon=True, name="dbgk-collective-watchdog")
                self.thread.start()
            sid = self.next_id
            self.next_id += 1
            st = {"name": name, "t0": time.perf_counter(), "limit": self.timeout_s(), "bytes_out": bytes_out, "bytes_in": bytes_in, "peers": peers}
            if on_gpu:
                st["start_event"] = torch.cuda.Event(enable_timing=True)
                st["start_event"].record()
            self.open[sid] = st
            return sid

    def end(self, sid):
        with self.lock:
            st = self.open.get(sid)
            if st is None:
                return
            if "start_event" in st:   # asynchronous transport: the stage stays open until the device has passed this point
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                st["end_event"] = ev
            else:
                self._close(sid, (time.perf_counter() - st["t0"]) * 1e3)

    def drain(self):
        """stages completed so far, summed by name (transfers still in flight stay open); the log is emptied"""
        with self.lock:
            for sid, st in list(self.open.items()):
                ev = st.get("end_event")
                if ev is not None and ev.query():
                    self._close(sid, st["start_event"].elapsed_time(ev))
            out = {}
            for e in self.log:
                d = out.setdefault(e["stage"], {"calls": 0, "ms": 0.0, "bytes_out": 0, "bytes_in": 0, "peers": e["peers"]})
                d["calls"] += 1
                d["ms"] += e["ms"]
                d["bytes_out"] += e["bytes_out"]
                d["bytes_in"] += e["bytes_in"]
            self.log = []
            return out


_STAGES = _Stages()


class stage:
    """with stage("exchange piece 3", bytes_out=..., bytes_in=..., peers=7, on_gpu=True): <collective calls>"""

    def __init__(self, name, bytes_out=0, bytes_in=0, peers=0, on_gpu=False):
        self.args = (name, bytes_out, bytes_in, peers, on_gpu)

    def __enter__(self):
        self.sid = _STAGES.begin(*self.args)
        return self

    def __exit__(self, exc_type, exc, tb):
        _STAGES.end(self.sid)
        return False


def stage_summary():
    """{stage name: calls, ms, bytes_out, bytes_in, peers} of the collectives completed since the last call"""
    return _STAGES.drain()


def _nbytes(t):
    return int(t.numel()) * int(t.element_size())


def _async(t, group):
    return bool(t.is_cuda) and dist.get_backend(group) == "nccl"


def all_reduce(t, op, group=None, name="all-reduce"):
    with stage(name, _nbytes(t), _nbytes(t), dist.get_world_size(group) - 1, _async(t, group)):
        if _host_staged(t, group):
            h = t.cpu()
            dist.all_reduce(h, op=op, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op, group=group)


def all_to_all_single(out, inp, group=None, name="all-to-all"):
    world = dist.get_world_size(group)
    with stage(name, _nbytes(inp) * (world - 1) // world, _nbytes(out) * (world - 1) // world, world - 1, _async(out, group)):
        if _host_staged(out, group):
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h, inp.cpu(), group=group)
            out.copy_(h)
        else:
            dist.all_to_all_single(out, inp, group=group)


def all_gather(outs, t, group=None, name="all-gather"):
    with stage(name, _nbytes(t) * (len(outs) - 1), sum(_nbytes(o) for o in outs) - _nbytes(t), len(outs) - 1, _async(t, group)):
        if _host_staged(t, group):
            hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
            dist.all_gather(hs, t.cpu(), group=group)
            for o, h in zip(outs, hs):
                o.copy_(h)
        else:
            dist.all_gather(outs, t, group=group)


def broadcast(buf, src, group=None, name="broadcast"):
    with stage(name, _nbytes(buf), _nbytes(buf), dist.get_world_size(group) - 1, _async(buf, group)):
        if _host_staged(buf, group):
            h = buf.cpu()
            dist.broadcast(h, src=src, group=group)
            buf.copy_(h)
        else:
            dist.broadcast(buf, src=src, group=group)


def _run_p2p(sends, recvs, group, name="point-to-point batch"):
    """sends / recvs: lists of (tensor view, peer).  One batch of point-to-point transfers, complete on return (NCCL: the
    current stream is ordered behind it)."""
    if not sends and not recvs:
        return
    first = (sends or recvs)[0][0]
    staged = _host_staged(first, group)
    with stage(name, sum(_nbytes(t) for t, _ in sends), sum(_nbytes(t) for t, _ in recvs), len({p for _, p in sends} | {p for _, p in recvs}), _async(first, group)):
        ops, landing = [], []
        for t, peer in sends:
            ops.append(dist.P2POp(dist.isend, t.cpu() if staged else t, peer, group))
        for t, peer in recvs:
            h = torch.empty(t.shape, dtype=t.dtype) if staged else t
            landing.append((t, h))
            ops.append(dist.P2POp(dist.irecv, h, peer, group))
        for work in dist.batch_isend_irecv(ops):
            work.wait()  # NCCL: orders the current stream behind the transfer; gloo: blocks until done
        if staged:
            for t, h in landing:
                t.copy_(h)


def exchange_slices(pairs, rank, group=None, name="exchange"):
    """pairs: list of (send_view, recv_view, peer) of uint8 tensors, equal sizes on both ends of a pair.
    Moves send_view of every pair to the peer's recv_view."""
    longest = max([int(sv.numel()) for sv, _, _ in pairs] + [0])
    for off in range(0, longest, MAX_MESSAGE_BYTES):
        sends, recvs = [], []
        for sv, rv, peer in pairs:
            n = int(sv.numel())
            if off >= n:
                continue
            hi = min(n, off + MAX_MESSAGE_BYTES)
            if peer == rank:
                rv[off:hi].copy_(sv[off:hi])
            else:
                sends.append((sv[off:hi], peer))
                recvs.append((rv[off:hi], peer))
        _run_p2p(sends, recvs, group, name)


def _exchange_uneven(pairs, rank, group=None, name="exchange (uneven)"):
    """like exchange_slices, but the two directions of a pair have their own lengths"""
    longest = max([max(int(sv.numel()), int(rv.numel())) for sv, rv, _ in pairs] + [0])
    for off in range(0, longest, MAX_MESSAGE_BYTES):
        sends, recvs = [], []
        for sv, rv, peer in pairs:
            sn, rn = int(sv.numel()), int(rv.numel())
            if peer == rank:
                if off < sn:
                    hi = min(sn, off + MAX_MESSAGE_BYTES)
                    rv[off:hi].copy_(sv[off:hi])
                continue
            if off < sn:
                sends.append((sv[off:min(sn, off + MAX_MESSAGE_BYTES)], peer))
            if off < rn:
                recvs.append((rv[off:min(rn, off + MAX_MESSAGE_BYTES)], peer))
        _run_p2p(sends, recvs, group, name)


def exchange_and_merge(engine, group=None):
    """Run collectives 1-3 for an engine whose local table is finalized.  Returns a dict with the
    global totals and this rank's owned node count.  After the call the engine's table holds
    exactly the keys this rank owns, fully merged."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    NODE_BYTES = getattr(engine, "node_bytes", 16)
    st = engine.local_stats()
    local_reads, local_kmers, local_stored = int(st.total_reads), int(st.total_kmers), int(st.stored_kmers)

    # 1. bucket-count all-reduce
    counts = engine.partition_counts(world)  # nodes of MY table owned by each rank
    dev = getattr(engine, "device", "cpu")
    mat = torch.zeros((world, world), dtype=torch.int64, device=dev)
    mat[rank] = torch.as_tensor(counts, dtype=torch.int64, device=dev)
    all_reduce(mat, op=dist.ReduceOp.SUM, group=group)
    mat_h = mat.cpu().numpy()
    send_counts = mat_h[rank]           # what I send to each owner
    recv_counts = mat_h[:, rank]        # what each rank sends to me

    # 2. all-to-all of the aggregated nodes
    send = engine.partition_export(world, int(send_counts.sum()))
    recv = engine.new_buffer(int(recv_counts.sum()))
    engine.sync()
    n_send, n_recv = int(send_counts.sum()), int(recv_counts.sum())
    pairs, so, ro = [], 0, 0
    for peer in range(world):
        sn, rn = int(send_counts[peer]) * NODE_BYTES, int(recv_counts[peer]) * NODE_BYTES
        # a pair's two ends have different lengths here (what I send to the peer / what it sends to me):
        # one entry per direction, the other side empty
        pairs.append((send[so:so + sn], recv[ro:ro + rn], peer))
        so += sn
        ro += rn
    _exchange_uneven(pairs, rank, group)
    if recv.is_cuda:
        torch.cuda.synchronize()  # device-wide: also the communication stream RCCL works on

    # owner-side merge into a fresh table
    engine.reset_table()
    engine.merge(recv, n_recv)
    owned = engine.finish()

    # 3. scalar totals.  Every rank's owned count includes one key-0 node (rank 0 owns the real
    # one); the global graph has exactly one, so subtract the world-1 placeholders.
    tot = torch.tensor([local_reads, local_kmers, local_stored, int(owned.count)], dtype=torch.int64, device=dev)
    all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    tot = tot.cpu().numpy()
    return {"total_reads": int(tot[0]), "total_kmers": int(tot[1]), "stored_kmers": int(tot[2]),
            "count": int(tot[3]) - (world - 1), "owned_count": int(owned.count),
            "sent_nodes": n_send, "recv_nodes": n_recv}


# ==================================================================================================
# Slot-range ownership (the scalable path; bench.py uses it for N > 1)
# ==================================================================================================
# exchange_and_merge above ships locally AGGREGATED nodes, which pays when every rank sees the whole
# genome at high coverage.  With reads sharded by record, a rank's coverage is only 1/N of the job's,
# local aggregation finds few duplicates and the per-rank table would have to hold nearly every
# occurrence.  Here no local table exists at all: every rank level-1-partitions its k-mer records by
# slot range of ONE global table (PARTITION engine), rank d owns the buckets [d*B, (d+1)*B), and
#
#   1. all-reduce (sum) of the per-bucket record counts  -> global bucket fills (capacity check),
#   2. all-to-all of the bucket fill counts and of the level-1 bucket stores themselves: chunk d of
#      every rank's store goes to rank d (zero-copy slices of the library's own buffers).  The records
#      travel in `exchange_chunks` pieces (ranges of own buckets, point-to-point sends/receives queued
#      up front): the build of a piece starts as soon as it has arrived and overlaps the transfer of
#      the next one -- every record crosses xGMI once (8 B x 7/8 of all k-mer occurrences per GPU),
#      which takes about as long as the build itself,
#   3. each rank builds only its slot range (level-2 partition + LDS region build),
#   4. hand-offs that are normally empty or tiny: nodes whose probe ran off the end of a shard go
#      to the next rank (ring), bucket-overflow observations are offered to every rank,
#   5. key-0 links gathered onto rank 0, scalar totals all-reduced.

class _DevMem:
    """zero-copy view of raw device memory for torch (CUDA array interface v2)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def wrap_device_memory(ptr, nbytes, device):
    return torch.as_tensor(_DevMem(ptr, nbytes), device=device)


def _exchange_range(send, recv, info, j0, j1, world, rank, group, name="level-1 exchange"):
    """buckets [j0, j1) of every destination's chunk -> the same buckets of chunk `rank` at the destination"""
    pairs = []
    for peer in range(world):
        lo, hi = peer * info.chunk_bytes + j0 * info.bucket_bytes, peer * info.chunk_bytes + j1 * info.bucket_bytes
        pairs.append((send[lo:hi], recv[lo:hi], peer))
    exchange_slices(pairs, rank, group, name)


def _exchange_level1(g, device, group, wrap, exchange_chunks, verify):
    """steps 1-2 for the level-1 record stores of the current step (64-bit engine) or pass (WIDE engine): all-reduce of the
    per-bucket record counts, all-to-all of the fill counts and of the bucket stores in pieces, every piece built
    (dbgk_shard_build_range) as soon as it has arrived.  verify: every rank checksums what it sends to each peer and what it
    received from each peer (64-bit wrapping sums over the transferred slices) and the two sides are compared -- a transfer
    that silently moved only part of a message (seen once on this RCCL build with a single 3.5 GB all_to_all_single) fails
    loudly instead of building a wrong table.  Returns the number of records of the whole job in this step / pass."""
    on_gpu = torch.device(device).type == "cuda"
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    info = g.shard_info()
    assert info.n_ranks == world and info.rank == rank, (info.n_ranks, info.rank, world, rank)
    g.sync()
    send = wrap(info.d_send, world * info.chunk_bytes, device)
    recv = wrap(info.d_recv, world * info.chunk_bytes, device)
    send_cnt = wrap(info.d_send_cnt, world * info.cnt_chunk_bytes, device)
    recv_cnt = wrap(info.d_recv_cnt, world * info.cnt_chunk_bytes, device)

    # 1. per-bucket k-mer counts of the whole job
    bucket_counts = send_cnt.view(torch.int32).to(torch.int64)
    all_reduce(bucket_counts, op=dist.ReduceOp.SUM, group=group, name="bucket-count all-reduce")
    records_global = int(bucket_counts.sum().item())

    # 2. the exchange
    all_to_all_single(recv_cnt, send_cnt, group=group, name="fill-count all-to-all")
    # every own bucket must have received exactly what the all-reduce says the job holds for it (counts are capped by the
    # bucket capacity on both sides alike): a wrong fill count would silently drop or invent records
    B = int(info.buckets_per_rank)
    per = int(info.cnt_chunk_bytes) // 4   # fill counters per rank chunk (buckets x sub-stores)
    mine = recv_cnt.view(torch.int32).to(torch.int64).view(world, per).sum(dim=0)
    want = bucket_counts.view(world, per)[rank]
    flag = torch.tensor([0 if bool(torch.equal(mine, want)) else 1], dtype=torch.int64, device=device)
    all_reduce(flag, op=dist.ReduceOp.MAX, group=group, name="fill-count agreement")
    if int(flag.item()):
        raise RuntimeError("rank %d: the exchanged bucket fill counts disagree with the all-reduced totals (here or on another rank)" % rank)
    n_chunks = max(1, min(int(exchange_chunks), B))
    if n_chunks <= 1:
        _exchange_range(send, recv, info, 0, B, world, rank, group, "level-1 exchange (one piece)")
        if on_gpu:
            torch.cuda.synchronize()  # device-wide: also the communication stream RCCL works on
    else:
        # every rank cuts the SAME bucket ranges (buckets_per_rank is common; ranks that own fewer
        # buckets build a shorter range but still take part in every transfer)
        per = (B + n_chunks - 1) // n_chunks
        ranges = [(j0, min(j0 + per, B)) for j0 in range(0, B, per)]
        arrived = []
        for piece, (j0, j1) in enumerate(ranges):  # all transfers are queued before anything is waited for
            _exchange_range(send, recv, info, j0, j1, world, rank, group, "level-1 exchange piece %d of %d" % (piece, len(ranges)))
            if on_gpu:
                ev = torch.cuda.Event()
                ev.record()
                arrived.append(ev)
        planned = False
        for i, (j0, j1) in enumerate(ranges):
            if on_gpu:
                arrived[i].synchronize()  # the fill counts (queued first) have arrived as well
            if not planned:
                g.shard_plan()
                planned = True
            own0, own1 = min(j0, int(info.own_buckets)), min(j1, int(info.own_buckets))
            if own1 > own0:
                g.shard_build_range(own0, own1)  # queues level 2 + build on the library's streams and returns
    if verify:
        if on_gpu:
            torch.cuda.synchronize()
        cb = int(info.chunk_bytes)
        sent = torch.stack([send[p * cb:(p + 1) * cb].view(torch.int64).sum() for p in range(world)])
        got = torch.stack([recv[p * cb:(p + 1) * cb].view(torch.int64).sum() for p in range(world)])
        theirs = torch.empty_like(sent)
        all_to_all_single(theirs, sent, group=group, name="exchange checksums")   # theirs[p] = checksum of what rank p sent to me
        bad = [p for p in range(world) if int(theirs[p]) != int(got[p])]
        flag = torch.tensor([1 if bad else 0], dtype=torch.int64, device=device)
        all_reduce(flag, op=dist.ReduceOp.MAX, group=group, name="exchange checksum agreement")   # every rank stops, not only the one that saw it (no rank is left waiting)
        if int(flag.item()):
            raise RuntimeError("rank %d: the level-1 record buckets received from rank(s) %r differ from what was sent "
                               "(a truncated or corrupted transfer%s)" % (rank, bad, "" if bad else " seen by another rank"))
    g.shard_mark_exchanged()
    return records_global


def _gather_lists(ptr, n, node_bytes, device, group, wrap, on_gpu, name="list gather"):
    """all-gather of one device list per rank (n entries of node_bytes): -> (sizes per rank, padded tensors per rank)"""
    world = dist.get_world_size(group)
    sizes = torch.tensor([n], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    all_gather(all_sizes, sizes, group=group, name=name + ": sizes")
    all_sizes = torch.stack(all_sizes).cpu().numpy()[:, 0]
    longest = int(all_sizes.max())
    if not longest:
        return all_sizes, None
    mine = torch.zeros(longest * node_bytes, dtype=torch.uint8, device=device)
    if n:
        mine[:n * node_bytes] = wrap(ptr, n * node_bytes, device)
    lists = [torch.empty_like(mine) for _ in range(world)]
    all_gather(lists, mine, group=group, name=name)
    if on_gpu:
        torch.cuda.synchronize()  # device-wide: also the communication stream RCCL works on
    return all_sizes, lists


def _agree(ok, what, device, group):
    """every rank learns whether ANY rank failed a local step, so that all of them leave together (a rank that raised alone
    would leave the others waiting in the next collective): all-reduce (min) of an ok flag, then the same exception everywhere"""
    flag = torch.tensor([1 if ok is None else 0], dtype=torch.int64, device=device)
    all_reduce(flag, op=dist.ReduceOp.MIN, group=group, name="agreement: " + what)
    if int(flag.item()) == 0:
        raise RuntimeError("%s failed on %s" % (what, ("this rank: %s" % ok) if ok is not None else "another rank"))


def _hand_offs(g, device, group, wrap, node_bytes):
    """step 4: bucket-overflow observations offered to every rank, the aggregated surplus of heavy hitters broadcast, nodes
    whose probe ran off the end of a shard handed to the next rank (ring).  Overflow observations first: merging them can
    push further nodes off the end of a shard, so the outgoing lists are read only afterwards.  The ring hand-off is REPEATED
    until an all-reduce says that nothing moved any more (a node can pass through a shard that is full behind its first slots
    and leave it again at the far end: dbgk_comm_* does the same); a node still travelling after `world` rounds has been
    offered to every shard -- the table is full.  Every local failure is turned into an all-reduced flag (_agree): all ranks
    raise together."""
    on_gpu = torch.device(device).type == "cuda"
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)

    def attempt(fn):
        try:
            fn()
            if on_gpu or hasattr(g, "sync"):
                g.sync()
            return None
        except Exception as e:  # noqa: BLE001 -- reported through _agree on every rank
            return "%s: %s" % (type(e).__name__, e)

    def fetch(fn, what):
        """a local read of one of the handle's lists (can fail: a list that overran its capacity) -- agreed on before the collective"""
        box = {}
        _agree(attempt(lambda: box.update(v=fn())), what, device, group)
        return box["v"]

    p_ovf, n_ovf = fetch(g.shard_overflow, "reading the bucket-overflow list")
    ovf_sizes, lists = _gather_lists(p_ovf, n_ovf, node_bytes, device, group, wrap, on_gpu, "bucket-overflow observations")
    if lists is not None:
        def merge_observations():
            for src in range(world):
                if ovf_sizes[src]:
                    g.shard_merge(lists[src].data_ptr(), int(ovf_sizes[src]), is_triple=True)
        _agree(attempt(merge_observations), "merging the bucket-overflow observations", device, group)
    # the surplus of heavy hitters (beyond a rank's overflow list) sits aggregated in its side table: rare, broadcast whole
    p_hh, n_hh = fetch(g.shard_heavy, "reading the heavy-hitter side table")
    sizes = torch.tensor([n_hh], dtype=torch.int64, device=device)
    hh_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    all_gather(hh_sizes, sizes, group=group, name="heavy-hitter table sizes")
    hh_sizes = torch.stack(hh_sizes).cpu().numpy()[:, 0]
    for src in range(world):
        if hh_sizes[src]:
            nbytes = int(hh_sizes[src]) * node_bytes
            buf = wrap(p_hh, nbytes, device).clone() if rank == src else torch.empty(nbytes, dtype=torch.uint8, device=device)
            broadcast(buf, src=src, group=group, name="heavy-hitter table of rank %d" % src)
            if on_gpu:
                torch.cuda.synchronize()
            _agree(attempt(lambda: g.shard_merge(buf.data_ptr(), int(hh_sizes[src]))), "merging a heavy-hitter side table", device, group)
    # the ring: what left shard d continues at the first slot of shard d + 1; the outgoing list only grows (it stays valid until
    # the next reset), so every round ships the entries that were appended since the last one
    delivered, handed, prev = 0, 0, (rank - 1) % world
    for rnd in range(world + 1):
        p_out, n_out = fetch(g.shard_outgoing, "reading the list of nodes that left the shard")
        fresh = int(n_out) - delivered
        out_sizes, lists = _gather_lists(int(p_out) + delivered * node_bytes if fresh else p_out, fresh, node_bytes, device, group, wrap, on_gpu,
                                         "ring hand-off round %d" % rnd)
        moved = int(out_sizes.sum())   # (the same number on every rank: all of them leave the loop, or fail, together)
        if moved == 0:
            break
        if rnd == world:
            raise RuntimeError("%d nodes are still looking for a slot after a whole round over all %d shards: the table is full" % (moved, world))
        handed += moved
        delivered = int(n_out)
        err = None
        if out_sizes[prev]:
            err = attempt(lambda: g.shard_merge(lists[prev].data_ptr(), int(out_sizes[prev]), from_previous_shard=True))
        _agree(err, "merging the nodes handed over by the previous shard", device, group)
    return int(ovf_sizes.sum()), handed


def sharded_finalize(g, device, group=None, wrap=None, exchange_chunks=8, verify_exchange=False):
    """g: a sharded capi.Graph (shard_count == world, shard_index == rank) that has received all its
    pushes.  Performs steps 1-5 and returns the global totals; afterwards g's table holds this
    rank's slot range of the global table.  `wrap(ptr, nbytes, device)` turns the library's buffers
    into tensors (default: zero-copy CUDA array interface; the gloo CPU test passes a host wrapper).
    exchange_chunks <= 1: one all-to-all of the whole stores, then the build."""
    wrap = wrap or globals()["wrap_device_memory"]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    records_global = _exchange_level1(g, device, group, wrap, exchange_chunks, verify_exchange)

    # 3. build this rank's slot range
    st = g.finalize()
    local = (int(st.total_reads), int(st.total_kmers), int(st.stored_kmers))

    # 4. hand-offs
    n_ovf, n_handed = _hand_offs(g, device, group, wrap, NODE_BYTES)

    # 5. key-0 node onto rank 0, totals
    links = torch.tensor([int(st.polyA_l_link), int(st.polyA_r_link)], dtype=torch.int64, device=device)
    all_links = [torch.zeros_like(links) for _ in range(world)]
    all_gather(all_links, links, group=group, name="key-0 links gather")
    if rank == 0:
        for src in range(1, world):
            l, r = (int(x) for x in all_links[src].cpu().numpy())
            if l or r:
                g.add_polyA(l, r)
    owned = g.refresh_stats()
    tot = torch.tensor([local[0], local[1], local[2], int(owned.count)], dtype=torch.int64, device=device)
    all_reduce(tot, op=dist.ReduceOp.SUM, group=group, name="totals all-reduce")
    tot = tot.cpu().numpy()
    return {"total_reads": int(tot[0]), "total_kmers": int(tot[1]), "stored_kmers": int(tot[2]), "count": int(tot[3]),
            "owned_count": int(owned.count), "records_global": records_global,
            "handed_over_nodes": n_handed, "overflow_observations": n_ovf}


WIDE_NODE_BYTES = 32


def wide_sharded_build(g, device, push_all, group=None, wrap=None, exchange_chunks=8, verify_exchange=False):
    """The slot-range flow for 128-bit keys (k <= 63; WIDE engine through 16-byte records).  g: a sharded WIDE capi.Graph
    (shard_count == world, shard_index == rank, expected_kmers > 0), reset.  push_all(g) pushes ALL of this rank's reads and is
    called once per pass: a job whose records do not fit the level-1 fan-out (1024 buckets over all ranks) or the device reads
    its input several times, each pass completing a part of every rank's slot range (dbgk_wide_begin_pass).  Per pass: steps
    1-3 of sharded_finalize; then the hand-offs, and the few nodes that live outside the table (keys whose low word is 0, the
    key-0 node: a side table per handle) are gathered onto rank 0.  Returns the global totals."""
    wrap = wrap or globals()["wrap_device_memory"]
    on_gpu = torch.device(device).type == "cuda"
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_passes, _ = g.wide_pass_info()
    records_global = 0
    for p in range(n_passes):
        g.wide_begin_pass(p)
        push_all(g)
        records_global += _exchange_level1(g, device, group, wrap, exchange_chunks, verify_exchange)
        g.wide_end_pass()
    st = g.finalize()
    local = (int(st.total_reads), int(st.total_kmers), int(st.stored_kmers))
    n_ovf, n_handed = _hand_offs(g, device, group, wrap, WIDE_NODE_BYTES)
    # side tables (keys with a zero low word) and key-0 links: onto rank 0, cleared everywhere else
    p_side, n_side = g.shard_side_export()
    side_sizes, lists = _gather_lists(p_side, n_side, WIDE_NODE_BYTES, device, group, wrap, on_gpu)
    if rank == 0:
        for src in range(1, world):
            g.wide_merge_nodes(lists[src].data_ptr(), int(side_sizes[src]))
        g.sync()
    else:
        g.shard_side_clear()
    owned = g.refresh_stats()
    tot = torch.tensor([local[0], local[1], local[2], int(owned.count)], dtype=torch.int64, device=device)
    all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    tot = tot.cpu().numpy()
    return {"total_reads": int(tot[0]), "total_kmers": int(tot[1]), "stored_kmers": int(tot[2]), "count": int(tot[3]),
            "owned_count": int(owned.count), "records_global": records_global, "passes": n_passes,
            "handed_over_nodes": n_handed, "overflow_observations": n_ovf}


# ---- the k-mer frequency table on several GPUs -----------------------------------------------
# Every rank counts its share of the reads into a whole 4^k-counter table (ENGINE_KFREQ, finalized).  The
# combination is a reduce-scatter whose operator is the per-byte saturating add -- RCCL's sum would wrap -- so it
# is written as chunked all-to-alls plus the library's merge kernel: rank d ends up owning the counters of the
# k-mer values [bounds[d], bounds[d+1]).

def kfreq_slice_bounds(n_counts, world):
    return [n_counts if d == world else (n_counts * d // world) & ~1023 for d in range(world + 1)]


def kfreq_reduce(g, device, group=None, wrap=None, chunk_bytes=256 << 20):
    """g: this rank's finalized KFREQ capi.Graph.  Returns (lo, hi): the range of k-mer values whose counters in
    g's table are now those of the whole job."""
    wrap_device_memory = wrap or globals()["wrap_device_memory"]
    on_gpu = torch.device(device).type == "cuda"
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ptr, n_counts = g.kfreq_device_counts()
    table = wrap_device_memory(ptr, n_counts, device)
    bounds = kfreq_slice_bounds(n_counts, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    if world == 1:
        return lo, hi
    longest = max(bounds[d + 1] - bounds[d] for d in range(world))
    chunk = min(int(chunk_bytes), longest) & ~1023 or 1024
    stage = torch.empty(world * chunk, dtype=torch.uint8, device=device)
    for off in range(0, longest, chunk):
        send, recv = [], []
        for d in range(world):  # piece [off, off + chunk) of every owner's range
            a, b = min(bounds[d] + off, bounds[d + 1]), min(bounds[d] + off + chunk, bounds[d + 1])
            send.append(table[a:b])
        mine = max(0, min(lo + off + chunk, hi) - min(lo + off, hi))
        for s in range(world):
            recv.append(stage[s * chunk: s * chunk + mine])
        _exchange_uneven([(send[p], recv[p], p) for p in range(world) if p != rank], rank, group)  # ranges differ in length
        if on_gpu:
            torch.cuda.synchronize()
        for s in range(world):
            if s != rank and mine:
                g.kfreq_merge_counts(recv[s].data_ptr(), lo + off, mine)
        g.sync()
    return lo, hi
