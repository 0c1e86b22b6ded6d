// dbgk_comm.h -- several GPUs inside ONE process: N sharded handles of one global table, the exchange done with
// peer copies over xGMI (hipMemcpyPeerAsync), driven by a single host thread.  Included at the end of dbgk.hip.
//
// This is what the C++ host layer uses for DBGK_GPUS=N (host/DBGgraph.cpp); bench.py's one-process-per-GPU flow
// over RCCL (dbg_assembly_amd/multigpu.py: sharded_finalize) follows the same protocol with collectives.  The
// ownership rule is the reference's per-thread one (`kmer % threadNum`, DBG_contig/DBGgraph.cpp:148) turned into
// slot ranges: reads shard by record, every rank level-1-partitions its k-mer records by GLOBAL slot range,
// rank d owns the level-1 buckets [d*B, (d+1)*B) and builds only that part of the table.
//
//   flush / finalize:
//     1. bucket fill counts  s -> d              (tiny)
//     2. level-1 record buckets of d's slot range, from every s, in pieces of buckets: the copy of piece j+1
//        (copy stream of d) overlaps level 2 + region build of piece j (the handle's two streams)
//     3. every shard builds its slot range (incremental when the table already holds nodes)
//     4. hand-offs, normally empty: overflow observations and the heavy-hitter side tables are offered to every
//        shard, nodes that probed past the end of a shard continue at the start of the next one
//     5. finalize only: key-0 links folded onto shard 0, totals summed
#pragma once

struct dbgk_comm {
	std::vector<dbgk_handle *> h;
	std::vector<hipStream_t> copy_stream;      // per destination handle
	std::vector<std::vector<hipEvent_t>> ev;   // per destination handle: piece arrived
	std::vector<hipEvent_t> cnt_ev;            // per destination handle: fill counts arrived
	std::vector<uint64_t> delivered;           // per handle: outgoing nodes already handed to the next shard
	uint32_t next_push = 0;
	uint32_t pieces = 8;
	bool finalized = false;
	bool host_staging = false;                 // some pair of member GPUs are no peers (or DBGK_COMM_HOST_STAGING=1): copies between
	                                           // different devices go through a pinned host buffer instead of hipMemcpyPeerAsync
	void *stage = nullptr;                     // pinned, kStageBytes
	dbgk_config cfg;                           // what the members were created from (dbgk_comm_resize)
	std::vector<int32_t> devices;
	// KFREQ communicators: every member counts its reads into a whole table of its own; at finalize member d
	// becomes the owner of the k-mer values [kf_lo[d], kf_lo[d+1]) and adds the other members' slices to its own
	bool wide = false;                         // members are sharded WIDE handles (k <= 63, 16-byte records)
	bool kfreq = false;
	std::vector<uint64_t> kf_lo;
	uint64_t kf_distinct = 0;
};

constexpr size_t kCommStageBytes = 64ull << 20;
static thread_local dbgk_comm *g_comm_ctx = nullptr; // the communicator whose copies are being queued (host staging needs its buffer)

static int comm_copy(dbgk_handle *dst, void *d_dst, dbgk_handle *src, const void *d_src, size_t bytes, hipStream_t stream)
{
	if (bytes == 0) return DBGK_OK;
	if (dst->device == src->device && !(g_comm_ctx && g_comm_ctx->host_staging && getenv("DBGK_COMM_HOST_STAGING"))) {
		HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, stream));
		return DBGK_OK;
	}
	if (!g_comm_ctx || !g_comm_ctx->host_staging) {
		HIPCHK(hipMemcpyPeerAsync(d_dst, dst->device, d_src, src->device, bytes, stream));
		return DBGK_OK;
	}
	// no peer access: through pinned host memory, piece by piece, synchronously (slow but correct; ordered behind what `stream` holds)
	HIPCHK(hipStreamSynchronize(stream));
	for (size_t off = 0; off < bytes; off += kCommStageBytes) {
		const size_t len = std::min(kCommStageBytes, bytes - off);
		HIPCHK(hipSetDevice(src->device));
		HIPCHK(hipMemcpy(g_comm_ctx->stage, static_cast<const char *>(d_src) + off, len, hipMemcpyDeviceToHost));
		HIPCHK(hipSetDevice(dst->device));
		HIPCHK(hipMemcpy(static_cast<char *>(d_dst) + off, g_comm_ctx->stage, len, hipMemcpyHostToDevice));
	}
	return DBGK_OK;
}

struct CommScope { // every entry point that copies sets the context for comm_copy
	dbgk_comm *prev;
	explicit CommScope(dbgk_comm *c) : prev(g_comm_ctx) { g_comm_ctx = c; }
	~CommScope() { g_comm_ctx = prev; }
};

extern "C" int dbgk_comm_destroy(dbgk_comm *c)
{
	if (!c) return DBGK_ERR_ARG;
	for (size_t i = 0; i < c->h.size(); i++) {
		if (!c->h[i]) continue;
		(void)hipSetDevice(c->h[i]->device);
		if (i < c->copy_stream.size() && c->copy_stream[i]) {
			(void)hipStreamSynchronize(c->copy_stream[i]);
			(void)hipStreamDestroy(c->copy_stream[i]);
		}
		if (i < c->ev.size())
			for (hipEvent_t e : c->ev[i]) (void)hipEventDestroy(e);
		if (i < c->cnt_ev.size() && c->cnt_ev[i]) (void)hipEventDestroy(c->cnt_ev[i]);
		free_handle(c->h[i]);
	}
	if (c->stage) (void)hipHostFree(c->stage);
	delete c;
	return DBGK_OK;
}

extern "C" int dbgk_comm_create(const dbgk_config *cfg, const int32_t *devices, uint32_t n, dbgk_comm **out)
{
	if (!cfg || !devices || !out || n < 1 || n > 64) return DBGK_ERR_ARG;
	*out = nullptr;
	dbgk_comm *c = new (std::nothrow) dbgk_comm();
	if (!c) return DBGK_ERR_NOMEM;
	c->h.assign(n, nullptr);
	c->copy_stream.assign(n, nullptr);
	c->ev.assign(n, {});
	c->cnt_ev.assign(n, nullptr);
	c->delivered.assign(n, 0);
	if (const char *e = getenv("DBGK_COMM_PIECES")) c->pieces = (uint32_t)std::max(1, atoi(e));
	c->cfg = *cfg;
	c->devices.assign(devices, devices + n);
	c->host_staging = getenv("DBGK_COMM_HOST_STAGING") != nullptr; // tests / diagnosis: never use device-to-device copies between members
	for (uint32_t i = 0; i < n; i++) {
		dbgk_config one = *cfg;
		one.device_id = devices[i];
		if (cfg->engine == DBGK_ENGINE_KFREQ) {
			c->kfreq = true;
			one.shard_count = 0;
			one.shard_index = 0;
		} else if (cfg->engine == DBGK_ENGINE_WIDE) { // 128-bit keys: slot-range shards of 16-byte records (one pass: the reads stream through once)
			c->wide = true;
			one.shard_count = n;
			one.shard_index = i;
			one.n_passes = 1;
		} else {
			one.engine = DBGK_ENGINE_PARTITION;
			one.shard_count = n;
			one.shard_index = i;
		}
		int rc = dbgk_create(&one, &c->h[i]);
		if (rc == DBGK_OK && c->wide && c->h[i]->wgeom.n_passes != 1) {
			g_last_error = "dbgk_comm (WIDE): this table needs several passes over the input; a communicator streams its reads once -- "
			               "use the per-handle protocol (dbgk_wide_begin_pass) or fewer level-1 buckets (smaller table / fewer shards)";
			rc = DBGK_ERR_ARG;
		}
		if (rc == DBGK_OK && hipStreamCreateWithFlags(&c->copy_stream[i], hipStreamNonBlocking) != hipSuccess) rc = DBGK_ERR_HIP;
		if (rc == DBGK_OK && hipEventCreateWithFlags(&c->cnt_ev[i], hipEventDisableTiming) != hipSuccess) rc = DBGK_ERR_HIP;
		if (rc != DBGK_OK) {
			dbgk_comm_destroy(c);
			return rc;
		}
	}
	for (uint32_t i = 0; i < n; i++) // peer access between distinct devices (already enabled is fine)
		for (uint32_t j = 0; j < n; j++)
			if (devices[i] != devices[j]) {
				(void)hipSetDevice(devices[i]);
				const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
				if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) c->host_staging = true; // not peers: copies go through pinned host memory
				(void)hipGetLastError();
			}
	if (c->host_staging && hipHostMalloc(&c->stage, kCommStageBytes, hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		dbgk_comm_destroy(c);
		return DBGK_ERR_NOMEM;
	}
	*out = c;
	return DBGK_OK;
}

extern "C" uint32_t dbgk_comm_size(const dbgk_comm *c) { return c ? (uint32_t)c->h.size() : 0; }
extern "C" dbgk_handle *dbgk_comm_handle(dbgk_comm *c, uint32_t i) { return (c && i < c->h.size()) ? c->h[i] : nullptr; }

// steps 1-4 of the header comment on every shard
static int comm_exchange_and_build(dbgk_comm *c)
{
	CommScope scope(c);
	const uint32_t n = (uint32_t)c->h.size();
	int rc;
	for (dbgk_handle *h : c->h) { // every rank's level-1 store is complete
		rc = dbgk_sync(h);
		if (rc) return rc;
	}
	const PartGeom &G0 = c->h[0]->geom;
	const uint64_t bucket_bytes = (uint64_t)G0.n_sub * G0.cap1 * 8, chunk_bytes = (uint64_t)G0.B * bucket_bytes;
	const uint64_t cnt_chunk_bytes = (uint64_t)G0.B * G0.n_sub * 4;
	for (dbgk_handle *h : c->h)
		if (h->geom.cap1 != G0.cap1 || h->geom.B != G0.B || h->geom.size != G0.size) {
			g_last_error = "dbgk_comm: the shards do not share one bucket geometry";
			return DBGK_ERR_STATE;
		}
	// 1. fill counts
	for (uint32_t d = 0; d < n; d++) {
		dbgk_handle *D = c->h[d];
		rc = use_device(D);
		if (rc) return rc;
		for (uint32_t s = 0; s < n; s++) {
			dbgk_handle *S = c->h[s];
			rc = comm_copy(D, reinterpret_cast<char *>(D->inbox_cnt) + s * cnt_chunk_bytes, S,
			               reinterpret_cast<const char *>(S->store.cnt1) + d * cnt_chunk_bytes, cnt_chunk_bytes, c->copy_stream[d]);
			if (rc) return rc;
		}
		HIPCHK(hipEventRecord(c->cnt_ev[d], c->copy_stream[d]));
	}
	// 2. the record buckets, piece by piece; all copies are queued up front
	const uint32_t pieces = std::max(1u, std::min(c->pieces, G0.B));
	const uint32_t per = (G0.B + pieces - 1) / pieces;
	std::vector<std::pair<uint32_t, uint32_t>> ranges;
	for (uint32_t j0 = 0; j0 < G0.B; j0 += per) ranges.push_back({j0, std::min(j0 + per, G0.B)});
	for (uint32_t d = 0; d < n; d++) {
		dbgk_handle *D = c->h[d];
		rc = use_device(D);
		if (rc) return rc;
		while (c->ev[d].size() < ranges.size()) {
			hipEvent_t e;
			HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			c->ev[d].push_back(e);
		}
		for (size_t p = 0; p < ranges.size(); p++) {
			const uint64_t off = (uint64_t)ranges[p].first * bucket_bytes, len = (uint64_t)(ranges[p].second - ranges[p].first) * bucket_bytes;
			for (uint32_t s = 0; s < n; s++) {
				dbgk_handle *S = c->h[s];
				rc = comm_copy(D, reinterpret_cast<char *>(D->inbox) + s * chunk_bytes + off, S,
				               reinterpret_cast<const char *>(S->store.l1) + d * chunk_bytes + off, len, c->copy_stream[d]);
				if (rc) return rc;
			}
			HIPCHK(hipEventRecord(c->ev[d][p], c->copy_stream[d]));
		}
	}
	// 3. plan + build, piece by piece as the records arrive
	for (uint32_t d = 0; d < n; d++) {
		dbgk_handle *D = c->h[d];
		rc = use_device(D);
		if (rc) return rc;
		HIPCHK(hipStreamWaitEvent(D->stream, c->cnt_ev[d], 0));
		rc = part_plan(D);
		if (rc) return rc;
	}
	for (size_t p = 0; p < ranges.size(); p++)
		for (uint32_t d = 0; d < n; d++) {
			dbgk_handle *D = c->h[d];
			rc = use_device(D);
			if (rc) return rc;
			HIPCHK(hipStreamWaitEvent(D->stream, c->ev[d][p], 0));
			const uint32_t own0 = std::min(ranges[p].first, D->geom.nb_own), own1 = std::min(ranges[p].second, D->geom.nb_own);
			if (own1 > own0) {
				rc = part_build_range(D, own0, own1, true);
				if (rc) return rc;
			}
		}
	for (dbgk_handle *D : c->h) {
		rc = use_device(D);
		if (rc) return rc;
		D->exchanged = true;
		rc = build_from_records(D); // joins the streams, merges the shard's own spill nodes
		if (rc) return rc;
	}
	for (uint32_t d = 0; d < n; d++) {
		rc = use_device(c->h[d]);
		if (rc) return rc;
		HIPCHK(hipStreamSynchronize(c->h[d]->stream));
		HIPCHK(hipStreamSynchronize(c->copy_stream[d]));
	}
	{ // what every shard received must be what the others extracted for it: the fill counts, summed over the whole job on both sides
		unsigned long long sent = 0, received = 0;
		std::vector<uint32_t> buf((size_t)n * G0.B * G0.n_sub);
		for (uint32_t d = 0; d < n; d++) {
			rc = use_device(c->h[d]);
			if (rc) return rc;
			HIPCHK(hipMemcpy(buf.data(), c->h[d]->store.cnt1, buf.size() * 4, hipMemcpyDeviceToHost));
			for (uint32_t v : buf) sent += v;
			HIPCHK(hipMemcpy(buf.data(), c->h[d]->inbox_cnt, buf.size() * 4, hipMemcpyDeviceToHost));
			for (uint32_t v : buf) received += v;
		}
		if (sent != received) {
			g_last_error = "dbgk_comm: records sent (" + std::to_string(sent) + ") != records received (" + std::to_string(received) + "): a copy between shards went wrong";
			return DBGK_ERR_STATE;
		}
	}
	// 4. hand-offs.  A list of rank s is merged straight from s's memory when s and d share a device, through a
	// scratch copy otherwise.
	auto offer = [&](dbgk_handle *S, const Node *list, uint64_t count, int is_triple, int from_prev, dbgk_handle *D) -> int {
		if (count == 0) return DBGK_OK;
		int r = use_device(D);
		if (r) return r;
		const Node *src = list;
		Node *scratch = nullptr;
		if (S->device != D->device) {
			if (hipMalloc(&scratch, count * sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
			r = comm_copy(D, scratch, S, list, count * sizeof(Node), D->stream);
			src = scratch;
		}
		if (r == DBGK_OK) {
			hipLaunchKernelGGL(k_merge_sharded, dim3(grid_for(D, count)), dim3(kBlock), 0, D->stream, src, (const unsigned long long *)nullptr, count, count,
			                   is_triple, from_prev, D->geom, D->store, D->table, D->d_ctr);
			if (hipGetLastError() != hipSuccess) r = DBGK_ERR_HIP;
		}
		if (hipStreamSynchronize(D->stream) != hipSuccess) r = DBGK_ERR_HIP;
		if (scratch) (void)hipFree(scratch);
		return r;
	};
	auto read_u64 = [&](dbgk_handle *H, const unsigned long long *d_ptr, uint64_t &v) -> int {
		int r = use_device(H);
		if (r) return r;
		unsigned long long x = 0;
		HIPCHK(hipMemcpyAsync(&x, d_ptr, 8, hipMemcpyDeviceToHost, H->stream));
		HIPCHK(hipStreamSynchronize(H->stream));
		v = x;
		return DBGK_OK;
	};
	for (uint32_t s = 0; s < n; s++) { // overflow observations (and, beyond the list, the side table) of rank s: every rank keeps its own
		dbgk_handle *S = c->h[s];
		uint64_t v = 0;
		rc = read_u64(S, &S->store.ovf_n[0], v);
		if (rc) return rc;
		const uint64_t n_list = std::min<uint64_t>(v, S->store.ovf_cap);
		for (uint32_t d = 0; d < n; d++) {
			rc = offer(S, S->store.ovf, n_list, 1, 0, c->h[d]);
			if (rc) return rc;
			if (v > S->store.ovf_cap && S->store.hh) {
				rc = offer(S, S->store.hh, S->store.hh_size, 0, 0, c->h[d]);
				if (rc) return rc;
			}
		}
	}
	std::fill(c->delivered.begin(), c->delivered.end(), 0);
	for (uint32_t round = 0; round <= n; round++) { // nodes that ran off the end of shard s continue in shard s+1 (and, rarely, further)
		bool any = false;
		for (uint32_t s = 0; s < n; s++) {
			dbgk_handle *S = c->h[s];
			uint64_t v = 0;
			rc = read_u64(S, S->store.outgoing_n, v);
			if (rc) return rc;
			if (v > S->store.outgoing_cap) {
				g_last_error = "dbgk_comm: more nodes left a shard than its hand-over list holds (table nearly full?)";
				return DBGK_ERR_CAPACITY;
			}
			if (v > c->delivered[s]) {
				rc = offer(S, S->store.outgoing + c->delivered[s], v - c->delivered[s], 0, 1, c->h[(s + 1) % n]);
				if (rc) return rc;
				c->delivered[s] = v;
				any = true;
			}
		}
		if (!any) break;
		if (round == n) {
			g_last_error = "dbgk_comm: handed-over nodes went round all shards without finding a slot (table full)";
			return DBGK_ERR_TABLE_FULL;
		}
	}
	return DBGK_OK;
}

// The same steps for a communicator of WIDE handles (16-byte records, one pass): fill counts, record buckets in pieces
// overlapped with level 2 + region build, then -- finalize only, a WIDE store is built once -- the hand-offs with 32-byte
// entries and the side tables (keys with a zero low word, key-0 links) gathered onto shard 0.
static int comm_wide_finalize(dbgk_comm *c)
{
	CommScope scope(c);
	const uint32_t n = (uint32_t)c->h.size();
	int rc;
	for (dbgk_handle *h : c->h) {
		rc = dbgk_sync(h);
		if (rc) return rc;
	}
	const WPartGeom &G0 = c->h[0]->wgeom;
	const uint64_t bucket_bytes = G0.cap1 * 16, chunk_bytes = (uint64_t)G0.Bp * bucket_bytes, cnt_chunk_bytes = (uint64_t)G0.Bp * 4;
	for (dbgk_handle *h : c->h)
		if (!h->wpart || h->wgeom.cap1 != G0.cap1 || h->wgeom.Bp != G0.Bp || h->wgeom.size != G0.size || !h->wpass_open) {
			g_last_error = "dbgk_comm (WIDE): the shards do not share one bucket geometry";
			return DBGK_ERR_STATE;
		}
	if (n > 1) {
		for (uint32_t d = 0; d < n; d++) { // 1. fill counts
			dbgk_handle *D = c->h[d];
			rc = use_device(D);
			if (rc) return rc;
			for (uint32_t s = 0; s < n; s++) {
				dbgk_handle *S = c->h[s];
				rc = comm_copy(D, reinterpret_cast<char *>(D->winbox_cnt) + s * cnt_chunk_bytes, S,
				               reinterpret_cast<const char *>(S->wstore.cnt1) + d * cnt_chunk_bytes, cnt_chunk_bytes, c->copy_stream[d]);
				if (rc) return rc;
			}
			HIPCHK(hipEventRecord(c->cnt_ev[d], c->copy_stream[d]));
		}
	}
	// 2. the record buckets, piece by piece; all copies are queued up front
	const uint32_t pieces = std::max(1u, std::min(c->pieces, G0.Bp));
	const uint32_t per = (G0.Bp + pieces - 1) / pieces;
	std::vector<std::pair<uint32_t, uint32_t>> ranges;
	for (uint32_t j0 = 0; j0 < G0.Bp; j0 += per) ranges.push_back({j0, std::min(j0 + per, G0.Bp)});
	for (uint32_t d = 0; d < n; d++) {
		dbgk_handle *D = c->h[d];
		rc = use_device(D);
		if (rc) return rc;
		while (c->ev[d].size() < ranges.size()) {
			hipEvent_t e;
			HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			c->ev[d].push_back(e);
		}
		for (size_t p = 0; p < ranges.size(); p++) {
			const uint64_t off = (uint64_t)ranges[p].first * bucket_bytes, len = (uint64_t)(ranges[p].second - ranges[p].first) * bucket_bytes;
			if (n > 1)
				for (uint32_t s = 0; s < n; s++) {
					dbgk_handle *S = c->h[s];
					rc = comm_copy(D, reinterpret_cast<char *>(D->winbox) + s * chunk_bytes + off, S,
					               reinterpret_cast<const char *>(S->wstore.l1) + d * chunk_bytes + off, len, c->copy_stream[d]);
					if (rc) return rc;
				}
			HIPCHK(hipEventRecord(c->ev[d][p], c->copy_stream[d]));
		}
	}
	// 3. plan + build, piece by piece as the records arrive
	for (uint32_t d = 0; d < n; d++) {
		dbgk_handle *D = c->h[d];
		rc = use_device(D);
		if (rc) return rc;
		if (n > 1) HIPCHK(hipStreamWaitEvent(D->stream, c->cnt_ev[d], 0));
		rc = wide_plan_pass(D);
		if (rc) return rc;
	}
	for (size_t p = 0; p < ranges.size(); p++)
		for (uint32_t d = 0; d < n; d++) {
			dbgk_handle *D = c->h[d];
			rc = use_device(D);
			if (rc) return rc;
			HIPCHK(hipStreamWaitEvent(D->stream, c->ev[d][p], 0));
			const uint32_t nb = wide_pass_buckets(D);
			const uint32_t own0 = std::min(ranges[p].first, nb), own1 = std::min(ranges[p].second, nb);
			if (own1 > own0) {
				rc = wide_build_range(D, own0, own1);
				if (rc) return rc;
			}
		}
	for (uint32_t d = 0; d < n; d++) {
		dbgk_handle *D = c->h[d];
		D->exchanged = true;
		rc = dbgk_finalize(D, nullptr); // ends the pass, merges the shard's own spill nodes, reads the counters
		if (rc) return rc;
		rc = use_device(D);
		if (rc) return rc;
		HIPCHK(hipStreamSynchronize(c->copy_stream[d]));
	}
	if (n > 1) { // what every shard received must be what the others extracted for it (as in the 64-bit flow): the fill counts of the pass,
	             // summed over the whole job on both sides -- a copy between shards that moved only part of a message fails here
		unsigned long long sent = 0, received = 0;
		std::vector<uint32_t> buf((size_t)G0.n_l1);
		for (uint32_t d = 0; d < n; d++) {
			rc = use_device(c->h[d]);
			if (rc) return rc;
			HIPCHK(hipMemcpy(buf.data(), c->h[d]->wstore.cnt1, buf.size() * 4, hipMemcpyDeviceToHost));
			for (uint32_t v : buf) sent += v;
			HIPCHK(hipMemcpy(buf.data(), c->h[d]->winbox_cnt, buf.size() * 4, hipMemcpyDeviceToHost));
			for (uint32_t v : buf) received += v;
		}
		if (sent != received) {
			g_last_error = "dbgk_comm (WIDE): records sent (" + std::to_string(sent) + ") != records received (" + std::to_string(received) + "): a copy between shards went wrong";
			return DBGK_ERR_STATE;
		}
	}
	// 4. hand-offs: a list of shard s is merged straight from s's memory when s and d share a device, through a scratch copy otherwise
	auto offer = [&](dbgk_handle *S, const void *list, uint64_t count, int is_obs, int from_prev, dbgk_handle *D, bool side_nodes) -> int {
		if (count == 0) return DBGK_OK;
		int r = use_device(D);
		if (r) return r;
		const void *src = list;
		void *scratch = nullptr;
		if (S->device != D->device) {
			if (hipMalloc(&scratch, count * sizeof(dbgk_node32)) != hipSuccess) return DBGK_ERR_NOMEM;
			r = comm_copy(D, scratch, S, list, count * sizeof(dbgk_node32), D->stream);
			src = scratch;
		}
		if (r == DBGK_OK)
			r = side_nodes ? dbgk_wide_merge_nodes(D, reinterpret_cast<const dbgk_node32 *>(src), count)
			               : dbgk_shard_merge(D, reinterpret_cast<const dbgk_node *>(src), count, is_obs, from_prev);
		if (hipStreamSynchronize(D->stream) != hipSuccess) r = DBGK_ERR_HIP;
		if (scratch) (void)hipFree(scratch);
		return r;
	};
	for (uint32_t s = 0; s < n; s++) { // overflow observations of shard s: every shard keeps its own
		dbgk_node *list = nullptr;
		uint64_t cnt = 0;
		rc = dbgk_shard_overflow(c->h[s], &list, &cnt);
		if (rc) return rc;
		for (uint32_t d = 0; d < n; d++) {
			rc = offer(c->h[s], list, cnt, 1, 0, c->h[d], false);
			if (rc) return rc;
		}
	}
	std::fill(c->delivered.begin(), c->delivered.end(), 0);
	for (uint32_t round = 0; round <= n; round++) { // nodes that ran off the end of shard s continue in shard s+1 (and, rarely, further)
		bool any = false;
		for (uint32_t s = 0; s < n; s++) {
			dbgk_node *list = nullptr;
			uint64_t v = 0;
			rc = dbgk_shard_outgoing(c->h[s], &list, &v);
			if (rc) return rc;
			if (v > c->delivered[s]) {
				rc = offer(c->h[s], reinterpret_cast<const dbgk_node32 *>(list) + c->delivered[s], v - c->delivered[s], 0, 1, c->h[(s + 1) % n], false);
				if (rc) return rc;
				c->delivered[s] = v;
				any = true;
			}
		}
		if (!any) break;
		if (round == n) {
			g_last_error = "dbgk_comm: handed-over nodes went round all shards without finding a slot (table full)";
			return DBGK_ERR_TABLE_FULL;
		}
	}
	// 5. the nodes that live outside the table (zero low word, key 0): onto shard 0
	for (uint32_t s = 1; s < n; s++) {
		dbgk_node32 *list = nullptr;
		uint64_t cnt = 0;
		rc = dbgk_shard_side_export(c->h[s], &list, &cnt);
		if (rc) return rc;
		rc = offer(c->h[s], list, cnt, 0, 0, c->h[0], true);
		if (rc) return rc;
		rc = dbgk_shard_side_clear(c->h[s]);
		if (rc) return rc;
	}
	for (dbgk_handle *h : c->h) {
		rc = use_device(h);
		if (rc) return rc;
		rc = read_counters(h);
		if (rc) return rc;
		if (h->h_ctr->error & 1u) return DBGK_ERR_TABLE_FULL;
		if (h->h_ctr->error & 2u) return DBGK_ERR_CAPACITY;
	}
	return DBGK_OK;
}

extern "C" int dbgk_comm_flush(dbgk_comm *c)
{
	if (!c || c->finalized) return DBGK_ERR_STATE;
	if (c->wide) {
		bool pending = false;
		for (dbgk_handle *h : c->h) pending = pending || h->pending_kmers > 0;
		if (!pending) return DBGK_OK;
		g_last_error = "dbgk_comm (WIDE): the record stores are built once, at dbgk_comm_finalize (expected_kmers sizes them for the whole input)";
		return DBGK_ERR_CAPACITY;
	}
	if (c->kfreq) { // members are independent until finalize
		for (dbgk_handle *h : c->h) {
			const int rc = dbgk_flush(h);
			if (rc) return rc;
		}
		return DBGK_OK;
	}
	bool pending = false;
	for (dbgk_handle *h : c->h) pending = pending || h->pending_kmers > 0;
	if (!pending) return DBGK_OK;
	int rc = comm_exchange_and_build(c);
	if (rc) return rc;
	for (dbgk_handle *h : c->h) {
		rc = use_device(h);
		if (rc) return rc;
		rc = read_counters(h); // surfaces capacity / table-full conditions of this round
		if (rc) return rc;
		if (h->h_ctr->error & 1u) return DBGK_ERR_TABLE_FULL;
		if (h->h_ctr->error & 2u) return DBGK_ERR_CAPACITY;
		h->incr = true;
		rc = clear_record_store(h);
		if (rc) return rc;
	}
	return DBGK_OK;
}

// enlarge_kmerset_parallel (kmerSet.cpp:132-189) for a table that lives on several GPUs: new shards of a table of new_slots
// slots are created, every node of every old shard is re-seated into the new shard that owns its new home slot (each new shard
// scans the old shards and keeps its own keys: k_merge_sharded), nodes that run off the end of a shard are handed on, the totals
// and the key-0 links carry over, the old shards are freed.  The record stores must be empty: pending records are flushed first.
extern "C" int dbgk_comm_resize(dbgk_comm *c, uint64_t new_slots)
{
	if (!c || new_slots < 3) return DBGK_ERR_ARG;
	if (c->finalized || c->kfreq || c->wide) return DBGK_ERR_STATE;
	int rc = dbgk_comm_flush(c);
	if (rc) return rc;
	if (new_slots == c->h[0]->size) return DBGK_OK;
	CommScope scope(c);
	const uint32_t n = (uint32_t)c->h.size();
	// A shard that has never been through a region build (nothing pushed, or nothing flushed yet: dbgk_comm_flush returns early
	// when no records wait) holds NO table -- its memory is whatever hipMalloc returned (zero_pending): nothing to carry over
	// from it, and scanning it would re-seat garbage.  resize_partition_table guards the single handle the same way.
	bool any_built = false;
	for (dbgk_handle *S : c->h) any_built = any_built || S->incr;
	std::vector<dbgk_handle *> fresh(n, nullptr);
	auto drop_fresh = [&]() {
		for (dbgk_handle *f : fresh)
			if (f) free_handle(f);
	};
	for (uint32_t i = 0; i < n && rc == DBGK_OK; i++) {
		dbgk_config one = c->cfg;
		one.device_id = c->devices[i];
		one.engine = DBGK_ENGINE_PARTITION;
		one.shard_count = n;
		one.shard_index = i;
		one.table_slots = new_slots;
		rc = dbgk_create(&one, &fresh[i]);
		if (rc == DBGK_OK) rc = use_device(fresh[i]);
		if (rc == DBGK_OK && any_built) {
			rc = zero_table_now(fresh[i]);                // nodes arrive through the atomic path
			if (rc == DBGK_OK) fresh[i]->incr = true;    // later region builds load them back
		}                                                 // (else: a fresh shard as dbgk_comm_create leaves it -- the first build writes every slot)
	}
	if (rc) {
		drop_fresh();
		return rc;
	}
	const uint64_t chunk_nodes = 16ull << 20; // 256 MiB of nodes at a time when source and destination sit on different GPUs
	for (uint32_t d = 0; d < n && rc == DBGK_OK; d++) {
		dbgk_handle *D = fresh[d];
		Node *scratch = nullptr;
		for (uint32_t s = 0; s < n && rc == DBGK_OK; s++) {
			dbgk_handle *S = c->h[s];
			if (!S->incr) continue; // never built: no nodes
			for (uint64_t off = 0; off < S->tslots && rc == DBGK_OK; off += chunk_nodes) {
				const uint64_t cnt = std::min(chunk_nodes, S->tslots - off);
				const Node *src = S->table + off;
				rc = use_device(D);
				if (rc) break;
				if (S->device != D->device || c->host_staging) {
					if (!scratch && hipMalloc(&scratch, chunk_nodes * sizeof(Node)) != hipSuccess) { rc = DBGK_ERR_NOMEM; break; }
					rc = comm_copy(D, scratch, S, src, cnt * sizeof(Node), D->stream);
					src = scratch;
					if (rc) break;
				}
				hipLaunchKernelGGL(k_merge_sharded, dim3(grid_for(D, cnt)), dim3(kBlock), 0, D->stream, src, (const unsigned long long *)nullptr, cnt, cnt, 0, 0, D->geom,
				                   D->store, D->table, D->d_ctr);
				if (hipGetLastError() != hipSuccess || hipStreamSynchronize(D->stream) != hipSuccess) rc = DBGK_ERR_HIP;
			}
		}
		if (scratch) (void)hipFree(scratch);
	}
	// nodes that ran off the end of a new shard continue in the next one
	std::vector<uint64_t> delivered(n, 0);
	for (uint32_t round = 0; round <= n && rc == DBGK_OK; round++) {
		bool any = false;
		for (uint32_t s = 0; s < n && rc == DBGK_OK; s++) {
			dbgk_handle *S = fresh[s], *D = fresh[(s + 1) % n];
			unsigned long long v = 0;
			rc = use_device(S);
			if (rc) break;
			if (hipMemcpy(&v, S->store.outgoing_n, 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = DBGK_ERR_HIP; break; }
			if (v > S->store.outgoing_cap) { rc = DBGK_ERR_CAPACITY; break; }
			if (v <= delivered[s]) continue;
			const uint64_t cnt = v - delivered[s];
			const Node *src = S->store.outgoing + delivered[s];
			Node *scratch = nullptr;
			rc = use_device(D);
			if (rc) break;
			if (S->device != D->device || c->host_staging) {
				if (hipMalloc(&scratch, cnt * sizeof(Node)) != hipSuccess) { rc = DBGK_ERR_NOMEM; break; }
				rc = comm_copy(D, scratch, S, src, cnt * sizeof(Node), D->stream);
				src = scratch;
			}
			if (rc == DBGK_OK) {
				hipLaunchKernelGGL(k_merge_sharded, dim3(grid_for(D, cnt)), dim3(kBlock), 0, D->stream, src, (const unsigned long long *)nullptr, cnt, cnt, 0, 1, D->geom,
				                   D->store, D->table, D->d_ctr);
				if (hipGetLastError() != hipSuccess || hipStreamSynchronize(D->stream) != hipSuccess) rc = DBGK_ERR_HIP;
			}
			if (scratch) (void)hipFree(scratch);
			delivered[s] = v;
			any = true;
		}
		if (!any) break;
		if (round == n) rc = DBGK_ERR_TABLE_FULL;
	}
	// totals and key-0 links carry over shard by shard (they belong to the reads a shard extracted, not to its slot range)
	for (uint32_t i = 0; i < n && rc == DBGK_OK; i++) {
		dbgk_handle *O = c->h[i], *F = fresh[i];
		Counters co, cf;
		rc = use_device(O);
		if (rc) break;
		if (hipMemcpy(&co, O->d_ctr, sizeof co, hipMemcpyDeviceToHost) != hipSuccess) { rc = DBGK_ERR_HIP; break; }
		rc = use_device(F);
		if (rc) break;
		if (hipMemcpy(&cf, F->d_ctr, sizeof cf, hipMemcpyDeviceToHost) != hipSuccess) { rc = DBGK_ERR_HIP; break; }
		if (cf.error & 1u) { rc = DBGK_ERR_TABLE_FULL; break; }
		if (cf.error & 2u) { rc = DBGK_ERR_CAPACITY; break; }
		cf.total_reads = co.total_reads;
		cf.total_kmers = co.total_kmers;
		cf.stored_kmers = co.stored_kmers;
		cf.polyA_links = co.polyA_links;
		cf.other_bytes = co.other_bytes;
		cf.other_seen = co.other_seen;
		cf.n_conflict += co.n_conflict;
		if (hipMemcpy(F->d_ctr, &cf, sizeof cf, hipMemcpyHostToDevice) != hipSuccess) { rc = DBGK_ERR_HIP; break; }
		if (hipMemsetAsync(F->store.outgoing_n, 0, 8, F->stream) != hipSuccess || hipStreamSynchronize(F->stream) != hipSuccess) { rc = DBGK_ERR_HIP; break; }
		F->total_reads = O->total_reads;
		F->host_other_bytes = O->host_other_bytes;
	}
	if (rc) {
		drop_fresh();
		return rc;
	}
	for (uint32_t i = 0; i < n; i++) {
		free_handle(c->h[i]);
		c->h[i] = fresh[i];
	}
	c->cfg.table_slots = new_slots;
	return DBGK_OK;
}

// `packed` != null: the batch is 2 bits per base (dbgk_push_reads_packed), `bases` is unused
static int comm_push_impl(dbgk_comm *c, const char *bases, const uint32_t *packed, const uint64_t *offsets, uint64_t n_reads, uint64_t other_bytes)
{
	if (!c || !offsets) return DBGK_ERR_ARG;
	if (c->finalized) return DBGK_ERR_STATE;
	dbgk_handle *h = c->h[c->next_push];
	auto push = [&](dbgk_handle *m, const uint64_t *off, uint64_t n, uint64_t other) {
		return packed ? dbgk_push_reads_packed(m, packed, off, n, other) : dbgk_push_reads(m, bases, off, n);
	};
	if (c->kfreq) { // the member streams through its own record store
		c->next_push = (c->next_push + 1) % (uint32_t)c->h.size();
		return push(h, offsets, n_reads, other_bytes);
	}
	uint64_t windows = 0;
	const uint64_t K = (uint64_t)h->cfg.kmer_size, max_len = (uint64_t)h->cfg.max_read_len;
	for (uint64_t i = 0; i < n_reads; i++) {
		const uint64_t len = offsets[i + 1] - offsets[i], rl = len > max_len ? max_len : len;
		if (rl >= K) windows += rl - K + 1;
	}
	const uint32_t n = (uint32_t)c->h.size();
	// A large batch is cut into one piece per member and the pieces are handed over by as many host threads at once (each member
	// has its own pinned staging buffers, stream and PCIe link): one thread copying batch after batch tops out near 30 GB/s whatever
	// the number of GPUs.  Small batches go to one member, round robin.
	static const bool serial = DBGK_EXPERIMENT_ENV("DBGK_COMM_SERIAL_PUSH") != nullptr;
	const uint64_t bytes = offsets[n_reads] - offsets[0];
	if (!serial && n > 1 && n_reads >= 4ull * n && bytes >= (8ull << 20)) {
		uint64_t worst = 0;
		for (dbgk_handle *m : c->h) worst = std::max(worst, m->pending_kmers);
		if (worst > 0 && worst + windows / n + (windows >> 4) > h->store_capacity) { // some store would run full: everybody exchanges and builds
			int rc = dbgk_comm_flush(c);
			if (rc) return rc;
		}
		std::vector<uint64_t> cut(n + 1, n_reads); // cut[i]: first read of member i's piece (equal byte shares)
		cut[0] = 0;
		for (uint32_t i = 1; i < n; i++) {
			const uint64_t want = offsets[0] + bytes * i / n;
			cut[i] = (uint64_t)(std::lower_bound(offsets, offsets + n_reads + 1, want) - offsets);
			cut[i] = std::max(cut[i - 1], std::min(cut[i], n_reads));
		}
		std::vector<int> rcs(n, DBGK_OK);
		std::vector<std::string> errs(n);
		auto push_piece = [&](uint32_t i) {
			dbgk_handle *m = c->h[(c->next_push + i) % n];
			if (cut[i + 1] > cut[i]) rcs[i] = push(m, offsets + cut[i], cut[i + 1] - cut[i], i == 0 ? other_bytes : 0);
			if (rcs[i]) errs[i] = g_last_error; // (the message is per thread)
		};
		std::vector<std::thread> th;
		for (uint32_t i = 1; i < n; i++) th.emplace_back(push_piece, i);
		push_piece(0);
		for (auto &t : th) t.join();
		c->next_push = (c->next_push + 1) % n;
		for (uint32_t i = 0; i < n; i++)
			if (rcs[i]) {
				g_last_error = errs[i];
				return rcs[i];
			}
		return DBGK_OK;
	}
	if (h->pending_kmers > 0 && h->pending_kmers + windows > h->store_capacity) { // this rank's store is full: everybody exchanges and builds
		int rc = dbgk_comm_flush(c);
		if (rc) return rc;
	}
	c->next_push = (c->next_push + 1) % n;
	return push(h, offsets, n_reads, other_bytes);
}

extern "C" int dbgk_comm_push_reads(dbgk_comm *c, const char *bases, const uint64_t *offsets, uint64_t n_reads)
{
	return comm_push_impl(c, bases, nullptr, offsets, n_reads, 0);
}

extern "C" int dbgk_comm_push_reads_packed(dbgk_comm *c, const uint32_t *packed, const uint64_t *offsets, uint64_t n_reads, uint64_t other_bytes)
{
	if (!packed && n_reads && offsets && offsets[n_reads] != offsets[0]) return DBGK_ERR_ARG;
	static const uint32_t none = 0;
	return comm_push_impl(c, nullptr, packed ? packed : &none, offsets, n_reads, other_bytes);
}

// KFREQ: the reduce-scatter of SURVEY 8(e)-4 with the exact combination rule (saturating byte add) instead of a
// wrapping sum.  Owner d pulls the slice of every other member through two staging buffers of kStage bytes on its
// own GPU: the peer copy of chunk j+1 (copy stream) runs while chunk j is added (the handle's stream).
static int comm_kfreq_finalize(dbgk_comm *c)
{
	CommScope scope(c);
	const uint32_t n = (uint32_t)c->h.size();
	for (dbgk_handle *h : c->h) {
		const int rc = dbgk_finalize(h, nullptr);
		if (rc) return rc;
	}
	const uint64_t total = c->h[0]->n_counts;
	c->kf_lo.assign(n + 1, 0);
	for (uint32_t d = 0; d <= n; d++) c->kf_lo[d] = d == n ? total : (uint64_t)((unsigned __int128)total * d / n) & ~(uint64_t)1023; // 4^k >= 1024 for k >= 5
	c->kf_distinct = 0;
	if (n == 1) {
		c->kf_distinct = c->h[0]->kf_distinct;
		return DBGK_OK;
	}
	const uint64_t kStage = dbgk_hook("comm_kfreq_stage_mb") ? (uint64_t)std::max(1, atoi(dbgk_hook("comm_kfreq_stage_mb"))) << 20 : 256ull << 20;
	struct Owner {
		uint8_t *stage[2] = {nullptr, nullptr};
		hipEvent_t arrived[2] = {nullptr, nullptr}, merged[2] = {nullptr, nullptr};
		uint64_t chunks = 0;
	};
	std::vector<Owner> own(n);
	int rc = DBGK_OK;
	auto fail = [&](hipError_t e, int line) { if (e != hipSuccess && rc == DBGK_OK) rc = hip_fail(e, "kfreq reduce", line); return e != hipSuccess; };
	for (uint32_t d = 0; d < n && rc == DBGK_OK; d++) {
		if (use_device(c->h[d])) { rc = DBGK_ERR_HIP; break; }
		for (int b = 0; b < 2 && rc == DBGK_OK; b++) {
			if (hipMalloc(&own[d].stage[b], kStage) != hipSuccess) { (void)hipGetLastError(); rc = DBGK_ERR_NOMEM; break; }
			if (fail(hipEventCreateWithFlags(&own[d].arrived[b], hipEventDisableTiming), __LINE__)) break;
			if (fail(hipEventCreateWithFlags(&own[d].merged[b], hipEventDisableTiming), __LINE__)) break;
		}
	}
	// all owners advance together, one chunk of one source at a time, so that every link carries traffic
	for (uint32_t step = 1; step < n && rc == DBGK_OK; step++)
		for (uint64_t off = 0;; off += kStage) {
			bool any = false;
			for (uint32_t d = 0; d < n && rc == DBGK_OK; d++) {
				const uint64_t lo = c->kf_lo[d] + off, hi = c->kf_lo[d + 1];
				if (lo >= hi) continue;
				any = true;
				dbgk_handle *hd = c->h[d], *hs = c->h[(d + step) % n];
				Owner &o = own[d];
				const uint64_t bytes = std::min(kStage, hi - lo);
				const int b = (int)(o.chunks & 1);
				if (use_device(hd)) { rc = DBGK_ERR_HIP; break; }
				if (o.chunks >= 2 && fail(hipStreamWaitEvent(c->copy_stream[d], o.merged[b], 0), __LINE__)) break;
				rc = comm_copy(hd, o.stage[b], hs, hs->counts + lo, bytes, c->copy_stream[d]);
				if (rc) break;
				if (fail(hipEventRecord(o.arrived[b], c->copy_stream[d]), __LINE__)) break;
				if (fail(hipStreamWaitEvent(hd->stream, o.arrived[b], 0), __LINE__)) break;
				hipLaunchKernelGGL(k_counts_merge, dim3(grid_for(hd, bytes >> 4)), dim3(kBlock), 0, hd->stream, hd->counts + lo, o.stage[b], bytes);
				if (fail(hipGetLastError(), __LINE__)) break;
				if (fail(hipEventRecord(o.merged[b], hd->stream), __LINE__)) break;
				o.chunks++;
			}
			if (!any || rc) break;
		}
	for (uint32_t d = 0; d < n; d++) {
		if (use_device(c->h[d])) continue;
		(void)hipStreamSynchronize(c->copy_stream[d]);
		(void)hipStreamSynchronize(c->h[d]->stream);
		for (int b = 0; b < 2; b++) {
			if (own[d].stage[b]) (void)hipFree(own[d].stage[b]);
			if (own[d].arrived[b]) (void)hipEventDestroy(own[d].arrived[b]);
			if (own[d].merged[b]) (void)hipEventDestroy(own[d].merged[b]);
		}
	}
	if (rc) return rc;
	for (uint32_t d = 0; d < n; d++) {
		rc = use_device(c->h[d]);
		if (rc) return rc;
		unsigned long long res[2];
		rc = kfreq_summary(c->h[d], c->kf_lo[d], c->kf_lo[d + 1] - c->kf_lo[d], res);
		if (rc) return rc;
		c->kf_distinct += res[0];
	}
	return DBGK_OK;
}

// the table of the whole job: every range is read from the member that owns it
extern "C" int dbgk_comm_kfreq_export_counts(dbgk_comm *c, uint64_t first_kmer, uint64_t n, uint8_t *host_out)
{
	if (!c || !host_out) return DBGK_ERR_ARG;
	if (!c->kfreq || !c->finalized) return DBGK_ERR_STATE;
	const uint64_t total = c->h[0]->n_counts;
	if (first_kmer > total || n > total - first_kmer) return DBGK_ERR_ARG;
	for (size_t d = 0; d < c->h.size(); d++) {
		const uint64_t lo = std::max(first_kmer, c->kf_lo[d]), hi = std::min(first_kmer + n, c->kf_lo[d + 1]);
		if (lo >= hi) continue;
		const int rc = dbgk_kfreq_export_counts(c->h[d], lo, hi - lo, host_out + (lo - first_kmer));
		if (rc) return rc;
	}
	return DBGK_OK;
}

extern "C" int dbgk_comm_kfreq_export_bits(dbgk_comm *c, uint32_t cutoff, uint64_t first_byte, uint64_t n_bytes, uint8_t *host_out)
{
	if (!c || !host_out) return DBGK_ERR_ARG;
	if (!c->kfreq || !c->finalized) return DBGK_ERR_STATE;
	const uint64_t total_bytes = c->h[0]->n_counts >> 3;
	if (first_byte > total_bytes || n_bytes > total_bytes - first_byte) return DBGK_ERR_ARG;
	for (size_t d = 0; d < c->h.size(); d++) { // slice bounds are multiples of 1024 k-mers = 128 bytes of the bit table
		const uint64_t lo = std::max(first_byte, c->kf_lo[d] >> 3), hi = std::min(first_byte + n_bytes, c->kf_lo[d + 1] >> 3);
		if (lo >= hi) continue;
		const int rc = dbgk_kfreq_export_bits(c->h[d], cutoff, lo, hi - lo, host_out + (lo - first_byte));
		if (rc) return rc;
	}
	return DBGK_OK;
}

static void comm_sum_stats(dbgk_comm *c, dbgk_stats *out)
{
	memset(out, 0, sizeof *out);
	for (dbgk_handle *h : c->h) {
		dbgk_stats st;
		fill_stats(h, &st);
		out->total_reads += st.total_reads;
		out->total_kmers += st.total_kmers;
		out->stored_kmers += st.stored_kmers;
		out->count += st.count; // only shard 0 counts the key-0 node
		out->count_conflict += st.count_conflict;
		out->other_bytes += st.other_bytes;
	}
	out->table_slots = c->h[0]->size;
	if (c->kfreq) {
		out->count = c->kf_distinct;
		out->count_conflict = 0;
		out->table_slots = c->h[0]->n_counts;
		return;
	}
	const Counters &c0 = *c->h[0]->h_ctr;
	out->polyA_l_link = (uint32_t)(c0.polyA_links & 0xFFFFFFFFu);
	out->polyA_r_link = (uint32_t)(c0.polyA_links >> 32);
}

extern "C" int dbgk_comm_refresh_stats(dbgk_comm *c, dbgk_stats *out)
{
	if (!c || !out) return DBGK_ERR_ARG;
	for (dbgk_handle *h : c->h) {
		int rc = use_device(h);
		if (rc) return rc;
		rc = read_counters(h);
		if (rc) return rc;
	}
	comm_sum_stats(c, out);
	return DBGK_OK;
}

extern "C" int dbgk_comm_finalize(dbgk_comm *c, dbgk_stats *out)
{
	if (!c) return DBGK_ERR_ARG;
	if (c->finalized) return DBGK_ERR_STATE;
	if (c->kfreq) {
		int rc = comm_kfreq_finalize(c);
		if (rc) return rc;
		c->finalized = true;
		if (out) comm_sum_stats(c, out);
		return DBGK_OK;
	}
	if (c->wide) {
		int wrc = comm_wide_finalize(c);
		if (wrc) return wrc;
		c->finalized = true;
		if (out) comm_sum_stats(c, out);
		return DBGK_OK;
	}
	int rc = comm_exchange_and_build(c);
	if (rc) return rc;
	for (dbgk_handle *h : c->h) {
		rc = use_device(h);
		if (rc) return rc;
		rc = read_counters(h);
		if (rc) return rc;
		h->finalized = true;
		h->pending_kmers = 0;
		if (h->h_ctr->error & 1u) return DBGK_ERR_TABLE_FULL;
		if (h->h_ctr->error & 2u) return DBGK_ERR_CAPACITY;
	}
	// 5. the key-0 node lives on shard 0
	for (size_t s = 1; s < c->h.size(); s++) {
		const unsigned long long links = c->h[s]->h_ctr->polyA_links;
		if (links) {
			rc = dbgk_add_polyA(c->h[0], (uint32_t)(links & 0xFFFFFFFFu), (uint32_t)(links >> 32));
			if (rc) return rc;
		}
	}
	rc = use_device(c->h[0]);
	if (rc) return rc;
	rc = read_counters(c->h[0]);
	if (rc) return rc;
	c->finalized = true;
	if (out) comm_sum_stats(c, out);
	return DBGK_OK;
}

extern "C" int dbgk_comm_digest(dbgk_comm *c, uint64_t *digest)
{
	if (!c || !digest) return DBGK_ERR_ARG;
	if (!c->finalized) return DBGK_ERR_STATE;
	uint64_t sum = 0;
	for (dbgk_handle *h : c->h) {
		uint64_t d = 0;
		int rc = dbgk_digest(h, &d);
		if (rc) return rc;
		sum += d;
	}
	*digest = sum;
	return DBGK_OK;
}

extern "C" int dbgk_comm_link_stats(dbgk_comm *c, int32_t cutoff, dbgk_link_stats *out)
{
	if (!c || !out) return DBGK_ERR_ARG;
	if (!c->finalized) return DBGK_ERR_STATE;
	memset(out, 0, sizeof *out);
	for (dbgk_handle *h : c->h) {
		dbgk_link_stats one;
		int rc = dbgk_link_stats_device(h, cutoff, &one);
		if (rc) return rc;
		for (int i = 0; i < 256; i++) out->depth_stat[i] += one.depth_stat[i];
		out->total_nodes += one.total_nodes;
		out->deleted_lowfreq += one.deleted_lowfreq;
		out->linear_nodes += one.linear_nodes;
		out->tip_nodes += one.tip_nodes;
		out->branch_nodes += one.branch_nodes;
	}
	return DBGK_OK;
}

// The host KmerSet of the whole job.  host_size == the global table size: the shards ARE pieces of that table,
// their slices are copied side by side (slot ranges start at multiples of 2^r >= 2^20: byte-aligned in nul_flag) and
// the key-0 node is put on key 0's probe chain (add_node_to_kmerset, kmerSet.cpp:253-273).  Any other size: every
// node is re-seated on the host by linear probing (the reference's own insert rule, DBGgraph.cpp:167-205).
extern "C" int dbgk_comm_export_host_table(dbgk_comm *c, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag)
{
	if (!c || !array || !nul_flag || host_size < 3) return DBGK_ERR_ARG;
	if (!c->finalized || c->wide || c->kfreq) return DBGK_ERR_STATE; // (WIDE: dbgk_comm_wide_export_*)
	dbgk_stats tot;
	comm_sum_stats(c, &tot);
	if (tot.count > host_size) return DBGK_ERR_TABLE_FULL;
	auto set_flag = [&](uint64_t i) { nul_flag[i >> 3] |= (uint8_t)(128u >> (i & 7u)); };
	auto flag_set = [&](uint64_t i) { return (nul_flag[i >> 3] & (uint8_t)(128u >> (i & 7u))) != 0; };
	const uint64_t global = c->h[0]->size;
	if (host_size == global) {
		memset(nul_flag, 0, host_size / 8 + 1);
		for (dbgk_handle *h : c->h) {
			const uint64_t lo = h->geom.slot_lo, len = h->tslots;
			std::vector<uint8_t> fl(len / 8 + 1);
			int rc = dbgk_export_host_table(h, len, array + lo, fl.data());
			if (rc) return rc;
			memcpy(nul_flag + lo / 8, fl.data(), (len + 7) / 8); // the last shard ends the table: its spare bits are zero
		}
	} else {
		memset(array, 0, host_size * sizeof(dbgk_node));
		memset(nul_flag, 0, host_size / 8 + 1);
		for (dbgk_handle *h : c->h) { // only the occupied nodes of a shard travel (compacted on the device), not its whole slot range
			dbgk_stats st;
			fill_stats(h, &st);
			std::vector<dbgk_node> part(st.count ? st.count : 1);
			uint64_t n = 0;
			int rc = dbgk_export_sorted(h, part.data(), part.size(), &n);
			if (rc) return rc;
			for (uint64_t i = 0; i < n; i++) {
				if (part[i].kmer == 0) continue; // shard 0 lists the key-0 node: placed last, below
				uint64_t hc = hash_code(part[i].kmer) % host_size;
				while (flag_set(hc)) hc = (hc + 1 == host_size) ? 0 : hc + 1;
				array[hc] = part[i];
				set_flag(hc);
			}
		}
	}
	uint64_t hc = hash_code(0ull) % host_size; // the key-0 node: first slot without a flag on key 0's chain
	while (flag_set(hc)) hc = (hc + 1 == host_size) ? 0 : hc + 1;
	array[hc].kmer = 0;
	array[hc].l_link = tot.polyA_l_link;
	array[hc].r_link = tot.polyA_r_link;
	set_flag(hc);
	return DBGK_OK;
}

// dbgk_export_host_table_links for the table of a communicator (calculate_kmer_links' first pass, DBG_contig/contig.cpp:107-181,
// for exactly the table that is handed over).  Slot numbers -- and with them the ascending order of tip_nodes / branch_nodes -- are
// those of ONE table of host_size slots, so the shards' nodes are first brought together in such a table on the first member's
// device (a temporary handle: the shard tables travel there in pieces and are merged in, the device counterpart of the re-seating
// dbgk_comm_export_host_table does on the host when host_size differs from the device table); everything after that is the
// single-handle export, bit for bit.  Needs host_size * 16 bytes (+ the link arrays) free on that device.
extern "C" int dbgk_comm_export_host_table_links(dbgk_comm *c, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag, int32_t kmer_freq_cutoff,
                                                 uint16_t *klink, uint8_t *del_flag, uint64_t *tip_nodes, uint64_t tip_capacity, uint64_t *n_tips,
                                                 uint64_t *branch_nodes, uint64_t branch_capacity, uint64_t *n_branches, dbgk_link_stats *stats)
{
	if (!c || !array || !nul_flag || !klink || !del_flag || !n_tips || !n_branches || host_size < 3) return DBGK_ERR_ARG;
	if (!c->finalized || c->wide || c->kfreq) return DBGK_ERR_STATE;
	dbgk_stats tot;
	comm_sum_stats(c, &tot);
	if (tot.count > host_size) return DBGK_ERR_TABLE_FULL;
	dbgk_config one = c->cfg;
	one.device_id = c->devices[0];
	one.engine = DBGK_ENGINE_DIRECT;
	one.shard_count = one.shard_index = 0;
	one.table_slots = host_size;
	one.expected_kmers = 0;
	one.flags = 0;
	dbgk_handle *T = nullptr;
	int rc = dbgk_create(&one, &T);
	if (rc) return rc;
	const uint64_t chunk_nodes = 16ull << 20; // 256 MiB of slots at a time when a shard sits on another GPU
	Node *scratch = nullptr;
	for (dbgk_handle *S : c->h) {
		for (uint64_t off = 0; off < S->tslots && rc == DBGK_OK; off += chunk_nodes) {
			const uint64_t cnt = std::min(chunk_nodes, S->tslots - off);
			const Node *src = S->table + off;
			rc = use_device(T);
			if (rc) break;
			if (S->device != T->device || c->host_staging) {
				if (!scratch && hipMalloc(&scratch, chunk_nodes * sizeof(Node)) != hipSuccess) { rc = DBGK_ERR_NOMEM; break; }
				rc = comm_copy(T, scratch, S, src, cnt * sizeof(Node), T->stream);
				src = scratch;
				if (rc) break;
			}
			rc = dbgk_merge_nodes(T, reinterpret_cast<const dbgk_node *>(src), cnt); // (empty slots are all-zero entries: they add nothing)
			if (rc == DBGK_OK) rc = dbgk_sync(T); // the scratch buffer / the shard's memory is read by the kernel just queued
		}
		if (rc) break;
	}
	if (scratch) {
		(void)hipSetDevice(T->device);
		(void)hipFree(scratch);
	}
	if (rc == DBGK_OK) rc = dbgk_add_polyA(T, tot.polyA_l_link, tot.polyA_r_link);
	dbgk_stats st;
	if (rc == DBGK_OK) rc = dbgk_finalize(T, &st);
	if (rc == DBGK_OK && st.count != tot.count) {
		g_last_error = "dbgk_comm_export_host_table_links: the assembled table holds " + std::to_string(st.count) + " nodes, the shards " + std::to_string(tot.count);
		rc = DBGK_ERR_STATE;
	}
	if (rc == DBGK_OK)
		rc = dbgk_export_host_table_links(T, host_size, array, nul_flag, kmer_freq_cutoff, klink, del_flag, tip_nodes, tip_capacity, n_tips, branch_nodes,
		                                  branch_capacity, n_branches, stats);
	dbgk_destroy(T);
	return rc;
}

// ---- WIDE communicators: results as 32-byte nodes ----------------------------------------------------------------------
// all nodes of the job sorted by (kmer_hi, kmer_lo), the key-0 node first; capacity >= stats.count
extern "C" int dbgk_comm_wide_export_sorted(dbgk_comm *c, dbgk_node32 *out, uint64_t capacity, uint64_t *n_out)
{
	if (!c || !out || !n_out) return DBGK_ERR_ARG;
	if (!c->finalized || !c->wide) return DBGK_ERR_STATE;
	uint64_t at = 0;
	for (dbgk_handle *h : c->h) {
		uint64_t got = 0;
		int rc = dbgk_wide_export_sorted(h, out + at, capacity - at, &got);
		if (rc) return rc;
		at += got;
	}
	*n_out = at;
	std::sort(out, out + at, [](const dbgk_node32 &a, const dbgk_node32 &b) {
		return a.kmer_hi < b.kmer_hi || (a.kmer_hi == b.kmer_hi && a.kmer_lo < b.kmer_lo);
	});
	return DBGK_OK;
}

// the host table of the whole job (host_size == the global table size): the shards' slices side by side, then the nodes that
// live outside the table on the devices (zero low word, key 0) put on their probe chains by add_node_to_kmerset's rule
// (kmerSet.cpp:253-273) -- the contract of dbgk_wide_export_host_table for ONE table over several GPUs
extern "C" int dbgk_comm_wide_export_host_table(dbgk_comm *c, uint64_t host_size, dbgk_node32 *array, uint8_t *nul_flag)
{
	if (!c || !array || !nul_flag) return DBGK_ERR_ARG;
	if (!c->finalized || !c->wide) return DBGK_ERR_STATE;
	if (host_size != c->h[0]->size) return DBGK_ERR_ARG;
	dbgk_stats tot;
	comm_sum_stats(c, &tot);
	if (tot.count > host_size) return DBGK_ERR_TABLE_FULL;
	memset(nul_flag, 0, host_size / 8 + 1);
	for (dbgk_handle *h : c->h) {
		const uint64_t lo = h->wgeom.slot_lo, len = h->tslots;
		std::vector<uint8_t> fl(len / 8 + 1);
		int rc = dbgk_wide_export_host_table(h, len, array + lo, fl.data());
		if (rc) return rc;
		memcpy(nul_flag + lo / 8, fl.data(), (len + 7) / 8); // slot ranges start at multiples of 2^r: byte-aligned
	}
	dbgk_handle *h0 = c->h[0];
	int rc = use_device(h0);
	if (rc) return rc;
	std::vector<WNode> side(kWideSideSlots);
	HIPCHK(hipMemcpyAsync(side.data(), h0->wside, kWideSideSlots * sizeof(WNode), hipMemcpyDeviceToHost, h0->stream));
	HIPCHK(hipStreamSynchronize(h0->stream));
	auto place = [&](dbgk_node32 nd) {
		uint64_t hc = dbgk_wide::hash128(dbgk_wide::Key128{nd.kmer_hi, nd.kmer_lo}) % host_size;
		while (nul_flag[hc >> 3] & (uint8_t)(128u >> (hc & 7u))) hc = (hc + 1 == host_size) ? 0 : hc + 1;
		array[hc] = nd;
		nul_flag[hc >> 3] |= (uint8_t)(128u >> (hc & 7u));
	};
	for (const WNode &sd : side)
		if (sd.hi1) place(dbgk_node32{sd.hi1 - 1ull, 0ull, (uint32_t)sd.links, (uint32_t)(sd.links >> 32), 0});
	place(dbgk_node32{0, 0, tot.polyA_l_link, tot.polyA_r_link, 0}); // DBGgraph.cpp:418
	return DBGK_OK;
}
