// dbgk_host_build.h -- part of libdbgk.so's host side (one translation unit: included by dbgk.hip, in this order).
// level 2 and the region build: launches, EARLY level-2 rounds between the batches, ranged finalize, fix-ups
#pragma once

static int read_counters(dbgk_handle *h)
{
	HIPCHK(hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	for (auto &s : h->slots) s.busy = false;
	return collect_spans(h);
}

static void fill_stats(const dbgk_handle *h, dbgk_stats *out)
{
	const Counters &c = *h->h_ctr;
	out->total_reads = h->total_reads;
	out->total_kmers = c.total_kmers;
	out->stored_kmers = c.stored_kmers;
	// + the key-0 node, always present (DBGgraph.cpp:418); of a sharded table only shard 0 reports it
	out->count = c.n_new + ((h->sharded && h->shard_rank != 0) ? 0 : 1);
	if (h->seed) out->count = c.n_new + (c.polyA_links ? 1 : 0); // key 0 is an ordinary key of the seed index
	out->count_conflict = c.n_conflict;
	out->table_slots = h->tslots;
	out->polyA_l_link = (uint32_t)(c.polyA_links & 0xFFFFFFFFu);
	out->polyA_r_link = (uint32_t)(c.polyA_links >> 32);
	out->other_bytes = c.other_bytes + h->host_other_bytes;
}

// PARTITION engine: records -> final buckets -> table regions, then the stragglers
template <int DBG>
static void launch_l2(dbgk_handle *h, int grid, uint32_t j0, uint32_t j1)
{
	if (h->three) { // mid pass, plan of the mid buckets filled so far, final pass (see the handle's comment)
		hipLaunchKernelGGL(k_scatter_l2<0>, dim3(grid), dim3(l2_threads(kMaxBuckets)), sizeof(ScatterLdsL2), h->stream, h->g_mid, h->s_mid, h->tile_prefix, h->d_ctr, j0, j1);
		hipLaunchKernelGGL(k_plan_l2, dim3(1), dim3(kMaxBuckets), 0, h->stream, h->g_fin, h->s_fin, h->tile_prefix2);
		hipLaunchKernelGGL(k_scatter_l2<0>, dim3(grid), dim3(l2_threads(kMaxBuckets)), sizeof(ScatterLdsL2), h->stream, h->g_fin, h->s_fin, h->tile_prefix2, h->d_ctr,
		                   j0 * h->fan_mid, j1 * h->fan_mid);
		return;
	}
	if (h->geom.n2 > 2048u) // tables of 2^33 slots and more
		hipLaunchKernelGGL((k_scatter_l2<0, 4096>), dim3(grid), dim3(l2_threads(4096)), sizeof(ScatterLdsL2T<4096>), h->stream, h->geom, h->store, h->tile_prefix, h->d_ctr, j0, j1);
	else if (h->geom.n2 > (uint32_t)kMaxBuckets) // 2^32 .. 2^33 slots
		hipLaunchKernelGGL((k_scatter_l2<0, 2048>), dim3(grid), dim3(l2_threads(2048)), sizeof(ScatterLdsL2T<2048>), h->stream, h->geom, h->store, h->tile_prefix, h->d_ctr, j0, j1);
	else if (h->geom.kf == 2u) // KFREQ, direct blocks: 32-bit level-1 records (n2 <= 1024 always)
		hipLaunchKernelGGL((k_scatter_l2<0, kMaxBuckets, true>), dim3(grid), dim3(l2_threads(kMaxBuckets)), sizeof(ScatterLdsL2), h->stream, h->geom, h->store, h->tile_prefix, h->d_ctr, j0, j1);
	else
		hipLaunchKernelGGL(k_scatter_l2<DBG>, dim3(grid), dim3(l2_threads(kMaxBuckets)), sizeof(ScatterLdsL2), h->stream, h->geom, h->store, h->tile_prefix, h->d_ctr, j0, j1);
}

static RedoList redo_list(dbgk_handle *h)
{
	return RedoList{h->region_cursor + kMaxBuildLaunches + 2, h->region_cursor + kMaxBuildLaunches + 1, h->geom.n_regions_own};
}

// FAST form of the insert (four records per thread in flight, plain adds on the link words, regions whose counters pass 255
// left to the exact pass) unless DBGK_BUILD_EXACT=1 asks for the saturating CAS loops everywhere
static bool build_fast()
{
	const bool exact = dbgk_hook("build_exact") && atoi(dbgk_hook("build_exact")) != 0;
	return !exact;
}

template <int DBG>
static void launch_build(dbgk_handle *h, hipStream_t stream, uint32_t first_region, uint32_t n_regions, unsigned int *cursor)
{
	static const int per_cu = DBGK_EXPERIMENT_ENV("DBGK_BUILD_PER_CU") ? std::max(1, atoi(DBGK_EXPERIMENT_ENV("DBGK_BUILD_PER_CU"))) : 2; // tuning knob
	const uint32_t grid = std::min<uint32_t>(n_regions, (uint32_t)h->n_cu * (uint32_t)per_cu); // persistent: two 66-KiB workgroups fit a CU
	const RedoList redo = redo_list(h);
	if (h->geom.kf == 2u) { // KFREQ, direct blocks
#define DBGK_KFB(INCR, FAST) \
	hipLaunchKernelGGL((k_kf_build_blocks<INCR, FAST>), dim3(grid), dim3(kBuildThreads), sizeof(KfBlockLds), stream, h->geom, h->store, h->counts, \
	                   h->d_ctr, first_region, n_regions, cursor, redo)
		if (build_fast()) { if (h->incr) DBGK_KFB(true, true); else DBGK_KFB(false, true); }
		else { if (h->incr) DBGK_KFB(true, false); else DBGK_KFB(false, false); }
#undef DBGK_KFB
		return;
	}
	Node *counts = reinterpret_cast<Node *>(h->counts);
#define DBGK_BUILD(D, KF, INCR, FAST, TABLE) \
	hipLaunchKernelGGL((k_build_regions<D, KF, INCR, FAST>), dim3(grid), dim3(kBuildThreads), sizeof(BuildLds), stream, h->geom, h->store, TABLE, h->d_ctr, \
	                   first_region, n_regions, cursor, redo)
	if (build_fast()) {
		if (h->kfreq && h->incr) DBGK_BUILD(0, true, true, true, counts);
		else if (h->kfreq) DBGK_BUILD(0, true, false, true, counts);
		else if (h->incr && DBG == 0) DBGK_BUILD(0, false, true, true, h->table);
		else DBGK_BUILD(DBG, false, false, true, h->table);
	} else {
		if (h->kfreq && h->incr) DBGK_BUILD(0, true, true, false, counts);
		else if (h->kfreq) DBGK_BUILD(0, true, false, false, counts);
		else if (h->incr && DBG == 0) DBGK_BUILD(0, false, true, false, h->table);
		else DBGK_BUILD(DBG, false, false, false, h->table);
	}
#undef DBGK_BUILD
}

// the exact pass over the regions the fast launches flagged (normally none: the kernel finds an empty list and returns)
static void launch_build_redo(dbgk_handle *h, hipStream_t stream)
{
	const uint32_t grid = (uint32_t)h->n_cu * 2u;
	const RedoList redo = redo_list(h);
	unsigned int *cursor = h->region_cursor + kMaxBuildLaunches;
	if (h->geom.kf == 2u) {
		if (h->incr)
			hipLaunchKernelGGL((k_kf_build_blocks<true, false, true>), dim3(grid), dim3(kBuildThreads), sizeof(KfBlockLds), stream, h->geom, h->store,
			                   h->counts, h->d_ctr, 0u, 0u, cursor, redo);
		else
			hipLaunchKernelGGL((k_kf_build_blocks<false, false, true>), dim3(grid), dim3(kBuildThreads), sizeof(KfBlockLds), stream, h->geom, h->store,
			                   h->counts, h->d_ctr, 0u, 0u, cursor, redo);
		return;
	}
	Node *counts = reinterpret_cast<Node *>(h->counts);
#define DBGK_REDO(KF, INCR, TABLE) \
	hipLaunchKernelGGL((k_build_regions<0, KF, INCR, false, true>), dim3(grid), dim3(kBuildThreads), sizeof(BuildLds), stream, h->geom, h->store, TABLE, \
	                   h->d_ctr, 0u, 0u, cursor, redo)
	if (h->kfreq && h->incr) DBGK_REDO(true, true, counts);
	else if (h->kfreq) DBGK_REDO(true, false, counts);
	else if (h->incr) DBGK_REDO(false, true, h->table);
	else DBGK_REDO(false, false, h->table);
#undef DBGK_REDO
}

// Level 2 is bound by the memory system (8 waves per CU and 72 KiB of LDS reach the same time as a
// full CU), the region build by instruction issue: they run CONCURRENTLY.  The own level-1 buckets
// are cut into chunks; level 2 of chunk c+1 runs on `stream` while the regions of chunk c are built
// on `stream2` (one 512-thread level-2 workgroup and one 1024-thread build workgroup fit a CU together).
// ---- pieces of the finalize of the PARTITION engine ---------------------------------------------
// part_plan: level-2 tile plan for all own buckets (needs every inbox fill count);
// part_build_range: level 2 + region build of the own buckets [j0, j1), asynchronous, level 2 on
// `stream`, the build behind it on `stream2`; part_finish: join, spill / overflow fix-ups.
// EARLY level 2.  Level 2 is an append into the final buckets, so it does not have to wait for the end of the input: every
// push first queues a level-2 round over the records the batches before it left in the level-1 buckets (tile plan from the fill
// counts minus what earlier rounds took, P.l2_done), then its own level 1.  A job whose batches come over the link (the reference
// overlaps reading and parsing the same way, DBGgraph.cpp:233-296) keeps the GPU busy with level 2 while the next batch travels;
// after the last batch only that batch's level 2 and the region build remain.  A job that is pushed in one piece (bench.py's
// resident step) is unchanged: its only round runs at dbgk_finalize.  Rounds are only worth their tiles' fixed costs when there
// is something to scatter: at least kEarlyL2Min occurrences since the last one.
constexpr uint64_t kEarlyL2Min = 8ull << 20;
static int early_l2(dbgk_handle *h)
{
	if (!h->part || !h->l2_done || h->part_planned || h->part_built) return DBGK_OK;
	const char *e_min = dbgk_hook("early_l2_min"); // (read per call: the tests ask for a round after every small batch)
	const uint64_t min_kmers = e_min ? strtoull(e_min, nullptr, 10) : kEarlyL2Min;
	if (h->pending_kmers < h->l2_seen_kmers + std::max<uint64_t>(min_kmers, 1)) return DBGK_OK;
	const PartGeom &G = h->geom;
	static const int l2_grid_env = DBGK_EXPERIMENT_ENV("DBGK_L2_GRID") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_L2_GRID")) : 0;
	const int l2_grid = l2_grid_env >= 8 ? (l2_grid_env & ~7) : h->n_cu;
	TimedSpan sp;
	int rc = span_begin(h, PH_PARTITION, sp);
	if (rc) return rc;
	hipLaunchKernelGGL(k_plan_l2, dim3(1), dim3(kMaxBuckets), 0, h->stream, h->geom, h->store, h->tile_prefix);
	launch_l2<0>(h, l2_grid, 0, G.nb_own);
	HIPCHK(hipGetLastError());
	h->l2_seen_kmers = h->pending_kmers;
	return span_end(h, sp);
}

static int part_plan(dbgk_handle *h)
{
	if (h->part_planned) return DBGK_OK;
	if (!h->stream2) {
		HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
		HIPCHK(hipEventCreateWithFlags(&h->join_ev, hipEventDisableTiming));
	}
	hipLaunchKernelGGL(k_plan_l2, dim3(1), dim3(kMaxBuckets), 0, h->stream, h->geom, h->store, h->tile_prefix);
	HIPCHK(hipGetLastError());
	h->cursors_used = 0; // (the work cursors were zeroed with the record stores' control arrays: clear_record_store)
	int rc = span_begin(h, PH_L2_BUILD_WALL, h->wall_span);
	if (rc) return rc;
	h->part_planned = true;
	h->next_bucket = 0;
	h->chunks_used = 0;
	return DBGK_OK;
}

static int part_build_range(dbgk_handle *h, uint32_t j0, uint32_t j1, bool two_streams)
{
	const PartGeom &G = h->geom;
	static const int l2_grid_env = DBGK_EXPERIMENT_ENV("DBGK_L2_GRID") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_L2_GRID")) : 0; // tuning knob: level-2 workgroups (multiple of 8)
	const int l2_grid = l2_grid_env >= 8 ? (l2_grid_env & ~7) : h->n_cu;
	static const int dbg_l2 = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_L2") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_L2")) : 0;       // timing experiments, results are wrong
	static const int dbg_build = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_BUILD") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_BUILD")) : 0;
	if (j0 >= j1) return DBGK_OK;
	hipStream_t bstream = two_streams ? h->stream2 : h->stream;
	TimedSpan sp;
	int rc = span_begin(h, PH_PARTITION, sp);
	if (rc) return rc;
	switch (dbg_l2) {
#ifdef DBGK_EXPERIMENTS
		case 1: launch_l2<1>(h, l2_grid, j0, j1); break;
		case 2: launch_l2<2>(h, l2_grid, j0, j1); break;
		case 3: launch_l2<3>(h, l2_grid, j0, j1); break;
#endif
		default: launch_l2<0>(h, l2_grid, j0, j1); break;
	}
	HIPCHK(hipGetLastError());
	rc = span_end(h, sp);
	if (rc) return rc;
	h->next_bucket = j1;
	if (dbg_l2) return DBGK_OK; // never build regions from the garbage a timing experiment leaves behind
	if (two_streams) {
		if (h->chunk_ev.size() <= h->chunks_used) {
			hipEvent_t e;
			HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			h->chunk_ev.push_back(e);
		}
		hipEvent_t ev = h->chunk_ev[h->chunks_used++];
		HIPCHK(hipEventRecord(ev, h->stream));
		HIPCHK(hipStreamWaitEvent(bstream, ev, 0));
	}
	const uint32_t r0 = j0 * G.n2, r1 = std::min(j1 * G.n2, G.n_regions_own);
	if (r1 <= r0) return DBGK_OK;
	rc = span_begin(h, PH_BUILD, sp, bstream);
	if (rc) return rc;
	if (h->cursors_used >= kMaxBuildLaunches) {
		g_last_error = "too many build ranges in one step";
		return DBGK_ERR_STATE;
	}
	unsigned int *cursor = h->region_cursor + h->cursors_used++;
	switch (dbg_build) {
#ifdef DBGK_EXPERIMENTS
		case 1: launch_build<1>(h, bstream, r0, r1 - r0, cursor); break;
		case 2: launch_build<2>(h, bstream, r0, r1 - r0, cursor); break;
		case 3: launch_build<3>(h, bstream, r0, r1 - r0, cursor); break;
#endif
		default: launch_build<0>(h, bstream, r0, r1 - r0, cursor); break;
	}
	HIPCHK(hipGetLastError());
	return span_end(h, sp, bstream);
}

// Level 2 is bound by the memory system (8 waves per CU and 72 KiB of LDS reach the same time as a
// full CU), the region build by instruction issue: they run CONCURRENTLY.  The own level-1 buckets
// are cut into chunks; level 2 of chunk c+1 runs on `stream` while the regions of chunk c are built
// on `stream2` (one 512-thread level-2 workgroup and one 1024-thread build workgroup fit a CU together).
static int build_from_records(dbgk_handle *h)
{
	const PartGeom &G = h->geom;
	static const int dbg_l2 = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_L2") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_L2")) : 0;
	static const int dbg_build = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_BUILD") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_BUILD")) : 0;
	// 1 = level 2, then the build.  Round 5: with the leaner build and 16-byte record loads either kernel alone runs close to what the
	// memory system gives this traffic (4.1 + 3.7 ms for 2 x 19.2 GB), side by side they only share it: 1 / 2 / 3 / 4 / 6 / 12 chunk pairs ->
	// 12.86-12.92 / 12.86 / 13.00 / 13.03 / 13.07 / 13.06-13.24 ms per cfg2 step (profiles/r05_build_lean_walk_and_chunks_ab.txt; rounds 2-4: 12)
	static const int want_chunks = DBGK_EXPERIMENT_ENV("DBGK_OVERLAP_CHUNKS") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_OVERLAP_CHUNKS")) : 1;
	uint32_t n_chunks = ((dbg_l2 | dbg_build) != 0 || want_chunks < 1) ? 1u : (uint32_t)want_chunks;
	int rc = part_plan(h);
	if (rc) return rc;
	// whatever the caller has not built by ranges yet (dbgk_shard_build_range): all of it, normally
	const uint32_t left = G.nb_own - std::min(h->next_bucket, G.nb_own);
	if (n_chunks > left) n_chunks = left ? left : 1u;
	const bool two_streams = n_chunks > 1 || h->chunks_used > 0;
	const uint32_t per = (left + n_chunks - 1) / n_chunks;
	for (uint32_t c = 0; c < n_chunks && h->next_bucket < G.nb_own; c++) {
		const uint32_t j0 = h->next_bucket, j1 = std::min(j0 + per, G.nb_own);
		rc = part_build_range(h, j0, j1, two_streams);
		if (rc) return rc;
	}
	if (h->chunks_used > 0) { // everything after this point is ordered behind the last build on `stream` again
		HIPCHK(hipEventRecord(h->join_ev, h->stream2));
		HIPCHK(hipStreamWaitEvent(h->stream, h->join_ev, 0));
	}
	rc = span_end(h, h->wall_span);
	if (rc) return rc;
	h->part_planned = false;
	if ((dbg_l2 | dbg_build) != 0) {
		g_last_error = "DBGK_DEBUG_L2 / DBGK_DEBUG_BUILD set: timing experiment, no valid table was built";
		return DBGK_ERR_STATE;
	}
	TimedSpan sp;
	h->zero_pending = false; // every slot has just been written
	rc = span_begin(h, PH_FIXUP, sp);
	if (rc) return rc;
	if (build_fast()) launch_build_redo(h, h->stream); // regions with a link counter beyond 255: rebuilt exactly (before their spill nodes are merged)
	if (h->kfreq) {
		Counters *track = h->kf_blocks ? h->d_ctr : nullptr; // direct blocks: the table summary is kept as the table is written
		hipLaunchKernelGGL(k_kf_apply, dim3(h->n_cu), dim3(kBlock), 0, h->stream, h->store.spill, &h->store.ovf_n[1], h->store.spill_cap, 0,
		                   reinterpret_cast<uint32_t *>(h->counts), track);
		hipLaunchKernelGGL(k_kf_apply, dim3(h->n_cu), dim3(kBlock), 0, h->stream, h->store.ovf, &h->store.ovf_n[0], h->store.ovf_cap, 1,
		                   reinterpret_cast<uint32_t *>(h->counts), track);
		hipLaunchKernelGGL(k_kf_apply_table, dim3(grid_for(h, h->store.hh_size)), dim3(kBlock), 0, h->stream, h->store.hh, h->store.hh_size,
		                   reinterpret_cast<uint32_t *>(h->counts), track);
		hipLaunchKernelGGL(k_kf_key0, dim3(1), dim3(64), 0, h->stream, h->d_ctr, h->counts, h->kf_blocks ? 1 : 0);
	} else if (!h->sharded) {
		hipLaunchKernelGGL(k_merge_spill, dim3(h->n_cu), dim3(kBlock), 0, h->stream, h->store.spill, &h->store.ovf_n[1], h->store.spill_cap,
		                   h->tref(), h->d_ctr);
		hipLaunchKernelGGL(k_insert_triples, dim3(h->n_cu), dim3(kBlock), 0, h->stream, h->store.ovf, &h->store.ovf_n[0], h->store.ovf_cap,
		                   h->tref(), h->d_ctr);
		// the aggregated surplus of heavy hitters (empty slots are all-zero records and add nothing)
		hipLaunchKernelGGL(k_merge_nodes, dim3(grid_for(h, h->store.hh_size)), dim3(kBlock), 0, h->stream, h->store.hh, h->store.hh_size,
		                   h->tref(), h->d_ctr, (const unsigned long long *)h->store.ovf_n, (unsigned long long)h->store.ovf_cap); // (in use only once the overflow list is full)
	} else {
		// spill nodes of this shard's regions stay in the shard unless they run off its end (-> outgoing);
		// overflow triples may belong to any shard: the caller exchanges them (dbgk_shard_overflow)
		hipLaunchKernelGGL(k_merge_sharded, dim3(h->n_cu), dim3(kBlock), 0, h->stream, h->store.spill, &h->store.ovf_n[1], (uint64_t)0,
		                   h->store.spill_cap, 0, 0, G, h->store, h->table, h->d_ctr);
	}
	HIPCHK(hipGetLastError());
	rc = span_end(h, sp);
	h->part_built = true;
	return rc;
}

static int kfreq_summary(dbgk_handle *h, uint64_t first, uint64_t n, unsigned long long res[2]);
