// dbgk_host_wide.h -- part of libdbgk.so's host side (one translation unit: included by dbgk.hip, in this order).
// WIDE engine (k <= 63), host side of its record path: geometry, record stores, passes, level 2 + build per chunk
#pragma once

// ---- WIDE through radix-partitioned records (dbgk_wide_partition.h) ---------------------------------
// geometry: level-1 bucket = slot >> r, final bucket = slot >> 11 (one 2048-slot region), n2 = 2^(r - 11) <= 2048; the second
// half of a record holds q = hash / size, r slot bits, 6 neighbour bits.  The level-1 kernel fans out to at most 1024 store
// entries per pass: rank x own-bucket index inside the pass's window (WPartGeom).  One rank, one pass, n1 <= 1024: the
// round-2 form.  *err: the configuration ASKS for shards / passes and cannot have them.
static bool plan_wide_partition(dbgk_handle *h, bool *err)
{
	*err = false;
	const bool off = dbgk_hook("wide_direct") != nullptr; // always the atomic kernels
	const uint32_t n_ranks = h->cfg.shard_count > 1 ? h->cfg.shard_count : 1;
	const bool want_shard = h->cfg.shard_count >= 1;
	const uint32_t want_passes = (uint32_t)h->cfg.n_passes;
	auto refuse = [&](const char *why) {
		if (want_shard || want_passes > 1) {
			g_last_error = why;
			*err = true;
		}
		return false;
	};
	if (off || h->cfg.expected_kmers == 0) return refuse("a sharded / multi-pass WIDE handle needs expected_kmers > 0 (the record path)");
	if (h->size < (1ull << 26) || h->size >= (1ull << 34)) return refuse("the WIDE record path needs 2^26 <= table_slots < 2^34");
	if (want_shard && h->cfg.shard_index >= n_ranks) return refuse("shard_index >= shard_count");
	const uint64_t qmax = ~0ull / h->size;
	int qbits = 0;
	while (qbits < 64 && (qmax >> qbits)) qbits++;
	uint32_t r = 21;
	if (const char *e = DBGK_EXPERIMENT_ENV("DBGK_WIDE_R")) r = (uint32_t)std::max(kWRegionBits + 1, std::min(22, atoi(e)));
	if (!want_shard && want_passes <= 1)
		while (r < 22u && ((h->size + (1ull << r) - 1) >> r) > 1024ull) r++; // one pass if the fan-out allows it
	// passes: at least as many as keep the level-1 fan-out (ranks x buckets of a pass) within 1024
	auto passes_at = [&](uint32_t rr) {
		const uint64_t nn1 = (h->size + (1ull << rr) - 1) >> rr;
		const uint32_t BB = (uint32_t)((nn1 + n_ranks - 1) / n_ranks);
		uint32_t pp = std::max<uint32_t>(1u, want_passes);
		while ((uint64_t)n_ranks * ((BB + pp - 1) / pp) > 1024ull) pp++;
		return pp;
	};
	// every pass extracts the whole input again: beyond two passes the wider level-1 buckets of r = 22 (half the passes, level 2
	// fanning out 2048 ways) are the better trade
	if (!DBGK_EXPERIMENT_ENV("DBGK_WIDE_R") && r == 21u && passes_at(21u) > 2u && passes_at(22u) < passes_at(21u)) r = 22u;
	while (r > (uint32_t)kWRegionBits + 1u && qbits + (int)r + 6 > 64) r--;
	const uint64_t n1 = (h->size + (1ull << r) - 1) >> r;
	if ((1u << (r - kWRegionBits)) > 2048u || qbits + (int)r + 6 > 64 || n1 >= 65536ull) return refuse("no feasible wide record geometry for this table size");
	const uint32_t B = (uint32_t)((n1 + n_ranks - 1) / n_ranks);
	uint32_t n_passes = passes_at(r);
	if (n_passes > B) return refuse("shard_count too large for this table size");
	// Several passes are a PROTOCOL (begin_pass / push everything / end_pass, per pass): a caller who asked for neither shards nor
	// passes (n_passes == 0, the plain create / push / finalize flow) never gets it -- a table whose fan-out one pass cannot cover
	// is then built by the atomic kernels, as before the record path existed.  n_passes >= 1 says "I follow the protocol".
	if (n_passes > 1 && !want_shard && want_passes == 0) {
		// (not an error -- the handle works -- but a large performance step: say so where a caller can find it, once on stderr too)
		g_last_error = "WIDE handle: this table's level-1 fan-out needs several passes over the input and dbgk_config.n_passes is 0 -- "
		               "built by the atomic kernels; set n_passes (dbgk_wide_pass_info) for the record path";
		static std::atomic<bool> told{false};
		if (!told.exchange(true)) fprintf(stderr, "dbgk: %s\n", g_last_error.c_str());
		return false;
	}
	WPartGeom &G = h->wgeom;
	memset(&G, 0, sizeof G);
	G.size = h->size;
	G.magic = h->magic;
	if (h->size < (1ull << 32)) G.div = make_div32_magic((uint32_t)h->size);
	G.r = r;
	G.n1 = (uint32_t)n1;
	G.n2 = 1u << (r - kWRegionBits);
	G.n_regions = (uint32_t)((h->size + kWRegionSlots - 1) >> kWRegionBits);
	G.n_ranks = n_ranks;
	G.rank = want_shard ? h->cfg.shard_index : 0;
	G.B = B;
	G.bmagic = (uint32_t)(((1ull << 32) + B - 1) / B);
	G.b_lo = std::min(G.rank * B, G.n1);
	G.nb_own = std::min(B, G.n1 - G.b_lo);
	if (G.nb_own == 0) return refuse("shard_count too large for this table size");
	G.n_passes = n_passes;
	G.Bp = (B + n_passes - 1) / n_passes;
	G.pass = 0;
	G.pass_j0 = 0;
	G.n_l1 = n_ranks * G.Bp;
	G.slot_lo = (uint64_t)G.b_lo << r;
	G.slot_hi = std::min<uint64_t>(h->size, ((uint64_t)G.b_lo + G.nb_own) << r);
	// expected_kmers = occurrences THIS handle extracts (per pass over its input); a region receives the global density
	const double per_slot_mine = (double)h->cfg.expected_kmers / (double)h->size;
	G.cap1 = (uint64_t)(per_slot_mine * (double)(1ull << r) * 1.05) + 65536;
	G.cap2 = (uint64_t)(per_slot_mine * (double)n_ranks * (double)kWRegionSlots * 1.25) + 512;
	// bucket strides of an ODD number of 128-byte lines: the append points of neighbouring buckets then fall on neighbouring lines
	// (modulo any power of two), whatever the memory system's channel interleave
	G.cap1 = ((G.cap1 + 7u) & ~7ull) | 8u;
	G.cap2 = ((G.cap2 + 7u) & ~7ull) | 8u;
	G.chunk_buckets = (G.Bp + 7u) / 8u;
	h->wmulti = want_shard || n_passes > 1;
	h->sharded = n_ranks > 1;
	h->shard_rank = G.rank;
	h->tslots = G.slot_hi - G.slot_lo;
	return true;
}

static int setup_wide_partition(dbgk_handle *h)
{
	const WPartGeom &G = h->wgeom;
	WPartStore &P = h->wstore;
	memset(&P, 0, sizeof P);
	P.ovf_cap = h->cfg.expected_kmers / 16 + (1ull << 20);
	P.spill_cap = (uint64_t)((h->tslots + kWRegionSlots - 1) >> kWRegionBits) * 8 + (1ull << 16);
	P.outgoing_cap = 1ull << 16;
	const size_t l1_bytes = (size_t)G.n_l1 * G.cap1 * 16, l2_bytes = (size_t)G.chunk_buckets * G.n2 * G.cap2 * 16;
	bool ok = hipMalloc(&P.l1, l1_bytes) == hipSuccess && hipMalloc(&P.l2, l2_bytes) == hipSuccess &&
	          hipMalloc(&P.cnt1, (size_t)G.n_l1 * 4) == hipSuccess && hipMalloc(&P.cnt2, (size_t)G.chunk_buckets * G.n2 * 4) == hipSuccess &&
	          hipMalloc(&P.ovf, P.ovf_cap * sizeof(dbgk_node32)) == hipSuccess && hipMalloc(&P.spill, P.spill_cap * sizeof(dbgk_node32)) == hipSuccess &&
	          hipMalloc(&P.ovf_n, 16) == hipSuccess && hipMalloc(&h->w_tile_prefix, ((size_t)G.n_l1 + 1) * 4) == hipSuccess &&
	          hipMalloc(&h->w_cursor, (3 + (size_t)G.chunk_buckets * G.n2) * 4) == hipSuccess && // fast cursor, redo cursor, redo count, redo list
	          hipMalloc(&P.outgoing, P.outgoing_cap * sizeof(dbgk_node32)) == hipSuccess &&
	          hipMalloc(&P.outgoing_n, 8) == hipSuccess && hipMalloc(&h->w_side_out, ((size_t)kWideSideSlots + 1) * sizeof(dbgk_node32)) == hipSuccess &&
	          hipMalloc(&h->w_side_n, 8) == hipSuccess;
	if (ok && h->sharded) ok = hipMalloc(&h->winbox, l1_bytes) == hipSuccess && hipMalloc(&h->winbox_cnt, (size_t)G.n_l1 * 4) == hipSuccess;
	if (!ok) {
		(void)hipGetLastError();
		g_last_error = "hipMalloc of the wide record stores failed";
		return DBGK_ERR_NOMEM;
	}
	P.inbox = h->sharded ? h->winbox : P.l1;
	P.inbox_cnt = h->sharded ? h->winbox_cnt : P.cnt1;
	HIPCHK(hipMemsetAsync(P.cnt1, 0, (size_t)G.n_l1 * 4, h->stream));
	HIPCHK(hipMemsetAsync(P.ovf_n, 0, 16, h->stream));
	HIPCHK(hipMemsetAsync(P.outgoing_n, 0, 8, h->stream));
	DBGK_LDS_ATTR((k_wide_scatter_l1<false, 0>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1<true, 0>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1<false, 1>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1<true, 1>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1<false, 2>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1<true, 2>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1_uniform<0, false>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1_uniform<1, false>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1_uniform<2, false>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1_uniform<0, true>), sizeof(WL1PipeLds));
	DBGK_LDS_ATTR((k_wide_scatter_l1_uniform<1, true>), sizeof(WL1PipeLds));
	DBGK_LDS_ATTR((k_wide_scatter_l1_uniform<2, true>), sizeof(WL1PipeLds));
	DBGK_LDS_ATTR(k_wide_scatter_l2<1024>, sizeof(WL2Lds<1024>));
	DBGK_LDS_ATTR(k_wide_scatter_l2<2048>, sizeof(WL2Lds<2048>));
	DBGK_LDS_ATTR((k_wide_build_regions<true, false>), sizeof(WBuildLds));
	DBGK_LDS_ATTR((k_wide_build_regions<false, false>), sizeof(WBuildLds));
	DBGK_LDS_ATTR((k_wide_build_regions<false, true>), sizeof(WBuildLds));
	h->store_capacity = h->cfg.expected_kmers;
	h->pending_kmers = 0;
	h->wpass_open = true; // pass 0 is open from the start
	return DBGK_OK;
}

// the main table of a WIDE handle that was reset without a memset (the region build rewrites every slot)
static int wide_ensure_zero(dbgk_handle *h)
{
	if (!h->wzero_pending) return DBGK_OK;
	HIPCHK(hipMemsetAsync(h->wnodes, 0, h->tslots * sizeof(WNode), h->stream));
	h->wzero_pending = false;
	return DBGK_OK;
}

// own-bucket indices of the current pass that exist on this rank (the last rank / pass may have fewer)
static uint32_t wide_pass_buckets(const dbgk_handle *h)
{
	const WPartGeom &G = h->wgeom;
	return G.pass_j0 >= G.nb_own ? 0u : std::min(G.Bp, G.nb_own - G.pass_j0);
}

static int wide_plan_pass(dbgk_handle *h)
{
	if (h->wplanned) return DBGK_OK;
	hipLaunchKernelGGL(k_wide_l2_plan, dim3(1), dim3(1024), 0, h->stream, h->wgeom, h->wstore.inbox_cnt, h->w_tile_prefix);
	HIPCHK(hipGetLastError());
	h->wplanned = true;
	h->wnext = 0;
	return DBGK_OK;
}

// level 2 + region build of the own-bucket indices [j0, j1) of the current pass, chunk by chunk of level-1 buckets
// through the (small) level-2 store
static int wide_build_range(dbgk_handle *h, uint32_t j0, uint32_t j1)
{
	const WPartGeom &G = h->wgeom;
	const WPartStore &P = h->wstore;
	TimedSpan sp;
	int rc = DBGK_OK;
	for (uint32_t c0 = j0; c0 < j1; c0 += G.chunk_buckets) {
		const uint32_t c1 = std::min(c0 + G.chunk_buckets, j1);
		rc = span_begin(h, PH_PARTITION, sp);
		if (rc) return rc;
		HIPCHK(hipMemsetAsync(P.cnt2, 0, (size_t)G.chunk_buckets * G.n2 * 4, h->stream));
		if (G.n2 > 1024u)
			hipLaunchKernelGGL(k_wide_scatter_l2<2048>, dim3(h->n_cu & ~7), dim3(kWL2Threads), sizeof(WL2Lds<2048>), h->stream, G, P, h->w_tile_prefix, h->d_ctr, c0, c1);
		else
			hipLaunchKernelGGL(k_wide_scatter_l2<1024>, dim3((h->n_cu * 2) & ~7), dim3(kWL2Threads), sizeof(WL2Lds<1024>), h->stream, G, P, h->w_tile_prefix, h->d_ctr, c0, c1);
		HIPCHK(hipGetLastError());
		rc = span_end(h, sp);
		if (rc) return rc;
		const uint32_t n_regions = (c1 - c0) * G.n2;
		rc = span_begin(h, PH_BUILD, sp);
		if (rc) return rc;
		HIPCHK(hipMemsetAsync(h->w_cursor, 0, 12, h->stream)); // both cursors and the redo count
		const WRedoList redo{h->w_cursor + 3, h->w_cursor + 2, G.chunk_buckets * G.n2};
		const uint32_t bgrid = std::min<uint32_t>(n_regions, (uint32_t)h->n_cu * 3u);
		if (dbgk_hook("build_exact") && atoi(dbgk_hook("build_exact")) != 0) {
			hipLaunchKernelGGL((k_wide_build_regions<false, false>), dim3(bgrid), dim3(kWBuildThreads), sizeof(WBuildLds), h->stream, G, P, h->wnodes, h->d_ctr,
			                   h->w_cursor, c0, n_regions, redo);
		} else {
			// the fast insert, then the exact pass over the regions it flagged (normally none: the kernel finds an empty list and returns);
			// right here, while the level-2 store still holds this chunk's records
			hipLaunchKernelGGL((k_wide_build_regions<true, false>), dim3(bgrid), dim3(kWBuildThreads), sizeof(WBuildLds), h->stream, G, P, h->wnodes, h->d_ctr,
			                   h->w_cursor, c0, n_regions, redo);
			hipLaunchKernelGGL((k_wide_build_regions<false, true>), dim3(std::min<uint32_t>(bgrid, (uint32_t)h->n_cu)), dim3(kWBuildThreads), sizeof(WBuildLds), h->stream, G, P,
			                   h->wnodes, h->d_ctr, h->w_cursor + 1, c0, n_regions, redo);
		}
		HIPCHK(hipGetLastError());
		rc = span_end(h, sp);
		if (rc) return rc;
	}
	h->wnext = j1;
	return DBGK_OK;
}

// the rest of the current pass: whatever the caller has not built by ranges
static int wide_end_pass(dbgk_handle *h)
{
	if (!h->wpass_open) return DBGK_OK;
	if (h->sharded && !h->exchanged) {
		g_last_error = "sharded WIDE handle: exchange the level-1 buckets of this pass (dbgk_shard_buffers) and call dbgk_shard_mark_exchanged first";
		return DBGK_ERR_STATE;
	}
	int rc = wide_plan_pass(h);
	if (rc) return rc;
	const uint32_t nb = wide_pass_buckets(h);
	if (h->wnext < nb) {
		rc = wide_build_range(h, h->wnext, nb);
		if (rc) return rc;
	}
	if (h->wgeom.pass > 0) { // the input was read again: its totals were counted in pass 0
		HIPCHK(hipMemcpyAsync(&h->d_ctr->total_kmers, h->wsaved_totals, 16, hipMemcpyHostToDevice, h->stream));
		HIPCHK(hipMemcpyAsync(&h->d_ctr->other_bytes, &h->wsaved_other, 8, hipMemcpyHostToDevice, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		h->total_reads = h->wsaved_reads;
		h->host_other_bytes = h->wsaved_host_other;
	}
	h->wpass_open = false;
	h->wpasses_done = h->wgeom.pass + 1;
	h->pending_kmers = 0;
	return DBGK_OK;
}

static int wide_begin_pass(dbgk_handle *h, uint32_t p)
{
	WPartGeom &G = h->wgeom;
	if (h->wpass_open || p != h->wpasses_done || p >= G.n_passes) {
		g_last_error = "dbgk_wide_begin_pass: passes run in order, each ended (dbgk_wide_end_pass) before the next begins";
		return DBGK_ERR_STATE;
	}
	G.pass = p;
	G.pass_j0 = p * G.Bp;
	HIPCHK(hipMemsetAsync(h->wstore.cnt1, 0, (size_t)G.n_l1 * 4, h->stream));
	if (p > 0) {
		HIPCHK(hipMemcpyAsync(h->wsaved_totals, &h->d_ctr->total_kmers, 16, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(&h->wsaved_other, &h->d_ctr->other_bytes, 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		h->wsaved_reads = h->total_reads;
		h->wsaved_host_other = h->host_other_bytes;
	}
	h->wpass_open = true;
	h->wplanned = false;
	h->exchanged = false;
	h->wnext = 0;
	h->pending_kmers = 0;
	return DBGK_OK;
}

// after the last pass: region spill-over nodes and bucket-overflow observations through the atomic kernels
static int wide_finish_records(dbgk_handle *h)
{
	const WPartGeom &G = h->wgeom;
	const WPartStore &P = h->wstore;
	TimedSpan sp;
	h->wzero_pending = false; // every slot has just been written
	int rc = span_begin(h, PH_FIXUP, sp);
	if (rc) return rc;
	if (h->wmulti && G.n_ranks > 1) {
		// spill nodes stay in the shard unless they run off its end (-> outgoing); overflow observations may belong to any
		// shard: the caller exchanges them (dbgk_shard_overflow)
		hipLaunchKernelGGL(k_wide_merge_sharded, dim3(h->n_cu), dim3(kBlock), 0, h->stream, P.spill, &P.ovf_n[1], (uint64_t)0, P.spill_cap, 0, 0, G, P, h->wnodes,
		                   h->d_ctr);
	} else {
		hipLaunchKernelGGL(k_wide_merge_spill, dim3(h->n_cu), dim3(kBlock), 0, h->stream, P.spill, &P.ovf_n[1], P.spill_cap, h->wref(), h->d_ctr);
		hipLaunchKernelGGL(k_wide_insert_obs, dim3(h->n_cu), dim3(kBlock), 0, h->stream, P.ovf, &P.ovf_n[0], P.ovf_cap, h->wref(), h->d_ctr);
	}
	HIPCHK(hipGetLastError());
	rc = span_end(h, sp);
	if (rc) return rc;
	h->wbuilt = true;
	h->pending_kmers = 0;
	return DBGK_OK;
}

// records -> table (one rank, one pass).  Afterwards the handle is an ordinary WIDE handle: whatever is pushed later goes
// through the atomic kernels onto the table built here.
static int wide_build_from_records(dbgk_handle *h)
{
	if (!h->wpart || h->wbuilt) return DBGK_OK;
	if (h->wmulti) {
		g_last_error = "the record store of a sharded / multi-pass WIDE handle is full: expected_kmers too small (or more passes needed)";
		return DBGK_ERR_CAPACITY;
	}
	int rc = wide_end_pass(h);
	if (rc) return rc;
	return wide_finish_records(h);
}
