// dbgk_host_level1.h -- part of libdbgk.so's host side (one translation unit: included by dbgk.hip, in this order).
// one batch on the device: which level-1 / insert kernel form a batch takes (launch_batch) and its pre-passes
#pragma once

// queue mark + insert for a batch that is already in device memory
// The lane-per-chunk-of-valid-windows level-1 kernel (k_extract_scatter_uniform) applies when nothing is
// trimmed (longest read <= maxReadLen), the longest read has at least 64 windows (a tile's byte range
// must fit its LDS image) and either
//   * every read has that length and they lie back to back from offset 0 (EQUAL), or
//   * lengths differ, but giving every read the lane count of the longest one still needs fewer lane
//     slots than the flat kernel has positions (RAGGED: reads mostly full length, some shorter).
// Returns 0 = flat kernel, 1 = equal, 2 = ragged; fills U.
// lin (out): the linear level-1 form (8 windows per lane, linear copy-out) is to be used -- many level-1 buckets
static int uniform_mode(const dbgk_handle *h, int64_t uniform_len, uint64_t len_max, uint64_t n_reads, uint64_t n_bases, int has_long,
                        UniformGeom &U, bool &c15, bool &lin, bool &lin12)
{
	lin = lin12 = false;
	static const bool off = DBGK_EXPERIMENT_ENV("DBGK_L1_FLAT") != nullptr; // force the general kernel
	static const int dbg_mode = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE")) : 0;
	if (off || has_long || n_reads == 0) return 0;
	if (dbg_mode && (uniform_len != 150 || h->cfg.kmer_size != 31 || h->geom.size >= (1ull << 31))) return 0; // debug builds: cfg2's shape only
	const uint64_t L = uniform_len > 0 ? (uint64_t)uniform_len : len_max, k = (uint64_t)h->cfg.kmer_size;
	if (L > (uint64_t)h->cfg.max_read_len || L < k + 63 || L >= (1ull << 24)) return 0;
	const uint32_t W = (uint32_t)(L - k + 1);
	// 16 or 15 windows per lane, whichever leaves fewer empty slots at the end of a full-length read (W = 120: 15 -> none)
	const uint32_t q16 = (W + 15u) / 16u, q15 = (W + 14u) / 15u;
	c15 = q15 * 15u - W < q16 * 16u - W;
	const uint64_t Q = c15 ? q15 : q16, C = c15 ? 15 : 16;
	if (Q >= 2048 || n_reads * Q >= (1ull << 32)) return 0;
	if (((uint64_t)kL1Threads / Q + 2) * L + 96 > (uint64_t)kPkWords * 16) return 0; // bytes a tile of kL1Threads lanes can touch
	U.L = (uint32_t)L;
	U.W = W;
	U.Q = (uint32_t)Q;
	U.qmagic = ((1u << 22) + U.Q - 1u) / U.Q;
	U.n_lanes = n_reads * Q;
	U.lq = 0;
	U.tile_blocks = 0; // 0: no regular tiles
	// (the regular form also assumes k >= 17 -- the rolls then only touch the high words -- and a graph handle: no KFREQ neighbour
	// codes; reads that fill their lanes exactly, Q C == W, take its FULL instantiation, the others test every position's validity)
	if ((Q & (Q - 1)) == 0 && Q <= (uint64_t)kL1Threads && (((uint64_t)kL1Threads / Q) * L) % 16 == 0 &&
	    ((uint64_t)kL1Threads / Q) * L / 16 + 8 <= (uint64_t)kPkWords && k >= 17 && !h->kfreq) {
		while ((1ull << U.lq) < Q) U.lq++;
		U.tile_blocks = (uint32_t)(((uint64_t)kL1Threads / Q) * L / 16);
	}
	int mode;
	if (uniform_len > 0) mode = n_bases == n_reads * L ? 1 : 0;
	else mode = (double)(n_reads * Q * C) <= 0.93 * (double)n_bases ? 2 : 0; // mostly full-length reads: the ragged form
	if (mode == 0) return 0;
	// Many level-1 buckets (big tables; every rank of a multi-GPU job partitions by the GLOBAL table's buckets): the linear
	// form, 12 (or 8) windows per lane.  Measured on cfg2's reads, level 1 in ms, wave-per-bucket / linear with 8 / with 12
	// windows: n1 = 143: 5.57 / 6.76 / 6.05, 573: 8.13 / 7.57 / 6.81, 1023: 10.61 / 7.76 / 6.99 (287: 6.34 / 6.77 / -).
	// DBGK_L1_LINEAR=0/1 forces the choice.
	const int force = dbgk_hook("l1_linear") ? atoi(dbgk_hook("l1_linear")) : -1; // (read per batch: tests switch it)
	// 12 or 8 windows per lane, whichever leaves fewer empty slots at the end of a read (W = 120: both none -> 12)
	const uint64_t q12 = (W + 11u) / 12u, q8 = (W + 7u) / 8u;
	lin12 = q12 * 12u - W <= q8 * 8u - W;
	if (const char *e = DBGK_EXPERIMENT_ENV("DBGK_L1_LINEAR_C")) lin12 = atoi(e) == 12; // measurements
	const uint64_t QL = lin12 ? q12 : q8, CL = lin12 ? 12 : 8;
	bool fits = QL < 2048 && n_reads * QL < (1ull << 32) && ((uint64_t)kL1Threads / QL + 2) * L + 96 <= (uint64_t)kPkWords * 16;
	if (mode == 2) fits = fits && (double)(n_reads * QL * CL) <= 0.93 * (double)n_bases;
	// (round 5, both forms pipelined, profiles/r05_l1_forms_by_buckets.txt, wave-per-bucket / linear with 12 windows: n1 = 144: 4.34 / 5.08,
	// 287: 4.80 / 5.26, 573: 6.83 / 5.87, 1023: 12.27 / 6.11 -- the lines cross near 330)
	if (fits && (force == 1 || (force < 0 && h->geom.n1 > 320u))) {
		lin = true;
		U.tile_blocks = 0;
		U.Q = (uint32_t)QL;
		U.qmagic = ((1u << 22) + U.Q - 1u) / U.Q;
		U.n_lanes = n_reads * QL;
	}
	return mode;
}

// WIDE record path: can this batch take the equal-length level-1 kernel (k_wide_scatter_l1_uniform)?
static int wide_uniform_mode(const dbgk_handle *h, int64_t uniform_len, uint64_t n_reads, uint64_t n_bases, int has_long, WUniformGeom &U)
{
	static const bool off = DBGK_EXPERIMENT_ENV("DBGK_L1_FLAT") != nullptr; // force the general kernel
	if (off || has_long || n_reads == 0 || uniform_len <= 0) return 0;
	const uint64_t L = (uint64_t)uniform_len, k = (uint64_t)h->cfg.kmer_size;
	if (L > (uint64_t)h->cfg.max_read_len || L < k || L >= (1ull << 24) || n_bases != n_reads * L) return 0;
	const uint32_t W = (uint32_t)(L - k + 1);
	const uint64_t Q = (W + 7u) / 8u;
	if (Q >= 2048 || n_reads * Q >= (1ull << 40)) return 0;
	if (((uint64_t)kWL1Threads / Q + 2) * L + 96 > (uint64_t)(kWPkWords - 8u) * 16) return 0; // bytes a tile of 1024 lanes can touch
	if ((double)(n_reads * Q * 8) > 0.93 * (double)n_bases) return 0; // (k small against L: the flat kernel wastes little)
	U.L = (uint32_t)L;
	U.W = W;
	U.Q = (uint32_t)Q;
	U.qmagic = ((1u << 22) + U.Q - 1u) / U.Q;
	U.n_lanes = n_reads * Q;
	return 1;
}

// scratch of the prefix form for a batch of n_reads reads / n_bases bases (grown as needed; the stream is idle when it grows)
static int ensure_prefix_scratch(dbgk_handle *h, uint64_t n_reads, uint64_t n_bases, bool need_packed)
{
	const uint64_t tiles = (n_bases / 15 + n_reads) / kL1Threads + 2; // a read of W windows has at most W / 15 + 1 lanes
	if (n_reads > h->pf_cap_reads || tiles > h->pf_cap_tiles) {
		HIPCHK(hipStreamSynchronize(h->stream));
		for (void *q : {(void *)h->pf_ent, (void *)h->pf_tile_first, (void *)h->pf_tiles, (void *)h->pf_bsum})
			if (q) (void)hipFree(q);
		h->pf_ent = nullptr; h->pf_tile_first = nullptr; h->pf_tiles = nullptr; h->pf_bsum = nullptr;
		h->pf_cap_reads = h->pf_cap_tiles = 0;
		const uint64_t cr = std::max(n_reads, h->cap_reads), ct = std::max(tiles, (h->cap_bases / 15 + h->cap_reads) / kL1Threads + 2);
		const uint64_t blocks = (cr + kPrefixBlock * kPrefixItems - 1) / (kPrefixBlock * kPrefixItems) + 1;
		if (hipMalloc(&h->pf_ent, cr * sizeof(ReadLanes)) != hipSuccess || hipMalloc(&h->pf_tile_first, (ct + 1) * 4) != hipSuccess ||
		    hipMalloc(&h->pf_tiles, ct * sizeof(PrefixTile)) != hipSuccess || hipMalloc(&h->pf_bsum, blocks * 8) != hipSuccess)
			return DBGK_ERR_NOMEM;
		h->pf_cap_reads = cr;
		h->pf_cap_tiles = ct;
	}
	if (!h->pf_tot && hipMalloc(&h->pf_tot, sizeof(PrefixTotals)) != hipSuccess) return DBGK_ERR_NOMEM;
	const uint64_t words = (n_bases + 15) / 16 + 16;
	if (need_packed && words > h->pf_packed_words) {
		HIPCHK(hipStreamSynchronize(h->stream));
		if (h->pf_packed) (void)hipFree(h->pf_packed);
		h->pf_packed = nullptr;
		h->pf_packed_words = 0;
		const uint64_t cw = std::max(words, h->cap_bases / 16 + 16);
		if (hipMalloc(&h->pf_packed, cw * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		h->pf_packed_words = cw;
	}
	return DBGK_OK;
}

static int early_l2(dbgk_handle *h);

static int launch_batch(dbgk_handle *h, const char *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                        uint64_t n_bases, uint32_t *d_start, uint32_t *d_dead, int has_long /* 0,1 or -1 = ask device */,
                        int64_t uniform_len = -1 /* every read this long; 0 = lengths differ; -1 = ask device */,
                        uint64_t len_max = 0 /* longest read of the batch (with uniform_len >= 0) */,
                        const uint32_t *d_packed = nullptr /* the batch as 2-bit codes instead of d_bases (then null) */)
{
	if (n_reads == 0) return DBGK_OK;
	if (h->seed && d_packed) return DBGK_ERR_ARG; // the seed index cuts its windows at 'N', which two bits cannot say
	if (h->seed) has_long = 1; // the dead bitmap carries the 'N' positions
	const uint64_t words = bitmap_words(n_bases);
	TimedSpan sp;
	int rc = early_l2(h); // level 2 of what the batches before this one stored: queued IN FRONT of this batch's level 1, i.e. it runs while
	if (rc) return rc;    // this batch is still on the link (the level-1 launch below waits for the copy, the level-2 round does not)
	rc = span_begin(h, PH_MARK, sp);
	if (rc) return rc;
	static_assert(offsetof(Counters, len_max) + sizeof(unsigned long long) - offsetof(Counters, any_dead) == 20, "per-batch fields are contiguous");
	if (d_offsets) HIPCHK(hipMemsetAsync(&h->d_ctr->any_dead, 0, 20, h->stream)); // any_dead, len_min_inv, len_max: what k_mark reports per batch
	// The read-boundary bitmaps are what the general kernels navigate by; the PARTITION engine's level-1 kernel for
	// (nearly) equal-length reads does without them, so for such a batch only the statistics are taken.  A device
	// batch tells its shape only after those statistics: the bitmaps follow in a second pass if they are needed.
	UniformGeom U{};
	bool c15 = false, lin8 = false, lin12 = false;
	int umode = -1; // not decided yet
	auto mark_bits = [&](int with_stats) -> int {
		HIPCHK(hipMemsetAsync(d_start, 0, words * 4, h->stream));
		if (has_long != 0) HIPCHK(hipMemsetAsync(d_dead, 0, words * 4, h->stream));
		hipLaunchKernelGGL(k_mark, dim3(grid_for(h, n_reads)), dim3(kBlock), 0, h->stream, d_offsets, n_reads, n_bases,
		                   h->cfg.kmer_size, h->cfg.max_read_len, d_start, has_long != 0 ? d_dead : nullptr, h->d_ctr, with_stats);
		return DBGK_OK;
	};
	const bool wrec = h->wide && h->wpart && !h->wbuilt; // WIDE handle that is still collecting records
	WUniformGeom WU{};
	auto decide_umode = [&]() {
		return wrec ? wide_uniform_mode(h, uniform_len, n_reads, n_bases, has_long, WU) : uniform_mode(h, uniform_len, len_max, n_reads, n_bases, has_long, U, c15, lin8, lin12);
	};
	// A batch of equal-length reads that came WITHOUT offsets (dbgk_push_reads_packed_uniform*): the equal-length level-1 forms of the
	// PARTITION / WIDE record engines never look at offsets -- the totals are added by a one-thread kernel and nothing else runs in
	// front of level 1; any other consumer gets the offsets made on the device first.
	const bool no_offsets = d_offsets == nullptr;
	auto make_offsets = [&]() -> int {
		if (n_reads + 1 > h->uni_cap) {
			HIPCHK(hipStreamSynchronize(h->stream));
			if (h->uni_offsets) (void)hipFree(h->uni_offsets);
			h->uni_offsets = nullptr;
			h->uni_cap = 0;
			const uint64_t cap = std::max(n_reads + 1, h->cap_reads + 1);
			if (hipMalloc(&h->uni_offsets, cap * 8) != hipSuccess) return DBGK_ERR_NOMEM;
			h->uni_cap = cap;
		}
		hipLaunchKernelGGL(k_iota_offsets, dim3(grid_for(h, n_reads + 1)), dim3(kBlock), 0, h->stream, h->uni_offsets, n_reads, (uint64_t)uniform_len);
		HIPCHK(hipMemsetAsync(&h->d_ctr->any_dead, 0, 20, h->stream));
		d_offsets = h->uni_offsets;
		return DBGK_OK;
	};
	const bool may_skip_bits = (h->part && !h->seed) || wrec;
	// The PREFIX form of level 1 (k_extract_scatter_prefix: every read exactly the lanes its windows need, reads of any lengths,
	// trimmed ones included) takes what would otherwise go through the flat kernel -- a fifth of whose positions straddle a
	// read boundary at 150 bases and k = 31 -- and the batches of the ragged form as well; not with many level-1 buckets (the
	// linear forms), not for reads of more than 4 M windows.  DBGK_L1_PREFIX=0 switches it off (ragged / flat as before).
	const int prefix_env = dbgk_hook("l1_prefix") ? atoi(dbgk_hook("l1_prefix")) : -1; // (read per batch: tests switch it)
	bool use_prefix = false;
	auto prefix_wanted = [&](int um) {
		static const bool flat_only = DBGK_EXPERIMENT_ENV("DBGK_L1_FLAT") != nullptr;
		static const bool dbg = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE") != nullptr;
		const int force_lin = dbgk_hook("l1_linear") ? atoi(dbgk_hook("l1_linear")) : -1;
		if (!h->part || h->seed || wrec || flat_only || dbg || prefix_env == 0) return false;
		if (force_lin == 1 || (force_lin < 0 && h->geom.n1 > 320u)) return false;
		if (len_max > (uint64_t)kPrefixMaxW || n_bases / 15 + n_reads >= (1ull << 32)) return false;
		return um == 0 || um == 2; // (measured on cfg2t, level 1 per step: prefix 5.32 ms, ragged 5.72, flat 6.05 + 0.14 of bitmaps: profiles/r04_cfg2t_level1_forms_ab.json)
	};
	if (may_skip_bits && has_long >= 0 && uniform_len >= 0) {
		umode = decide_umode();
		use_prefix = prefix_wanted(umode);
	}
	if (no_offsets && umode == 1) { // (umode 1 = equal lengths, nothing trimmed: every read has uniform_len - k + 1 windows)
		const unsigned long long w = uniform_len >= h->cfg.kmer_size ? (unsigned long long)(uniform_len - h->cfg.kmer_size + 1) * n_reads : 0ull;
		hipLaunchKernelGGL(k_add_totals, dim3(1), dim3(64), 0, h->stream, h->d_ctr, w, w);
	} else if (no_offsets) {
		rc = make_offsets();
		if (rc) return rc;
	}
	if (no_offsets && umode == 1) {
	} else if (may_skip_bits && (umode > 0 || umode < 0 || use_prefix)) {
		hipLaunchKernelGGL(k_mark, dim3(grid_for(h, n_reads)), dim3(kBlock), 0, h->stream, d_offsets, n_reads, n_bases, h->cfg.kmer_size,
		                   h->cfg.max_read_len, (uint32_t *)nullptr, (uint32_t *)nullptr, h->d_ctr, 1); // statistics only
	} else {
		rc = mark_bits(1);
		if (rc) return rc;
	}
	if (h->seed && n_bases)
		hipLaunchKernelGGL(k_mark_n, dim3(grid_for(h, (n_bases + 31) >> 5)), dim3(kBlock), 0, h->stream, d_bases, n_bases, d_dead);
	HIPCHK(hipGetLastError());
	if (has_long < 0 || (uniform_len < 0 && (h->part || wrec))) {
		HIPCHK(hipMemcpyAsync(&h->h_ctr->any_dead, &h->d_ctr->any_dead, 20, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		if (has_long < 0) has_long = h->h_ctr->any_dead ? 1 : 0;
		if (uniform_len < 0) {
			uniform_len = ~h->h_ctr->len_min_inv == h->h_ctr->len_max ? (int64_t)h->h_ctr->len_max : 0;
			len_max = h->h_ctr->len_max;
		}
	}
	if (may_skip_bits && umode < 0) {
		umode = decide_umode();
		use_prefix = prefix_wanted(umode);
		if (umode == 0 && !use_prefix) { // the general kernel after all: it needs the bitmaps
			rc = mark_bits(0);
			if (rc) return rc;
			HIPCHK(hipGetLastError());
		}
	}
	rc = span_end(h, sp);
	if (rc) return rc;
	const uint64_t id_base = h->total_reads; // contig index of the batch's first sequence (SEEDIDX)
	h->total_reads += n_reads;
	if (n_bases == 0) return DBGK_OK;

	ReadBatch rb{d_bases, n_bases, d_start, has_long ? d_dead : nullptr, h->cfg.kmer_size, d_packed, &h->d_ctr->other_seen};
	const uint64_t n_chunks = (n_bases + 15) >> 4;
	rc = span_begin(h, PH_INSERT, sp);
	if (rc) return rc;
	if (h->seed) {
		hipLaunchKernelGGL(k_seed_insert, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, d_offsets, n_reads, id_base, h->tref(), h->d_ctr);
	} else if (wrec && umode > 0) {
		h->uniform_launches++;
		const int grid = (int)std::min<uint64_t>((WU.n_lanes + kWL1Threads - 1) / kWL1Threads, (uint64_t)h->n_cu);
		const int wd = h->size >= (1ull << 32) ? 2 : (h->size >= (1ull << 31) ? 1 : 0); // how hash / size is computed
		// the pipelined form keeps a tile's packed words beside the stage buffer: taken when the bytes a tile of 1024 lanes can touch fit there
		const bool wpipe = !dbgk_hook("wide_l1_plain") &&
		                   ((uint64_t)kWL1Threads / WU.Q + 2) * WU.L + 96 <= (uint64_t)(kWPipePkWords - 8u) * 16;
#define DBGK_LAUNCH_WIDE_L1U(WD)                                                                                                                    \
	do {                                                                                                                                            \
		if (wpipe)                                                                                                                                  \
			hipLaunchKernelGGL((k_wide_scatter_l1_uniform<WD, true>), dim3(grid), dim3(kWL1Threads), sizeof(WL1PipeLds), h->stream, rb, WU, h->wgeom, \
			                   h->wstore, h->wref(), h->d_ctr);                                                                                     \
		else                                                                                                                                        \
			hipLaunchKernelGGL((k_wide_scatter_l1_uniform<WD, false>), dim3(grid), dim3(kWL1Threads), sizeof(WL1Lds), h->stream, rb, WU, h->wgeom,    \
			                   h->wstore, h->wref(), h->d_ctr);                                                                                     \
	} while (0)
		if (wd == 2) DBGK_LAUNCH_WIDE_L1U(2); else if (wd == 1) DBGK_LAUNCH_WIDE_L1U(1); else DBGK_LAUNCH_WIDE_L1U(0);
#undef DBGK_LAUNCH_WIDE_L1U
	} else if (wrec) {
		const int grid = (int)std::min<uint64_t>((n_chunks + kWL1Threads - 1) / kWL1Threads, (uint64_t)h->n_cu);
		const int wd = h->size >= (1ull << 32) ? 2 : (h->size >= (1ull << 31) ? 1 : 0); // how hash / size is computed
#define DBGK_LAUNCH_WIDE_L1(DEAD, WD)                                                                                                          \
	hipLaunchKernelGGL((k_wide_scatter_l1<DEAD, WD>), dim3(grid), dim3(kWL1Threads), sizeof(WL1Lds), h->stream, rb, h->wgeom, h->wstore, h->wref(), \
	                   h->d_ctr)
		if (has_long) {
			if (wd == 2) DBGK_LAUNCH_WIDE_L1(true, 2); else if (wd == 1) DBGK_LAUNCH_WIDE_L1(true, 1); else DBGK_LAUNCH_WIDE_L1(true, 0);
		} else {
			if (wd == 2) DBGK_LAUNCH_WIDE_L1(false, 2); else if (wd == 1) DBGK_LAUNCH_WIDE_L1(false, 1); else DBGK_LAUNCH_WIDE_L1(false, 0);
		}
#undef DBGK_LAUNCH_WIDE_L1
	} else if (h->wide) {
		rc = wide_ensure_zero(h);
		if (rc) return rc;
		if (has_long)
			hipLaunchKernelGGL(k_wide_extract_insert<true>, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->wref(), h->d_ctr);
		else
			hipLaunchKernelGGL(k_wide_extract_insert<false>, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->wref(), h->d_ctr);
	} else if (h->kfreq && !h->part) {
		if (has_long)
			hipLaunchKernelGGL(k_extract_count<true>, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, reinterpret_cast<uint32_t *>(h->counts));
		else
			hipLaunchKernelGGL(k_extract_count<false>, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, reinterpret_cast<uint32_t *>(h->counts));
	} else if (h->part && use_prefix) {
		// lane prefix of the batch (three short kernels over the offsets), the batch itself as 2-bit words, then level 1
		h->prefix_launches++;
		rc = ensure_prefix_scratch(h, n_reads, n_bases, d_packed == nullptr);
		if (rc) return rc;
		if (!d_packed) {
			hipLaunchKernelGGL(k_pack_bases, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, d_bases, n_bases, h->pf_packed, h->d_ctr);
			rb.packed = h->pf_packed;
		}
		const uint32_t W_max = (uint32_t)std::min<uint64_t>(len_max, (uint64_t)h->cfg.max_read_len) >= (uint32_t)h->cfg.kmer_size
		                           ? (uint32_t)std::min<uint64_t>(len_max, (uint64_t)h->cfg.max_read_len) - (uint32_t)h->cfg.kmer_size + 1u : 1u;
		const bool pc15 = (W_max + 14u) / 15u * 15u - W_max < (W_max + 15u) / 16u * 16u - W_max; // 15 or 16 windows per lane: fewer empty slots for a full-length read
		const uint32_t n_blocks = (uint32_t)((n_reads + kPrefixBlock * kPrefixItems - 1) / (kPrefixBlock * kPrefixItems));
		const uint32_t tiles_max = (uint32_t)std::min<uint64_t>((n_bases / 15 + n_reads) / kL1Threads + 1, h->pf_cap_tiles);
		const int wide = (h->geom.kf == 2u || h->geom.size >= (1ull << 32)) ? 2 : (h->geom.size >= (1ull << 31) ? 1 : 0);
		const int grid_p = (int)((uint64_t)h->n_cu * l1_wgs_per_cu());
#define DBGK_LAUNCH_PREFIX(WIDE, CC)                                                                                                              \
	do {                                                                                                                                          \
		hipLaunchKernelGGL((k_prefix_count<CC>), dim3(n_blocks), dim3(kPrefixBlock), 0, h->stream, d_offsets, n_reads, (uint32_t)h->cfg.kmer_size,   \
		                   (uint32_t)h->cfg.max_read_len, h->pf_bsum);                                                                            \
		hipLaunchKernelGGL(k_prefix_blocks, dim3(1), dim3(kPrefixBlock), 0, h->stream, h->pf_bsum, n_blocks, h->pf_tot);                          \
		hipLaunchKernelGGL((k_prefix_emit<CC>), dim3(n_blocks), dim3(kPrefixBlock), 0, h->stream, d_offsets, n_reads, (uint32_t)h->cfg.kmer_size,    \
		                   (uint32_t)h->cfg.max_read_len, h->pf_bsum, h->pf_ent, h->pf_tile_first, h->d_ctr);                                     \
		hipLaunchKernelGGL((k_prefix_tiles<CC>), dim3((tiles_max + 255) / 256), dim3(256), 0, h->stream, h->pf_ent, h->pf_tile_first, h->pf_tot,    \
		                   (uint32_t)h->cfg.kmer_size, n_bases, h->pf_tiles);                                                                     \
		if (h->cfg.kmer_size >= 17 && h->geom.kf != 2u && h->geom.n1 <= kPrefixPipeMaxB && !dbgk_hook("l1_plain")) /* the pipelined tile loop */   \
			hipLaunchKernelGGL((k_extract_scatter_prefix<WIDE, CC, true>), dim3(grid_p), dim3(kL1Threads), sizeof(PrefixLds), h->stream, rb, h->pf_ent, \
			                   h->pf_tiles, h->pf_tot, h->geom, h->store, h->d_ctr);                                                              \
		else                                                                                                                                      \
			hipLaunchKernelGGL((k_extract_scatter_prefix<WIDE, CC, false>), dim3(grid_p), dim3(kL1Threads), sizeof(PrefixLds), h->stream, rb, h->pf_ent, \
			                   h->pf_tiles, h->pf_tot, h->geom, h->store, h->d_ctr);                                                              \
	} while (0)
		if (pc15) { if (wide == 2) DBGK_LAUNCH_PREFIX(2, 15); else if (wide == 1) DBGK_LAUNCH_PREFIX(1, 15); else DBGK_LAUNCH_PREFIX(0, 15); }
		else { if (wide == 2) DBGK_LAUNCH_PREFIX(2, 16); else if (wide == 1) DBGK_LAUNCH_PREFIX(1, 16); else DBGK_LAUNCH_PREFIX(0, 16); }
#undef DBGK_LAUNCH_PREFIX
	} else if (h->part && umode > 0) {
		h->uniform_launches++;
		const int wide = (h->geom.kf == 2u || h->geom.size >= (1ull << 32)) ? 2 : (h->geom.size >= (1ull << 31) ? 1 : 0); // (direct blocks: the 64-bit slot path)
		const bool ragged = umode == 2;
		// equal lengths, ragged: the pipelined tile loop (32-bit rolls; 64-bit records -- a KFREQ handle with direct blocks has its own instantiation)
		const bool k17 = h->cfg.kmer_size >= 17 && h->geom.kf != 2u && !dbgk_hook("l1_plain");
#define DBGK_LAUNCH_UNIFORM(WIDE, CC, RAG)                                                                                                   \
	hipLaunchKernelGGL((k_extract_scatter_uniform<0, WIDE, CC, RAG>), dim3(grid), dim3(kL1Threads), sizeof(UniformLds), h->stream, rb, U, \
	                   d_offsets, h->geom, h->store, h->d_ctr)
#define DBGK_LAUNCH_UNIFORM_K17(WIDE, CC, RAG)                                                                                                    \
	hipLaunchKernelGGL((k_extract_scatter_uniform<0, WIDE, CC, RAG, false, false, false, true, true>), dim3(grid), dim3(kL1Threads), sizeof(UniformLds), \
	                   h->stream, rb, U, d_offsets, h->geom, h->store, h->d_ctr)
#define DBGK_LAUNCH_UNIFORM_W(WIDE)                                    \
	do {                                                               \
		if (k17 && c15 && ragged) DBGK_LAUNCH_UNIFORM_K17(WIDE, 15, true); \
		else if (k17 && ragged) DBGK_LAUNCH_UNIFORM_K17(WIDE, 16, true);   \
		else if (k17 && c15) DBGK_LAUNCH_UNIFORM_K17(WIDE, 15, false);     \
		else if (k17) DBGK_LAUNCH_UNIFORM_K17(WIDE, 16, false);            \
		else if (c15 && ragged) DBGK_LAUNCH_UNIFORM(WIDE, 15, true);       \
		else if (ragged) DBGK_LAUNCH_UNIFORM(WIDE, 16, true);              \
		else if (c15) DBGK_LAUNCH_UNIFORM(WIDE, 15, false);                \
		else DBGK_LAUNCH_UNIFORM(WIDE, 16, false);                     \
	} while (0)
#define DBGK_LAUNCH_UNIFORM8(WIDE, CC, RAG)                                                                                                               \
	do {                                                                                                                                                  \
		if (k17) /* the pipelined linear form */                                                                                                          \
			hipLaunchKernelGGL((k_extract_scatter_uniform<0, WIDE, CC, RAG, true, false, false, true, true>), dim3(grid), dim3(kL1Threads),                \
			                   sizeof(UniformLdsLin<CC>), h->stream, rb, U, d_offsets, h->geom, h->store, h->d_ctr);                                      \
		else                                                                                                                                              \
			hipLaunchKernelGGL((k_extract_scatter_uniform<0, WIDE, CC, RAG, true>), dim3(grid), dim3(kL1Threads), sizeof(UniformLdsLin<CC>), h->stream, rb, \
			                   U, d_offsets, h->geom, h->store, h->d_ctr);                                                                                \
	} while (0)
		static const int dbg_mode_u = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE")) : 0;
		static const bool no_reg = DBGK_EXPERIMENT_ENV("DBGK_L1_NO_REG") != nullptr; // A/B: the general form everywhere
		bool rest_only = false;
		if (!dbg_mode_u && !no_reg && !dbgk_hook("l1_plain") && umode == 1 && !lin8 && U.tile_blocks) {
			// regular tiles: kL1Threads / Q whole reads each, from a 16-byte boundary; the reads behind the last whole tile
			// (fewer than kL1Threads / Q) go through the general form below
			const uint64_t reads_per_tile = (uint64_t)kL1Threads / U.Q, full_tiles = n_reads / reads_per_tile;
			if (full_tiles) {
				UniformGeom UR = U;
				UR.n_lanes = full_tiles * kL1Threads;
				ReadBatch rr = rb;
				rr.n_bases = full_tiles * reads_per_tile * U.L;
				const int grid_r = (int)std::min<uint64_t>(full_tiles, (uint64_t)h->n_cu * l1_wgs_per_cu());
				const bool full = (uint64_t)U.Q * (c15 ? 15u : 16u) == (uint64_t)U.W; // the reads fill their lanes exactly
#define DBGK_LAUNCH_REG(WIDE, CC, PK)                                                                                                                  \
	do {                                                                                                                                               \
		if (full)                                                                                                                                      \
			hipLaunchKernelGGL((k_extract_scatter_uniform<0, WIDE, CC, false, false, true, PK, true>), dim3(grid_r), dim3(kL1Threads), sizeof(UniformLds),  \
			                   h->stream, rr, UR, d_offsets, h->geom, h->store, h->d_ctr);                                                            \
		else                                                                                                                                           \
			hipLaunchKernelGGL((k_extract_scatter_uniform<0, WIDE, CC, false, false, true, PK, false>), dim3(grid_r), dim3(kL1Threads), sizeof(UniformLds), \
			                   h->stream, rr, UR, d_offsets, h->geom, h->store, h->d_ctr);                                                            \
	} while (0)
				if (d_packed) {
					if (c15) { if (wide == 2) DBGK_LAUNCH_REG(2, 15, true); else if (wide == 1) DBGK_LAUNCH_REG(1, 15, true); else DBGK_LAUNCH_REG(0, 15, true); }
					else { if (wide == 2) DBGK_LAUNCH_REG(2, 16, true); else if (wide == 1) DBGK_LAUNCH_REG(1, 16, true); else DBGK_LAUNCH_REG(0, 16, true); }
				} else if (c15) { if (wide == 2) DBGK_LAUNCH_REG(2, 15, false); else if (wide == 1) DBGK_LAUNCH_REG(1, 15, false); else DBGK_LAUNCH_REG(0, 15, false); }
				else { if (wide == 2) DBGK_LAUNCH_REG(2, 16, false); else if (wide == 1) DBGK_LAUNCH_REG(1, 16, false); else DBGK_LAUNCH_REG(0, 16, false); }
#undef DBGK_LAUNCH_REG
				const uint64_t done_reads = full_tiles * reads_per_tile;
				if (d_packed) rb.packed += done_reads * U.L / 16; // (a whole number of words: a tile is a multiple of 16 bases)
				else rb.bases += done_reads * U.L;
				rb.n_bases -= done_reads * U.L;
				U.n_lanes = (n_reads - done_reads) * U.Q;
				rest_only = U.n_lanes == 0;
			}
		}
		const uint64_t n_tiles_rest = (U.n_lanes + kL1Threads - 1) / kL1Threads;
		const int grid = (int)std::min<uint64_t>(std::max<uint64_t>(n_tiles_rest, 1), (uint64_t)h->n_cu * l1_wgs_per_cu());
		if (rest_only) {
		}
#ifdef DBGK_EXPERIMENTS
		else if (dbg_mode_u == 1)   // timing experiments on cfg2's shape (C = 15, equal lengths, size < 2^31): results are wrong
			hipLaunchKernelGGL((k_extract_scatter_uniform<1, 0, 15, false>), dim3(grid), dim3(kL1Threads), sizeof(UniformLds), h->stream, rb, U, d_offsets, h->geom, h->store, h->d_ctr);
		else if (dbg_mode_u == 2)
			hipLaunchKernelGGL((k_extract_scatter_uniform<2, 0, 15, false>), dim3(grid), dim3(kL1Threads), sizeof(UniformLds), h->stream, rb, U, d_offsets, h->geom, h->store, h->d_ctr);
		else if (dbg_mode_u == 3)
			hipLaunchKernelGGL((k_extract_scatter_uniform<3, 0, 15, false>), dim3(grid), dim3(kL1Threads), sizeof(UniformLds), h->stream, rb, U, d_offsets, h->geom, h->store, h->d_ctr);
#endif
		else if (lin8 && lin12 && ragged) {
			if (wide == 2) DBGK_LAUNCH_UNIFORM8(2, 12, true); else if (wide == 1) DBGK_LAUNCH_UNIFORM8(1, 12, true); else DBGK_LAUNCH_UNIFORM8(0, 12, true);
		} else if (lin8 && lin12) {
			if (wide == 2) DBGK_LAUNCH_UNIFORM8(2, 12, false); else if (wide == 1) DBGK_LAUNCH_UNIFORM8(1, 12, false); else DBGK_LAUNCH_UNIFORM8(0, 12, false);
		} else if (lin8 && ragged) {
			if (wide == 2) DBGK_LAUNCH_UNIFORM8(2, 8, true); else if (wide == 1) DBGK_LAUNCH_UNIFORM8(1, 8, true); else DBGK_LAUNCH_UNIFORM8(0, 8, true);
		} else if (lin8) {
			if (wide == 2) DBGK_LAUNCH_UNIFORM8(2, 8, false); else if (wide == 1) DBGK_LAUNCH_UNIFORM8(1, 8, false); else DBGK_LAUNCH_UNIFORM8(0, 8, false);
		} else if (wide == 2 && h->geom.kf == 2u && !ragged) { // KFREQ, direct blocks: the instantiation without hash, division and neighbour codes
			if (c15) DBGK_LAUNCH_UNIFORM(3, 15, false); else DBGK_LAUNCH_UNIFORM(3, 16, false);
		} else if (wide == 2) DBGK_LAUNCH_UNIFORM_W(2);
		else if (wide == 1) DBGK_LAUNCH_UNIFORM_W(1);
		else DBGK_LAUNCH_UNIFORM_W(0);
#undef DBGK_LAUNCH_UNIFORM8
#undef DBGK_LAUNCH_UNIFORM_W
#undef DBGK_LAUNCH_UNIFORM_K17
#undef DBGK_LAUNCH_UNIFORM
	} else if (h->part) {
		const uint64_t n_tiles = (n_chunks + kL1Threads - 1) / kL1Threads;
		const int grid = (int)std::min<uint64_t>(n_tiles, (uint64_t)h->n_cu * l1_wgs_per_cu()); // 140 KiB of LDS: one workgroup per CU
		static const int dbg_mode = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE")) : 0;
		const int wide_d = (h->geom.kf == 2u || h->geom.size >= (1ull << 32)) ? 2 : (h->geom.size >= (1ull << 31) ? 1 : 0); // how hash / size is computed
		const int force_lin = dbgk_hook("l1_linear") ? atoi(dbgk_hook("l1_linear")) : -1;
		if (!dbg_mode && (force_lin == 1 || (force_lin < 0 && h->geom.n1 > 320u))) { // many level-1 buckets: the linear form
#define DBGK_LAUNCH_FLAT_LIN(DEAD, WD)                                                                                                       \
	hipLaunchKernelGGL((k_extract_scatter_lin<DEAD, WD>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLdsLin<8>), h->stream, rb, h->geom, h->store, \
	                   h->d_ctr)
			if (has_long) {
				if (wide_d == 2) DBGK_LAUNCH_FLAT_LIN(true, 2); else if (wide_d == 1) DBGK_LAUNCH_FLAT_LIN(true, 1); else DBGK_LAUNCH_FLAT_LIN(true, 0);
			} else {
				if (wide_d == 2) DBGK_LAUNCH_FLAT_LIN(false, 2); else if (wide_d == 1) DBGK_LAUNCH_FLAT_LIN(false, 1); else DBGK_LAUNCH_FLAT_LIN(false, 0);
			}
#undef DBGK_LAUNCH_FLAT_LIN
		} else if (wide_d == 2 && has_long)
			hipLaunchKernelGGL((k_extract_scatter<true, 0, 2>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else if (wide_d == 2)
			hipLaunchKernelGGL((k_extract_scatter<false, 0, 2>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else if (wide_d && has_long)
			hipLaunchKernelGGL((k_extract_scatter<true, 0, 1>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else if (wide_d)
			hipLaunchKernelGGL((k_extract_scatter<false, 0, 1>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
#ifdef DBGK_EXPERIMENTS
		else if (dbg_mode == 1)
			hipLaunchKernelGGL((k_extract_scatter<false, 1>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else if (dbg_mode == 2)
			hipLaunchKernelGGL((k_extract_scatter<false, 2>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else if (dbg_mode == 3)
			hipLaunchKernelGGL((k_extract_scatter<false, 3>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
#endif
		else if (has_long)
			hipLaunchKernelGGL(k_extract_scatter<true>, dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else
			hipLaunchKernelGGL(k_extract_scatter<false>, dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
	} else if (h->track) {
		if (has_long)
			hipLaunchKernelGGL((k_extract_insert<true, true>), dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->tref(), h->d_ctr,
			                   h->first_pos, h->pos_base);
		else
			hipLaunchKernelGGL((k_extract_insert<false, true>), dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->tref(), h->d_ctr,
			                   h->first_pos, h->pos_base);
		h->pos_base += n_bases;
	} else if (has_long) {
		hipLaunchKernelGGL((k_extract_insert<true, false>), dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->tref(), h->d_ctr,
		                   (unsigned long long *)nullptr, (uint64_t)0);
	} else {
		hipLaunchKernelGGL((k_extract_insert<false, false>), dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->tref(), h->d_ctr,
		                   (unsigned long long *)nullptr, (uint64_t)0);
	}
	// bytes outside ACGTNacgtn were read as 'A'; if a kernel met one (Counters::other_seen) this batch's are counted now --
	// every workgroup of the launch leaves at once otherwise.  A packed batch has none: its packer counted them.
	if (!d_packed) hipLaunchKernelGGL(k_count_other_bytes, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, d_bases, n_bases, h->d_ctr);
	HIPCHK(hipGetLastError());
	return span_end(h, sp);
}
