// dbgk_host_partition.h -- part of libdbgk.so's host side (one translation unit: included by dbgk.hip, in this order).
// PARTITION engine, host side: reset, geometry (plan_partition), record stores and kernel attributes (setup_partition)
#pragma once

static int reset_state(dbgk_handle *h)
{
	bool counters_done = false;
	// (direct-block form: the build writes every block of the table, also the empty ones -- nothing to zero)
	if (h->kfreq && !(h->part && h->kf_blocks)) HIPCHK(hipMemsetAsync(h->counts, 0, h->n_counts, h->stream));
	if (h->wide) {
		if (h->wpart) { // the region build overwrites every slot of the main table: zero it only if something writes it before
			h->wzero_pending = true;
			h->wbuilt = false;
			h->pending_kmers = 0;
			h->wgeom.pass = 0;
			h->wgeom.pass_j0 = 0;
			h->wpass_open = true;
			h->wplanned = false;
			h->wnext = 0;
			h->wpasses_done = 0;
			h->exchanged = false;
			HIPCHK(hipMemsetAsync(h->wstore.cnt1, 0, (size_t)h->wgeom.n_l1 * 4, h->stream));
			HIPCHK(hipMemsetAsync(h->wstore.ovf_n, 0, 16, h->stream));
			HIPCHK(hipMemsetAsync(h->wstore.outgoing_n, 0, 8, h->stream));
		} else {
			HIPCHK(hipMemsetAsync(h->wnodes, 0, h->tslots * sizeof(WNode), h->stream));
		}
		HIPCHK(hipMemsetAsync(h->wside, 0, kWideSideSlots * sizeof(WNode), h->stream));
	} else if (h->kfreq && !h->part) {
	} else if (h->part) {
		// the region build of finalize overwrites every slot, so the 16 B/slot memset is only needed
		// if a direct-path write (merge) happens first
		h->zero_pending = true;
		h->incr = false;
		int rc = clear_record_store(h, true); // (one launch: the stores' control arrays and the counters)
		if (rc) return rc;
		counters_done = true;
	} else {
		int rc = zero_table_now(h);
		if (rc) return rc;
	}
	if (h->track) HIPCHK(hipMemsetAsync(h->first_pos, 0xFF, h->tslots * 8, h->stream));
	h->pos_base = 0;
	if (!counters_done) {
		HIPCHK(hipMemsetAsync(h->d_ctr, 0, sizeof(Counters), h->stream));
		HIPCHK(hipMemsetAsync(&h->d_ctr->polyA_slot, 0xFF, sizeof(unsigned long long), h->stream));
	}
	h->finalized = false;
	h->total_reads = 0;
	h->host_other_bytes = 0;
	return DBGK_OK;
}

// Decide whether the PARTITION engine is used and allocate its record stores.
//   geometry: level-1 bucket = slot >> r, r >= 20 chosen so that n1 = ceil(size / 2^r) <= 1024;
//   final bucket = slot >> 12 (one 4096-slot region); n2 = 2^(r-12) sub-buckets per level-1 bucket.
//   An 8-byte record must hold q = hash / size, r slot bits and 6 neighbour bits.
static int plan_partition(dbgk_handle *h)
{
	h->part = false;
	h->sharded = false;
	memset(&h->store, 0, sizeof h->store);
	memset(&h->geom, 0, sizeof h->geom);
	h->tslots = h->size;
	if (h->kfreq && h->kf_blocks) { // direct blocks: `size` = 4^k, slot = the key with its block index permuted (kf_slot_of_key)
		PartGeom &G = h->geom;
		const uint32_t bits = 2u * (uint32_t)h->cfg.kmer_size; // >= 26
		// level-1 bucket = slot >> r: 256 buckets where the table allows (the wave-per-bucket level-1 kernel), level 2 then
		// splits a bucket into its 2^(r - 16) <= 1024 blocks in one pass
		const uint32_t r = std::max(20u, std::min(26u, bits - 8u));
		G.size = h->size;
		G.magic = h->magic;
		if (h->size < (1ull << 32)) G.div = make_div32_magic((uint32_t)h->size);
		G.r = r;
		G.n1 = (uint32_t)(h->size >> r);
		G.n2 = 1u << (r - kKfBlockBits);
		G.n_final = (uint32_t)(h->size >> kKfBlockBits);
		G.n_ranks = 1;
		G.rank = 0;
		G.B = G.n1;
		G.n_sub = kSubStores;
		G.b_lo = 0;
		G.nb_own = G.n1;
		G.slot_lo = 0;
		G.slot_hi = h->size;
		G.n_regions_own = G.n_final;
		const double per_slot = (double)h->cfg.expected_kmers / (double)h->size;
		// a level-1 bucket sums 2^(r - 16) blocks of very different weight (canonical k-mers favour small key values): more slack
		// than the hashed form's 5 %; a single block may hold 2.2 times the average
		G.cap1 = (uint64_t)(per_slot * (double)(1ull << r) * 1.2 / (double)G.n_sub) + 65536 / G.n_sub + 8192;
		G.cap2 = ((uint64_t)(per_slot * (double)(1ull << kKfBlockBits) * 2.6) + 1024 + 3) & ~3ull; // (16-bit records, read four at a time)
		G.r_rec = r;
		G.l2_shift = kKfBlockBits - (uint32_t)kRegionBits; // level 2 splits by block, not by 4096-slot region
		G.kf = 2u;
		G.l2_records = 16u * (uint32_t)l2_threads((int)G.n2);
		G.kf_mask = (uint32_t)((1ull << (bits - kKfBlockBits)) - 1ull);
		if (G.n1 > (uint32_t)kL1MaxB || G.n2 > (uint32_t)kMaxBuckets) return DBGK_OK; // (cannot happen for 13 <= k <= 18)
		h->shard_rank = 0;
		h->part = true;
		h->three = false;
		return DBGK_OK;
	}
	const uint32_t n_ranks = h->cfg.shard_count > 1 ? h->cfg.shard_count : 1;
	const int want = h->cfg.engine == DBGK_ENGINE_SEEDIDX ? DBGK_ENGINE_DIRECT // the seed index uses the plain table
	                 : h->cfg.engine == DBGK_ENGINE_KFREQ ? DBGK_ENGINE_AUTO   // KFREQ: only if the geometry is feasible
	                                                      : h->cfg.engine;
	const bool want_shard = h->cfg.shard_count >= 1; // shard_count == 1: one-rank sharded handle (same protocol, for testing)
	if (want_shard && h->cfg.shard_index >= n_ranks) return DBGK_ERR_ARG;
	if (!want_shard) {
		if (want == DBGK_ENGINE_DIRECT) return DBGK_OK;
		if (want == DBGK_ENGINE_AUTO && h->cfg.expected_kmers == 0) return DBGK_OK; // streaming use: total unknown
	}
	uint32_t r = 22; // measured on cfg2 (round 2, profiles/r02_r_sweep.txt): r = 20 / 21 / 22 -> 17.3 / 16.5 / 16.3 ms per step
	if (const char *e = DBGK_EXPERIMENT_ENV("DBGK_PART_R")) r = (uint32_t)std::max(20, std::min(24, atoi(e))); // tuning knob: level-1 bucket = slot >> r
	while (((h->size + (1ull << r) - 1) >> r) > (uint64_t)kL1MaxB) r++;
	const uint64_t qmax = ~0ull / h->size;
	int qbits = 0;
	while (qbits < 64 && (qmax >> qbits)) qbits++;
	while (r > 20 && qbits + (int)r + 6 > 64 && ((h->size + (1ull << (r - 1)) - 1) >> (r - 1)) <= (uint64_t)kL1MaxB) r--; // small tables: q needs the bits
	// level 1 fans out to <= 1024 buckets, level 2 to 2^(r-12) <= 4096 final buckets per level-1 bucket: 2^34 slots
	const bool feasible = (1u << (r - kRegionBits)) <= (uint32_t)kMaxBucketsL2 && (qbits + (int)r + 6) <= 64 && h->size >= (1ull << 26) &&
	                      h->size < (1ull << 34);
	if (!feasible || (want_shard && want == DBGK_ENGINE_DIRECT)) {
		if (want == DBGK_ENGINE_PARTITION || want_shard) {
			g_last_error = "PARTITION engine (and any sharded handle) needs 2^26 <= table_slots < 2^34";
			return DBGK_ERR_ARG;
		}
		return DBGK_OK;
	}
	for (uint64_t x : {0ull, 1ull, 0x0123456789ABCDEFull, ~0ull, 488296166657017542ull}) {
		if (hash_code_inverse(hash_code(x)) != x) {
			g_last_error = "hash_code_inverse self-check failed";
			return DBGK_ERR_STATE;
		}
	}
	PartGeom &G = h->geom;
	G.size = h->size;
	G.magic = h->magic;
	if (h->size < (1ull << 32)) G.div = make_div32_magic((uint32_t)h->size);
	G.r = r;
	G.n1 = (uint32_t)((h->size + (1ull << r) - 1) >> r);
	G.n2 = 1u << (r - kRegionBits);
	G.n_final = (uint32_t)((h->size + kRegionSlots - 1) >> kRegionBits);
	G.n_ranks = n_ranks;
	G.rank = want_shard ? h->cfg.shard_index : 0;
	h->shard_rank = G.rank;
	G.B = (G.n1 + n_ranks - 1) / n_ranks;
	G.n_sub = kSubStores;
	G.b_lo = std::min(G.rank * G.B, G.n1);
	G.nb_own = std::min(G.B, G.n1 - G.b_lo);
	G.slot_lo = (uint64_t)G.b_lo << r;
	G.slot_hi = std::min<uint64_t>(h->size, ((uint64_t)G.b_lo + G.nb_own) << r);
	G.n_regions_own = (uint32_t)((G.slot_hi - G.slot_lo + kRegionSlots - 1) >> kRegionBits);
	if (G.nb_own == 0 || (uint64_t)n_ranks * G.B * G.n_sub > (uint64_t)kMaxInboxEntries) {
		g_last_error = "shard_count too large for this table size";
		return DBGK_ERR_ARG;
	}
	// expected_kmers = occurrences THIS handle extracts; a region receives the global density
	const uint64_t expected = h->cfg.expected_kmers ? h->cfg.expected_kmers : h->size * 2 / n_ranks;
	const double per_slot = (double)expected / (double)h->size;
	G.cap1 = (uint64_t)(per_slot * (double)(1ull << r) * 1.05 / (double)G.n_sub) + 65536 / G.n_sub + (G.n_sub > 1 ? 8192 : 0); // per sub-store
	G.cap2 = (uint64_t)(per_slot * (double)n_ranks * (double)kRegionSlots * 1.15) + 512;
	// every bucket starts on a 128-byte line (the 16-byte record loads of level 2 and of the build are aligned), and a bucket is an ODD
	// number of lines long: the append points of neighbouring buckets then fall on neighbouring lines modulo any power of two, whatever
	// the memory system's channel interleave (on one box WIDE level 2 moved between 44 and 50 ms from process to process until its strides
	// were odd; other boxes run it at 44 or at 49 whatever the strides, profiles/r05_cfg5_wide_l2_variants.txt)
	G.cap1 = ((G.cap1 + 15u) & ~15ull) | 16u;
	G.cap2 = ((G.cap2 + 15u) & ~15ull) | 16u;
	G.r_rec = r;
	G.l2_shift = 0;
	G.l2_records = 16u * (uint32_t)l2_threads((int)G.n2);
	h->tslots = G.slot_hi - G.slot_lo;
	h->sharded = want_shard;
	h->part = true;
	static const bool no_three = DBGK_EXPERIMENT_ENV("DBGK_THREE_LEVEL") && atoi(DBGK_EXPERIMENT_ENV("DBGK_THREE_LEVEL")) == 0; // measurements
	// measured with cfg2's 1.2 G records, level 2 alone: n2 = 4096 (8.6 G slots) 17.6 ms in one pass, 9.7 ms in two;
	// n2 = 2048 (5 G slots) 6.9 ms in one pass, 9.5 in two -- so only the 4096-way fan-out is split
	h->three = G.n2 > 2048u && !no_three;
	if (h->three) {
		h->fan_mid = G.n2 / 64u; // 64
		uint32_t lg = 0;
		while ((1u << lg) < h->fan_mid) lg++;
		const uint64_t cap_mid = ((uint64_t)(per_slot * (double)n_ranks * (double)(1ull << (r - lg)) * 1.08) + 8192 + 15u) & ~15ull;
		h->g_mid = G;
		h->g_mid.n2 = h->fan_mid;
		h->g_mid.l2_records = 16u * (uint32_t)l2_threads((int)h->fan_mid);
		h->g_mid.l2_shift = 6;      // the low 6 bits of the final-bucket index are left to the final pass
		h->g_mid.cap2 = cap_mid;
		h->g_fin = G;
		h->g_fin.n_ranks = 1;       // its input is this handle's own mid store
		h->g_fin.rank = 0;
		h->g_fin.n_sub = 1;
		h->g_fin.r = r - lg;
		h->g_fin.n1 = G.n1 * h->fan_mid;
		h->g_fin.B = G.nb_own * h->fan_mid;
		h->g_fin.b_lo = G.b_lo * h->fan_mid;
		h->g_fin.nb_own = G.nb_own * h->fan_mid;
		h->g_fin.n2 = 64;
		h->g_fin.l2_records = 16u * (uint32_t)l2_threads(64);
		h->g_fin.cap1 = cap_mid;
	}
	return DBGK_OK;
}

// allocate the record stores of the PARTITION engine (geometry already planned)
static int setup_partition(dbgk_handle *h)
{
	if (!h->part) return DBGK_OK;
	const PartGeom &G = h->geom;
	PartStore &P = h->store;
	const uint64_t expected = h->cfg.expected_kmers ? h->cfg.expected_kmers : h->size * 2 / G.n_ranks;
	// records that find their bucket full are kept as {key, lb, rb} triples and inserted through the global
	// path after the build: a key that occurs more often than a final bucket holds (cap2, ~1.15x the mean
	// bucket fill) sends its surplus here, so this bounds the share of occurrences that may belong to such
	// heavy hitters (high-copy repeats): 1/16 of the input + 1 M; beyond it finalize returns DBGK_ERR_CAPACITY
	P.ovf_cap = expected / 16 + (1ull << 20);
	P.spill_cap = (uint64_t)G.n_regions_own * 8 + (1ull << 16);
	P.outgoing_cap = 1ull << 16;
	const size_t n_entries = (size_t)G.n_ranks * G.B * G.n_sub;
	const size_t l1_bytes = n_entries * G.cap1 * 8, l2_bytes = (size_t)G.nb_own * G.n2 * G.cap2 * (G.kf == 2u ? 2 : 8); // (direct blocks: 16-bit records)
	// (+ 64 bytes: level 2 and the build load their records in pairs, and the second half of a bucket's last pair may lie behind the bucket)
	bool ok = hipMalloc(&P.l1, l1_bytes + 64) == hipSuccess && hipMalloc(&P.l2, l2_bytes + 64) == hipSuccess &&
	          hipMalloc(&P.cnt1, n_entries * 4) == hipSuccess && hipMalloc(&P.cnt2, (size_t)G.nb_own * G.n2 * 4) == hipSuccess &&
	          hipMalloc(&P.ovf, P.ovf_cap * sizeof(Node)) == hipSuccess && hipMalloc(&P.spill, P.spill_cap * sizeof(Node)) == hipSuccess &&
	          hipMalloc(&P.ovf_n, 16) == hipSuccess && hipMalloc(&h->tile_prefix, (n_entries + 1) * 4) == hipSuccess &&
	          hipMalloc(&h->region_cursor, ((size_t)kMaxBuildLaunches + 2 + G.n_regions_own) * sizeof(unsigned int)) == hipSuccess && // + redo cursor, count, list

	          hipMalloc(&P.outgoing, P.outgoing_cap * sizeof(Node)) == hipSuccess && hipMalloc(&P.outgoing_n, 8) == hipSuccess;
	if (ok && h->sharded)
		ok = hipMalloc(&h->inbox, l1_bytes + 64) == hipSuccess && hipMalloc(&h->inbox_cnt, n_entries * 4) == hipSuccess;
	if (!ok) {
		g_last_error = "hipMalloc of the PARTITION record stores failed";
		return DBGK_ERR_NOMEM;
	}
	// side table for the surplus of heavy hitters; on a sharded handle it may hold keys of any shard and is
	// offered to every rank after the build (dbgk_shard_heavy), like the overflow list
	if (hipMalloc(&P.hh, kHeavyHitterSlots * sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
	HIPCHK(hipMemsetAsync(P.hh, 0, kHeavyHitterSlots * sizeof(Node), h->stream)); // once: afterwards it is zeroed again only when it was used (clear_record_store)
	P.hh_size = kHeavyHitterSlots;
	P.hh_magic = make_mod_magic(kHeavyHitterSlots);
	P.inbox = h->sharded ? h->inbox : P.l1;
	P.inbox_cnt = h->sharded ? h->inbox_cnt : P.cnt1;
	// EARLY level 2 (early_l2): a handle that extracts into its own inbox scatters what earlier batches stored while the next
	// batch is on the link.  Not for shards (their inbox is filled by the exchange), the three-level form, 32-bit KFREQ records
	const bool no_early = dbgk_hook("early_l2") && atoi(dbgk_hook("early_l2")) == 0; // (read per handle: measurements, and the tests compare the two)
	P.l2_done = P.l2_upto = nullptr;
	if (!h->sharded && !h->three && G.kf != 2u && !no_early) {
		if (hipMalloc(&h->l2_done, 2 * n_entries * 4) != hipSuccess) return DBGK_ERR_NOMEM; // done[] and upto[]
		HIPCHK(hipMemsetAsync(h->l2_done, 0, 2 * n_entries * 4, h->stream));
		P.l2_done = h->l2_done;
		P.l2_upto = h->l2_done + n_entries;
	}
	h->l2_seen_kmers = 0;
	h->store_capacity = expected;
	if (h->three) {
		const size_t n_mid = (size_t)G.nb_own * h->fan_mid;
		if (hipMalloc(&h->mid, n_mid * h->g_mid.cap2 * 8 + 64) != hipSuccess || hipMalloc(&h->cnt_mid, n_mid * 4) != hipSuccess ||
		    hipMalloc(&h->tile_prefix2, (n_mid + 1) * 4) != hipSuccess) {
			g_last_error = "hipMalloc of the mid-level record store failed";
			return DBGK_ERR_NOMEM;
		}
		h->s_mid = P;           // reads the inbox like level 2, writes the mid store
		h->s_mid.l2 = h->mid;
		h->s_mid.cnt2 = h->cnt_mid;
		h->s_fin = P;           // reads the mid store, writes the final buckets where level 2 would
		h->s_fin.inbox = h->mid;
		h->s_fin.inbox_cnt = h->cnt_mid;
	}
#define DBGK_UNIFORM_ATTRS(W)                                                  \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 16, false>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 16, true>), sizeof(UniformLds));  \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 15, false>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 15, true>), sizeof(UniformLds));  \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 15, false, false, false, false, true, true>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 16, false, false, false, false, true, true>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 15, true, false, false, false, true, true>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 16, true, false, false, false, true, true>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 16, false, false, true, false, true>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 15, false, false, true, false, true>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 16, false, false, true, true, true>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 15, false, false, true, true, true>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 16, false, false, true, false, false>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 15, false, false, true, false, false>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 16, false, false, true, true, false>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 15, false, false, true, true, false>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter<true, 0, W>), sizeof(ScatterLds));              \
	DBGK_LDS_ATTR((k_extract_scatter<false, 0, W>), sizeof(ScatterLds))
#ifdef DBGK_EXPERIMENTS // timing experiments (DBGK_DEBUG_MODE, results are wrong): not in the product library
	DBGK_LDS_ATTR((k_extract_scatter_uniform<1, 0, 15, false>), sizeof(UniformLds));
	DBGK_LDS_ATTR((k_extract_scatter_uniform<2, 0, 15, false>), sizeof(UniformLds));
	DBGK_LDS_ATTR((k_extract_scatter_uniform<3, 0, 15, false>), sizeof(UniformLds));
	DBGK_LDS_ATTR((k_extract_scatter<false, 1>), sizeof(ScatterLds));
	DBGK_LDS_ATTR((k_extract_scatter<false, 2>), sizeof(ScatterLds));
	DBGK_LDS_ATTR((k_extract_scatter<false, 3>), sizeof(ScatterLds));
	DBGK_LDS_ATTR((k_scatter_l2<1>), sizeof(ScatterLdsL2));
	DBGK_LDS_ATTR((k_scatter_l2<2>), sizeof(ScatterLdsL2));
	DBGK_LDS_ATTR((k_scatter_l2<3>), sizeof(ScatterLdsL2));
#endif
	DBGK_UNIFORM_ATTRS(0);
	DBGK_UNIFORM_ATTRS(1);
	DBGK_UNIFORM_ATTRS(2);
#define DBGK_LIN_ATTRS(W)                                                                          \
	DBGK_LDS_ATTR((k_extract_scatter_lin<false, W>), sizeof(ScatterLdsLin<8>));                   \
	DBGK_LDS_ATTR((k_extract_scatter_lin<true, W>), sizeof(ScatterLdsLin<8>));                    \
	DBGK_LIN_ATTRS_U(W)
#define DBGK_LIN_ATTRS_U(W)                                                                          \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 8, false, true>), sizeof(UniformLdsLin<8>));   \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 8, true, true>), sizeof(UniformLdsLin<8>));    \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 12, false, true>), sizeof(UniformLdsLin<12>)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 12, true, true>), sizeof(UniformLdsLin<12>)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 8, false, true, false, false, true, true>), sizeof(UniformLdsLin<8>));   \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 8, true, true, false, false, true, true>), sizeof(UniformLdsLin<8>));    \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 12, false, true, false, false, true, true>), sizeof(UniformLdsLin<12>)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 12, true, true, false, false, true, true>), sizeof(UniformLdsLin<12>))
	DBGK_LIN_ATTRS(0);
	DBGK_LIN_ATTRS(1);
	DBGK_LIN_ATTRS(2);
#undef DBGK_LIN_ATTRS
#undef DBGK_LIN_ATTRS_U
#undef DBGK_UNIFORM_ATTRS
	DBGK_LDS_ATTR((k_scatter_l2<0>), sizeof(ScatterLdsL2));
	DBGK_LDS_ATTR((k_scatter_l2<0, kMaxBuckets, true>), sizeof(ScatterLdsL2));
	DBGK_LDS_ATTR((k_scatter_l2<0, 2048>), sizeof(ScatterLdsL2T<2048>));
	DBGK_LDS_ATTR((k_scatter_l2<0, 4096>), sizeof(ScatterLdsL2T<4096>));
#define DBGK_BUILD_ATTR(...) DBGK_LDS_ATTR((k_build_regions<__VA_ARGS__>), sizeof(BuildLds))
	DBGK_BUILD_ATTR(0, false, false, false); DBGK_BUILD_ATTR(0, true, false, false); DBGK_BUILD_ATTR(0, false, true, false); DBGK_BUILD_ATTR(0, true, true, false);
	DBGK_BUILD_ATTR(0, false, false, true);  DBGK_BUILD_ATTR(0, true, false, true);  DBGK_BUILD_ATTR(0, false, true, true);  DBGK_BUILD_ATTR(0, true, true, true);
	DBGK_BUILD_ATTR(0, false, false, false, true); DBGK_BUILD_ATTR(0, true, false, false, true); DBGK_BUILD_ATTR(0, false, true, false, true);
	DBGK_BUILD_ATTR(0, true, true, false, true);
#ifdef DBGK_EXPERIMENTS
	DBGK_BUILD_ATTR(1, false, false, false); DBGK_BUILD_ATTR(2, false, false, false); DBGK_BUILD_ATTR(3, false, false, false);
	DBGK_BUILD_ATTR(1, false, false, true);  DBGK_BUILD_ATTR(2, false, false, true);  DBGK_BUILD_ATTR(3, false, false, true);
#endif
#undef DBGK_BUILD_ATTR
	DBGK_LDS_ATTR((k_kf_build_blocks<false, true>), sizeof(KfBlockLds));
	DBGK_LDS_ATTR((k_kf_build_blocks<true, true>), sizeof(KfBlockLds));
	DBGK_LDS_ATTR((k_kf_build_blocks<false, false>), sizeof(KfBlockLds));
	DBGK_LDS_ATTR((k_kf_build_blocks<true, false>), sizeof(KfBlockLds));
	DBGK_LDS_ATTR((k_kf_build_blocks<false, false, true>), sizeof(KfBlockLds));
	DBGK_LDS_ATTR((k_kf_build_blocks<true, false, true>), sizeof(KfBlockLds));
	return DBGK_OK;
}

static int ensure_slot(dbgk_handle *h, StageSlot &s);
