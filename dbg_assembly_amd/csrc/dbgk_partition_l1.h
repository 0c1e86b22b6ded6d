// dbgk_partition_l1.h -- part of the PARTITION engine (dbgk_partition.h includes it, in this order: _l1, _l2, _build): level 1 -- reads -> records, scattered into the level-1 buckets: the flat forms, the equal-length forms (regular tiles, general, ragged, linear), the pipelined tile loop (L1Pipe), the lane-prefix form for reads of any lengths
#pragma once

namespace dbgk {

// ---- level 1: extraction fused with the first scatter ------------------------------------------
// (A variant without LDS staging -- every lane storing its own 8-byte records into the reserved
// bucket ranges -- was measured at 24.5 ms against 10.4 ms for the staged form: uncoalesced 8-byte
// stores are the wrong trade on this chip even though the L2 merges them into lines.)
// DBG != 0 are timing experiments selected with DBGK_DEBUG_MODE (results are wrong): 1 = extraction
// only, 2 = no copy-out, 3 = copy-out into a small window
// all 16 positions of one lane: canonical k-mer, neighbour codes, hash, slot, record; the record is
// parked in the (still unused) stage buffer, column i of this thread, and ranked right away with an LDS
// histogram atomic (bkt[i] = (bucket << 16) | rank).  Returns true when some canonical k-mer of the lane
// is 0 (poly-A / poly-T): rare, the caller then feeds the key-0 side node.
// WIDE_D: how hash / size is computed -- 0: size < 2^31 (one multiply-high with a 32-bit remainder fix-up), 1: size < 2^32
// (two 2-by-1 division steps), 2: any size (64-bit multiply-high by floor(2^64 / size), 64-bit remainder)
// SPECIAL (regular tiles of the equal-length kernel, k >= 17, every lane's NPOS windows valid): the rolls work on the 32-bit halves
// -- with 2k > 32 the head mask only touches the high word and the entering complement base only the high word of rc -- and the
// per-position validity test is gone.
// ROLL32 (k >= 17, any validity pattern -- the prefix form of reads of any lengths): the two savings of SPECIAL that do not depend on
// every window being valid -- the 32-bit rolls, and the neighbour codes taken after the strand select
// IN_REGS (pipelined regular tiles, SPECIAL): the records stay in registers (rec[]) instead of being parked in the stage buffer --
// which still holds the sorted records of the tile before -- the ranks are taken in `hist` (one of two histograms), and mid(i) runs
// in front of position i: the caller copies one run of the tile before out of the stage buffer there, its LDS read and its store in
// the shadow of the position's arithmetic
struct NoMid { __device__ __forceinline__ void operator()(uint32_t) const {} };
template <int WIDE_D, int NPOS = 16, class LDS = ScatterLds, bool SPECIAL = false, bool ROLL32 = SPECIAL, bool IN_REGS = false, class Mid = NoMid>
__device__ __forceinline__ bool l1_positions(LDS &L, const PartGeom &G, Chunk16 c, uint32_t tid, uint64_t head_mask, uint32_t rc_shift,
                                             uint32_t rel_mask, uint32_t q_shift, uint32_t (&bkt)[16], uint64_t *rec = nullptr,
                                             uint32_t hist_off = 0u, Mid mid = Mid()) // hist_off: the histogram in use, in words from L.hist
{
	static_assert(!IN_REGS || SPECIAL || ROLL32 || WIDE_D == 3, "records in registers: regular tiles of a graph handle, or a KFREQ handle with direct blocks (nothing to patch)");
	uint32_t *const hh = L.hist + (IN_REGS ? hist_off : 0u);
	// The per-position path assumes both neighbours exist; the ~2 % of positions at a read's
	// first / last window are patched afterwards (rare per lane, so the loop stays lean).
	// Complemented neighbour codes (3 - x == x ^ 3) for all 16 positions at once:
	const uint32_t lwc = ~c.lw, nbc = ~c.nb;
	uint32_t rev_mask = 0;
	bool zero_any = false; // some canonical k-mer of this lane is 0 (the key-0 node is kept apart, DBGgraph.cpp:418)
	uint32_t zero_acc = 0u;
#pragma unroll
	for (uint32_t i = NPOS; i < 16; i++) bkt[i] = (uint32_t)kL1MaxB << 16; // lanes own NPOS positions: the rest never holds a record
#pragma unroll
	for (uint32_t i = 0; i < (uint32_t)NPOS; i++) {
		mid(i);
		const uint32_t sh = 30u - 2u * i;
		const uint32_t left = (c.lw >> sh) & 3u, right = (c.nb >> sh) & 3u;
		const bool rev = c.rc < c.kbit;                         // tie -> forward (DBGgraph.cpp:80)
		const uint64_t key = rev ? c.rc : c.kbit;
		// forward: (left, right); reverse strand: (comp(right), comp(left))  (DBGgraph.cpp:82-97)
		uint32_t links = 4u; // (WIDE_D == 3, KFREQ: (lb, rb) = (0, none) and nothing below is needed)
		if constexpr (WIDE_D == 3) {
		} else if constexpr (ROLL32) { // the packed words are selected, the two codes extracted once
			uint32_t wa = rev ? nbc : c.lw, wb = rev ? lwc : c.nb;
			asm volatile("" : "+v"(wa), "+v"(wb)); // two selects, not a branch
			links = (((wa >> sh) & 3u) << 3) | ((wb >> sh) & 3u);
			if constexpr (!SPECIAL) links = G.kf ? 4u : links; // KFREQ: (lb, rb) = (0, none)  (SPECIAL: never a KFREQ handle, the host keeps those on the general form)
		} else {
			uint32_t lf = (left << 3) | right, lr = (((nbc >> sh) & 3u) << 3) | ((lwc >> sh) & 3u);
			asm volatile("" : "+v"(lf), "+v"(lr)); // both sides are cheap: a select, not a branch
			links = G.kf ? 4u : (rev ? lr : lf); // KFREQ: (lb, rb) = (0, none)
		}
		if (WIDE_D != 3 && (!SPECIAL || i == 0u || i == (uint32_t)NPOS - 1u)) { // (SPECIAL: only a read's first and last window are ever patched below)
			uint32_t rev_bit = rev ? 1u : 0u;
			asm volatile("" : "+v"(rev_bit)); // accumulate in a VGPR now instead of parking 16 condition masks in SGPRs
			rev_mask = SPECIAL ? (rev_mask | (rev_bit << ((uint32_t)NPOS - 1u - i))) : ((rev_mask << 1) | rev_bit);  // position i ends up at bit NPOS - 1 - i
		}
		uint64_t q;
		uint32_t slot, bucket; // slot: its low 32 bits (r <= 24 of them are recorded); bucket = slot >> r
		if constexpr (WIDE_D == 3) { // compiled for KFREQ with direct blocks only: no hash, no division, no neighbour codes
			const uint64_t s64 = kf_slot_of_key(key, G.kf_mask);
			q = 0ull;
			slot = (uint32_t)s64;
			bucket = (uint32_t)(s64 >> G.r);
		} else if (WIDE_D == 2) {
			uint64_t s64;
			if (G.kf == 2u) { // KFREQ, direct blocks: the slot IS the key, its block index permuted (wave-uniform branch)
				s64 = kf_slot_of_key(key, G.kf_mask);
				q = 0ull;
			} else {
				s64 = fast_divmod(hash_code(key), G.magic, q);
			}
			slot = (uint32_t)s64;
			bucket = (uint32_t)(s64 >> G.r);
		} else {
			slot = WIDE_D ? divmod_u64_u32(hash_code(key), G.div, q) : divmod_magic_small(hash_code(key), G.magic.m, (uint32_t)G.magic.d, q);
			bucket = slot >> G.r;
		}
		// only one packed register per position stays live across the tile
		const uint32_t q_lo = (uint32_t)q, q_hi = (uint32_t)(q >> 32);
		// ((q << r | place in the bucket) << 6) | links, as two shift-or instructions (q_shift = r + 6)
		const uint32_t rec_lo = (((q_lo << (q_shift - 6u)) | (slot & rel_mask)) << 6) | links;
		const uint32_t rec_hi = __builtin_amdgcn_alignbit(q_hi, q_lo, 32u - q_shift);
		if constexpr (IN_REGS) {
			// (built HERE: left to itself the compiler sinks the packing below the loop and keeps q, slot and links alive instead --
			// four registers per position for two; the rank of the position before is folded into its bucket word one position late,
			// when the LDS atomic has long returned)
			uint32_t lo_now = rec_lo, hi_now = rec_hi;
			if constexpr (WIDE_D == 3) { // (32-bit records: q = 0)
				asm volatile("" : "+v"(lo_now));
				rec[i] = lo_now;
			} else {
				asm volatile("" : "+v"(lo_now), "+v"(hi_now));
				rec[i] = ((uint64_t)hi_now << 32) | lo_now;
			}
			if (i > 0u) asm volatile("" : "+v"(bkt[i - 1u]));
		} else {
			L.stage[i * kL1Threads + tid] = ((uint64_t)rec_hi << 32) | rec_lo;
		}
		const bool valid = SPECIAL ? true : (bool)((c.valid >> i) & 1u);
		const bool zero = key == 0ull;
		if constexpr (IN_REGS) { // (a select per position into one register: with the copy-out's branches between the positions the compiler
			// would keep fifteen condition masks alive and OR them behind the loop)
			zero_acc = zero ? 1u : zero_acc;
			asm volatile("" : "+v"(zero_acc));
		} else {
			zero_any = zero_any || zero; // (the compare is needed below anyway: an OR of condition masks)
		}
		// positions without a record rank themselves in a per-lane dummy bin: no exec juggling around the LDS atomic
		const uint32_t b = (valid && !zero) ? bucket : (uint32_t)kL1MaxB + (tid & 63u);
		bkt[i] = (b << 16) | atomicAdd(&hh[b], 1u);
		// roll to the next position (DBGgraph.cpp:71-73)
		if constexpr (ROLL32) {
			const uint32_t klo = (uint32_t)c.kbit, khi = (uint32_t)(c.kbit >> 32);
			const uint32_t nhi = __builtin_amdgcn_alignbit(khi, klo, 30u) & (uint32_t)(head_mask >> 32), nlo = (klo << 2) | right;
			c.kbit = ((uint64_t)nhi << 32) | nlo;
			const uint64_t r2 = c.rc >> 2;
			c.rc = ((uint64_t)((uint32_t)(r2 >> 32) | ((right ^ 3u) << (rc_shift - 32u))) << 32) | (uint32_t)r2;
		} else {
			c.kbit = ((c.kbit << 2) | right) & head_mask;
			c.rc = (c.rc >> 2) | ((uint64_t)(3u - right) << rc_shift);
		}
	}
	// windows without a left / right neighbour: that side's code becomes 4 = none
	const uint32_t no_l = ~c.has_l & 0xFFFFu, no_r = ~c.has_r & 0xFFFFu;
	if constexpr (IN_REGS && !SPECIAL && WIDE_D != 3) { // any position may be a read's first or last window: a predicated pass over the registers
		if (const uint32_t fix = G.kf ? 0u : (no_l | no_r) & c.valid) { // (KFREQ through hashed regions: (lb, rb) = (0, none) everywhere)
#pragma unroll
			for (uint32_t i = 0; i < (uint32_t)NPOS; i++) {
				if (!((fix >> i) & 1u)) continue;
				const bool nl = (no_l >> i) & 1u, nr = (no_r >> i) & 1u;
				const bool fwd = !((rev_mask >> ((uint32_t)NPOS - 1u - i)) & 1u);
				uint32_t lb = ((uint32_t)rec[i] >> 3) & 7u, rbb = (uint32_t)rec[i] & 7u;
				if (fwd ? nl : nr) lb = 4u;
				if (fwd ? nr : nl) rbb = 4u;
				rec[i] = (rec[i] & ~63ull) | (lb << 3) | rbb;
			}
		}
		return zero_acc != 0u;
	} else if constexpr (IN_REGS) { // (SPECIAL: only a read's first window, a lane's position 0, and its last one, a lane's position NPOS - 1, lack a side)
#pragma unroll
		for (uint32_t e = 0; e < (WIDE_D == 3 ? 0u : 2u); e++) {
			const uint32_t i = e ? (uint32_t)NPOS - 1u : 0u;
			if (e && NPOS == 1) break;
			const bool nl = (no_l >> i) & 1u, nr = (no_r >> i) & 1u;
			if (nl || nr) {
				const bool fwd = !((rev_mask >> ((uint32_t)NPOS - 1u - i)) & 1u);
				uint32_t lb = ((uint32_t)rec[i] >> 3) & 7u, rbb = (uint32_t)rec[i] & 7u;
				if (fwd ? nl : nr) lb = 4u;
				if (fwd ? nr : nl) rbb = 4u;
				rec[i] = (rec[i] & ~63ull) | (lb << 3) | rbb;
			}
		}
		return zero_acc != 0u;
	}
	for (uint32_t fix = (WIDE_D == 3 || (!SPECIAL && G.kf)) ? 0u : ((no_l | no_r) & c.valid); fix; fix &= fix - 1u) {
		const uint32_t i = (uint32_t)__builtin_ctz(fix);
		const bool fwd = !((rev_mask >> ((uint32_t)NPOS - 1u - i)) & 1u), nl = (no_l >> i) & 1u, nr = (no_r >> i) & 1u;
		uint64_t rec = L.stage[i * kL1Threads + tid];
		uint32_t lb = ((uint32_t)rec >> 3) & 7u, rbb = (uint32_t)rec & 7u;
		if (fwd ? nl : nr) lb = 4u;
		if (fwd ? nr : nl) rbb = 4u;
		L.stage[i * kL1Threads + tid] = (rec & ~63ull) | (lb << 3) | rbb;
	}
	return zero_any;
}

// after the positions of a tile: reserve, scan, move the parked records into sorted order, copy out
template <int DBG, bool KF32_POSSIBLE = true, int N_SURE = 0>
__device__ __forceinline__ void l1_scatter_tail(ScatterLds &L, const PartGeom &G, const PartStore &P, Counters *ctr, uint32_t tid,
                                                const uint32_t (&bkt)[16], bool sure = false)
{
	lds_barrier(); // hist complete
	// the parked records come back into registers BEFORE the reservation and the scan: the barriers inside the scan then
	// also say that every thread has its records, and the stage buffer may be overwritten in sorted order right after
	// (one barrier less per tile; with the one dropped after the copy-out: 5.53 -> 5.47 ms)
	uint64_t rec[16];
#pragma unroll
	for (int u = 0; u < 16; u++) rec[u] = (N_SURE == 0 || u < N_SURE) ? L.stage[u * kL1Threads + tid] : 0ull; // (N_SURE = C: a lane owns C positions, the rest never holds a record)
	uint32_t my_gbase[ScatterLds::kBpt];
	const uint32_t sub = blockIdx.x % G.n_sub; // this workgroup's sub-store (its XCD under round-robin dispatch)
	scatter_reserve_scan(L, G.n1, P.cnt1 + sub, my_gbase, G.n_sub);
	scatter_stage_copy<16, DBG, false, KF32_POSSIBLE, N_SURE>(L, rec, bkt, my_gbase, G.n1, P.l1 + (uint64_t)sub * G.cap1, G.cap1, 0u, true, G, P, ctr, G.n_sub, 0u, sure);
}

// LINEAR form for MANY level-1 buckets (large tables, and every rank of a multi-GPU job: the level-1 buckets are those
// of the GLOBAL table).  With several hundred buckets a 16 K-record tile holds only a few dozen records per bucket and
// the wave-per-bucket copy-out above issues one mostly empty store per bucket (level 1 at n1 = 1023: 10.6 ms against 5.5
// at n1 = 143).  Here a tile is 8 or 12 records per thread, every staged record carries a 16-bit bucket tag, and the copy-out
// walks the sorted stage linearly, every lane busy -- its cost no longer depends on the number of buckets.
template <int C> // C = records per thread and tile: 8 or 12
struct ScatterLdsLin {
	static constexpr int kThreads = kL1Threads;
	static constexpr int kRecords = kL1Threads * C;
	static constexpr int kMaxB = kL1MaxB;
	static constexpr int kBpt = kL1MaxB / kL1Threads;
	using Desc = uint32_t;
	uint64_t stage[kRecords];
	uint32_t hist[kL1MaxB + 64];
	uint32_t lbase[kL1MaxB];
	uint32_t desc[kL1MaxB];
	uint32_t wave_tot[kL1Threads / 64];
	uint16_t bucket_of[kRecords];
};

template <int DBG, int C, bool KF32_POSSIBLE = true>
__device__ __forceinline__ void l1_scatter_tail_linear(ScatterLdsLin<C> &L, const PartGeom &G, const PartStore &P, Counters *ctr, uint32_t tid,
                                                       const uint32_t (&bkt)[16])
{
	using ScatterLds8 = ScatterLdsLin<C>;
	lds_barrier(); // hist complete
	uint64_t rec[C];
#pragma unroll
	for (int u = 0; u < C; u++) rec[u] = L.stage[u * kL1Threads + tid];
	uint32_t my_gbase[ScatterLds8::kBpt];
	const uint32_t sub = blockIdx.x % G.n_sub;
	scatter_reserve_scan(L, G.n1, P.cnt1 + sub, my_gbase, G.n_sub);
	const uint32_t total = L.lbase[G.n1 - 1u] + L.hist[G.n1 - 1u]; // (read now: the next tile zeroes the histogram while slower waves still copy out)
#pragma unroll
	for (int u = 0; u < C; u++) {
		const uint32_t b = bkt[u] >> 16;
		if (b < (uint32_t)kL1MaxB) {
			const uint32_t at = L.lbase[b] + (bkt[u] & 0xFFFFu);
			L.stage[at] = rec[u];
			L.bucket_of[at] = (uint16_t)b;
		}
	}
#pragma unroll
	for (int j = 0; j < ScatterLds8::kBpt; j++) L.desc[ScatterLds8::kBpt * tid + j] = my_gbase[j];
	lds_barrier();
	if (DBG != 2) {
		uint64_t *out = P.l1 + (uint64_t)sub * G.cap1; // bucket b lives at out + b * n_sub * cap1
		// (two neighbouring records per lane and 16-byte stores where both fall into one bucket were measured slower: with
		// many buckets most pairs straddle a boundary -- 8.5 against 7.75 ms at n1 = 1023, 6.80 against 6.65 at n1 = 143)
#pragma unroll
		for (int u = 0; u < C; u++) {
			const uint32_t p = (uint32_t)u * kL1Threads + fresh_tid();
			if (p >= total) continue;
			const uint64_t rcd = L.stage[p];
			const uint32_t b = L.bucket_of[p];
			const uint64_t off = (uint64_t)L.desc[b] + (p - L.lbase[b]);
			if (off < G.cap1) {
				if (KF32_POSSIBLE && G.kf == 2u) reinterpret_cast<uint32_t *>(out)[(uint64_t)b * G.cap1 + off] = (uint32_t)rcd; // (32-bit level-1 records, scatter_stage_copy)
				else out[(uint64_t)b * G.n_sub * G.cap1 + off] = rcd;
			} else { // the bucket is full: records beyond its capacity go to the overflow list
				push_overflow(P, record_key(rcd, b, G), (uint32_t)(rcd >> 3) & 7u, (uint32_t)rcd & 7u, ctr);
			}
		}
	}
	// (no barrier here, as above: the next tile meets its first barrier before any record is parked in the stage buffer again,
	// and lbase / desc / bucket_of are only rewritten after its scan)
}

template <bool HAS_DEAD, int DBG = 0, int WIDE_D = 0>
__global__ __launch_bounds__(kL1Threads) void k_extract_scatter(ReadBatch rb, PartGeom G, PartStore P, Counters *__restrict__ ctr)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	ScatterLds &L = *reinterpret_cast<ScatterLds *>(lds_raw);
	const uint64_t n_chunks = (rb.n_bases + 15u) >> 4;
	const uint64_t n_tiles = (n_chunks + kL1Threads - 1) / kL1Threads;
	const uint32_t k = (uint32_t)rb.k;
	const uint64_t head_mask = (k < 32u) ? ((1ull << (2u * k)) - 1ull) : ~0ull;  // KmerHeadMaskVal, DBGgraph.cpp:371
	const uint32_t rc_shift = 2u * k - 2u;                                      // KmerRCOrVal[b] = (3-b) << (2k-2), :373-376
	const uint32_t rel_mask = (1u << G.r) - 1u, q_shift = G.r + 6u;             // record = q << (r+6) | slot_rel << 6 | lb << 3 | rb

	// a tile is INTERIOR when every chunk it touches (incl. the two halo chunks past its end) is a
	// full 16 bytes inside the buffer: its loads need no guards and are issued back to back
	auto interior = [&](uint64_t tile) { return ((tile + 1u) * kL1Threads + 2u) * 16u <= rb.n_bases; };
	auto fetch = [&](uint64_t tile) {
		const uint64_t ch = tile * kL1Threads + fresh_tid();
		if (tile >= n_tiles) return RawChunk{};
		return interior(tile) ? load_raw<HAS_DEAD, false>(rb, ch, n_chunks) : load_raw<HAS_DEAD, true>(rb, ch, n_chunks);
	};
	// The loads of tile i+1 are issued when the extraction of tile i is done and are complete (they
	// precede the reservation atomics, whose results staging waits for) before tile i's copy-out
	// stores are issued: nothing ever waits for those stores, they drain during the next extraction.
	RawChunk raw = fetch(blockIdx.x);
	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint32_t tid = fresh_tid();
		const uint64_t chunk = tile * kL1Threads + tid;
		uint32_t bkt[16]; // (bucket << 16) | rank once the position has been processed
#pragma unroll
		for (int j = 0; j < ScatterLds::kBpt; j++) L.hist[ScatterLds::kBpt * tid + j] = 0;
		lds_barrier();
		Chunk16 c = decode_chunk16<HAS_DEAD>(raw, rb, chunk);
		if (chunk >= n_chunks) c.valid = 0u;
		const bool zero_seen = l1_positions<WIDE_D>(L, G, c, tid, head_mask, rc_shift, rel_mask, q_shift, bkt);
		if (zero_seen && chunk < n_chunks) { // key-0 side node (DBGgraph.cpp:153-164): rare; redo the chunk on the plain path, rolled
			// (lanes past the last chunk hold 'A' padding, i.e. key 0 everywhere: nothing of theirs is valid)
			LaneWindow w = load_lane_window<HAS_DEAD>(rb, chunk);
#pragma unroll 1
			for (uint32_t i = 0; i < 16; i++) {
				const Triple tr = next_triple<HAS_DEAD>(w, i, rb.k, rb.n_bases);
				if (tr.valid && tr.key == 0ull)
					links_cas_observe(&ctr->polyA_links, *reinterpret_cast<volatile unsigned long long *>(&ctr->polyA_links),
					                  G.kf ? 0u : tr.lb, G.kf ? 4u : tr.rb);
			}
		}
		if (DBG == 1) {
			uint64_t x = 0;
#pragma unroll
			for (int u = 0; u < 16; u++) x ^= L.stage[u * kL1Threads + tid] + bkt[u];
			if (x == 0x1234567u) P.l1[threadIdx.x] = x;
			lds_barrier();
			raw = fetch(tile + gridDim.x);
			continue;
		}
		const RawChunk nxt = fetch(tile + gridDim.x);
		l1_scatter_tail<DBG, (WIDE_D >= 2)>(L, G, P, ctr, tid, bkt);
		raw = nxt;
	}
}

// The flat kernel in its LINEAR form (many level-1 buckets, see ScatterLdsLin): a lane's 16 positions are handled as two
// tiles of 8; between them the chunk state moves on by 8 positions.
template <bool HAS_DEAD, int WIDE_D = 0>
__global__ __launch_bounds__(kL1Threads) void k_extract_scatter_lin(ReadBatch rb, PartGeom G, PartStore P, Counters *__restrict__ ctr)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	using LDS = ScatterLdsLin<8>;
	LDS &L = *reinterpret_cast<LDS *>(lds_raw);
	const uint64_t n_chunks = (rb.n_bases + 15u) >> 4;
	const uint64_t n_tiles = (n_chunks + kL1Threads - 1) / kL1Threads;
	const uint32_t k = (uint32_t)rb.k;
	const uint64_t head_mask = (k < 32u) ? ((1ull << (2u * k)) - 1ull) : ~0ull;
	const uint32_t rc_shift = 2u * k - 2u;
	const uint32_t rel_mask = (1u << G.r) - 1u, q_shift = G.r + 6u;
	auto interior = [&](uint64_t tile) { return ((tile + 1u) * kL1Threads + 2u) * 16u <= rb.n_bases; };
	auto fetch = [&](uint64_t tile) {
		const uint64_t ch = tile * kL1Threads + fresh_tid();
		if (tile >= n_tiles) return RawChunk{};
		return interior(tile) ? load_raw<HAS_DEAD, false>(rb, ch, n_chunks) : load_raw<HAS_DEAD, true>(rb, ch, n_chunks);
	};
	RawChunk raw = fetch(blockIdx.x);
	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint64_t chunk = tile * kL1Threads + fresh_tid();
		Chunk16 c = decode_chunk16<HAS_DEAD>(raw, rb, chunk);
		if (chunk >= n_chunks) c.valid = 0u;
		raw = fetch(tile + gridDim.x); // in flight during both tiles
		bool zero_any = false;
#pragma unroll 1
		for (uint32_t half = 0; half < 2u; half++) {
			const uint32_t tid = fresh_tid();
			uint32_t bkt[16];
#pragma unroll
			for (int j = 0; j < LDS::kBpt; j++) L.hist[LDS::kBpt * tid + j] = 0;
			lds_barrier();
			Chunk16 h8 = c; // the 8 positions of this tile
			h8.valid &= 0xFFu;
			h8.has_l = (h8.has_l & 0xFFu) | 0xFF00u; // (positions 8..15 are not this tile's: never "fixed up")
			h8.has_r = (h8.has_r & 0xFFu) | 0xFF00u;
			zero_any = l1_positions<WIDE_D, 8, LDS>(L, G, h8, tid, head_mask, rc_shift, rel_mask, q_shift, bkt) || zero_any;
			// move on by 8 positions: the 8 entering bases are the top half of nb
			c.kbit = ((c.kbit << 16) | (uint64_t)(c.nb >> 16)) & head_mask;
			c.rc = revcomp_kbit(c.kbit, (int)k);
			c.lw <<= 16;
			c.nb <<= 16;
			c.valid >>= 8;
			c.has_l >>= 8;
			c.has_r >>= 8;
			l1_scatter_tail_linear<0, 8, (WIDE_D >= 2)>(L, G, P, ctr, tid, bkt);
		}
		if (zero_any && chunk < n_chunks) { // key-0 side node (DBGgraph.cpp:153-164): rare; redo the chunk on the plain path, rolled
			LaneWindow w = load_lane_window<HAS_DEAD>(rb, chunk);
#pragma unroll 1
			for (uint32_t i = 0; i < 16; i++) {
				const Triple tr = next_triple<HAS_DEAD>(w, i, rb.k, rb.n_bases);
				if (tr.valid && tr.key == 0ull)
					links_cas_observe(&ctr->polyA_links, *reinterpret_cast<volatile unsigned long long *>(&ctr->polyA_links),
					                  G.kf ? 0u : tr.lb, G.kf ? 4u : tr.rb);
			}
		}
	}
}

// ---- level 1 for batches of EQUAL-LENGTH reads ----------------------------------------------------
// The flat kernel above gives every lane 16 consecutive base positions, so a fifth of the positions of
// 150-base reads at k = 31 are windows that straddle a read boundary (hashed, divided, ranked in a dummy
// bin and thrown away).  When every read of a batch has the same length L (the usual case for short
// reads; decided per batch, the flat kernel stays the general path) lanes are mapped to chunks of VALID
// windows instead: read r = lane / Q, chunk c = lane % Q, Q = ceil(W / 16), W = L - k + 1 windows per
// read.  The lanes of a tile no longer own aligned 16-byte blocks, so the tile's byte range is packed
// to 2 bits per base cooperatively into LDS first (one aligned 16-byte load per lane, as before) and each
// lane funnels its own window out of five packed words; the boundary predicates are arithmetic.
struct UniformGeom {
	uint32_t L;          // length of every read of the batch (RAGGED: of the longest), k + 63 <= L <= maxReadLen (no trimming in this mode)
	uint32_t W;          // windows of a read of length L = L - k + 1, >= 64
	uint32_t Q;          // lanes per read = ceil(W / C), < 2048; C = 16 or 15 windows per lane, whichever wastes fewer slots
	uint32_t qmagic;     // ceil(2^22 / Q): (x * qmagic) >> 22 == x / Q for x < 2048 + Q
	uint64_t n_lanes;    // n_reads * Q
	// REGULAR tiles (k_extract_scatter_uniform<..., REG = true>): equal-length reads whose lane count Q divides the tile (a
	// power of two) and whose tile -- kL1Threads / Q whole reads -- is a multiple of 16 bytes: every tile starts at a
	// 16-byte boundary on a read start, its byte range and every lane's place in it are the same for all tiles
	uint32_t lq;            // log2(Q)
	uint32_t tile_blocks;   // 16-byte blocks of a tile = (kL1Threads / Q) * L / 16
};

constexpr int kPkWords = 1792 * kTileThreads / 1024; // packed words of one tile's byte range: <= 1024 lanes * 16 (1 + (k - 1) / W) bases + slack
struct UniformLds {
	ScatterLds s;
	uint32_t pk[kPkWords];
	uint32_t gbase[kL1MaxB]; // pipelined regular tiles: a bucket's reserved place, from the thread that reserved it to the wave that copies the run
};
template <int C>
struct UniformLdsLin {
	ScatterLdsLin<C> s;
	uint32_t pk[kPkWords];
};

__device__ __forceinline__ uint32_t funnel_left(uint32_t hi, uint32_t lo, uint32_t sh) // ({hi,lo} << sh) >> 32, sh in 0..30 (even)
{
	return sh ? ((hi << sh) | (lo >> (32u - sh))) : hi;
}

// the side node of key 0 for one lane, from its decoded window (rare path, rolled)
__device__ __forceinline__ void l1_key0_from_chunk(Chunk16 c, uint64_t head_mask, uint32_t rc_shift, uint32_t kf, Counters *ctr)
{
#pragma unroll 1
	for (uint32_t i = 0; i < 16; i++) {
		const uint32_t sh = 30u - 2u * i;
		const uint32_t left = (c.lw >> sh) & 3u, right = (c.nb >> sh) & 3u;
		const bool rev = c.rc < c.kbit;
		const uint64_t key = rev ? c.rc : c.kbit;
		if (((c.valid >> i) & 1u) && key == 0ull) {
			const uint32_t lc = ((c.has_l >> i) & 1u) ? left : 4u, rcd = ((c.has_r >> i) & 1u) ? right : 4u;
			const uint32_t lb = rev ? (rcd == 4u ? 4u : 3u - rcd) : lc, rb = rev ? (lc == 4u ? 4u : 3u - lc) : rcd;
			links_cas_observe(&ctr->polyA_links, *reinterpret_cast<volatile unsigned long long *>(&ctr->polyA_links), kf ? 0u : lb, kf ? 4u : rb);
		}
		c.kbit = ((c.kbit << 2) | right) & head_mask;
		c.rc = (c.rc >> 2) | ((uint64_t)(3u - right) << rc_shift);
	}
}

// ---- the pipelined tile loop of level 1 (k_extract_scatter_uniform, k_extract_scatter_prefix) ------------------------------------
// (round 5: level 1 4.74 -> 3.92 ms on cfg2, profiles/r05_l1_pipelined_ab.txt.)  The copy-out of a tile runs INSIDE the position loop
// of the next one -- a run's LDS read and its store between two positions' arithmetic, instead of a phase of its own in which the
// vector ALUs idle.  The records of a tile stay in registers until they are staged in sorted order (nothing is parked in the stage
// buffer, which holds the tile before), the ranks of consecutive tiles go to two histograms in turn (the second one lives in the
// descriptor array: the copying wave holds its buckets' descriptors in registers -- lane l of wave w copies bucket w + 16 l; count and
// first staged index it reads itself, the reserved place comes from the reserving thread through `gbase` in LDS), the next tile is
// opened in the tail, and a tile costs TWO barriers: (C) ranks complete and the stage buffer read out, (E) records staged, the next
// tile's words and cleared histogram in place.  Per tile:
//     positions (l1_positions<..., IN_REGS>, mid(i) = copy_run(i)); copy_rest; fetch of the tile after; barrier (C);
//     tail(rec, bkt, ..., open_next)   -- reserve, scan, stage, hand-over, open_next(histogram to clear), barrier (E)
// and after the last tile copy_rest(0).  REC32: a KFREQ handle with direct blocks -- 32-bit records, four per lane and store, the
// stage buffer used as 32-bit words.
template <bool REC32>
struct L1Pipe {
	static constexpr uint32_t kWaves = kL1Threads / 64;
	static constexpr uint32_t kHist2 = (uint32_t)((offsetof(ScatterLds, desc) - offsetof(ScatterLds, hist)) / 4u);
	static_assert(sizeof(ScatterLds::desc) >= sizeof(ScatterLds::hist), "the second histogram lives in the descriptor array");
	static_assert(kL1MaxB <= kL1Threads, "thread b reserves bucket b");
	ScatterLds &L;
	uint32_t *const gbase; // LDS, >= n1 words: a bucket's reserved place, from the thread that reserved it to the wave that copies the run
	const PartGeom &G;
	const PartStore &P;
	Counters *const ctr;
	uint32_t tid, lane, wave, per_wave, mine;
	uint64_t *out;
	uint32_t *cnt;
	uint64_t bucket_stride;
	uint32_t d_lo = 0u, d_gb = 0u; // the runs this wave copies out of the stage buffer: lane l = records << 16 | first staged index, and the global base
	uint32_t cur = 0u;             // the histogram of the current tile, in words from L.hist: 0 or kHist2

	__device__ __forceinline__ L1Pipe(ScatterLds &L_, uint32_t *gbase_, const PartGeom &G_, const PartStore &P_, Counters *ctr_)
	    : L(L_), gbase(gbase_), G(G_), P(P_), ctr(ctr_)
	{
		tid = fresh_tid();
		lane = tid & 63u;
		wave = __builtin_amdgcn_readfirstlane(tid >> 6);
		per_wave = (G.n1 + kWaves - 1u - wave) / kWaves; // buckets wave + kWaves * l < n1  (<= 64)
		mine = wave + kWaves * lane;
		const uint32_t sub = blockIdx.x % G.n_sub; // this workgroup's sub-store (l1_scatter_tail)
		out = P.l1 + (uint64_t)sub * G.cap1;
		cnt = P.cnt1 + sub;
		bucket_stride = (uint64_t)G.n_sub * G.cap1;
	}
	__device__ __forceinline__ uint32_t *stage32() const { return reinterpret_cast<uint32_t *>(L.stage); }

	__device__ __forceinline__ void copy_run(uint32_t kk, uint64_t &slow) const // (kk: wave-uniform)
	{
		const uint32_t lo = __builtin_amdgcn_readlane(d_lo, kk), dst = __builtin_amdgcn_readlane(d_gb, kk);
		const uint32_t n = lo >> 16, src = lo & 0xFFFFu;
		if (n == 0u) return;
		if ((uint64_t)dst + n > G.cap1) { // the bucket is full: after the positions, record by record
			slow |= 1ull << kk;
			return;
		}
		// (the bucket's address is scalar arithmetic redone per run: hoisted out of the tile loop, fifteen 64-bit bases cost more
		// registers than the kernel has)
		uint32_t b = wave + kWaves * kk;
		asm volatile("" : "+s"(b));
		if constexpr (REC32) { // (a level-1 record is (place in the bucket) << 6 | 4 and travels as 32 bits, scatter_stage_copy)
			static_assert(kSubStores == 1, "the 32-bit level-1 store is addressed without sub-stores");
			uint32_t *o32 = reinterpret_cast<uint32_t *>(out) + (uint64_t)b * G.cap1 + dst;
			const uint32_t *s32 = stage32();
			typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
			if (n >= 4u) { // (wave-uniform) the lane whose four would reach past the run takes the run's LAST four instead: one store
				// instruction per 256 records whatever the run's length, a few records written twice with the same value
				for (uint32_t i = 4u * lane; i < n; i += 256u) {
					const uint32_t j = min(i, n - 4u);
					const u32x4_a4 v = {s32[src + j], s32[src + j + 1u], s32[src + j + 2u], s32[src + j + 3u]};
					*reinterpret_cast<u32x4_a4 *>(o32 + j) = v;
				}
			} else if (lane < n) {
				o32[lane] = s32[src + lane];
			}
			return;
		}
		uint64_t *o = out + (uint64_t)b * bucket_stride + dst;
		typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
		if (n >= 2u) { // (wave-uniform) two records per lane and store instruction; the lane left with the odd last record takes the
			// run's last TWO instead (one record written twice with the same value: no 8-byte store instruction behind the others)
			for (uint32_t i = 2u * lane; i < n; i += 128u) {
				const uint32_t j = min(i, n - 2u);
				const uint64_t a = L.stage[src + j], b2 = L.stage[src + j + 1u];
				const u32x4_a8 v = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b2, (uint32_t)(b2 >> 32)};
				*reinterpret_cast<u32x4_a8 *>(o + j) = v;
			}
		} else if (lane == 0u) {
			o[0] = L.stage[src];
		}
	}
	__device__ __forceinline__ void copy_slow(uint64_t slow) const
	{
		while (slow) {
			const uint32_t kk = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_ctzll(slow));
			slow &= slow - 1ull;
			const uint32_t lo = __builtin_amdgcn_readlane(d_lo, kk), dst = __builtin_amdgcn_readlane(d_gb, kk);
			const uint32_t n = lo >> 16, src = lo & 0xFFFFu, b = wave + kWaves * kk;
			uint64_t *o = out + (uint64_t)b * G.n_sub * G.cap1 + dst;
			uint32_t *o32 = reinterpret_cast<uint32_t *>(out) + (uint64_t)b * G.cap1 + dst;
			for (uint32_t i = lane; i < n; i += 64u) {
				const uint64_t rcd = REC32 ? (uint64_t)stage32()[src + i] : L.stage[src + i];
				if ((uint64_t)dst + i < G.cap1) {
					if constexpr (REC32) o32[i] = (uint32_t)rcd;
					else o[i] = rcd;
				} else push_overflow(P, record_key(rcd, b, G), (uint32_t)(rcd >> 3) & 7u, (uint32_t)rcd & 7u, ctr);
			}
		}
	}
	// the wave's runs from the `from`-th on (behind the positions: more than 16 C buckets; behind the last tile: all), then the full buckets
	__device__ __forceinline__ void copy_rest(uint32_t from, uint64_t slow) const
	{
		for (uint32_t kk = from; kk < per_wave; kk++) copy_run(__builtin_amdgcn_readfirstlane(kk), slow);
		if (slow) copy_slow(slow);
	}
	// behind barrier (C): reserve, scan, stage the tile's records, hand the reservations over, open the next tile, barrier (E).
	// SURE: the caller's lanes hold C records each, all with a bucket, unless a lane has seen a zero key (regular tiles that fill their lanes)
	template <int C, bool SURE, class Open>
	__device__ __forceinline__ void tail(const uint64_t (&rec)[16], const uint32_t (&bkt)[16], bool zero_seen, Open open_next)
	{
		// one reservation per non-empty bucket, by thread b as everywhere else: consecutive lanes, consecutive counters -- a handful of
		// atomic REQUESTS per tile.  (Reserved by the copying lanes themselves -- bucket w + 16 l, sixteen instructions of nine scattered
		// lanes -- the same 144 atomics took 23 ms instead of 4.7: the counters' five cache lines saw sixteen times the requests.)
		const uint32_t c_t = tid < G.n1 ? L.hist[cur + tid] : 0u;
		const uint32_t g_t = c_t ? atomicAdd(&cnt[tid * G.n_sub], c_t) : 0u;
		const uint32_t c_mine = lane < per_wave ? L.hist[cur + mine] : 0u;
		scan_hist_per_wave(L, G.n1, cur);
		d_lo = (c_mine << 16) | (lane < per_wave ? L.lbase[mine] : 0u);
		if (SURE && __builtin_amdgcn_ballot_w64(zero_seen) == 0ull) { // (wave-uniform) every record of every lane has a bucket
			uint32_t at[C];
#pragma unroll
			for (int u = 0; u < C; u++) at[u] = L.lbase[bkt[u] >> 16];
#pragma unroll
			for (int u = 0; u < C; u++) L.stage[at[u] + (bkt[u] & 0xFFFFu)] = rec[u];
		} else {
#pragma unroll
			for (int u = 0; u < C; u++)
				if ((bkt[u] >> 16) < (uint32_t)kL1MaxB) {
					if constexpr (REC32) stage32()[L.lbase[bkt[u] >> 16] + (bkt[u] & 0xFFFFu)] = (uint32_t)rec[u];
					else L.stage[L.lbase[bkt[u] >> 16] + (bkt[u] & 0xFFFFu)] = rec[u];
				}
		}
		if (tid < G.n1) gbase[tid] = g_t;
		open_next(cur ^ kHist2);
		lds_barrier(); // (E) the tile is staged, the next one's packed words and cleared histogram are in place
		d_gb = lane < per_wave ? gbase[mine] : 0u;
		cur ^= kHist2;
	}
};

// RAGGED: the reads are NOT all L long.  Every read still gets Q = ceil(W_max / C) lanes (W_max from the
// longest read of the batch, L holds its length), a read's own offset and length come from `offsets`, and
// the lanes past a shorter read's last window stay empty.  Worth it when most reads have (nearly) the full
// length -- the host compares n_reads * Q * C lane slots with the n_bases positions of the flat kernel.
// LIN: the linear form (C = 8 or 12 windows per lane, ScatterLdsLin<C>, l1_scatter_tail_linear) for many level-1 buckets
// REG: regular tiles (see UniformGeom) -- the per-tile address arithmetic (64-bit read offsets, divisions by Q, guarded loads)
// collapses to a few 32-bit operations; the host launches this form over the whole tiles of a batch and the general form over
// the reads that are left
// PACKED (regular tiles only; the other forms test rb.packed at run time): the batch came 2-bit packed -- a tile is tile_blocks
// WORDS, one or two per lane, and nothing is packed on the way into LDS
// FULL (regular tiles): the reads fill their lanes exactly (W = Q C: every window of every lane is valid); without it the last lane of
// a read holds fewer than C windows (151-base reads at k = 31: 121 = 7 * 16 + 9) and the positions test their validity
// K17 (the general equal-length form): k >= 17 -- the launch may take the pipelined tile loop with the 32-bit rolls (regular tiles
// always have it)
template <int DBG = 0, int WIDE_D = 0, int C = 16, bool RAGGED = false, bool LIN = false, bool REG = false, bool PACKED = false, bool FULL = true,
          bool K17 = REG>
__global__ __launch_bounds__(kL1Threads) void k_extract_scatter_uniform(ReadBatch rb, UniformGeom U, const uint64_t *__restrict__ offsets,
                                                                         PartGeom G, PartStore P, Counters *__restrict__ ctr)
{
	static_assert(!LIN || C == 8 || C == 12, "the linear form stages 8 or 12 records per thread");
	static_assert(!REG || (!RAGGED && !LIN), "regular tiles: equal-length reads, wave-per-bucket form");
	static_assert(!PACKED || REG, "the other forms test rb.packed at run time");
	static_assert(FULL || (REG && DBG == 0), "partly filled lanes: the pipelined regular tiles");
	static_assert(K17 || !REG, "regular tiles: k >= 17");
	using ULds = typename std::conditional<LIN, UniformLdsLin<LIN ? C : 8>, UniformLds>::type;
	using SLds = typename std::conditional<LIN, ScatterLdsLin<LIN ? C : 8>, ScatterLds>::type;
	extern __shared__ __align__(16) unsigned char lds_raw[];
	ULds &UL = *reinterpret_cast<ULds *>(lds_raw);
	SLds &L = UL.s;
	const uint64_t n_tiles = (U.n_lanes + kL1Threads - 1) / kL1Threads;
	const uint32_t k = (uint32_t)rb.k;
	const uint64_t head_mask = (k < 32u) ? ((1ull << (2u * k)) - 1ull) : ~0ull;
	const uint32_t rc_shift = 2u * k - 2u;
	const uint32_t rel_mask = (1u << G.r) - 1u, q_shift = G.r + 6u;

	// (read, chunk) of a tile's first lane; advanced by the grid stride without dividing again
	const uint64_t stride_lanes = (uint64_t)gridDim.x * kL1Threads;
	const uint64_t stride_r = stride_lanes / U.Q;
	const uint32_t stride_c = (uint32_t)(stride_lanes - stride_r * U.Q);
	uint64_t r0 = ((uint64_t)blockIdx.x * kL1Threads) / U.Q;
	uint32_t c0 = (uint32_t)((uint64_t)blockIdx.x * kL1Threads - r0 * U.Q);

	auto read_start = [&](uint64_t r) -> uint64_t { return RAGGED ? offsets[r] : r * U.L; };
	// everything a lane needs of a tile, requested one tile ahead: its 16-byte blocks of the tile's byte
	// range (block t and block t + 1024), where that range starts, and the lane's own read
	using Block = typename std::conditional<PACKED, uint32_t, uint4>::type; // what a lane holds of a 16-base block
	struct RawU {
		Block a, b;
		uint64_t B0;       // first byte of the tile's range (16-aligned)
		uint64_t p;        // flat position of the lane's first window
		uint32_t n_blocks; // 16-byte blocks of the range
		uint32_t cc;       // the lane's chunk inside its read
		uint32_t W;        // windows of the lane's read (0: lane beyond the batch or read shorter than k)
	};
	auto fetch = [&](uint64_t tile, uint64_t rr, uint32_t c_first) {
		RawU raw;
		if constexpr (PACKED) raw.a = raw.b = 0u;
		else raw.a = raw.b = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
		raw.B0 = raw.p = 0;
		raw.n_blocks = raw.cc = raw.W = 0;
		if (tile >= n_tiles) return raw;
		const uint32_t t = fresh_tid();
		if constexpr (REG) { // every tile: kL1Threads / Q whole reads from a 16-byte boundary, all of them inside the buffer
			raw.B0 = tile * ((uint64_t)U.tile_blocks * 16u);
			raw.n_blocks = U.tile_blocks;
			if constexpr (PACKED) {
				const uint32_t *words = rb.packed + (raw.B0 >> 4);
				if (t < raw.n_blocks) raw.a = words[t];
				if (t + kL1Threads < raw.n_blocks) raw.b = words[t + kL1Threads];
			} else {
				const uint4 *blocks = reinterpret_cast<const uint4 *>(rb.bases + raw.B0);
				if (t < raw.n_blocks) raw.a = blocks[t];
				if (t + kL1Threads < raw.n_blocks) raw.b = blocks[t + kL1Threads];
			}
			raw.cc = t & (U.Q - 1u);
			raw.p = raw.B0 + (t >> U.lq) * U.L + (uint32_t)C * raw.cc;
			raw.W = U.W;
			return raw;
		}
		// the tile's byte range: from one base before its first lane's first window to the end of its last lane's window
		// RAGGED: a tile may start in the empty tail lanes of a read shorter than C * c_first; the range then
		// starts at the next read (the first live lane's window), never past it
		const uint64_t p_first = RAGGED ? min(offsets[rr] + (uint32_t)C * c_first, offsets[rr + 1]) : read_start(rr) + (uint32_t)C * c_first;
		raw.B0 = (p_first ? p_first - 1u : 0u) & ~15ull;
		const uint64_t lane_last = min((tile + 1u) * kL1Threads, U.n_lanes) - 1u;
		const uint32_t xl = c_first + (uint32_t)(lane_last - tile * kL1Threads);
		const uint32_t drl = (xl * U.qmagic) >> 22;
		uint64_t end = read_start(rr + drl) + (uint32_t)C * (xl - drl * U.Q) + (uint32_t)C + k + 2u;
		end = min(end, (rb.n_bases + 15u) & ~15ull);
		raw.n_blocks = end > raw.B0 ? min((uint32_t)((end - raw.B0 + 15u) >> 4), (uint32_t)kPkWords) : 0u;
		if constexpr (!PACKED) {
			if (rb.packed) { // (wave-uniform) .x holds the packed word
				if (t < raw.n_blocks) raw.a.x = packed_word(rb, (raw.B0 >> 4) + t);
				if (t + kL1Threads < raw.n_blocks) raw.b.x = packed_word(rb, (raw.B0 >> 4) + t + kL1Threads);
			} else {
				if (t < raw.n_blocks) raw.a = load_ascii16(rb.bases, rb.n_bases, (raw.B0 >> 4) + t);
				if (t + kL1Threads < raw.n_blocks) raw.b = load_ascii16(rb.bases, rb.n_bases, (raw.B0 >> 4) + t + kL1Threads);
			}
		}
		// this lane
		const uint32_t x = c_first + t;
		const uint32_t dr = (x * U.qmagic) >> 22;
		raw.cc = x - dr * U.Q;
		if (tile * kL1Threads + t < U.n_lanes) {
			const uint64_t r = rr + dr;
			const uint64_t s = read_start(r);
			const uint64_t len = RAGGED ? offsets[r + 1] - s : (uint64_t)U.L;
			raw.p = s + (uint32_t)C * raw.cc;
			raw.W = len >= k ? (uint32_t)(len - k + 1u) : 0u;
		}
		return raw;
	};

	RawU raw = fetch(blockIdx.x, r0, c0);
	// the tile's bytes (block tid, block tid + 1024) go into LDS packed 16 bases per word, the histogram is cleared
	auto open_tile = [&](const RawU &rw, uint32_t hist_off = 0u) { // hist_off: the histogram to clear, in words from L.hist (the pipelined form has two)
		const uint32_t tid = fresh_tid();
		if constexpr (PACKED) {
			if (tid < rw.n_blocks) UL.pk[tid] = rw.a;
			if (tid + kL1Threads < rw.n_blocks) UL.pk[tid + kL1Threads] = rw.b;
		} else if (rb.packed) {
			if (tid < rw.n_blocks) UL.pk[tid] = rw.a.x;
			if (tid + kL1Threads < rw.n_blocks) UL.pk[tid + kL1Threads] = rw.b.x;
		} else {
			if (tid < rw.n_blocks) UL.pk[tid] = pack16_ascii(rw.a, rb.other_seen);
			if (tid + kL1Threads < rw.n_blocks) UL.pk[tid + kL1Threads] = pack16_ascii(rw.b, rb.other_seen);
		}
#pragma unroll
		for (int j = 0; j < SLds::kBpt; j++) L.hist[hist_off + SLds::kBpt * tid + j] = 0;
	};
	// a lane's window of 16 positions out of the tile's packed words
	auto decode = [&](const RawU &raw) {
		const uint64_t p = raw.p;                        // flat position of the lane's first window
		// the packed stream starts one base earlier (left neighbour) -- except at position 0, and (regular tiles, whose byte
		// range starts ON a read start) for a read's first chunk, whose first window has no left neighbour anyway
		const bool no_prev = REG ? raw.cc == 0u : p == 0u;
		const uint64_t s0 = no_prev ? p : p - 1u;
		const uint32_t first_w = (uint32_t)C * raw.cc;   // index of the lane's first window inside its read
		const bool live = first_w < raw.W;               // (raw.W == 0 for lanes beyond the batch)
		Chunk16 c;
		const uint32_t rel = live ? (uint32_t)(s0 - raw.B0) : 0u;
		const uint32_t d = min(rel >> 4, (uint32_t)kPkWords - 5u), sh = 2u * (rel & 15u);
		const uint32_t x0 = UL.pk[d], x1 = UL.pk[d + 1], x2 = UL.pk[d + 2], x3 = UL.pk[d + 3], x4 = UL.pk[d + 4];
		const uint32_t X0 = funnel_left(x0, x1, sh), X1 = funnel_left(x1, x2, sh), X2 = funnel_left(x2, x3, sh), X3 = funnel_left(x3, x4, sh);
		// stream Y starts at position p (X starts at p - 1 unless p == 0)
		const uint32_t adv = no_prev ? 0u : 2u;
		const uint32_t Y0 = funnel_left(X0, X1, adv), Y1 = funnel_left(X1, X2, adv), Y2 = funnel_left(X2, X3, adv), Y3 = X3 << adv;
		c.lw = no_prev ? (X0 >> 2) : X0; // bases p-1 .. p+14 (the base before position 0 does not exist: has_l excludes it)
		c.kbit = ((((uint64_t)Y0 << 32) | Y1)) >> (64u - 2u * k);
		c.rc = revcomp_kbit(c.kbit, (int)k);
		const uint32_t widx = k >> 4, wsh = 2u * (k & 15u); // bases p+k .. p+k+15 (wave-uniform selection)
		const uint32_t ya = widx == 0u ? Y0 : (widx == 1u ? Y1 : Y2), yb = widx == 0u ? Y1 : (widx == 1u ? Y2 : Y3);
		c.nb = funnel_left(ya, yb, wsh);
		const uint32_t nv = live ? min((uint32_t)C, raw.W - first_w) : 0u;
		const uint32_t nr = (live && first_w + 1u < raw.W) ? min((uint32_t)C, raw.W - 1u - first_w) : 0u;
		c.valid = (1u << nv) - 1u;
		c.has_r = (1u << nr) - 1u;            // the read's last window has no right neighbour
		c.has_l = raw.cc ? 0xFFFFu : 0xFFFEu; // its first window no left one
		return c;
	};
	// PIPELINED tile loop (L1Pipe): regular tiles, every batch of equal-length or mostly full-length reads from k = 17 on, and
	// a KFREQ handle with direct blocks (WIDE_D == 3, equal-length reads of any length: 32-bit records).
	constexpr bool kRec32 = WIDE_D == 3;
	constexpr bool kPipe = !LIN && DBG == 0 && (REG || K17 || (kRec32 && !RAGGED));
	if constexpr (kPipe) {
		L1Pipe<kRec32> pp(L, UL.gbase, G, P, ctr);
		const uint32_t tid = pp.tid;
		if (blockIdx.x < n_tiles) {
			open_tile(raw, 0u);
			lds_barrier();
		}
		for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
			const Chunk16 c = decode(raw);
			uint32_t bkt[16];
			uint64_t rec[16];
			uint64_t slow = 0ull;
			const bool zero_seen = l1_positions<WIDE_D, C, SLds, REG && FULL, K17, true>(L, G, c, tid, head_mask, rc_shift, rel_mask, q_shift, bkt, rec, pp.cur,
			                                                                    [&](uint32_t i) { pp.copy_run(i, slow); });
			pp.copy_rest((uint32_t)C, slow); // (more than 16 C buckets; the full ones)
			if (zero_seen) l1_key0_from_chunk(c, head_mask, rc_shift, G.kf, ctr);
			if constexpr (!REG) { // (regular tiles: fetch works from the tile index alone)
				r0 += stride_r;
				c0 += stride_c;
				if (c0 >= U.Q) { c0 -= U.Q; r0 += 1u; }
			}
			const RawU nxt = fetch(tile + gridDim.x, r0, c0);
			lds_barrier(); // (C) every rank of this tile has been taken, every wave has copied its runs of the tile before
			pp.template tail<C, (REG && FULL)>(rec, bkt, zero_seen, [&](uint32_t hist_next) { open_tile(nxt, hist_next); });
			raw = nxt;
		}
		pp.copy_rest(0u, 0ull); // the runs of the workgroup's last tile
		return;
	}
	// The LINEAR form pipelined (k >= 17, a graph handle): as L1Pipe, with the linear copy-out -- element u of the tile before (staged
	// record u * 1024 + tid, its bucket from the tag array, its place = desc[bucket] + index) goes out in front of position u; ONE
	// histogram, cleared by the thread that owns the entry right after it has read it; the classic scan (four barriers per tile).
	if constexpr (LIN && K17 && DBG == 0) {
		const uint32_t tid = fresh_tid();
		const uint32_t sub = blockIdx.x % G.n_sub;
		uint64_t *const out = P.l1 + (uint64_t)sub * G.cap1; // bucket b lives at out + b * n_sub * cap1
		uint32_t *const cnt = P.cnt1 + sub;
		uint32_t total_prev = 0u; // records of the tile before, sorted in the stage buffer
		static_assert(SLds::kBpt == 1, "thread b owns bucket b");
		auto copy_elem = [&](uint32_t u, uint32_t &slow) {
			const uint32_t p = u * kL1Threads + tid;
			if (p >= total_prev) return;
			const uint64_t rcd = L.stage[p];
			const uint32_t b = L.bucket_of[p];
			const uint64_t off = (uint64_t)(uint32_t)(L.desc[b] + p); // desc = reserved place - first staged index
			if (off < G.cap1) out[(uint64_t)b * G.n_sub * G.cap1 + off] = rcd;
			else slow |= 1u << u; // the bucket is full: behind the positions
		};
		auto copy_slow = [&](uint32_t slow) {
			for (; slow; slow &= slow - 1u) {
				const uint32_t p = (uint32_t)__builtin_ctz(slow) * kL1Threads + tid;
				const uint64_t rcd = L.stage[p];
				push_overflow(P, record_key(rcd, L.bucket_of[p], G), (uint32_t)(rcd >> 3) & 7u, (uint32_t)rcd & 7u, ctr);
			}
		};
		if (blockIdx.x < n_tiles) {
			open_tile(raw);
			if (tid < 64u) L.hist[kL1MaxB + tid] = 0u;
			lds_barrier();
		}
		for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
			const Chunk16 c = decode(raw);
			uint32_t bkt[16];
			uint64_t rec[16];
			uint32_t slow = 0u;
			const bool zero_seen = l1_positions<WIDE_D, C, SLds, false, true, true>(L, G, c, tid, head_mask, rc_shift, rel_mask, q_shift, bkt, rec, 0u,
			                                                                     [&](uint32_t i) { copy_elem(i, slow); });
			if (slow) copy_slow(slow);
			if (zero_seen) l1_key0_from_chunk(c, head_mask, rc_shift, G.kf, ctr);
			r0 += stride_r;
			c0 += stride_c;
			if (c0 >= U.Q) { c0 -= U.Q; r0 += 1u; }
			const RawU nxt = fetch(tile + gridDim.x, r0, c0);
			lds_barrier(); // (C) every rank taken; the stage buffer, the tags and the descriptors of the tile before read out; the packed words decoded
			const uint32_t c_t = L.hist[tid];
			L.hist[tid] = 0u; // (for the next tile; nobody else reads this entry)
			if (tid < 64u) L.hist[kL1MaxB + tid] = 0u; // (the bins of the positions without a record)
			const uint32_t g_t = (tid < G.n1 && c_t) ? atomicAdd(&cnt[tid * G.n_sub], c_t) : 0u;
			uint32_t inc = c_t;
			{
				const uint32_t lane = tid & 63u;
#pragma unroll
				for (int off = 1; off < 64; off <<= 1) {
					const uint32_t n = __shfl_up(inc, off, 64);
					if ((int)lane >= off) inc += n;
				}
				if (lane == 63u) L.wave_tot[tid >> 6] = inc;
			}
			lds_barrier();
			uint32_t run = inc - c_t, all = 0;
#pragma unroll
			for (uint32_t w = 0; w < (uint32_t)kL1Threads / 64u; w++) {
				const uint32_t wt = L.wave_tot[w];
				run += (w < (tid >> 6)) ? wt : 0u;
				all += wt;
			}
			L.lbase[tid] = run;
			lds_barrier(); // every bucket's first staged index is known; `all` = the tile's records with a bucket
#pragma unroll
			for (int u = 0; u < C; u++) {
				const uint32_t b = bkt[u] >> 16;
				if (b < (uint32_t)kL1MaxB) {
					const uint32_t at = L.lbase[b] + (bkt[u] & 0xFFFFu);
					L.stage[at] = rec[u];
					L.bucket_of[at] = (uint16_t)b;
				}
			}
			L.desc[tid] = g_t - run;
			open_tile(nxt); // (its histogram clearing repeats what the owners did above)
			lds_barrier(); // (E) the tile is staged, the next one's packed words are in place
			total_prev = all;
			raw = nxt;
		}
		{ // the workgroup's last tile
			uint32_t slow = 0u;
#pragma unroll 1
			for (uint32_t u = 0; u < (uint32_t)C; u++) copy_elem(u, slow);
			if (slow) copy_slow(slow);
		}
		return;
	}
	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint32_t tid = fresh_tid();
		uint32_t bkt[16];
		open_tile(raw);
		lds_barrier();
		const Chunk16 c = decode(raw);
		const bool zero_seen = l1_positions<WIDE_D, C, SLds, REG>(L, G, c, tid, head_mask, rc_shift, rel_mask, q_shift, bkt);
		if (zero_seen) l1_key0_from_chunk(c, head_mask, rc_shift, G.kf, ctr);
		// next tile
		r0 += stride_r;
		c0 += stride_c;
		if (c0 >= U.Q) { c0 -= U.Q; r0 += 1u; }
		if (DBG == 1) {
			uint64_t xx = 0;
#pragma unroll
			for (int u = 0; u < 16; u++) xx ^= L.stage[u * kL1Threads + tid] + bkt[u];
			if (xx == 0x1234567u) P.l1[threadIdx.x] = xx;
			lds_barrier();
			raw = fetch(tile + gridDim.x, r0, c0);
			continue;
		}
		const RawU nxt = fetch(tile + gridDim.x, r0, c0);
		if constexpr (LIN) l1_scatter_tail_linear<DBG, C, (WIDE_D >= 2)>(L, G, P, ctr, tid, bkt);
		else l1_scatter_tail<DBG, (WIDE_D >= 2), (REG ? C : 0)>(L, G, P, ctr, tid, bkt, REG && !zero_seen);
		raw = nxt;
	}
}

// ---- level 1 for reads of ANY lengths: lanes follow a per-read lane prefix --------------------------------------------------
// What debruijn_contig is really fed are quality-trimmed, corrected reads of mixed lengths (the reference's own recorded run: mean
// 243 of 250, test/02.build_contig/Ecoli_corrected_reads.contig.log:437-438).  The equal-length kernel above maps lanes to chunks of
// VALID windows by arithmetic; its RAGGED form gives every read the lane count of the longest one.  Here read r gets exactly
// Q_r = ceil(W_r / C) lanes, W_r = max(0, min(len_r, maxReadLen) - k + 1) windows (trimmed reads included), and a short pre-pass
// turns the offsets into what the tiles need:
//   k_prefix_count   v_r = (Q_r > 0) << 32 | Q_r summed per block of 4096 reads
//   k_prefix_blocks  exclusive scan of the block sums (one workgroup), totals
//   k_prefix_emit    per read with windows: ReadLanes {first base, first lane, windows} into a COMPACT list; tile_first[T] = the entry
//                    that covers lane 1024 * T
//   k_prefix_tiles   per tile: the 16-aligned start and the packed words of its byte range, its first entry
// The level-1 kernel then finds a lane's read with a 1024-bit "an entry starts at this lane" bitmap in LDS (popcount prefix), takes the
// read's start and windows from two small LDS arrays, and funnels its window out of the tile's packed words exactly like the
// equal-length kernel.  The input is always 2-bit PACKED (ASCII batches are packed first, k_pack_bases).  A tile whose byte range
// does not fit the LDS image (reads without a window in between, trimmed tails of long reads) reads its words from global memory.
struct ReadLanes {
	uint64_t start;   // first base of the read in the batch
	uint32_t lane0;   // its first lane (global lane index of the batch)
	uint32_t W;       // its windows
};
struct PrefixTile {
	uint64_t B0;       // first base of the tile's byte range (a multiple of 16)
	uint64_t base0;    // start of the tile's first entry: the LDS entries hold starts relative to it
	uint32_t n_words;  // packed words of the range; 0 = does not fit the LDS image: lanes read global memory
	uint32_t e0;       // first entry
	uint32_t cc0;      // chunk of the tile's first lane inside its read (the read may have begun in an earlier tile)
	uint32_t pad;
};
struct PrefixTotals {
	unsigned long long n_lanes, n_entries;
};
constexpr int kPrefixBlock = 1024, kPrefixItems = 4; // reads per workgroup of the pre-pass: 4096
constexpr uint32_t kPrefixMaxW = (1u << 22) - 2u;    // windows of one read the LDS entry format can say (longer reads: the flat kernel)

__device__ __forceinline__ uint32_t prefix_windows(const uint64_t *__restrict__ offsets, uint64_t r, uint32_t k, uint32_t max_read_len)
{
	const uint64_t len = offsets[r + 1] - offsets[r], rl = len > max_read_len ? max_read_len : len;
	return rl >= k ? (uint32_t)(rl - k + 1u) : 0u;
}

template <int C>
__global__ __launch_bounds__(kPrefixBlock) void k_prefix_count(const uint64_t *__restrict__ offsets, uint64_t n_reads, uint32_t k, uint32_t max_read_len,
                                                               unsigned long long *__restrict__ bsum)
{
	__shared__ unsigned long long red[kPrefixBlock / 64];
	const uint64_t first = ((uint64_t)blockIdx.x * kPrefixBlock + threadIdx.x) * kPrefixItems;
	unsigned long long v = 0;
#pragma unroll
	for (int j = 0; j < kPrefixItems; j++)
		if (first + j < n_reads) {
			const uint32_t W = prefix_windows(offsets, first + j, k, max_read_len), Q = (W + (uint32_t)C - 1u) / (uint32_t)C;
			v += (unsigned long long)Q | ((unsigned long long)(Q ? 1u : 0u) << 32);
		}
	const unsigned long long s = block_sum_n<kPrefixBlock>(v, red);
	if (threadIdx.x == 0) bsum[blockIdx.x] = s;
}

// exclusive scan of the block sums in place (n_blocks may exceed one workgroup: chunks with a carry); totals
__global__ __launch_bounds__(kPrefixBlock) void k_prefix_blocks(unsigned long long *__restrict__ bsum, uint32_t n_blocks, PrefixTotals *__restrict__ tot)
{
	__shared__ unsigned long long wave_tot[kPrefixBlock / 64];
	__shared__ unsigned long long carry;
	const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
	if (t == 0) carry = 0ull;
	__syncthreads();
	for (uint32_t base = 0; base < n_blocks; base += kPrefixBlock) {
		const uint32_t i = base + (uint32_t)t;
		const unsigned long long v = i < n_blocks ? bsum[i] : 0ull;
		unsigned long long inc = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const unsigned long long n = __shfl_up(inc, off, 64);
			if (lane >= off) inc += n;
		}
		if (lane == 63) wave_tot[wave] = inc;
		__syncthreads();
		unsigned long long before = carry;
		for (int w = 0; w < wave; w++) before += wave_tot[w];
		if (i < n_blocks) bsum[i] = before + inc - v;
		__syncthreads();
		if (t == kPrefixBlock - 1) carry = before + inc;
		__syncthreads();
	}
	if (t == 0) {
		tot->n_lanes = carry & 0xFFFFFFFFull;
		tot->n_entries = carry >> 32;
	}
}

template <int C>
__global__ __launch_bounds__(kPrefixBlock) void k_prefix_emit(const uint64_t *__restrict__ offsets, uint64_t n_reads, uint32_t k, uint32_t max_read_len,
                                                              const unsigned long long *__restrict__ bsum, ReadLanes *__restrict__ ent,
                                                              uint32_t *__restrict__ tile_first, Counters *__restrict__ ctr)
{
	__shared__ unsigned long long wave_tot[kPrefixBlock / 64];
	const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
	const uint64_t first = ((uint64_t)blockIdx.x * kPrefixBlock + threadIdx.x) * kPrefixItems;
	uint32_t W[kPrefixItems];
	unsigned long long v = 0;
#pragma unroll
	for (int j = 0; j < kPrefixItems; j++) {
		W[j] = first + j < n_reads ? prefix_windows(offsets, first + j, k, max_read_len) : 0u;
		const uint32_t Q = (W[j] + (uint32_t)C - 1u) / (uint32_t)C;
		v += (unsigned long long)Q | ((unsigned long long)(Q ? 1u : 0u) << 32);
	}
	unsigned long long inc = v;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const unsigned long long n = __shfl_up(inc, off, 64);
		if (lane >= off) inc += n;
	}
	if (lane == 63) wave_tot[wave] = inc;
	__syncthreads();
	unsigned long long run = bsum[blockIdx.x] + inc - v;
	for (int w = 0; w < wave; w++) run += wave_tot[w];
	bool too_long = false;
#pragma unroll
	for (int j = 0; j < kPrefixItems; j++) {
		const uint32_t Q = (W[j] + (uint32_t)C - 1u) / (uint32_t)C;
		if (Q) {
			const uint32_t lane0 = (uint32_t)run, e = (uint32_t)(run >> 32);
			ent[e] = ReadLanes{offsets[first + j], lane0, W[j]};
			too_long = too_long || W[j] > kPrefixMaxW;
			// the tiles whose first lane belongs to this read
			for (uint32_t T = (lane0 + (uint32_t)kL1Threads - 1u) / (uint32_t)kL1Threads; (uint64_t)T * kL1Threads < (uint64_t)lane0 + Q; T++) tile_first[T] = e;
			run += (unsigned long long)Q | (1ull << 32);
		}
	}
	if (too_long) atomicOr(&ctr->error, 4u); // (the host never sends such a batch here: reads of more than 4 M windows take the flat kernel)
}

template <int C>
__global__ __launch_bounds__(256) void k_prefix_tiles(const ReadLanes *__restrict__ ent, const uint32_t *__restrict__ tile_first,
                                                      const PrefixTotals *__restrict__ tot, uint32_t k, uint64_t n_bases, PrefixTile *__restrict__ tiles)
{
	const uint64_t n_lanes = tot->n_lanes, n_entries = tot->n_entries;
	const uint64_t n_tiles = (n_lanes + kL1Threads - 1) / kL1Threads;
	const uint64_t T = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (T >= n_tiles) return;
	const uint32_t e0 = tile_first[T];
	const ReadLanes E0 = ent[e0];
	const uint64_t lane_first = T * kL1Threads, lane_last = min((T + 1) * (uint64_t)kL1Threads, n_lanes) - 1u;
	uint32_t e1 = (uint32_t)n_entries - 1u;
	if (T + 1 < n_tiles) {
		e1 = tile_first[T + 1];
		if ((uint64_t)ent[e1].lane0 > lane_last) e1--;
	}
	const ReadLanes E1 = ent[e1];
	const uint32_t cc0 = (uint32_t)(lane_first - E0.lane0);
	const uint64_t p_first = E0.start + (uint64_t)C * cc0;
	PrefixTile M;
	M.B0 = (p_first ? p_first - 1u : 0u) & ~15ull;
	M.base0 = E0.start;
	uint64_t end = E1.start + (uint64_t)C * (uint32_t)(lane_last - E1.lane0) + (uint32_t)C + k + 2u;
	end = min(end, (n_bases + 15u) & ~15ull);
	const uint64_t words = end > M.B0 ? (end - M.B0 + 15u) >> 4 : 0u;
	// (the entries' starts are kept relative to base0 in 32 bits)
	M.n_words = (words <= (uint64_t)kPkWords - 8u && E1.start - E0.start < (1ull << 32)) ? (uint32_t)words : 0u;
	M.e0 = e0;
	M.cc0 = cc0;
	M.pad = 0u;
	tiles[T] = M;
}

struct PrefixLds {
	ScatterLds s;
	uint32_t pk[kPkWords];
	uint32_t ent_start[kL1Threads]; // start of the tile's i-th entry, relative to the tile's base0
	uint32_t meta[kL1Threads];      // its windows << 10 | the lane of the tile it begins at
	unsigned long long starts[2][kL1Threads / 64]; // bit l: an entry begins at lane l of the tile (two tiles in turn: the pipelined loop opens the next tile in the tail of this one)
};
static_assert(sizeof(PrefixLds) <= 160 * 1024, "one workgroup per CU");
// the pipelined loop's hand-over array lives behind the second histogram in the descriptor array: room for this many buckets
constexpr uint32_t kPrefixPipeMaxB = (uint32_t)(sizeof(ScatterLds::desc) / 4u) - (uint32_t)(kL1MaxB + 64);

template <int WIDE_D = 0, int C = 16, bool K17 = false> // K17: k >= 17 and n1 <= kPrefixPipeMaxB -- the 32-bit rolls of l1_positions and the pipelined tile loop (L1Pipe)
__global__ __launch_bounds__(kL1Threads) void k_extract_scatter_prefix(ReadBatch rb, const ReadLanes *__restrict__ ent, const PrefixTile *__restrict__ tiles,
                                                                        const PrefixTotals *__restrict__ tot, PartGeom G, PartStore P, Counters *__restrict__ ctr)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	PrefixLds &UL = *reinterpret_cast<PrefixLds *>(lds_raw);
	ScatterLds &L = UL.s;
	const uint64_t n_lanes = tot->n_lanes, n_entries = tot->n_entries;
	const uint64_t n_tiles = (n_lanes + kL1Threads - 1) / kL1Threads;
	const uint64_t n_words_total = (rb.n_bases + 15u) >> 4;
	const uint32_t k = (uint32_t)rb.k;
	const uint64_t head_mask = (k < 32u) ? ((1ull << (2u * k)) - 1ull) : ~0ull;
	const uint32_t rc_shift = 2u * k - 2u;
	const uint32_t rel_mask = (1u << G.r) - 1u, q_shift = G.r + 6u;

	struct RawP {
		uint32_t a, b;       // packed words tid and tid + 1024 of the tile's range
		uint32_t start_rel;  // the tile's tid-th entry: its start relative to the tile's base0,
		uint32_t meta;       // its windows << 10 | the lane of the tile it begins at; ~0: no such entry
	};
	auto tile_meta = [&](uint64_t tile) {
		PrefixTile M{};
		if (tile < n_tiles) M = tiles[tile]; // (uniform address: a scalar load)
		return M;
	};
	auto fetch = [&](uint64_t tile, const PrefixTile &M) {
		RawP raw;
		raw.a = raw.b = raw.start_rel = 0u;
		raw.meta = 0xFFFFFFFFu;
		if (tile >= n_tiles) return raw;
		const uint32_t t = fresh_tid();
		const uint64_t w0 = M.B0 >> 4;
		if (t < M.n_words && w0 + t < n_words_total) raw.a = rb.packed[w0 + t];
		if (t + kL1Threads < M.n_words && w0 + t + kL1Threads < n_words_total) raw.b = rb.packed[w0 + t + kL1Threads];
		if ((uint64_t)M.e0 + t < n_entries) {
			const ReadLanes E = ent[(uint64_t)M.e0 + t];
			const uint64_t lane_first = tile * kL1Threads, lane_end = min(lane_first + (uint64_t)kL1Threads, n_lanes);
			if ((uint64_t)E.lane0 < lane_end) { // (only the tile's first entry can begin before the tile)
				const uint32_t at = (uint64_t)E.lane0 > lane_first ? (uint32_t)(E.lane0 - lane_first) : 0u;
				raw.start_rel = (uint32_t)(E.start - M.base0);
				raw.meta = (E.W << 10) | at;
			}
		}
		return raw;
	};
	// a tile is opened: its packed words, its entries (bitmap `sb`, start and meta of entry tid), the histogram at hist_off cleared
	auto open = [&](const RawP &rw, const PrefixTile &Mx, uint32_t sb, uint32_t hist_off) {
		const uint32_t tid = fresh_tid();
		if (tid < Mx.n_words) UL.pk[tid] = rw.a;
		if (tid + kL1Threads < Mx.n_words) UL.pk[tid + kL1Threads] = rw.b;
		if (rw.meta != 0xFFFFFFFFu) {
			const uint32_t at = rw.meta & 1023u;
			atomicOr(&UL.starts[sb][at >> 6], 1ull << (at & 63u));
			UL.ent_start[tid] = rw.start_rel;
			UL.meta[tid] = rw.meta;
		}
#pragma unroll
		for (int j = 0; j < ScatterLds::kBpt; j++) L.hist[hist_off + ScatterLds::kBpt * tid + j] = 0;
	};
	// the lane's read -- the number of entries that begin at or before this lane -- and its window of C positions
	auto decode = [&](uint64_t tile, const PrefixTile &Mx, uint32_t sb) {
		const uint32_t tid = fresh_tid();
		const uint64_t lane_first = tile * kL1Threads, lane_end = min(lane_first + (uint64_t)kL1Threads, n_lanes);
		const bool live = lane_first + tid < lane_end;
		uint32_t idx;
		{
			const uint32_t lane = tid & 63u, wave = tid >> 6;
			const unsigned long long mine = UL.starts[sb][wave];
			uint32_t before = (lane < (uint32_t)(kL1Threads / 64) && lane < wave) ? (uint32_t)__popcll(UL.starts[sb][lane]) : 0u;
#pragma unroll
			for (int off = 8; off > 0; off >>= 1) before += __shfl_xor(before, off, 64); // lanes 0..15 hold the per-wave counts: sum over them
			before = __builtin_amdgcn_readfirstlane(before);
			idx = before + (uint32_t)__popcll(mine & ((2ull << lane) - 1ull)) - 1u; // (bit 0 of the tile is always set: idx >= 0 for live lanes)
		}
		uint64_t p = 0;        // flat position of the lane's first window
		uint32_t cc = 0, W = 0;
		if (live) {
			const uint32_t meta = UL.meta[idx];
			W = meta >> 10;
			cc = tid - (meta & 1023u) + (idx == 0u ? Mx.cc0 : 0u);
			p = Mx.base0 + UL.ent_start[idx] + (uint64_t)C * cc;
		}
		const bool no_prev = p == 0u;
		const uint64_t s0 = no_prev ? p : p - 1u;
		const uint32_t first_w = (uint32_t)C * cc;
		Chunk16 c;
		uint32_t x0, x1, x2, x3, x4;
		const uint32_t sh = 2u * ((uint32_t)s0 & 15u);
		if (Mx.n_words) { // (tile-uniform) the range sits in LDS
			const uint32_t rel = live ? (uint32_t)(s0 - Mx.B0) : 0u;
			const uint32_t d = min(rel >> 4, (uint32_t)kPkWords - 5u);
			x0 = UL.pk[d]; x1 = UL.pk[d + 1]; x2 = UL.pk[d + 2]; x3 = UL.pk[d + 3]; x4 = UL.pk[d + 4];
		} else {         // a range too long for the image: every lane reads its five words from global memory
			const uint64_t wi = live ? s0 >> 4 : 0ull, last = n_words_total ? n_words_total - 1u : 0u;
			x0 = rb.packed[min(wi, last)]; x1 = rb.packed[min(wi + 1u, last)]; x2 = rb.packed[min(wi + 2u, last)];
			x3 = rb.packed[min(wi + 3u, last)]; x4 = rb.packed[min(wi + 4u, last)];
		}
		const uint32_t X0 = funnel_left(x0, x1, sh), X1 = funnel_left(x1, x2, sh), X2 = funnel_left(x2, x3, sh), X3 = funnel_left(x3, x4, sh);
		const uint32_t adv = no_prev ? 0u : 2u;
		const uint32_t Y0 = funnel_left(X0, X1, adv), Y1 = funnel_left(X1, X2, adv), Y2 = funnel_left(X2, X3, adv), Y3 = X3 << adv;
		c.lw = no_prev ? (X0 >> 2) : X0;
		c.kbit = ((((uint64_t)Y0 << 32) | Y1)) >> (64u - 2u * k);
		c.rc = revcomp_kbit(c.kbit, (int)k);
		const uint32_t widx = k >> 4, wsh = 2u * (k & 15u);
		const uint32_t ya = widx == 0u ? Y0 : (widx == 1u ? Y1 : Y2), yb = widx == 0u ? Y1 : (widx == 1u ? Y2 : Y3);
		c.nb = funnel_left(ya, yb, wsh);
		const uint32_t nv = (live && first_w < W) ? min((uint32_t)C, W - first_w) : 0u;
		const uint32_t nr = (live && first_w + 1u < W) ? min((uint32_t)C, W - 1u - first_w) : 0u;
		c.valid = (1u << nv) - 1u;
		c.has_r = (1u << nr) - 1u;            // the read's last window (after trimming) has no right neighbour
		c.has_l = cc ? 0xFFFFu : 0xFFFEu;     // its first window no left one
		return c;
	};

	if (fresh_tid() < (uint32_t)(kL1Threads / 64)) UL.starts[0][fresh_tid()] = UL.starts[1][fresh_tid()] = 0ull;
	PrefixTile M = tile_meta(blockIdx.x), M_next = tile_meta((uint64_t)blockIdx.x + gridDim.x);
	RawP raw = fetch(blockIdx.x, M);
	lds_barrier();
	if constexpr (K17) { // the pipelined tile loop (L1Pipe)
		L1Pipe<false> pp(L, reinterpret_cast<uint32_t *>(L.desc) + (kL1MaxB + 64), G, P, ctr);
		const uint32_t tid = pp.tid;
		uint32_t sb = 0u; // the bitmap of the current tile
		if (blockIdx.x < n_tiles) {
			open(raw, M, 0u, 0u);
			lds_barrier();
		}
		for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
			const Chunk16 c = decode(tile, M, sb);
			uint32_t bkt[16];
			uint64_t rec[16];
			uint64_t slow = 0ull;
			const bool zero_seen = l1_positions<WIDE_D, C, ScatterLds, false, true, true>(L, G, c, tid, head_mask, rc_shift, rel_mask, q_shift, bkt, rec, pp.cur,
			                                                                           [&](uint32_t i) { pp.copy_run(i, slow); });
			pp.copy_rest((uint32_t)C, slow);
			if (zero_seen) l1_key0_from_chunk(c, head_mask, rc_shift, G.kf, ctr);
			// next tile: its words and entries travel across the barrier and the tail; the meta data of the tile after it as well
			const uint64_t t1 = tile + gridDim.x, t2 = t1 + gridDim.x;
			const RawP nxt = fetch(t1, M_next);
			const PrefixTile M_after = tile_meta(t2);
			lds_barrier(); // (C) every rank taken, the runs of the tile before copied, and every lane has read its bitmap, entries and words
			pp.template tail<C, false>(rec, bkt, zero_seen, [&](uint32_t hist_next) {
				if (tid < (uint32_t)(kL1Threads / 64)) UL.starts[sb][tid] = 0ull; // (this tile's bitmap: set again two tiles on)
				open(nxt, M_next, sb ^ 1u, hist_next);
			});
			sb ^= 1u;
			raw = nxt;
			M = M_next;
			M_next = M_after;
		}
		pp.copy_rest(0u, 0ull); // the runs of the workgroup's last tile
		return;
	}
	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint32_t tid = fresh_tid();
		open(raw, M, 0u, 0u);
		uint32_t bkt[16];
		lds_barrier();
		const Chunk16 c = decode(tile, M, 0u);
		const bool zero_seen = l1_positions<WIDE_D, C, ScatterLds, false, K17>(L, G, c, tid, head_mask, rc_shift, rel_mask, q_shift, bkt);
		if (zero_seen) l1_key0_from_chunk(c, head_mask, rc_shift, G.kf, ctr);
		// next tile: its words and entries travel during this tile's scatter; the meta data of the tile after it as well
		const uint64_t t1 = tile + gridDim.x, t2 = t1 + gridDim.x;
		const RawP nxt = fetch(t1, M_next);
		const PrefixTile M_after = tile_meta(t2);
		lds_barrier(); // hist complete (as l1_scatter_tail begins) -- and every lane has read starts / ent_start / meta
		if (tid < (uint32_t)(kL1Threads / 64)) UL.starts[0][tid] = 0ull; // (for the next tile: set again only after that tile's threads passed the barriers below)
		{
			uint64_t rec[16];
#pragma unroll
			for (int u = 0; u < 16; u++) rec[u] = L.stage[u * kL1Threads + tid];
			uint32_t my_gbase[ScatterLds::kBpt];
			const uint32_t sub = blockIdx.x % G.n_sub;
			scatter_reserve_scan(L, G.n1, P.cnt1 + sub, my_gbase, G.n_sub);
			scatter_stage_copy<16, 0, false, (WIDE_D >= 2)>(L, rec, bkt, my_gbase, G.n1, P.l1 + (uint64_t)sub * G.cap1, G.cap1, 0u, true, G, P, ctr, G.n_sub);
		}
		raw = nxt;
		M = M_next;
		M_next = M_after;
	}
}

} // namespace dbgk
