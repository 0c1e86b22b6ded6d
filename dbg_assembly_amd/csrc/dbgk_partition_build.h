// dbgk_partition_build.h -- part of the PARTITION engine (dbgk_partition.h includes it, in this order: _l1, _l2, _build): the region build (k_build_regions) and the KFREQ block build (k_kf_build_blocks)
#pragma once

namespace dbgk {

// ---- build: one workgroup per 4096-slot region ---------------------------------------------------
// (Measured alternatives, round 1: 512 threads per workgroup 9.7 ms; 1024 threads, two workgroups
// per CU -- this form -- 8.2 ms; a persistent one-workgroup-per-CU variant with 16-bit LDS counters
// and fire-and-forget ds_add_u64 instead of the CAS loops 9.7 ms: the kernel is bound by instruction
// issue in the divergent probe loops, ~1000 VALU + ~1200 SALU per wave per region, not by the
// counter updates.)
constexpr int kWalkQ = 64; // entries of a wave's walk queue (one record each): one flush of a full queue keeps every lane busy
struct BuildLds {
	unsigned long long ident[kRegionSlots + kSpillSlots]; // (record >> 6) + 1, 0 = empty
	unsigned long long links[kRegionSlots + kSpillSlots];
	unsigned long long red[kBuildThreads / 64];
	uint32_t next_region;
	uint32_t redo;                 // FAST build: some counter of the current region overflowed its byte, the region goes to the exact pass
	unsigned long long walkq[kBuildThreads / 64][kWalkQ]; // FAST build: per wave, the records whose home slot holds another key
};

// Regions the FAST build could not finish (a link counter passed 255, or the region and its spill area were full): rebuilt from
// their records by the exact form of the kernel (saturating LDS CAS) after all fast launches of the step.
struct RedoList {
	uint32_t *list;        // local region indices
	unsigned int *n;       // appended so far
	uint32_t cap;
};

// DBG (DBGK_DEBUG_BUILD, timing experiments, results are wrong): 1 = clear + load only, 2 = no emit,
// 3 = emit without recomputing the keys
// KF: KFREQ through this engine -- there is no node table; `table` is the direct-addressed 4^k byte
// table and an occupied LDS slot is emitted as counts[key] = its occurrence counter.
// INCR: the table already holds nodes (an earlier flush of a streaming build, dbgk_flush): before a region's
// records are inserted its 4096 table slots are loaded back into the LDS image.  A node whose home slot lies
// in the region gets the identity its records carry; a node that probed in from an earlier region (merged
// there by k_merge_spill) is a foreign blocker: it keeps its slot, takes no record of this region (its
// records go to its home region, run off that region's end again and are merged by k_merge_spill) and is
// left untouched by the emit.  KF + INCR: counts[key] += the occurrences of this flush, saturating.
// FAST: the insert keeps four records per thread in flight and never loops on a counter.  A slot is probed with ONE
// unconditional ds_cmpst (compare 0, swap in the identity: returns 0 = claimed, the identity = found, anything else = occupied
// by another key -- a read and a claim in one LDS round trip, four of them issued back to back), and the two observed neighbour
// counters are bumped with ONE fire-and-forget-style ds_add_rtn_u64 on the link word.  A plain add cannot saturate, so its
// return value is checked instead: if the byte it bumped already held 255 (kmerSet.cpp:253-273 stops there) the region is
// flagged, emits NOTHING (no table slots, no spill nodes, no counts, no overflow records) and is appended to `redo`; the exact
// form of this kernel (FAST = false: saturating CAS loops, FROM_LIST = true) rebuilds the flagged regions from their records
// afterwards.  Exact for any input; a region pays twice only when one of its k-mers has a neighbour seen more than 255 times.
template <int DBG = 0, bool KF = false, bool INCR = false, bool FAST = false, bool FROM_LIST = false>
__global__ __launch_bounds__(kBuildThreads, 8) void k_build_regions(PartGeom G, PartStore P, Node *__restrict__ table,
                                                                  Counters *__restrict__ ctr, uint32_t first_region, uint32_t n_regions,
                                                                  unsigned int *__restrict__ cursor, RedoList redo)
{
	static_assert(!(FAST && FROM_LIST), "the exact pass is what the list is for");
	extern __shared__ __align__(16) unsigned char lds_raw[];
	BuildLds &L = *reinterpret_cast<BuildLds *>(lds_raw);
	const int t = (int)fresh_tid();
	constexpr int kBatch = 4; // records per thread in one batch; one batch is inserted while the next is in flight
	constexpr uint32_t kNone = 0xFFFFFFFFu;

	// Persistent workgroups pull regions from a cursor (self-balancing whatever the residency: while the
	// level-2 kernel of the next bucket chunk shares the chip only one build workgroup fits a CU).  The
	// records are consumed as ONE stream of batches across regions: while batch i is inserted, batch
	// i+1 -- the next 4096 records of this region or the first ones of the next region -- is already
	// in flight, in the registers the previous batch has just vacated (the kernel must stay inside the
	// 64 VGPRs that two workgroups per CU allow).
	// (two neighbouring records per lane and load instruction: the memory pipe charges per instruction; the order of a region's
	// records does not matter to the insert.  Lane t of a batch holds records base + 2t, 2t + 1, base + 2048 + 2t, 2t + 1.)
	typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
	auto load_batch = [&](uint32_t f, uint32_t base, uint64_t (&recs)[kBatch]) {
		static_assert(kBatch == 4, "two pairs of records per thread");
#pragma unroll
		for (int u = 0; u < kBatch; u++) recs[u] = ~0ull;
		if (f == kNone) return;
		const uint32_t filled = (uint32_t)(P.cnt2[f] < G.cap2 ? P.cnt2[f] : G.cap2);
		const uint64_t *in = P.l2 + (uint64_t)f * G.cap2; // scalar base + 32-bit lane offset
		const uint32_t lane_rec = base + 2u * fresh_tid(); // opaque: lane addresses are not worth keeping alive across batches
#pragma unroll
		for (int u = 0; u < kBatch; u += 2) {
			// (cap2 is a multiple of 16 and i even: record i + 1 lies inside the bucket's storage even when it is not a record --
			// the consumer voids it, fix_odd_tail.  ONE guarded load per pair: an else-branch with an 8-byte load made the compiler
			// wait for every load before issuing the next)
			const uint32_t i = (uint32_t)u * kBuildThreads + lane_rec;
			if (i < filled) {
				const u32x4_a8 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a8 *>(in + i));
				recs[u] = ((uint64_t)v.y << 32) | v.x;
				recs[u + 1] = ((uint64_t)v.w << 32) | v.z;
			}
		}
	};
	// a region with an odd number of records: the second half of its last pair is not a record
	auto fix_odd_tail = [&](uint32_t base, uint32_t filled, uint64_t (&recs)[kBatch]) {
		if (!(filled & 1u) || filled - base > (uint32_t)kBatch * kBuildThreads) return; // (wave-uniform)
		const uint32_t lane_rec = base + 2u * fresh_tid();
#pragma unroll
		for (int u = 0; u < kBatch; u += 2)
			if ((uint32_t)u * kBuildThreads + lane_rec + 1u == filled) recs[u + 1] = ~0ull;
	};
	auto grab = [&]() { // one region index per workgroup, broadcast through LDS
		if (t == 0) {
			const unsigned int k = atomicAdd(cursor, 1u);
			if (FROM_LIST) {
				const unsigned int n_list = *redo.n < redo.cap ? *redo.n : redo.cap; // complete: every fast launch has finished
				L.next_region = k < n_list ? redo.list[k] : kNone;
			} else {
				L.next_region = k < n_regions ? first_region + k : kNone;
			}
		}
	};

	constexpr unsigned long long kForeign = 1ull << 63; // identity of a node that lives here but belongs to an earlier region
	auto load_image = [&](uint32_t ff) { // INCR, graph tables only: the image is empty, fill it from the table
		if (ff == kNone) return;
		const uint64_t rbase = (uint64_t)ff << kRegionBits, rslot0 = G.slot_lo + rbase;
		const uint32_t rlen = (uint32_t)((G.size - rslot0 < (uint64_t)kRegionSlots) ? G.size - rslot0 : kRegionSlots);
		for (uint32_t i = fresh_tid(); i < rlen; i += kBuildThreads) {
			const uint4 v = *reinterpret_cast<const uint4 *>(&table[rbase + i]);
			const uint64_t key = ((uint64_t)v.y << 32) | v.x;
			if (key == 0ull) continue;
			uint64_t q;
			const uint64_t home = fast_divmod(hash_code(key), G.magic, q);
			const bool native = home >= rslot0 && home < rslot0 + rlen;
			L.ident[i] = native ? ((q << G.r) | (home & ((1ull << G.r) - 1ull))) + 1ull : (kForeign | i);
			L.links[i] = ((uint64_t)v.w << 32) | v.z;
		}
	};

	for (int i = t; i < kRegionSlots + kSpillSlots; i += kBuildThreads) {
		L.ident[i] = 0ull;
		L.links[i] = 0ull;
	}
	if (t == 0) L.redo = 0u;
	grab();
	lds_barrier();
	uint32_t f = __builtin_amdgcn_readfirstlane(L.next_region); // scalar: everything derived from it stays in SGPRs
	if (INCR && !KF) load_image(f);
	lds_barrier();
	grab(); // the region after it
	lds_barrier();
	uint32_t f_after = __builtin_amdgcn_readfirstlane(L.next_region);
	uint32_t base = 0;
	uint64_t recs[kBatch];
	load_batch(f, 0, recs);
	uint32_t n_new = 0, n_conf = 0; // per thread: far below 2^32
	uint32_t n_new_r = 0, n_conf_r = 0; // FAST: of the current region, committed only when the region is emitted
	bool ovf = false;                   // FAST: this thread saw a counter overflow (or a full region) in the current region
	uint32_t sat = 0;                   // FAST, lean batches: the largest counter byte this thread bumped in the current region, in bits 31..24

	while (f != kNone) {
		const uint32_t b1 = G.b_lo + (f >> (G.r - kRegionBits)); // f = LOCAL final bucket == local region index == (slot - slot_lo) >> 12
		const uint64_t region_base = (uint64_t)f << kRegionBits;   // index into this shard's table
		const uint64_t region_slot0 = G.slot_lo + region_base;     // global slot of the region's first entry
		const uint32_t region_len = (uint32_t)((G.size - region_slot0 < (uint64_t)kRegionSlots) ? G.size - region_slot0 : kRegionSlots);
		const uint32_t filled = (uint32_t)(P.cnt2[f] < G.cap2 ? P.cnt2[f] : G.cap2);
		// the batch after this one
		const bool last_of_region = base + (uint32_t)kBatch * kBuildThreads >= filled;
		const uint32_t f_nxt = last_of_region ? f_after : f, base_nxt = last_of_region ? 0u : base + (uint32_t)kBatch * kBuildThreads;
		uint64_t nxt[kBatch];
		load_batch(f_nxt, base_nxt, nxt);
		fix_odd_tail(base, filled, recs);
		if (DBG == 1) {
			uint64_t x = 0;
#pragma unroll
			for (int u = 0; u < kBatch; u++) x ^= recs[u];
			if (x == 0x1234567ull) table[t].kmer = x;
		} else if constexpr (FAST) {
			// The first probe of a record is ONE unconditional compare-swap on its home slot (0 = claimed, its identity = found; the
			// four of a thread are in flight together).  A record whose home slot holds another key does not walk in its owner lane
			// -- a loop there runs as long as the busiest lane of the wave needs for all four of its records (a fifth of the records
			// walk at all), with every register of the four selected by index: measured as the largest cost centre of the kernel --
			// but is handed to the wave's QUEUE in LDS (ballot + mbcnt give its place); when the four records have been probed (or
			// the queue would overflow) the wave walks the queued records one per lane, all lanes busy with a loop of a read, a
			// compare and, on an empty slot, the claiming compare-swap.  Whoever ends a record's probe (owner or walker) bumps its
			// two neighbour counters with one ds_add_rtn_u64; a plain add cannot saturate, so the returned bytes are folded into
			// `sat` (the bumped byte moved to the top of a word, maximum over the thread's records of the region): >= 0xFF000000 at
			// the region's end says a counter that already held 255 was bumped -- the region then emits NOTHING and is rebuilt by
			// the exact form (kmerSet.cpp:253-273 stops at 255).  A wave without a record u skips that claim and that add (the
			// tail of a region's last batch; most of the batch in a flush of a streaming build): an LDS atomic costs the same with
			// 0 lanes as with 64.
			const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
			unsigned long long *const wq = L.walkq[__builtin_amdgcn_readfirstlane((uint32_t)t >> 6)];
			auto walk_queue = [&](uint32_t n_q) {
				if (lane < n_q) {
					const uint64_t rec = wq[lane];
					const unsigned long long id = (rec >> 6) + 1ull;
					const uint32_t home = (uint32_t)(rec >> 6) & (kRegionSlots - 1u);
					uint32_t at = home, claim = 0u;
					unsigned long long cur;
					do { // ONE exit, no breaks; `at` is the slot just looked at
						at++;
						cur = L.ident[at];
						if (cur == 0ull) {
							const unsigned long long prev = atomicCAS(&L.ident[at], 0ull, id);
							claim = prev == 0ull ? 1u : 0u;
							cur = prev == 0ull ? id : prev;
						}
					} while (cur != id && at < (uint32_t)(kRegionSlots + kSpillSlots - 1));
					const bool hit = cur == id;
					n_conf_r += at - home;
					n_new_r += at < region_len ? claim : 0u; // spilled nodes are counted when they are merged
					ovf = ovf || !hit;                       // region + spill area completely full: the exact pass sends it to the overflow list
					if (hit) {
						const uint32_t sh_l = (uint32_t)rec & 0x38u, sh_r = ((uint32_t)rec << 3) & 0x38u; // 8 * lb, 8 * rb; 32 = no neighbour on that side
						const uint32_t dl = (uint32_t)(0x01000000ull >> sh_l), dr = (uint32_t)(0x01000000ull >> sh_r); // A in bits 31..24 (kmerSet.cpp:56)
						const unsigned long long old = atomicAdd(&L.links[at], ((unsigned long long)dr << 32) | dl);
						const uint32_t bl = (uint32_t)((uint64_t)(uint32_t)old << sh_l), br = (uint32_t)((uint64_t)(uint32_t)(old >> 32) << sh_r);
						sat = max(sat, max(bl, br));
					}
				}
			};
			unsigned long long got[kBatch];
#pragma unroll
			for (int u = 0; u < kBatch; u++) {
				got[u] = 0ull;
				if (recs[u] != ~0ull) got[u] = atomicCAS(&L.ident[(uint32_t)(recs[u] >> 6) & (kRegionSlots - 1u)], 0ull, (recs[u] >> 6) + 1ull);
			}
			uint32_t n_q = 0; // wave-uniform
			bool own[kBatch];
#pragma unroll
			for (int u = 0; u < kBatch; u++) {
				const uint32_t home = (uint32_t)(recs[u] >> 6) & (kRegionSlots - 1u);
				// (bitwise: three compares and two scalar ANDs, no short-circuit branches)
				const bool live = recs[u] != ~0ull, fresh = got[u] == 0ull, same = got[u] == (recs[u] >> 6) + 1ull;
				const bool walk = live & !fresh & !same;
				const unsigned long long m = __builtin_amdgcn_ballot_w64(walk);
				const uint32_t n_w = (uint32_t)__builtin_popcountll(m);
				if (n_q + n_w > (uint32_t)kWalkQ) { // (wave-uniform; rare at the load factors the reference allows)
					walk_queue(n_q);
					n_q = 0;
				}
				if (walk) wq[n_q + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = recs[u];
				n_q += n_w;
				n_new_r += (live & fresh & (home < region_len)) ? 1u : 0u;
				// the owner's add: issued now, its returned bytes looked at after the walks (four adds in flight)
				own[u] = live & !walk;
				if (own[u]) {
					const uint32_t sh_l = (uint32_t)recs[u] & 0x38u, sh_r = ((uint32_t)recs[u] << 3) & 0x38u;
					const uint32_t dl = (uint32_t)(0x01000000ull >> sh_l), dr = (uint32_t)(0x01000000ull >> sh_r);
					got[u] = atomicAdd(&L.links[home], ((unsigned long long)dr << 32) | dl);
				}
			}
			walk_queue(n_q);
#pragma unroll
			for (int u = 0; u < kBatch; u++) {
				const uint32_t sh_l = (uint32_t)recs[u] & 0x38u, sh_r = ((uint32_t)recs[u] << 3) & 0x38u;
				const uint32_t bl = (uint32_t)((uint64_t)(uint32_t)got[u] << sh_l), br = (uint32_t)((uint64_t)(uint32_t)(got[u] >> 32) << sh_r);
				sat = own[u] ? max(sat, max(bl, br)) : sat;
			}
		} else {
#pragma unroll
			for (int u = 0; u < kBatch; u++) {
				// The loops below have ONE exit condition each and no breaks: the structurizer turns every
				// extra exit of a divergent loop into a dozen scalar mask instructions per iteration, and this
				// kernel is bound by instruction issue (profiles/dbg_modes_build.sh).
				const uint64_t rec = recs[u];
				const bool live = rec != ~0ull;
				const unsigned long long id = (rec >> 6) + 1ull;
				const uint32_t lb = (uint32_t)(rec >> 3) & 7u, rb = (uint32_t)rec & 7u;
				const uint32_t home = (uint32_t)(rec >> 6) & (kRegionSlots - 1u);
				uint32_t idx = home;
				unsigned long long old = L.links[home]; // usually the key sits in its home slot: fetch its counters together with the first probe
				bool probing = live, lost = false;
				while (probing) {
					unsigned long long cur = L.ident[idx];
					if (cur == 0ull) {
						const unsigned long long prev = atomicCAS(&L.ident[idx], 0ull, id);
						cur = prev == 0ull ? id : prev;
						n_new += (prev == 0ull && idx < region_len) ? 1u : 0u; // spilled nodes are counted when they are merged
					}
					const bool hit = cur == id;
					idx += hit ? 0u : 1u;
					n_conf += hit ? 0u : 1u;
					lost = idx >= (uint32_t)(kRegionSlots + kSpillSlots); // region + spill area completely full
					probing = !hit && !lost;
				}
				// saturating +1 on the observed neighbour bytes (add_node_to_kmerset's "if (< 255) ++", kmerSet.cpp:253-273):
				// both dwords at once, bytes that are already 255 masked out of the increment
				const uint32_t dl = (lb != 4u) ? (1u << (24u - 8u * lb)) : 0u, dr = (rb != 4u) ? (1u << (24u - 8u * rb)) : 0u;
				bool pending = live && !lost;
				if (pending && idx != home) old = L.links[idx];
				while (pending) {
					const uint32_t lo = (uint32_t)old, hi = (uint32_t)(old >> 32);
					const uint32_t sat_l = (((lo & 0x7F7F7F7Fu) + 0x01010101u) & lo & 0x80808080u) >> 7; // 0x01 in every byte that is 0xFF
					const uint32_t sat_r = (((hi & 0x7F7F7F7Fu) + 0x01010101u) & hi & 0x80808080u) >> 7;
					const unsigned long long upd = ((unsigned long long)(hi + (dr & ~sat_r)) << 32) | (lo + (dl & ~sat_l));
					unsigned long long prev = old;
					if (upd != old) prev = atomicCAS(&L.links[idx], old, upd);
					pending = prev != old;
					old = prev;
				}
				if (live && lost) push_overflow(P, record_key(rec, b1, G), lb, rb, ctr);
			}
		}
		if (last_of_region) {
			bool redo_region = false;
			if constexpr (FAST) {
				if (ovf || sat >= 0xFF000000u) L.redo = 1u;
				lds_barrier();
				redo_region = __builtin_amdgcn_readfirstlane(L.redo) != 0u;
				if (!redo_region) { n_new += n_new_r; n_conf += n_conf_r; }
				else if (t == 0) {
					const unsigned int j = atomicAdd(redo.n, 1u);
					if (j < redo.cap) redo.list[j] = f; else atomicOr(&ctr->error, 2u);
				}
				n_new_r = n_conf_r = 0u;
				ovf = false;
				sat = 0u;
			} else {
				lds_barrier();
			}
			if (DBG != 1 && DBG != 2 && !redo_region) {
				// emit the region: slot i of the table <- LDS slot i (key recomputed from (q, home slot)); the LDS
				// image is cleared on the way for the next region.
				if constexpr (FAST && !KF && !INCR && DBG == 0) {
					// The keys first, on FULL waves: a wave owns the slots tid + 1024 j, 37 % of them occupied (cfg2), and hash_code_inverse is
					// ~55 of the ~70 VALU instructions an occupied slot costs below -- executed for every slot of a wave that has one.  So each
					// wave lists its occupied slots (ballot + mbcnt, 16-bit slot indices in its idle walk queue), turns the identities of
					// the list into keys with every lane busy (two rounds instead of four) and leaves them in ident[]; the loop below then
					// only moves slot i to the table.  No barrier: a wave reads and writes its own slots only.
					// (Round 1 measured a compaction as "no change" when this kernel took 6.7 ms and waited on its LDS round trips; since the
					// lean insert of round 5 it is three quarters VALU-busy and the emit was half of its instructions.)
					static_assert(kBuildThreads / 64 * kWalkQ * 8 >= kBuildThreads / 64 * 256 * 2, "a wave's queue holds 256 slot indices");
					uint16_t *const cq = reinterpret_cast<uint16_t *>(L.walkq[__builtin_amdgcn_readfirstlane((uint32_t)t >> 6)]);
					uint32_t n_occ = 0; // wave-uniform
					const uint32_t t0 = fresh_tid();
#pragma unroll
					for (uint32_t j = 0; j < (uint32_t)kRegionSlots / kBuildThreads; j++) {
						const uint32_t i = t0 + j * kBuildThreads;
						const bool occ = i < region_len && L.ident[i] != 0ull;
						const unsigned long long m = __builtin_amdgcn_ballot_w64(occ);
						if (occ) cq[n_occ + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)i;
						n_occ += (uint32_t)__builtin_popcountll(m);
					}
					for (uint32_t e = t0 & 63u; e < n_occ; e += 64u) {
						const uint32_t i = cq[e];
						const uint64_t v = L.ident[i] - 1ull;
						const uint64_t slot = ((uint64_t)b1 << G.r) | (v & ((1ull << G.r) - 1ull));
						L.ident[i] = hash_code_inverse((v >> G.r) * G.size + slot); // (never 0: key 0 has no record)
					}
				}
				for (uint32_t i = fresh_tid(); i < region_len; i += kBuildThreads) { // (opaque: the lane's table address is not worth a register across the inserts)
					const unsigned long long id = L.ident[i];
					uint64_t key = 0ull, links = 0ull;
					if constexpr (FAST && !KF && !INCR && DBG == 0) { // ident[] holds the keys already
						if (id) {
							key = id;
							links = L.links[i];
							L.ident[i] = 0ull;
							L.links[i] = 0ull;
						}
						*reinterpret_cast<uint4 *>(&table[region_base + i]) =
						    make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)links, (uint32_t)(links >> 32));
						continue;
					}
					const bool foreign = INCR && !KF && (id & kForeign);
					if (id) {
						const uint64_t v = id - 1ull;
						const uint64_t slot = ((uint64_t)b1 << G.r) | (v & ((1ull << G.r) - 1ull));
						if (!foreign) key = (DBG == 3) ? v + slot : hash_code_inverse((v >> G.r) * G.size + slot);
						links = L.links[i];
						L.ident[i] = 0ull;
						L.links[i] = 0ull;
					}
					if (KF) { // every key is aggregated in exactly one region: a plain byte store, nobody else writes it during the build
						if (id) {
							uint8_t *cell = reinterpret_cast<uint8_t *>(table) + key;
							const uint32_t c = (uint32_t)links >> 24, sum = INCR ? min(255u, (uint32_t)*cell + c) : c;
							*cell = (uint8_t)sum;
						}
					} else if (foreign) { // the slot keeps the node another region's spill merge put there
					} else {
						*reinterpret_cast<uint4 *>(&table[region_base + i]) =
						    make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)links, (uint32_t)(links >> 32));
					}
				}
				// nodes that probed past the region end: re-inserted by k_merge_nodes after all regions exist
				for (uint32_t i = region_len + t; i < (uint32_t)(kRegionSlots + kSpillSlots); i += kBuildThreads) {
					const unsigned long long id = L.ident[i];
					if (!id) continue;
					const uint64_t v = id - 1ull;
					const uint64_t slot = ((uint64_t)b1 << G.r) | (v & ((1ull << G.r) - 1ull));
					const unsigned long long j = atomicAdd(&P.ovf_n[1], 1ull);
					if (j < P.spill_cap) {
						P.spill[j].kmer = hash_code_inverse((v >> G.r) * G.size + slot);
						P.spill[j].links = L.links[i];
					} else {
						atomicOr(&ctr->error, 2u);
					}
					L.ident[i] = 0ull;
					L.links[i] = 0ull;
				}
			} else {
				for (int i = t; i < kRegionSlots + kSpillSlots; i += kBuildThreads) {
					L.ident[i] = 0ull;
					L.links[i] = 0ull;
				}
			}
			grab(); // the region after the one whose first batch is already in flight
			lds_barrier(); // the image is empty again, next_region is visible
			f_after = __builtin_amdgcn_readfirstlane(L.next_region);
			if (FAST && t == 0) L.redo = 0u; // (every thread has read it: that happened before its emit, i.e. before the barrier above)
			if (INCR && !KF) load_image(f_nxt); // the region whose records come next
			lds_barrier();
		}
		f = f_nxt;
		base = base_nxt;
#pragma unroll
		for (int u = 0; u < kBatch; u++) recs[u] = nxt[u];
	}
	const unsigned long long a = block_sum_n<kBuildThreads>(n_new, L.red);
	const unsigned long long b = block_sum_n<kBuildThreads>(n_conf, L.red);
	if (t == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
	}
}

// ---- KFREQ, direct blocks: one workgroup per 64-KiB block of the count table ------------------------------------------
// (see kf_slot_of_key).  Final bucket f = permuted block index; its records carry the key's low 16 bits, the block's place in
// the table is kf_key_of_slot(f << 16).  FAST: one plain LDS add per occurrence on the 32-bit word that holds the byte; the
// returned word tells whether that byte already held 255 -- then the add has carried into its neighbour, the block is flagged,
// writes NOTHING and is rebuilt by the exact form (compare-swap loop that stops at 255, FROM_LIST) after all fast launches,
// like a region of the graph build.  INCR: the table already holds counts (an earlier flush): the block is loaded first.
// Every block of the launch's range is written, also those without a record: nothing zeroes the table beforehand.
struct KfBlockLds {
	uint32_t w[1u << (kKfBlockBits - 2u)];
	uint32_t next_region;
	uint32_t redo;
};

template <bool INCR, bool FAST, bool FROM_LIST = false>
__global__ __launch_bounds__(kBuildThreads) void k_kf_build_blocks(PartGeom G, PartStore P, uint8_t *__restrict__ counts, Counters *__restrict__ ctr,
                                                                   uint32_t first_region, uint32_t n_regions, unsigned int *__restrict__ cursor,
                                                                   RedoList redo)
{
	static_assert(!(FAST && FROM_LIST), "the exact pass is what the list is for");
	extern __shared__ __align__(16) unsigned char lds_raw[];
	KfBlockLds &L = *reinterpret_cast<KfBlockLds *>(lds_raw);
	const uint32_t t = fresh_tid();
	constexpr uint32_t kNone = 0xFFFFFFFFu;
	constexpr uint32_t kVec = (1u << kKfBlockBits) / 16u; // 16-byte vectors of a block
	constexpr int kBatch = 4;
	auto grab = [&]() { // one block index per workgroup, broadcast through LDS
		if (t == 0) {
			const unsigned int k = atomicAdd(cursor, 1u);
			if (FROM_LIST) {
				const unsigned int n_list = *redo.n < redo.cap ? *redo.n : redo.cap; // complete: every fast launch has finished
				L.next_region = k < n_list ? redo.list[k] : kNone;
			} else {
				L.next_region = k < n_regions ? first_region + k : kNone;
			}
		}
	};
	grab();
	lds_barrier();
	uint32_t f = __builtin_amdgcn_readfirstlane(L.next_region);
	uint4 *lw = reinterpret_cast<uint4 *>(L.w);
	// the table summary (non-zero counters, their sum) follows what is written: + what a block holds when it is emitted,
	// - what it held when it was loaded (INCR); committed once per wave at the end
	auto nz4 = [](const uint4 &v) {
		auto nz = [](uint32_t w) { return (uint32_t)__builtin_popcount((((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u); };
		return nz(v.x) + nz(v.y) + nz(v.z) + nz(v.w);
	};
	auto sum4 = [](const uint4 &v) {
		return __builtin_amdgcn_sad_u8(v.w, 0u, __builtin_amdgcn_sad_u8(v.z, 0u, __builtin_amdgcn_sad_u8(v.y, 0u, __builtin_amdgcn_sad_u8(v.x, 0u, 0u))));
	};
	unsigned long long d_nz = 0ull, d_sum = 0ull;
	while (f != kNone) {
		uint4 *blk = reinterpret_cast<uint4 *>(counts + kf_key_of_slot((uint64_t)f << kKfBlockBits, G.kf_mask));
		uint32_t had_nz = 0u, had_sum = 0u;
		for (uint32_t j = t; j < kVec; j += kBuildThreads) {
			const uint4 v = INCR ? blk[j] : make_uint4(0u, 0u, 0u, 0u);
			lw[j] = v;
			if (INCR) { had_nz += nz4(v); had_sum += sum4(v); }
		}
		if (t == 0) L.redo = 0u;
		lds_barrier(); // the image is there; everybody has read next_region
		grab();        // the block after this one (visible behind the next barrier)
		// level 2 left 16-bit records (the key's place in the block), cap2 of them per block (a multiple of 4): four per 8-byte load
		const uint32_t filled = (uint32_t)(P.cnt2[f] < G.cap2 ? P.cnt2[f] : G.cap2);
		const uint64_t *in = reinterpret_cast<const uint64_t *>(reinterpret_cast<const uint16_t *>(P.l2) + (uint64_t)f * G.cap2);
		bool ovf = false;
		for (uint32_t base = 0; base < filled; base += (uint32_t)kBatch * kBuildThreads) {
			const uint32_t first = base + (uint32_t)kBatch * t; // this lane's four records
			const uint64_t four = first < filled ? __builtin_nontemporal_load(in + (first >> 2)) : 0ull;
#pragma unroll
			for (int u = 0; u < kBatch; u++) {
				if (first + (uint32_t)u >= filled) continue; // (a wave without a record skips the LDS instruction altogether)
				const uint32_t idx = (uint32_t)(four >> (16 * u)) & ((1u << kKfBlockBits) - 1u), sh = 8u * (idx & 3u);
				if constexpr (FAST) {
					const uint32_t old = atomicAdd(&L.w[idx >> 2], 1u << sh);
					ovf = ovf || ((old >> sh) & 0xFFu) == 0xFFu;
				} else {
					uint32_t old = L.w[idx >> 2];
					bool pending = true;
					while (pending) {
						const bool full = ((old >> sh) & 0xFFu) == 0xFFu;
						uint32_t prev = old;
						if (!full) prev = atomicCAS(&L.w[idx >> 2], old, old + (1u << sh));
						pending = prev != old;
						old = prev;
					}
				}
			}
		}
		if (FAST && ovf) L.redo = 1u;
		lds_barrier(); // all adds have landed, next_region and the flag are visible
		const bool redo_block = FAST && __builtin_amdgcn_readfirstlane(L.redo) != 0u;
		const uint32_t f_next = __builtin_amdgcn_readfirstlane(L.next_region);
		if (!redo_block) {
			uint32_t now_nz = 0u, now_sum = 0u;
			for (uint32_t j = t; j < kVec; j += kBuildThreads) {
				const uint4 v = lw[j];
				blk[j] = v;
				now_nz += nz4(v);
				now_sum += sum4(v);
			}
			d_nz += (unsigned long long)now_nz - (unsigned long long)had_nz;
			d_sum += (unsigned long long)now_sum - (unsigned long long)had_sum;
		} else if (t == 0) {
			const unsigned int j = atomicAdd(redo.n, 1u);
			if (j < redo.cap) redo.list[j] = f; else atomicOr(&ctr->error, 2u);
		}
		lds_barrier(); // the image has been read: the next block may overwrite it
		f = f_next;
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		d_nz += __shfl_down(d_nz, off, 64);
		d_sum += __shfl_down(d_sum, off, 64);
	}
	if ((t & 63u) == 0u) {
		if (d_nz) atomicAdd(&ctr->kf_nonzero, d_nz);
		if (d_sum) atomicAdd(&ctr->kf_sum, d_sum);
	}
}

} // namespace dbgk
