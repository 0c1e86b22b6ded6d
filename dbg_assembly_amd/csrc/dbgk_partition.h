// dbgk_partition.h -- PARTITION engine: make the table traffic sequential.
//
// The DIRECT engine pays one random 64-byte sector read plus one 64-byte atomic write-back per
// k-mer occurrence.  Here every occurrence becomes an 8-byte RECORD that is
// radix-partitioned by the table slot it will finally live in, and each 4096-slot REGION of the
// reference-layout table (slot = hash_code(key) % size, DBGgraph.cpp:167) is then built inside LDS
// by one workgroup and written out once with coalesced 16-byte stores:
//
//   k_extract_scatter   reads -> records, scattered into n1 level-1 buckets (slot >> r)
//   k_scatter_l2        every level-1 bucket -> n2 = 2^(r-12) final buckets (slot >> 12); XCD-aware
//                       tile order so that every append point is fed through one L2
//   k_build_regions     one workgroup per final bucket: LDS open addressing on the region's own
//                       slots (ds_cmpst claims, LDS CAS saturating counters), emit the region;
//                       runs concurrently with k_scatter_l2 of the next bucket chunk (dbgk.hip)
//   k_insert_triples    the few records that found no room in their bucket, through the
//                       global-atomic path; region spill-over nodes go through k_merge_nodes
//
// RECORD (64 bit):  [ q | slot_rel : r bits | lb : 3 | rb : 3 ]   with hash_code(key) = q*size + slot,
// slot_rel = slot & (2^r - 1).  hash_code is a bijection on u64 (inverse below), so the key itself
// is not stored: it is recomputed from (q, slot) once per DISTINCT node when the region is emitted.
#pragma once

#include <type_traits>

#include "dbgk_kernels.h"

namespace dbgk {

#ifndef DBGK_REGION_BITS
#define DBGK_REGION_BITS 12
#endif
#ifndef DBGK_BUILD_THREADS
#define DBGK_BUILD_THREADS 1024
#endif
constexpr int kRegionBits = DBGK_REGION_BITS;   // 4096 slots = 64 KiB of nodes per region (experiments: 11)
constexpr int kRegionSlots = 1 << kRegionBits;
constexpr int kSpillSlots = 128;                // LDS slots past the region end: probe overflow
#ifndef DBGK_L1_THREADS
#define DBGK_L1_THREADS 1024
#endif
#ifndef DBGK_L1_MAXB
#define DBGK_L1_MAXB 1024
#endif
constexpr int kTileThreads = DBGK_L1_THREADS;   // level 1: 16 waves, 16384-record tiles (level 2: kL2Threads below)
constexpr int kL1MaxB = DBGK_L1_MAXB;           // level-1 fan-out limit (histogram size of the level-1 kernels)
constexpr int kMaxBuckets = 1024;               // per-level fan-out limit (LDS histogram size)
constexpr int kSubStores = 1;                   // level-1 sub-stores per bucket (8 = one per XCD was measured: 7.04 ms against 7.07, so off)
constexpr int kL1Threads = kTileThreads;
constexpr int kBuildThreads = DBGK_BUILD_THREADS; // 2 workgroups per CU (66 KiB LDS each) = 32 waves per CU, needs <= 64 VGPRs

// ---- inverse of hash_code ------------------------------------------------------------------
constexpr uint64_t mod_inverse_u64(uint64_t a) // a odd; Newton iteration doubles the correct bits
{
	uint64_t x = a;
	for (int i = 0; i < 6; i++) x *= 2 - a * x;
	return x;
}
constexpr uint64_t kInvM32 = mod_inverse_u64(1ull - (1ull << 32)); // k + ~(k << 32) == k * (1 - 2^32) - 1
constexpr uint64_t kInvM13 = mod_inverse_u64(1ull - (1ull << 13));
constexpr uint64_t kInvM27 = mod_inverse_u64(1ull - (1ull << 27));
constexpr uint64_t kInv9 = mod_inverse_u64(9ull);

__host__ __device__ __forceinline__ uint64_t hash_code_inverse(uint64_t h)
{
	h ^= h >> 31; h ^= h >> 62;                       // undo k ^= k >> 31
	h = (h + 1) * kInvM27;                            // undo k += ~(k << 27)
	h ^= h >> 15; h ^= h >> 30; h ^= h >> 60;         // undo k ^= k >> 15
	h *= kInv9;                                       // undo k += k << 3
	h ^= h >> 8; h ^= h >> 16; h ^= h >> 32;          // undo k ^= k >> 8
	h = (h + 1) * kInvM13;                            // undo k += ~(k << 13)
	h ^= h >> 22; h ^= h >> 44;                       // undo k ^= k >> 22
	h = (h + 1) * kInvM32;                            // undo k += ~(k << 32)
	return h;
}

// ---- geometry --------------------------------------------------------------------------------
struct PartGeom {
	uint64_t size;       // table slots
	ModMagic magic;
	uint32_t r;          // level-1 bucket = slot >> r
	uint32_t n1;         // ceil(size / 2^r)            <= kMaxBuckets
	uint32_t n2;         // 2^(r - 12)                  <= kMaxBucketsL2
	uint32_t n_final;    // ceil(size / 4096)
	uint64_t cap1;       // records per level-1 bucket
	uint64_t cap2;       // records per final bucket
	Div32Magic div;      // size < 2^32: exact 64/32 division (dbgk_device.h); larger tables divide by `magic`
	// sharding (multi-GPU): `size` is the GLOBAL table; this handle owns the level-1 buckets
	// [b_lo, b_lo + nb_own) = the contiguous slot range [slot_lo, slot_hi) and holds only that
	// part of the table.  n_ranks == 1: b_lo = 0, nb_own = n1, the whole table.
	uint32_t n_ranks, rank;
	uint32_t B;          // level-1 buckets per rank = ceil(n1 / n_ranks); the level-1 store has n_ranks * B buckets
	uint32_t n_sub;      // sub-stores per level-1 bucket (kSubStores): workgroup w appends to sub-store w % n_sub, i.e.
	                     // (round-robin dispatch) all appends to one sub-store come through ONE XCD's L2 and merge there
	uint32_t b_lo, nb_own;
	uint32_t n_regions_own;
	uint64_t slot_lo, slot_hi;
	uint32_t r_rec;      // the r of the RECORD format: q = (record >> 6) >> r_rec.  == r except in the final pass of a three-level
	                     // partition, whose buckets are finer than the level-1 buckets the records were made for
	uint32_t l2_shift;   // level-2 passes: bucket of a record = (record >> (6 + 12 + l2_shift)) & (n2 - 1); 0 except in the MID
	                     // pass of a three-level partition (tables of 2^33 slots and more, see launch_l2 in dbgk.hip)
	uint32_t kf;         // KFREQ through this engine: a record is one occurrence of the key, its neighbour fields are
	                     // fixed (lb = 0, rb = none), so the A counter of l_link is the saturating occurrence count.
	                     // 2: DIRECT BLOCKS -- `size` is 4^k itself and slot = kf_slot_of_key(key), see below
	uint32_t kf_mask;    // kf == 2: 2^(2k - 16) - 1, the mask of a block index
	uint32_t l2_records; // records of a level-2 tile = 16 x the threads of the level-2 kernel this geometry is scattered by (l2_threads(n2))
};

// ---- KFREQ, direct blocks (kf == 2) --------------------------------------------------------------------------------
// The frequency table is direct-addressed -- counts[key] -- so a "region" can be a BLOCK OF THE TABLE ITSELF: 2^16 consecutive
// key values = 64 KiB of byte counters, held in LDS as they are (no hash table, no identities, no probing: one LDS add per
// occurrence) and written back as one contiguous 64 KiB run (no random byte stores, no zeroing of the 4^k bytes beforehand,
// no hash to invert).  Canonical k-mers are not uniform over the key space (min(x, rc(x)) starts with A 1.75 times as often as
// the average and with T a quarter as often), so the block index is PERMUTED -- multiplied by an odd constant modulo the
// number of blocks -- before it becomes the slot's high bits: heavy and light blocks then spread evenly over the level-1 and
// level-2 buckets (1024 blocks per level-1 bucket at k = 17), and only a block's own record count varies (cap2 allows 2.6x).
constexpr uint32_t kKfBlockBits = 16;
constexpr uint32_t kKfPermMul = 0x9E3779B1u;                                   // odd
constexpr uint32_t kKfPermInv = (uint32_t)mod_inverse_u64((uint64_t)kKfPermMul); // inverse modulo 2^32, hence modulo every 2^n <= 2^32

__host__ __device__ __forceinline__ uint64_t kf_slot_of_key(uint64_t key, uint32_t block_mask)
{
	const uint32_t pb = ((uint32_t)(key >> kKfBlockBits) * kKfPermMul) & block_mask;
	return ((uint64_t)pb << kKfBlockBits) | (key & ((1ull << kKfBlockBits) - 1ull));
}

__host__ __device__ __forceinline__ uint64_t kf_key_of_slot(uint64_t slot, uint32_t block_mask)
{
	const uint32_t b = ((uint32_t)(slot >> kKfBlockBits) * kKfPermInv) & block_mask;
	return ((uint64_t)b << kKfBlockBits) | (slot & ((1ull << kKfBlockBits) - 1ull));
}

struct PartStore {
	uint64_t *l1;                 // [n_ranks * B][n_sub][cap1]: what this rank extracted, by GLOBAL level-1 bucket and sub-store
	uint32_t *cnt1;               // [n_ranks * B][n_sub] records appended (may exceed cap1: excess went to ovf)
	const uint64_t *inbox;        // [n_ranks][B][n_sub][cap1]: level-1 buckets of MY slot range from every rank
	const uint32_t *inbox_cnt;    // [n_ranks * B * n_sub]   (n_ranks == 1: inbox == l1, inbox_cnt == cnt1)
	uint32_t *l2_done;            // [n_ranks * B * n_sub] or null -- EARLY level 2 (dbgk.hip early_l2: a level-2 round over what the batches so far
	uint32_t *l2_upto;            // stored, queued in front of every push): the round in flight scatters records [l2_done, l2_upto) of each inbox entry;
	                              // both written by k_plan_l2 (done <- the last round's upto, upto <- the fill count now)
	uint64_t *l2;                 // [nb_own * n2][cap2], local final bucket = (b1 - b_lo) * n2 + b2
	uint32_t *cnt2;               // [nb_own * n2]
	Node *outgoing;               // nodes that probed past the end of this shard: for the next rank
	unsigned long long *outgoing_n;
	uint64_t outgoing_cap;
	Node *ovf;                    // overflow triples {key, lb | rb << 8}
	Node *spill;                  // nodes that probed past the end of their region
	unsigned long long *ovf_n;    // [0] = overflow triples, [1] = spill nodes
	uint64_t ovf_cap, spill_cap;
	// once the overflow list is full, further observations are AGGREGATED in a small open-addressed table
	// (global atomics): the surplus of heavy hitters -- a satellite repeat's k-mers occur millions of times
	// -- collapses onto a handful of nodes there.  Merged into the main table after the build.
	Node *hh;                     // null: not available (sharded handles)
	uint64_t hh_size;
	ModMagic hh_magic;
};

__device__ __forceinline__ uint64_t record_key(uint64_t rec, uint32_t b1, const PartGeom &G)
{
	const uint64_t v = rec >> 6;
	const uint64_t slot = ((uint64_t)b1 << G.r) | (v & ((1ull << G.r) - 1ull));
	if (G.kf == 2u) return kf_key_of_slot(slot, G.kf_mask);
	return hash_code_inverse((v >> G.r_rec) * G.size + slot);
}

__device__ __forceinline__ void push_overflow(const PartStore &P, uint64_t key, uint32_t lb, uint32_t rb, Counters *ctr)
{
	const unsigned long long i = atomicAdd(&P.ovf_n[0], 1ull);
	if (i < P.ovf_cap) {
		P.ovf[i].kmer = key;
		P.ovf[i].links = (uint64_t)lb | ((uint64_t)rb << 8);
	} else if (P.hh) {
		const TableRef T{P.hh, P.hh_size, P.hh_magic};
		const uint64_t slot = fast_mod(hash_code(key), T.magic);
		const uint4 v = *reinterpret_cast<const uint4 *>(&T.nodes[slot]);
		Node first;
		first.kmer = ((uint64_t)v.y << 32) | v.x;
		first.links = ((uint64_t)v.w << 32) | v.z;
		uint64_t guess;
		unsigned long long dummy_new = 0, dummy_conf = 0; // counted when the node is merged into the main table
		const uint64_t s = find_or_claim(T, key, slot, first, guess, dummy_new, dummy_conf);
		if (s == ~0ull)
			atomicOr(&ctr->error, 2u); // more distinct overflowing keys than the side table holds
		else
			links_cas_observe(reinterpret_cast<unsigned long long *>(&T.nodes[s].links), guess, lb, rb);
	} else {
		atomicOr(&ctr->error, 2u); // overflow store exhausted: results would be incomplete
	}
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)), which would stall every tile on its own global stores, prefetched
// loads and reservation atomics; all data exchanged between the threads of these kernels goes
// through LDS, so waiting for the LDS counter is sufficient.
__device__ __forceinline__ void lds_barrier()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// threadIdx.x as a value the optimiser cannot hoist: the persistent tile loops otherwise keep dozens of
// loop-invariant per-thread addresses alive across the whole iteration, which pushes these
// 1024-thread kernels (128-VGPR ceiling) into scratch spills -- and a scratch reload queues behind
// the tile's global stores (one in-order vmcnt).  Recomputing an address costs an instruction or two.
__device__ __forceinline__ uint32_t fresh_tid()
{
	uint32_t t = threadIdx.x;
	asm volatile("" : "+v"(t));
	return t;
}

// ---- workgroup-wide bucket scatter of up to 16 records per thread ------------------------------
// MAXB: fan-out limit of the level (histogram size).  Level 1 always has <= 1024 buckets; level 2 has
// n2 = 2^(r - 12) <= 1024 for tables below 2^32 slots and 2048 / 4096 for tables up to 2^33 / 2^34 slots
// (kernels instantiated per MAXB, chosen at launch).  FLAT (level 2): the copy-out needs only the global
// base of a bucket (32 bits); the wave-per-bucket copy-out of level 1 reads a 64-bit descriptor.
template <int THREADS, int MAXB = kMaxBuckets, bool FLAT_DESC = false>
struct ScatterLdsT {
	static constexpr int kThreads = THREADS;           // workgroup size
	static constexpr int kRecords = THREADS * 16;      // records staged per tile
	static constexpr int kMaxB = MAXB;
	static constexpr int kBpt = MAXB / THREADS;        // histogram entries owned by one thread
	using Desc = typename std::conditional<FLAT_DESC, uint32_t, uint64_t>::type;
	uint64_t stage[kRecords];
	uint32_t hist[MAXB + 64];        // + one dummy bin per lane (level 1: positions that yield no record)
	uint32_t lbase[MAXB];
	Desc desc[MAXB];                 // copy-out descriptor per bucket (scatter_stage_copy)
	uint32_t wave_tot[THREADS / 64];
};
using ScatterLds = ScatterLdsT<kTileThreads, kL1MaxB>;   // level 1: 16384-record tiles, one workgroup per CU
#ifndef DBGK_L2_THREADS
#define DBGK_L2_THREADS 512
#endif
// Level-2 workgroup: 1024 threads and 16 K-record tiles (140 KiB of LDS, one workgroup per CU) for fan-outs up to 1024 -- a final
// bucket then gets 16 records = one whole 128-byte line per tile instead of half a line (round 5: level 2 alone 4.48 -> 4.09 ms on a
// box whose HBM is on the slow side, profiles/r05_l2_1024_threads_ab.txt); the 2048- and 4096-way forms keep 512 threads and 8 K-record
// tiles (their histograms would not fit beside a 128 KiB stage).  The tile plan (k_plan_l2) takes the tile size from PartGeom.l2_records.
constexpr int l2_threads(int maxb) { return maxb <= 1024 ? 2 * DBGK_L2_THREADS : DBGK_L2_THREADS; }
template <int MAXB> using ScatterLdsL2T = ScatterLdsT<l2_threads(MAXB), MAXB, true>;
using ScatterLdsL2 = ScatterLdsL2T<kMaxBuckets>;
constexpr int kMaxBucketsL2 = 4096;             // largest level-2 fan-out (tables below 2^34 slots)

// exclusive prefix sum of hist[0..kMaxBuckets) into lbase; thread t owns entries LDS::kBpt*t .. LDS::kBpt*t+LDS::kBpt-1
// returns the number of records ranked in the tile (the sum of the whole histogram)
template <class LDS>
__device__ __forceinline__ uint32_t scan_hist(LDS &L)
{
	const int t = (int)fresh_tid(), lane = t & 63, wave = t >> 6;
	constexpr int kBpt = LDS::kBpt;
	uint32_t v[kBpt], sum = 0;
#pragma unroll
	for (int j = 0; j < LDS::kBpt; j++) { v[j] = L.hist[LDS::kBpt * t + j]; sum += v[j]; }
	uint32_t inc = sum;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t n = __shfl_up(inc, off, 64);
		if (lane >= off) inc += n;
	}
	if (lane == 63) L.wave_tot[wave] = inc;
	lds_barrier();
	uint32_t run = inc - sum, all = 0;
#pragma unroll
	for (int w = 0; w < LDS::kThreads / 64; w++) { // branch-free: a rolled loop gets vectorised into a register hog
		const uint32_t wt = L.wave_tot[w];
		run += (w < wave) ? wt : 0u;
		all += wt;
	}
#pragma unroll
	for (int j = 0; j < LDS::kBpt; j++) { L.lbase[LDS::kBpt * t + j] = run; run += v[j]; }
	return all;
}

// The same prefix sum computed by EVERY wave for itself (level 1, whose fan-out is small next to its 16 waves): lane l of a wave owns
// the buckets [E l, E l + E), E = ceil(n_buckets / 64) <= 16, scans them with wave shuffles and writes lbase -- all waves write the
// same values -- so no wave_tot exchange and no barrier inside the scan: a wave reads back only what it has written itself.
template <class LDS>
__device__ __forceinline__ void scan_hist_per_wave(LDS &L, uint32_t n_buckets, uint32_t hist_off = 0u) // hist_off: the histogram to scan, in words from L.hist (the pipelined level 1 has two)
{
	const uint32_t *hist = L.hist + hist_off;
	const uint32_t lane = fresh_tid() & 63u;
	const uint32_t E = (n_buckets + 63u) >> 6; // (wave-uniform)
	uint32_t sum = 0;
	for (uint32_t j = 0; j < E; j++) { // (E <= 16; buckets beyond n_buckets hold zero: the histogram is cleared up to kMaxB)
		const uint32_t b = E * lane + j;
		sum += b < (uint32_t)LDS::kMaxB ? hist[b] : 0u;
	}
	uint32_t inc = sum;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t n = __shfl_up(inc, off, 64);
		if ((int)lane >= off) inc += n;
	}
	uint32_t run = inc - sum;
	for (uint32_t j = 0; j < E; j++) {
		const uint32_t b = E * lane + j;
		if (b < (uint32_t)LDS::kMaxB) {
			const uint32_t c = hist[b];
			L.lbase[b] = run;
			run += c;
		}
	}
}

// Phases of the workgroup-wide bucket scatter of one tile (<= 16 records per thread).
//   br[u] = (bucket << 16) | rank-within-bucket, bucket >= kMaxBuckets = no record

// after every record has been ranked (hist complete): reserve global space, scan
template <class LDS>
__device__ __forceinline__ uint32_t scatter_reserve_scan(LDS &L, uint32_t n_buckets, uint32_t *__restrict__ cnt, uint32_t (&my_gbase)[LDS::kBpt],
                                                         uint32_t stride = 1u) // bucket b counts in cnt[b * stride]; returns the records of the tile
{
	const int t = (int)fresh_tid();
	// one global atomic per non-empty bucket per tile: issued now, consumed only at copy-out, so its
	// round trip overlaps the scan and the staging writes
#pragma unroll
	for (int j = 0; j < LDS::kBpt; j++) {
		const uint32_t b = LDS::kBpt * t + j, c = L.hist[b];
		my_gbase[j] = (b < n_buckets && c) ? atomicAdd(&cnt[b * stride], c) : 0u;
	}
	const uint32_t all = scan_hist(L);
	lds_barrier();
	return all;
}

// bucket-sorted staging of the records, then the coalesced copy-out.  Records beyond a bucket's
// capacity are decoded and pushed to the overflow list.
// FLAT (level 2): the bucket of a staged record can be recomputed from the record itself
// ((rec >> 18) & (n_buckets - 1)), so the copy-out walks the sorted stage linearly, one record per
// lane and every lane busy: 256 store instructions per tile instead of one (half-empty) store per
// bucket and wave.  The vector-memory pipe costs ~27 clocks per store instruction per CU whatever
// the number of active lanes (profiles/dbg_modes_l2.sh), which makes the instruction count the cost:
// level 2 went from 6.3 to 5.3 ms.
// KF32_POSSIBLE: the caller may run for a KFREQ handle with direct blocks (level 1 with the 64-bit slot path, WIDE_D >= 2): only
// those instantiations carry the 32-bit copy-out
// N_SURE > 0 and `sure`: the caller knows that the lane's first N_SURE records all have a bucket (regular level-1 tiles without a
// zero key); a wave in which every lane says so stages them without a test per record, the lbase reads issued together
template <int PER_THREAD, int DBG = 0, bool FLAT = false, bool KF32_POSSIBLE = true, int N_SURE = 0, class LDS>
__device__ __forceinline__ void scatter_stage_copy(LDS &L, const uint64_t (&rec)[PER_THREAD], const uint32_t (&br)[PER_THREAD],
                                                   const uint32_t (&my_gbase)[LDS::kBpt], uint32_t n_buckets, uint64_t *__restrict__ out,
                                                   uint64_t cap, uint32_t b1_of_bucket0, bool bucket_is_b1, const PartGeom &G,
                                                   const PartStore &P, Counters *ctr, uint32_t stride = 1u, // bucket b lives at out + b * stride * cap
                                                   uint32_t flat_total = 0u, // FLAT: the records of the tile (scatter_reserve_scan)
                                                   bool sure = false)
{
	const int t = (int)fresh_tid();
	if (FLAT) { // the histogram has been consumed by the scan: every thread clears its own entries for the NEXT tile now, which then
		// starts ranking without a zeroing pass and its barrier (scatter_tile)
#pragma unroll
		for (int j = 0; j < LDS::kBpt; j++) L.hist[LDS::kBpt * t + j] = 0;
	}
	bool staged = false;
	if constexpr (N_SURE > 0) {
		if (__builtin_amdgcn_ballot_w64(!sure) == 0ull) { // (wave-uniform)
			uint32_t at[N_SURE];
#pragma unroll
			for (int u = 0; u < N_SURE; u++) at[u] = L.lbase[br[u] >> 16];
#pragma unroll
			for (int u = 0; u < N_SURE; u++) L.stage[at[u] + (br[u] & 0xFFFFu)] = rec[u];
			staged = true;
		}
	}
	if (!staged) {
#pragma unroll
		for (int u = 0; u < PER_THREAD; u++) {
			if ((br[u] >> 16) < (uint32_t)LDS::kMaxB) L.stage[L.lbase[br[u] >> 16] + (br[u] & 0xFFFFu)] = rec[u];
			if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0); // four lbase reads in flight are enough; more costs VGPRs the callers do not have
		}
	}
	// one descriptor per bucket for the copy-out: global offset | records in this tile | first staged index
	// (FLAT: the global offset alone; lbase is read next to it)
#pragma unroll
	for (int j = 0; j < LDS::kBpt; j++) {
		const uint32_t b = LDS::kBpt * t + j;
		if (FLAT) L.desc[b] = (typename LDS::Desc)(my_gbase[j] - L.lbase[b]); // global place of staged record p of bucket b = desc[b] + p (modulo 2^32)
		else L.desc[b] = (typename LDS::Desc)(((uint64_t)my_gbase[j] << 32) | (L.hist[b] << 16) | L.lbase[b]);
	}
	lds_barrier();
	if (FLAT) {
		if (DBG != 2) {
			const uint32_t total = flat_total;
#pragma unroll
			for (int u = 0; u < PER_THREAD; u++) {
				const uint32_t p = (uint32_t)u * LDS::kThreads + (uint32_t)t;
				if (p >= total) continue;
				const uint64_t rcd = L.stage[p];
				const uint32_t b = (uint32_t)(rcd >> (6 + kRegionBits + G.l2_shift)) & (n_buckets - 1u);
				const uint64_t off = (uint32_t)((uint32_t)L.desc[b] + p);
				if (DBG == 3) {
					out[(uint64_t)blockIdx.x * 4096u + (((uint64_t)b * cap + off) & 4095ull)] = rcd;
				} else if (off < cap) {
					// KFREQ, direct blocks: all a block's build needs of a record is the key's place in the block -- 16 bits instead of 64
					if (G.kf == 2u) reinterpret_cast<uint16_t *>(out)[(uint64_t)b * cap + off] = (uint16_t)(rcd >> 6);
					else out[(uint64_t)b * cap + off] = rcd;
				} else { // the bucket is full: records beyond its capacity go to the overflow list
					const uint32_t b1 = bucket_is_b1 ? b : b1_of_bucket0;
					push_overflow(P, record_key(rcd, b1, G), (uint32_t)(rcd >> 3) & 7u, (uint32_t)rcd & 7u, ctr);
				}
			}
		}
		// (no barrier here: the next tile ranks into the histogram cleared above and meets three barriers -- ranks complete, inside
		// the scan, after the scan -- before it writes lbase, the stage buffer or desc again; the global stores keep draining)
		return;
	}
	// copy-out: wave w takes buckets w, w+16, ...  Lane l fetches the descriptor of the wave's l-th
	// bucket in ONE LDS read; the loop then broadcasts descriptor k with readlane, so every per-bucket
	// quantity is scalar and an iteration is an LDS read of the staged run plus one coalesced store.
	const uint32_t lane = t & 63, wave = __builtin_amdgcn_readfirstlane((uint32_t)t >> 6); // (scalar: every per-bucket address below is scalar arithmetic)
	constexpr uint32_t kWaves = LDS::kThreads / 64;
	const uint32_t per_wave = (DBG == 2) ? 0u : (n_buckets + kWaves - 1u - wave) / kWaves; // buckets wave + kWaves * k < n_buckets
	const uint32_t mine = wave + kWaves * lane;
	const uint64_t d = (lane < per_wave) ? (uint64_t)L.desc[mine] : 0ull; // per_wave <= kMaxBuckets / kWaves = 64
	const uint32_t d_lo = (uint32_t)d, d_hi = (uint32_t)(d >> 32);
	// (Unrolling this loop by four so that the staged runs are read back to back was measured slower:
	// 7.2 / 6.9 ms against 7.1 / 6.4 ms for level 1 / level 2.)
	for (uint32_t kk = 0; kk < per_wave; kk++) {
		const uint32_t k = __builtin_amdgcn_readfirstlane(kk);
		const uint32_t lo = __builtin_amdgcn_readlane(d_lo, k), dst = __builtin_amdgcn_readlane(d_hi, k);
		const uint32_t n = lo >> 16, src = lo & 0xFFFFu;
		if (n == 0) continue;
		const uint32_t b = wave + kWaves * k;
		if (KF32_POSSIBLE && G.kf == 2u) {
			// KFREQ, direct blocks: a level-1 record is (place in the bucket) << 6 | 4 -- 32 bits, its high word zero -- and travels
			// as 32 bits: half the level-1 store written here and read by level 2 (kSubStores == 1: `out` is the store itself)
			static_assert(kSubStores == 1, "the 32-bit level-1 store is addressed without sub-stores");
			uint32_t *o32 = reinterpret_cast<uint32_t *>(out) + (uint64_t)b * cap + dst;
			if ((uint64_t)dst + n <= cap) {
				typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
				if (n >= 4u) { // four records per lane and store instruction; the lane whose four would reach past the run takes the run's
					// LAST four (a few records written twice with the same value: no narrow store instructions behind the wide one)
					for (uint32_t i = 4u * lane; i < n; i += 256u) {
						const uint32_t j = min(i, n - 4u);
						const u32x4_a4 v = {(uint32_t)L.stage[src + j], (uint32_t)L.stage[src + j + 1u], (uint32_t)L.stage[src + j + 2u], (uint32_t)L.stage[src + j + 3u]};
						*reinterpret_cast<u32x4_a4 *>(o32 + j) = v;
					}
				} else if (lane < n) {
					o32[lane] = (uint32_t)L.stage[src + lane];
				}
			} else {
				for (uint32_t i = lane; i < n; i += 64) {
					const uint64_t rcd = L.stage[src + i];
					if ((uint64_t)dst + i < cap) o32[i] = (uint32_t)rcd;
					else push_overflow(P, record_key(rcd, bucket_is_b1 ? b : b1_of_bucket0, G), (uint32_t)(rcd >> 3) & 7u, (uint32_t)rcd & 7u, ctr);
				}
			}
			continue;
		}
		uint64_t *o = out + (uint64_t)b * stride * cap + dst;
		if (DBG == 3) { // timing experiment: same instruction stream, stores land in a 32 KiB window per workgroup (no HBM write traffic)
			for (uint32_t i = lane; i < n; i += 64) out[(uint64_t)blockIdx.x * 4096u + (((uint64_t)b * stride * cap + dst + i) & 4095ull)] = L.stage[src + i];
		} else if ((uint64_t)dst + n <= cap) {
			// two records per lane and store instruction (16 bytes, 8-byte aligned): the memory pipe charges per instruction,
			// whatever its lane count (level 1: 5.68 -> 5.54 ms against one record per lane, profiles/ab_bench.sh)
			typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
			if (n >= 2u) { // the lane left with the odd last record takes the run's last TWO (round 5: one store instruction per 128 records
				// whatever the run's length -- level 1 4.11 -> 3.92 ms in the pipelined form, profiles/r05_l1_overlap_store_ab.txt)
				for (uint32_t i = 2u * lane; i < n; i += 128u) {
					const uint32_t j = min(i, n - 2u);
					const uint64_t a = L.stage[src + j], b2 = L.stage[src + j + 1u];
					const u32x4_a8 v = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b2, (uint32_t)(b2 >> 32)};
					*reinterpret_cast<u32x4_a8 *>(o + j) = v;
				}
			} else if (lane == 0u) {
				o[0] = L.stage[src];
			}
		} else { // the bucket is full: records beyond its capacity go to the overflow list
			for (uint32_t i = lane; i < n; i += 64) {
				const uint64_t rcd = L.stage[src + i];
				if ((uint64_t)dst + i < cap) {
					o[i] = rcd;
				} else {
					const uint32_t b1 = bucket_is_b1 ? b : b1_of_bucket0;
					push_overflow(P, record_key(rcd, b1, G), (uint32_t)(rcd >> 3) & 7u, (uint32_t)rcd & 7u, ctr);
				}
			}
		}
	}
	// (level 1: no barrier here -- the next tile zeroes the histogram and packs its bytes, neither of which the
	// copy-out reads, and then meets its first barrier before any record is parked in the stage buffer again)
}

// whole scatter for records held in registers (level 2)
template <int PER_THREAD, int DBG = 0, bool FLAT = false, class LDS>
__device__ __forceinline__ void scatter_tile(LDS &L, const uint64_t (&rec)[PER_THREAD], uint32_t (&bkt)[PER_THREAD],
                                             uint32_t n_buckets, uint32_t *__restrict__ cnt, uint64_t *__restrict__ out,
                                             uint64_t cap, uint32_t b1_of_bucket0, bool bucket_is_b1, const PartGeom &G,
                                             const PartStore &P, Counters *ctr)
{
	// (the histogram is zero: cleared by the kernel before its first tile and by scatter_stage_copy of the tile before)
#pragma unroll
	for (int u = 0; u < PER_THREAD; u++) bkt[u] = (bkt[u] << 16) | ((bkt[u] != 0xFFFFu) ? atomicAdd(&L.hist[bkt[u]], 1u) : 0u);
	lds_barrier();
	if (DBG == 1) { // timing experiment: loads + ranking only
		uint64_t x = 0;
#pragma unroll
		for (int u = 0; u < PER_THREAD; u++) x ^= rec[u] + bkt[u];
		if (x == 0x1234567u) out[threadIdx.x] = x;
		return;
	}
	uint32_t my_gbase[LDS::kBpt];
	const uint32_t all = scatter_reserve_scan(L, n_buckets, cnt, my_gbase);
	static_assert(FLAT, "scatter_tile is level 2's: the flat copy-out clears the histogram for the next tile");
	scatter_stage_copy<PER_THREAD, DBG, FLAT, false>(L, rec, bkt, my_gbase, n_buckets, out, cap, b1_of_bucket0, bucket_is_b1, G, P, ctr, 1u, all); // (level 2: FLAT)
}

// ---- lean extraction for the partition path -----------------------------------------------------
// Same semantics as load_lane_window/next_triple (dbgk_kernels.h; DBGgraph.cpp:64-98) with the
// per-position work cut down: forward and reverse k-mers ROLL by one base per position as in the
// reference (:71-73) instead of being re-derived from the window; the three per-position predicates
// (window inside one read, has left / right neighbour) are computed once per 16 positions as bit
// masks with log-step sliding ORs over the boundary bitmap; nothing branches per position.

// bit i of the result = OR of bits i .. i+w-1 of x (w wave-uniform, 0..63)
__device__ __forceinline__ uint64_t sliding_or(uint64_t x, uint32_t w)
{
	uint64_t res = 0ull, cur = x;
	uint32_t done = 0;
#pragma unroll
	for (uint32_t j = 0; j < 6; j++) {
		if (w & (1u << j)) {
			res |= cur >> done;
			done += 1u << j;
		}
		cur |= cur >> (1u << j);
	}
	return res;
}

struct Chunk16 {
	uint64_t kbit, rc;   // forward / reverse-complement k-mer of the window at position 0
	uint32_t nb;         // bases k .. k+15: right neighbour of position i = entering base of position i+1
	uint32_t lw;         // bases -1 .. 14: left neighbour of position i
	uint32_t valid, has_l, has_r; // 16-bit masks, bit i <-> position p0 + i
};

__device__ __forceinline__ uint4 load_ascii16(const char *__restrict__ bases, uint64_t n_bases, uint64_t chunk)
{
	const uint64_t off = chunk * 16u;
	if (off + 16u <= n_bases) return *reinterpret_cast<const uint4 *>(bases + off);
	uint32_t w[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u}; // beyond the end: 'A'
	for (uint64_t i = 0; off + i < n_bases && i < 16; i++) {
		const uint32_t c = (uint8_t)bases[off + i];
		w[i >> 2] = (w[i >> 2] & ~(0xFFu << ((i & 3u) * 8u))) | (c << ((i & 3u) * 8u));
	}
	return make_uint4(w[0], w[1], w[2], w[3]);
}

// Raw (not yet decoded) inputs of the 16 positions of one lane.  Kept separate from the decode so
// that the loads of tile i+1 can be issued before tile i is processed: nothing in load_raw waits.
// Lane l of a wave owns chunk c0+l; the two following packed words come from lanes l+1, l+2 by
// shuffle, and from the HALO chunks c0+64, c0+65 (loaded by lanes 0 and 1) at the wave's end.
struct RawChunk {
	uint4 a0;        // 16 ASCII bases of this lane's chunk
	uint4 halo;      // lanes 0,1: chunks c0+64, c0+65
	uint32_t prevb;  // lane 0: the byte before the wave's first base
	uint32_t s0, s1, s2, d0, d1, d2; // boundary bitmap words covering positions p0 .. p0+63 (+31)
};

template <bool HAS_DEAD, bool GUARDED>
__device__ __forceinline__ RawChunk load_raw(const ReadBatch &rb, uint64_t chunk, uint64_t n_chunks)
{
	RawChunk r;
	const uint32_t lane = fresh_tid() & 63u;
	const uint64_t halo_chunk = chunk - lane + 64u + lane; // == chunk + 64 for lanes 0,1
	r.a0 = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
	r.halo = r.a0;
	r.prevb = 0x41u;
	if (rb.packed) { // a 2-bit packed batch (wave-uniform): .x holds the packed word itself, prevb the code of the base before
		r.a0.x = r.halo.x = r.prevb = 0u;
		if (GUARDED) {
			if (chunk < n_chunks) r.a0.x = packed_word(rb, chunk);
			if (lane < 2u && halo_chunk < n_chunks) r.halo.x = packed_word(rb, halo_chunk);
		} else {
			r.a0.x = rb.packed[chunk];
			if (lane < 2u) r.halo.x = rb.packed[halo_chunk];
		}
		if (lane == 0u && chunk > 0u && (!GUARDED || chunk * 16u - 1u < rb.n_bases)) r.prevb = rb.packed[chunk - 1u] & 3u;
	} else if (GUARDED) {
		if (chunk < n_chunks) r.a0 = load_ascii16(rb.bases, rb.n_bases, chunk);
		if (lane < 2u && halo_chunk < n_chunks) r.halo = load_ascii16(rb.bases, rb.n_bases, halo_chunk);
	} else {
		r.a0 = *reinterpret_cast<const uint4 *>(rb.bases + chunk * 16u);
		if (lane < 2u) r.halo = *reinterpret_cast<const uint4 *>(rb.bases + halo_chunk * 16u);
	}
	if (!rb.packed && lane == 0u && chunk > 0u && (!GUARDED || chunk * 16u - 1u < rb.n_bases)) r.prevb = (uint8_t)rb.bases[chunk * 16u - 1u];
	const uint64_t p0 = chunk * 16u;
	const uint64_t wi = p0 >> 5; // the bitmaps are padded by 4 words, safe for every chunk < n_chunks
	r.s0 = r.s1 = r.s2 = r.d0 = r.d1 = r.d2 = 0u;
	if (!GUARDED || chunk < n_chunks) {
		r.s0 = rb.start_bits[wi];
		r.s1 = rb.start_bits[wi + 1];
		r.s2 = rb.start_bits[wi + 2];
		if (HAS_DEAD) {
			r.d0 = rb.dead_bits[wi];
			r.d1 = rb.dead_bits[wi + 1];
			r.d2 = rb.dead_bits[wi + 2];
		}
	}
	return r;
}

__device__ __forceinline__ uint64_t bits64_from_words(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t sh)
{
	const uint64_t lo = (uint64_t)w0 | ((uint64_t)w1 << 32);
	return sh ? (lo >> sh) | ((uint64_t)w2 << (64u - sh)) : lo;
}

template <bool HAS_DEAD>
__device__ __forceinline__ Chunk16 decode_chunk16(const RawChunk &raw, const ReadBatch &rb, uint64_t chunk)
{
	Chunk16 c;
	const uint32_t k = (uint32_t)rb.k;
	const uint32_t lane = fresh_tid() & 63u;
	const uint64_t p0 = chunk * 16u;
	// only lanes 0 and 1 hold halo chunks: broadcast their raw words and pack them on the scalar unit
	// instead of packing a dummy in all 64 lanes
	const uint4 h0 = make_uint4(__builtin_amdgcn_readlane(raw.halo.x, 0), __builtin_amdgcn_readlane(raw.halo.y, 0),
	                            __builtin_amdgcn_readlane(raw.halo.z, 0), __builtin_amdgcn_readlane(raw.halo.w, 0));
	const uint4 h1 = make_uint4(__builtin_amdgcn_readlane(raw.halo.x, 1), __builtin_amdgcn_readlane(raw.halo.y, 1),
	                            __builtin_amdgcn_readlane(raw.halo.z, 1), __builtin_amdgcn_readlane(raw.halo.w, 1));
	uint32_t w0, hw0, hw1, pb;
	if (rb.packed) { // (wave-uniform) the words came packed
		w0 = raw.a0.x; hw0 = h0.x; hw1 = h1.x;
		pb = __builtin_amdgcn_readlane(raw.prevb, 0);
	} else {
		w0 = pack16_ascii(raw.a0, rb.other_seen);
		hw0 = pack16_ascii(h0, rb.other_seen); hw1 = pack16_ascii(h1, rb.other_seen);
		pb = code_ascii(__builtin_amdgcn_readlane(raw.prevb, 0), rb.other_seen);
	}
	uint32_t w1 = __shfl_down(w0, 1, 64), w2 = __shfl_down(w0, 2, 64);
	if (lane == 63u) { w1 = hw0; w2 = hw1; }
	if (lane == 62u) w2 = hw0;
	uint32_t prev = __shfl_up(w0, 1, 64) & 3u; // last base of the previous lane's chunk
	if (lane == 0u) prev = pb;
	if (chunk == 0u) prev = 0u;
	const uint64_t S = bits64_from_words(raw.s0, raw.s1, raw.s2, (uint32_t)(p0 & 31u));
	const uint64_t hi = ((uint64_t)w0 << 32) | w1, mid = ((uint64_t)w1 << 32) | w2;
	const uint32_t sh = 64u - 2u * k;                 // wave-uniform
	c.kbit = hi >> sh;
	c.rc = revcomp_kbit(c.kbit, (int)k);
	c.nb = (sh < 32u) ? (uint32_t)(mid >> sh) : (uint32_t)(hi >> (sh - 32u));
	c.lw = (prev << 30) | (w0 >> 2);
	// window i is inside one read iff no read starts at positions i+1 .. i+k-1
	uint64_t bad = sliding_or(S >> 1, k - 1u);
	uint64_t no_r = S >> k;
	if (HAS_DEAD) {
		const uint64_t D = bits64_from_words(raw.d0, raw.d1, raw.d2, (uint32_t)(p0 & 31u));
		bad |= sliding_or(D, k);
		no_r |= D >> k;
	}
	// positions whose window / right neighbour lie inside the buffer
	const uint64_t room = rb.n_bases > p0 ? rb.n_bases - p0 : 0ull;
	const uint32_t nv = room >= k ? (uint32_t)(room - k + 1u < 16u ? room - k + 1u : 16u) : 0u;
	const uint32_t nr = room > k ? (uint32_t)(room - k < 16u ? room - k : 16u) : 0u;
	c.valid = ~(uint32_t)bad & ((1u << nv) - 1u);
	c.has_r = ~(uint32_t)no_r & ((1u << nr) - 1u);
	c.has_l = ~(uint32_t)S & 0xFFFFu & (p0 ? 0xFFFFu : 0xFFFEu);
	return c;
}

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

// ---- level 2: split every level-1 bucket into its n2 final buckets -----------------------------
// Inbox entry e = (s * B + j) * n_sub + x: sub-store x of level-1 bucket j of this shard's slot range
// as extracted by rank s.  The tile list is flattened OWN-BUCKET-MAJOR, f = (j * n_ranks + s) * n_sub + x,
// so that a range of own buckets [j0, j1) is a contiguous range of tiles (the finalize runs level 2 and the build in chunks of
// buckets on two streams).  k_plan_l2: tile_prefix[f] = number of tiles in flat entries < f
// (entries with j >= nb_own are empty).  One workgroup, entries strided.
constexpr int kMaxInboxEntries = 16384; // n_ranks * B * n_sub <= (n1 + n_ranks) * n_sub

__device__ __forceinline__ uint32_t flat_to_entry(const PartGeom &G, uint32_t f, uint32_t &j_out)
{
	const uint32_t per_j = G.n_ranks * G.n_sub;
	const uint32_t j = f / per_j, rem = f - j * per_j, src = rem / G.n_sub, x = rem - src * G.n_sub;
	j_out = j;
	return (src * G.B + j) * G.n_sub + x;
}

__global__ __launch_bounds__(kMaxBuckets) void k_plan_l2(PartGeom G, PartStore P, uint32_t *__restrict__ tile_prefix)
{
	__shared__ uint32_t tot[kMaxBuckets / 64];
	__shared__ uint32_t carry;
	const int t = (int)fresh_tid(), lane = t & 63, wave = t >> 6;
	const uint32_t n_entries = G.n_ranks * G.B * G.n_sub;
	if (t == 0) carry = 0;
	__syncthreads();
	for (uint32_t base = 0; base < n_entries; base += kMaxBuckets) {
		const uint32_t e = base + (uint32_t)t; // flat index f
		uint32_t v = 0, own_j = 0;
		const uint32_t entry = e < n_entries ? flat_to_entry(G, e, own_j) : 0u;
		if (e < n_entries && own_j < G.nb_own) {
			const uint64_t filled = P.inbox_cnt[entry] < G.cap1 ? P.inbox_cnt[entry] : G.cap1;
			uint64_t done = 0;
			if (P.l2_done) { // EARLY level 2: this round takes what has arrived since the last one
				done = P.l2_upto[entry];
				P.l2_done[entry] = (uint32_t)done;
				P.l2_upto[entry] = (uint32_t)filled;
			}
			v = (uint32_t)((filled - done + G.l2_records - 1) / G.l2_records);
		}
		uint32_t inc = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const uint32_t n = __shfl_up(inc, off, 64);
			if (lane >= off) inc += n;
		}
		if (lane == 63) tot[wave] = inc;
		__syncthreads();
		uint32_t before = carry;
		for (int w = 0; w < wave; w++) before += tot[w];
		if (e < n_entries) tile_prefix[e] = before + inc - v;
		__syncthreads();
		if (t == kMaxBuckets - 1) carry = before + inc;
		__syncthreads();
	}
	if (t == 0) tile_prefix[n_entries] = carry;
}

// Persistent workgroups (one per CU) walk the flattened tile list; the records of tile i+1 are
// loaded into registers before tile i is scattered, so HBM reads, the LDS work and the (undrained)
// stores of consecutive tiles overlap.
// KF32: KFREQ with direct blocks, 32-bit level-1 records (an instantiation of its own: the graph kernel keeps its registers).  The
// records stay 32 bits wide in the registers they are prefetched into -- widened right behind the load, every load waited for its
// predecessor (one s_waitcnt vmcnt(0) per record: level 2 of cfg4 9.0 instead of 5.0 ms)
template <bool KF32>
using L2RecIn = typename std::conditional<KF32, uint32_t, uint64_t>::type;

template <bool KF32 = false, int THREADS = 512>
__device__ __forceinline__ void l2_load_tile(const PartGeom &G, const PartStore &P, const uint32_t *__restrict__ tile_prefix,
                                             uint32_t g, uint32_t n_tiles, L2RecIn<KF32> (&rec)[16], uint32_t &b1_out, uint32_t &n_out)
{
	b1_out = 0;
	n_out = 0; // records of the tile (graph records only: the 32-bit KFREQ records are loaded one by one)
#pragma unroll
	for (int u = 0; u < 16; u++) rec[u] = ~(L2RecIn<KF32>)0;
	if (g >= n_tiles) return;
	uint32_t lo = 0, hi = G.n_ranks * G.B * G.n_sub; // last flat entry with tile_prefix[f] <= g (empty entries repeat the prefix: take the last)
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (tile_prefix[mid] <= g) lo = mid; else hi = mid;
	}
	uint32_t own_j;
	const uint32_t e = flat_to_entry(G, lo, own_j);
	b1_out = own_j; // own bucket index j
	const uint64_t filled = P.l2_done ? (uint64_t)P.l2_upto[e] : (P.inbox_cnt[e] < G.cap1 ? P.inbox_cnt[e] : G.cap1); // (EARLY level 2: as planned)
	constexpr uint32_t kL2Threads = (uint32_t)THREADS, kL2Records = 16u * (uint32_t)THREADS; // (== G.l2_records: the host plans with the kernel's tile)
	const uint64_t first = (uint64_t)(g - tile_prefix[lo]) * kL2Records + (P.l2_done ? P.l2_done[e] : 0u);
	const uint64_t *in = P.inbox + (uint64_t)e * G.cap1;
	const uint32_t tid = fresh_tid();
	if constexpr (KF32) { // KFREQ, direct blocks: 32-bit level-1 records (scatter_stage_copy); never all ones -- the low six bits are 4
		const uint32_t *in32 = reinterpret_cast<const uint32_t *>(P.inbox) + (uint64_t)e * G.cap1;
#pragma unroll
		for (int u = 0; u < 16; u++) {
			const uint64_t i = first + (uint64_t)u * kL2Threads + tid;
			if (i < filled) rec[u] = __builtin_nontemporal_load(in32 + i);
		}
		return;
	}
	// coalesced, two neighbouring records per lane and load instruction (16 bytes): the memory pipe charges per instruction,
	// and the order of a tile's records does not matter to the scatter
	typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
#pragma unroll
	for (int u = 0; u < 16; u += 2) {
		// (cap1 is a multiple of 16 and i even: record i + 1 lies inside the bucket's storage even when it is not a record -- the
		// consumer voids it, l2_fix_odd_tail.  ONE guarded load per pair: an else-branch with an 8-byte load made the compiler wait
		// for every load before issuing the next, level 2 alone 6.1 instead of 4.4 ms)
		const uint64_t i = first + (uint64_t)u * kL2Threads + 2u * tid;
		if (i < filled) {
			const u32x4_a8 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a8 *>(in + i));
			rec[u] = ((uint64_t)v.y << 32) | v.x;
			rec[u + 1] = ((uint64_t)v.w << 32) | v.z;
		}
	}
	n_out = (uint32_t)(filled - first < (uint64_t)kL2Records ? filled - first : (uint64_t)kL2Records);
}

// a tile with an odd number of records (the last tile of a level-1 bucket, at most): the second half of its last pair is not a record
template <int THREADS>
__device__ __forceinline__ void l2_fix_odd_tail(uint64_t (&rec)[16], uint32_t n)
{
	constexpr uint32_t kL2Threads = (uint32_t)THREADS;
	if (!(n & 1u)) return; // (wave-uniform)
	const uint32_t tid = fresh_tid();
#pragma unroll
	for (int u = 0; u < 16; u += 2)
		if ((uint32_t)u * kL2Threads + 2u * tid + 1u == n) rec[u + 1] = ~0ull;
}

// (two workgroups of eight waves per CU = four waves per SIMD: at most 128 VGPRs, said to the compiler for the 1024-bucket form -- at
// 130 it silently halved the occupancy and the pair of level 2 and build went from 9.2 to 12.8 ms; the forms for more buckets have
// always run one workgroup per CU)
template <int DBG = 0, int MAXB = kMaxBuckets, bool KF32 = false>
__global__ __launch_bounds__(l2_threads(MAXB), MAXB <= kMaxBuckets ? 4 : 1) void k_scatter_l2(PartGeom G, PartStore P, const uint32_t *__restrict__ tile_prefix,
                                                             Counters *__restrict__ ctr, uint32_t j0, uint32_t j1)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	ScatterLdsL2T<MAXB> &L = *reinterpret_cast<ScatterLdsL2T<MAXB> *>(lds_raw);
	const uint32_t first_tile = tile_prefix[j0 * G.n_ranks * G.n_sub], n_tiles = tile_prefix[j1 * G.n_ranks * G.n_sub]; // tiles of the own buckets [j0, j1)
	// XCD-aware tile order: workgroups b and b + 8 share an XCD (round-robin dispatch; speed only, never
	// correctness), so XCD x takes the x-th eighth of the tile range -- whole level-1 buckets -- and every
	// append point of a final bucket is fed through ONE L2: neighbouring 128-byte runs merge into full
	// lines there instead of leaving eight XCDs as partial-line writes.
	const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3, n_local = gridDim.x >> 3; // gridDim.x is a multiple of 8
	const uint32_t span = n_tiles - first_tile;
	const uint32_t lo_tile = first_tile + (uint32_t)(((uint64_t)span * xcd) >> 3), hi_tile = first_tile + (uint32_t)(((uint64_t)span * (xcd + 1u)) >> 3);
	{ // the histogram starts out zero; every tile clears it for the next one (scatter_stage_copy)
		const int t = (int)fresh_tid();
#pragma unroll
		for (int j = 0; j < ScatterLdsL2T<MAXB>::kBpt; j++) L.hist[ScatterLdsL2T<MAXB>::kBpt * t + j] = 0;
		lds_barrier();
	}
	L2RecIn<KF32> nxt[16];
	uint32_t nxt_b1, nxt_n; // own level-1 bucket index j = b1 - b_lo; records of the tile
	l2_load_tile<KF32, l2_threads(MAXB)>(G, P, tile_prefix, lo_tile + local, hi_tile, nxt, nxt_b1, nxt_n);
	for (uint32_t g = lo_tile + local; g < hi_tile; g += n_local) {
		uint64_t rec[16];
		uint32_t bkt[16];
		const uint32_t j = nxt_b1;
		if constexpr (!KF32) l2_fix_odd_tail<l2_threads(MAXB)>(nxt, nxt_n);
#pragma unroll
		for (int u = 0; u < 16; u++) {
			if constexpr (KF32) rec[u] = nxt[u] == ~0u ? ~0ull : (uint64_t)nxt[u];
			else rec[u] = nxt[u];
			// an all-ones word is never a record: the neighbour fields only take the values 0..4
			bkt[u] = (rec[u] == ~0ull) ? 0xFFFFu : ((uint32_t)(rec[u] >> (6 + kRegionBits + G.l2_shift)) & (G.n2 - 1u));
		}
		l2_load_tile<KF32, l2_threads(MAXB)>(G, P, tile_prefix, g + n_local, hi_tile, nxt, nxt_b1, nxt_n); // in flight during the scatter below
		// (KFREQ, direct blocks: the final buckets hold 16-bit records -- the same index arithmetic on a quarter of the bytes)
		uint64_t *out = G.kf == 2u ? reinterpret_cast<uint64_t *>(reinterpret_cast<uint16_t *>(P.l2) + (uint64_t)j * G.n2 * G.cap2)
		                           : P.l2 + (uint64_t)j * G.n2 * G.cap2;
		scatter_tile<16, DBG, true>(L, rec, bkt, G.n2, P.cnt2 + (uint64_t)j * G.n2, out, G.cap2, G.b_lo + j, false, G, P, ctr);
	}
}

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

// ---- overflow triples through the global-atomic path --------------------------------------------
__global__ __launch_bounds__(kBlock) void k_insert_triples(const Node *__restrict__ in, const unsigned long long *__restrict__ n_ptr,
                                                           uint64_t cap, TableRef T, Counters *__restrict__ ctr)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint64_t n = *n_ptr < cap ? *n_ptr : cap;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
		const uint64_t key = in[i].kmer;
		const uint32_t lb = (uint32_t)in[i].links & 0xFFu, rb = (uint32_t)(in[i].links >> 8) & 0xFFu;
		uint64_t slot = fast_mod(hash_code(key), T.magic);
		const uint4 f = *reinterpret_cast<const uint4 *>(&T.nodes[slot]);
		Node first;
		first.kmer = ((uint64_t)f.y << 32) | f.x;
		first.links = ((uint64_t)f.w << 32) | f.z;
		uint64_t guess;
		const uint64_t s = find_or_claim(T, key, slot, first, guess, n_new, n_conf);
		if (s == ~0ull) { full = true; continue; }
		links_cas_observe(reinterpret_cast<unsigned long long *>(&T.nodes[s].links), guess, lb, rb);
	}
	const unsigned long long a = block_sum(n_new, red);
	const unsigned long long b = block_sum(n_conf, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
	}
	if (full) atomicOr(&ctr->error, 1u);
}

// merge the spill nodes: count known only on the device
__global__ __launch_bounds__(kBlock) void k_merge_spill(const Node *__restrict__ in, const unsigned long long *__restrict__ n_ptr,
                                                        uint64_t cap, TableRef T, Counters *__restrict__ ctr)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint64_t n = *n_ptr < cap ? *n_ptr : cap;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
		const uint64_t key = in[i].kmer, add = in[i].links;
		uint64_t slot = fast_mod(hash_code(key), T.magic);
		const uint4 f = *reinterpret_cast<const uint4 *>(&T.nodes[slot]);
		Node first;
		first.kmer = ((uint64_t)f.y << 32) | f.x;
		first.links = ((uint64_t)f.w << 32) | f.z;
		uint64_t guess;
		const uint64_t s = find_or_claim(T, key, slot, first, guess, n_new, n_conf);
		if (s == ~0ull) { full = true; continue; }
		links_cas_merge(reinterpret_cast<unsigned long long *>(&T.nodes[s].links), guess, add);
	}
	const unsigned long long a = block_sum(n_new, red);
	const unsigned long long b = block_sum(n_conf, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
	}
	if (full) atomicOr(&ctr->error, 1u);
}

// ---- sharded tables (multi-GPU): probing confined to the shard's slot range -----------------------
// Like find_or_claim (dbgk_kernels.h) but on the shard-local array and without wrap-around: returns
// the local index, or ~0 when the probe reaches slot_hi (the node then belongs to the next shard).
__device__ __forceinline__ uint64_t find_or_claim_range(Node *nodes, uint64_t slot_lo, uint64_t slot_hi, uint64_t key, uint64_t slot,
                                                        uint64_t &links_guess, unsigned long long &n_new, unsigned long long &n_conf)
{
	for (; slot < slot_hi; slot++) {
		const uint64_t idx = slot - slot_lo;
		const uint4 v = *reinterpret_cast<const uint4 *>(&nodes[idx]);
		uint64_t seen = ((uint64_t)v.y << 32) | v.x;
		uint64_t links = ((uint64_t)v.w << 32) | v.z;
		if (seen == 0ull) {
			seen = atomicCAS(reinterpret_cast<unsigned long long *>(&nodes[idx].kmer), 0ull, (unsigned long long)key);
			if (seen == 0ull) {
				n_new++;
				links_guess = 0ull;
				return idx;
			}
			links = 0ull;
		}
		if (seen == key) {
			links_guess = links;
			return idx;
		}
		n_conf++;
	}
	return ~0ull;
}

// Merge nodes (is_triple == 0: {key, links}) or single observations (is_triple == 1: {key, lb | rb << 8})
// into this shard.  A node whose home slot lies outside the shard is skipped unless start_foreign_at_lo
// (nodes handed over by the previous shard continue their probe at this shard's first slot).  Nodes
// whose probe runs off the end of the shard are appended to P.outgoing for the next rank.
__global__ __launch_bounds__(kBlock) void k_merge_sharded(const Node *__restrict__ in, const unsigned long long *__restrict__ n_ptr,
                                                          uint64_t n_direct, uint64_t cap, int is_triple, int start_foreign_at_lo,
                                                          PartGeom G, PartStore P, Node *__restrict__ table, Counters *__restrict__ ctr)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n_new = 0, n_conf = 0;
	uint64_t n = n_ptr ? (uint64_t)*n_ptr : n_direct;
	if (n > cap) n = cap;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
		const uint64_t key = in[i].kmer;
		uint64_t add = in[i].links;
		if (is_triple) add = links_observe(0ull, (uint32_t)add & 0xFFu, (uint32_t)(add >> 8) & 0xFFu);
		if (key == 0ull) { // key-0 node of another shard: folded into this handle's side node (an all-zero entry -- an empty slot of a whole table given as the list -- adds nothing)
			if (add) links_cas_merge(&ctr->polyA_links, 0ull, add);
			continue;
		}
		const uint64_t home = fast_mod(hash_code(key), G.magic);
		const bool mine = home >= G.slot_lo && home < G.slot_hi;
		if (!mine && !start_foreign_at_lo) continue;
		uint64_t guess;
		const uint64_t idx = find_or_claim_range(table, G.slot_lo, G.slot_hi, key, mine ? home : G.slot_lo, guess, n_new, n_conf);
		if (idx == ~0ull) {
			const unsigned long long j = atomicAdd(P.outgoing_n, 1ull);
			if (j < P.outgoing_cap) {
				P.outgoing[j].kmer = key;
				P.outgoing[j].links = add;
			} else {
				atomicOr(&ctr->error, 2u);
			}
			continue;
		}
		links_cas_merge(reinterpret_cast<unsigned long long *>(&table[idx].links), guess, add);
	}
	const unsigned long long a = block_sum(n_new, red);
	const unsigned long long b = block_sum(n_conf, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
	}
}

} // namespace dbgk
