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

} // namespace dbgk

#include "dbgk_partition_l1.h"
#include "dbgk_partition_l2.h"
#include "dbgk_partition_build.h"

namespace dbgk {

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
