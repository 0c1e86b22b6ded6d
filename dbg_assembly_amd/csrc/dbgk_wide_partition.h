// dbgk_wide_partition.h -- the PARTITION engine's scheme (dbgk_partition.h) for 128-bit keys: WIDE handles whose
// input size is known (dbgk_config.expected_kmers > 0) build their table from radix-partitioned records instead of
// one random 32-byte node access per k-mer occurrence.
//
// RECORD (16 bytes):  { key.hi | [ q | slot_rel : r | lb : 3 | rb : 3 ] }  with hash128(key) = q * size + slot.
//   hash128 (include/dbgk_wide.h) is hash_code(lo ^ (hi ? hash_code(hi) : 0)) and hash_code is a bijection on u64,
//   so (hi, q, slot) determines the key: lo = hash_code_inverse(q * size + slot) ^ (hi ? hash_code(hi) : 0).  Two
//   different keys may share all 64 bits of hash128 (both words of the record's second half): the region build
//   compares the high words as well.
//   Keys whose low word is 0 never become records: the device table marks an empty slot with lo == 0
//   (dbgk_wide_kernels.h), so they go to the side table / the key-0 side node through the atomic path right away.
//
//   k_wide_scatter_l1     reads -> records, scattered into n1 <= 1024 level-1 buckets (slot >> r); 8192-record tiles
//   k_wide_l2_plan        tiles per level-1 bucket
//   k_wide_scatter_l2     every level-1 bucket -> n2 = 2^(r-11) final buckets (slot >> 11); XCD-aware tile order
//   k_wide_build_regions  one 2048-slot region at a time in LDS, emitted as 64 KiB of device nodes
//   afterwards            region spill-over nodes through k_wide_merge_nodes, bucket-overflow observations through
//                         k_wide_insert_obs (both global atomics, normally a handful)
// Once the table has been built the handle continues with the atomic kernels (k_wide_extract_insert) for anything
// pushed later: the built table IS a valid table of that engine.
#pragma once

#include "dbgk_partition.h"
#include "dbgk_wide_kernels.h"

namespace dbgk {

constexpr int kWRegionBits = 11;                 // 2048 slots = 64 KiB of 32-byte nodes per region
constexpr int kWRegionSlots = 1 << kWRegionBits;
constexpr int kWSpillSlots = 128;
constexpr int kWL1Threads = 1024;                // level 1: 8 positions per lane and tile, 8192 records of 16 bytes
constexpr int kWL1Records = kWL1Threads * 8;
constexpr int kWL2Threads = 512;                 // level 2: 4096-record tiles
constexpr int kWL2Records = kWL2Threads * 8;
constexpr int kWBuildThreads = 512;              // three workgroups per CU (51 KiB of LDS each)

struct WPartGeom {
	uint64_t size;
	ModMagic magic;
	uint32_t r;          // level-1 bucket = slot >> r
	uint32_t n1;         // ceil(size / 2^r) <= 1024
	uint32_t n2;         // 2^(r - 11)       <= 2048
	uint32_t n_regions;  // ceil(size / 2048)
	uint32_t chunk_buckets; // level-1 buckets whose final buckets the level-2 store holds at a time (level 2 and the build
	                        // walk the level-1 buckets chunk by chunk: the level-2 store is 1/8 of the records, not all)
	uint64_t cap1, cap2; // records per level-1 / final bucket
};

typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));

struct WPartStore {
	ull2 *l1;                     // [n1][cap1]
	uint32_t *cnt1;               // [n1] records appended (may exceed cap1: the excess went to ovf)
	ull2 *l2;                     // [chunk_buckets * n2][cap2]: the final buckets of ONE chunk of level-1 buckets at a time
	uint32_t *cnt2;               // [chunk_buckets * n2]
	dbgk_node32 *ovf;             // observations that found their bucket full: {hi, lo, lb, rb}
	dbgk_node32 *spill;           // nodes that probed past the end of their region
	unsigned long long *ovf_n;    // [0] observations, [1] spill nodes
	uint64_t ovf_cap, spill_cap;
};

__device__ __forceinline__ Key128 wide_record_key(ull2 rec, uint32_t b1, const WPartGeom &G)
{
	const uint64_t v = rec.y >> 6;
	const uint64_t slot = ((uint64_t)b1 << G.r) | (v & ((1ull << G.r) - 1ull));
	const uint64_t x = hash_code_inverse((v >> G.r) * G.size + slot);
	return Key128{rec.x, rec.x ? (x ^ hash_code(rec.x)) : x};
}

__device__ __forceinline__ void wide_push_overflow(const WPartStore &P, Key128 key, uint32_t lb, uint32_t rb, Counters *ctr)
{
	const unsigned long long i = atomicAdd(&P.ovf_n[0], 1ull);
	if (i < P.ovf_cap) {
		dbgk_node32 o;
		o.kmer_hi = key.hi;
		o.kmer_lo = key.lo;
		o.l_link = lb;
		o.r_link = rb;
		o.reserved = 0;
		P.ovf[i] = o;
	} else {
		atomicOr(&ctr->error, 2u);
	}
}

template <int THREADS, int PER_THREAD, int MAXB, bool FLAT_DESC>
struct WScatterLds {
	static constexpr int kThreads = THREADS;
	static constexpr int kRecords = THREADS * PER_THREAD;
	static constexpr int kMaxB = MAXB;
	static constexpr int kBpt = MAXB / THREADS;
	using Desc = typename std::conditional<FLAT_DESC, uint32_t, uint64_t>::type;
	ull2 stage[kRecords];
	uint32_t hist[MAXB + 64];        // + one dummy bin per lane (positions that yield no record)
	uint32_t lbase[MAXB];
	Desc desc[MAXB];
	uint32_t wave_tot[THREADS / 64];
};
using WL1Lds = WScatterLds<kWL1Threads, 8, 1024, false>;
template <int MAXB> using WL2Lds = WScatterLds<kWL2Threads, 8, MAXB, true>;

// ---- level 1 --------------------------------------------------------------------------------------
// One lane owns 16 consecutive base positions (as in k_wide_extract_insert), handled as two tiles of 8: the
// 160-bit window state stays in registers between them.  Per position nothing branches except the rare keys
// with lo == 0; positions without a record rank themselves in a per-lane dummy bin.
template <bool HAS_DEAD>
__global__ __launch_bounds__(kWL1Threads) void k_wide_scatter_l1(ReadBatch rb, WPartGeom G, WPartStore P, WTable T, Counters *__restrict__ ctr)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	WL1Lds &L = *reinterpret_cast<WL1Lds *>(lds_raw);
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint32_t k = (uint32_t)rb.k; // 1..63
	const uint64_t n_chunks = (rb.n_bases + 15u) >> 4;
	const uint32_t rmask = (1u << G.r) - 1u;
	const uint64_t inner_mask = (k > 1u) ? ((1ull << (k - 1u)) - 1ull) : 0ull; // k - 1 <= 62 bits
	const uint64_t km = (1ull << k) - 1ull;                                      // k <= 63
	const uint32_t top = 2u * k - 2u;                                            // bit position of a window's first base
	for (uint64_t c0 = (uint64_t)blockIdx.x * kWL1Threads; c0 < n_chunks; c0 += (uint64_t)gridDim.x * kWL1Threads) {
		const uint32_t tid = fresh_tid();
		const uint64_t chunk = c0 + tid;
		const bool live = chunk < n_chunks;
		const uint64_t p0 = chunk * 16u;
		// 80 bases from p0 (window of the last position + its right neighbour: 15 + 63 + 1), MSB first
		uint64_t A = 0, B = 0, C = 0, S0 = 0, S1 = 0, D0 = 0, D1 = 0;
		uint32_t prev = 0;
		if (live) {
			A = ((uint64_t)load_packed_chunk(rb.bases, rb.n_bases, chunk) << 32) | load_packed_chunk(rb.bases, rb.n_bases, chunk + 1);
			B = ((uint64_t)load_packed_chunk(rb.bases, rb.n_bases, chunk + 2) << 32) | load_packed_chunk(rb.bases, rb.n_bases, chunk + 3);
			C = (uint64_t)load_packed_chunk(rb.bases, rb.n_bases, chunk + 4) << 32;
			prev = chunk ? pack4_ascii((uint32_t)(uint8_t)rb.bases[p0 - 1]) >> 6 : 0u;
			load_bits128(rb.start_bits, p0, rb.n_bases, S0, S1);
			if (HAS_DEAD) load_bits128(rb.dead_bits, p0, rb.n_bases, D0, D1);
		}
		// reverse complement of the first window; it then ROLLS by one base per position like the forward k-mer
		Key128 rc;
		{
			Key128 f0;
			if (2u * k <= 64u) {
				f0.hi = 0ull;
				f0.lo = A >> (64u - 2u * k);
			} else {
				const uint32_t sh = 128u - 2u * k;
				f0.hi = A >> sh;
				f0.lo = (B >> sh) | (A << (64u - sh));
			}
			rc = dbgk_wide::revcomp(f0, (int)k);
		}
#pragma unroll 1
		for (uint32_t half = 0; half < 2u; half++) {
			L.hist[tid] = 0u; // kBpt == 1
			if (tid < 64u) L.hist[1024u + tid] = 0u;
			lds_barrier();
			uint32_t bkt[8];
#pragma unroll
			for (uint32_t i = 0; i < 8u; i++) {
				const uint64_t p = p0 + half * 8u + i;
				Key128 fwd;
				uint32_t right;
				if (2u * k <= 64u) {
					fwd.hi = 0ull;
					fwd.lo = A >> (64u - 2u * k);
					right = (2u * k < 64u) ? (uint32_t)(A >> (62u - 2u * k)) & 3u : (uint32_t)(B >> 62);
				} else {
					const uint32_t sh = 128u - 2u * k; // 2..62
					fwd.hi = A >> sh;
					fwd.lo = (B >> sh) | (A << (64u - sh));
					right = (uint32_t)(B >> (sh - 2u)) & 3u;
				}
				// the window [p, p + k) lies inside one read (no read starts at p+1 .. p+k-1) and inside the trimmed part
				bool valid = live && (p + k <= rb.n_bases) && (((S0 >> 1) | (S1 << 63)) & inner_mask) == 0ull;
				const bool has_left = p > 0 && !(S0 & 1ull);
				bool has_right = (p + k < rb.n_bases) && !((S0 >> k) & 1ull); // k <= 63
				if (HAS_DEAD) {
					valid = valid && (D0 & km) == 0ull;
					has_right = has_right && !((D0 >> k) & 1ull);
				}
				// canonical pick: tie -> forward (DBGgraph.cpp:80); reverse strand: (comp(right), comp(left)) (:82-97)
				const bool rev = rc.hi < fwd.hi || (rc.hi == fwd.hi && rc.lo < fwd.lo);
				const Key128 key{rev ? rc.hi : fwd.hi, rev ? rc.lo : fwd.lo};
				const uint32_t cl = has_left ? prev : 4u, cr = has_right ? right : 4u;
				const uint32_t lb = rev ? (cr == 4u ? 4u : 3u - cr) : cl, rbb = rev ? (cl == 4u ? 4u : 3u - cl) : cr;
				if (valid && key.lo == 0ull) wide_insert(T, key, lb, rbb, ctr, n_new, n_conf, full); // rare: k-mer ends in 32 A's
				const bool has_rec = valid && key.lo != 0ull;
				uint64_t q;
				const uint64_t slot = fast_divmod(hash_code(key.hi ? (key.lo ^ hash_code(key.hi)) : key.lo), G.magic, q);
				const uint64_t w = (q << (G.r + 6u)) | ((uint64_t)((uint32_t)slot & rmask) << 6) | (lb << 3) | rbb;
				const uint32_t b = has_rec ? (uint32_t)(slot >> G.r) : 1024u + (tid & 63u);
				L.stage[i * kWL1Threads + tid] = ull2{key.hi, w};
				bkt[i] = (b << 16) | atomicAdd(&L.hist[b], 1u);
				// slide by one base
				prev = (uint32_t)(A >> 62);
				A = (A << 2) | (B >> 62);
				B = (B << 2) | (C >> 62);
				C <<= 2;
				S0 = (S0 >> 1) | (S1 << 63);
				S1 >>= 1;
				if (HAS_DEAD) {
					D0 = (D0 >> 1) | (D1 << 63);
					D1 >>= 1;
				}
				// reverse complement of the next window: drop the complement of the base that left, the complement of
				// the entering base (`right`) arrives at the top
				rc.lo = (rc.lo >> 2) | (rc.hi << 62);
				rc.hi >>= 2;
				const uint64_t comp = (uint64_t)(3u - right);
				if (top >= 64u) rc.hi |= comp << (top - 64u); else rc.lo |= comp << top;
			}
			lds_barrier(); // histogram complete, parked records visible
			ull2 rec[8];
#pragma unroll
			for (uint32_t i = 0; i < 8u; i++) rec[i] = L.stage[i * kWL1Threads + fresh_tid()];
			uint32_t my_gbase[1];
			scatter_reserve_scan(L, G.n1, P.cnt1, my_gbase);
#pragma unroll
			for (uint32_t i = 0; i < 8u; i++)
				if ((bkt[i] >> 16) < 1024u) L.stage[L.lbase[bkt[i] >> 16] + (bkt[i] & 0xFFFFu)] = rec[i];
			{
				const uint32_t t = fresh_tid();
				L.desc[t] = ((uint64_t)my_gbase[0] << 32) | (L.hist[t] << 16) | L.lbase[t]; // <= 8192 records per tile: 16 bits each
			}
			lds_barrier();
			// copy-out: wave w takes buckets w, w + 16, ...; one 16-byte record per lane and store
			{
				const uint32_t t = fresh_tid(), lane = t & 63u, wave = t >> 6;
				constexpr uint32_t kWaves = kWL1Threads / 64;
				const uint32_t per_wave = (G.n1 + kWaves - 1u - wave) / kWaves; // <= 64
				const uint64_t d = (lane < per_wave) ? L.desc[wave + kWaves * lane] : 0ull;
				const uint32_t d_lo = (uint32_t)d, d_hi = (uint32_t)(d >> 32);
				for (uint32_t kk = 0; kk < per_wave; kk++) {
					const uint32_t kq = __builtin_amdgcn_readfirstlane(kk);
					const uint32_t lo = __builtin_amdgcn_readlane(d_lo, kq), dst = __builtin_amdgcn_readlane(d_hi, kq);
					const uint32_t n = lo >> 16, src = lo & 0xFFFFu;
					if (n == 0) continue;
					const uint32_t b = wave + kWaves * kq;
					ull2 *o = P.l1 + (uint64_t)b * G.cap1 + dst;
					for (uint32_t j = lane; j < n; j += 64u) {
						const ull2 rcd = L.stage[src + j];
						if ((uint64_t)dst + j < G.cap1) {
							o[j] = rcd;
						} else { // the bucket is full
							wide_push_overflow(P, wide_record_key(rcd, b, G), (uint32_t)(rcd.y >> 3) & 7u, (uint32_t)rcd.y & 7u, ctr);
						}
					}
				}
			}
			lds_barrier(); // the next tile parks its records in the stage buffer again
		}
	}
	__shared__ unsigned long long red[kWL1Threads / 64];
	const unsigned long long a = block_sum_n<kWL1Threads>(n_new, red);
	const unsigned long long b = block_sum_n<kWL1Threads>(n_conf, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
	}
	if (full) atomicOr(&ctr->error, 1u);
}

// ---- level 2 --------------------------------------------------------------------------------------
// tile_prefix[b] = tiles of the level-1 buckets before b (kWL2Records records per tile), tile_prefix[n1] = all
__global__ __launch_bounds__(1024) void k_wide_l2_plan(WPartGeom G, const uint32_t *__restrict__ cnt1, uint32_t *__restrict__ tile_prefix)
{
	__shared__ uint32_t wave_tot[16];
	const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
	const uint64_t filled = t < G.n1 ? (cnt1[t] < G.cap1 ? cnt1[t] : G.cap1) : 0ull;
	const uint32_t tiles = (uint32_t)((filled + kWL2Records - 1) / kWL2Records);
	uint32_t inc = tiles;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t n = __shfl_up(inc, off, 64);
		if ((int)lane >= off) inc += n;
	}
	if (lane == 63u) wave_tot[wave] = inc;
	__syncthreads();
	uint32_t run = inc - tiles;
	for (uint32_t w = 0; w < wave; w++) run += wave_tot[w];
	if (t < G.n1) tile_prefix[t] = run;
	if (t == G.n1 - 1u) tile_prefix[G.n1] = run + tiles;
}

template <int MAXB>
__global__ __launch_bounds__(kWL2Threads) void k_wide_scatter_l2(WPartGeom G, WPartStore P, const uint32_t *__restrict__ tile_prefix, Counters *__restrict__ ctr,
                                                                uint32_t j0, uint32_t j1) // the level-1 buckets [j0, j1) of this chunk
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	WL2Lds<MAXB> &L = *reinterpret_cast<WL2Lds<MAXB> *>(lds_raw);
	constexpr int kBpt = WL2Lds<MAXB>::kBpt;
	const uint32_t first_tile = tile_prefix[j0], span = tile_prefix[j1] - first_tile;
	// XCD-aware tile order (see k_scatter_l2): XCD x takes the x-th eighth of the tile range, so that the appends to a
	// final bucket come through ONE L2 and merge into full lines there
	const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3, n_local = gridDim.x >> 3; // gridDim.x is a multiple of 8
	const uint32_t lo_tile = first_tile + (uint32_t)(((uint64_t)span * xcd) >> 3), hi_tile = first_tile + (uint32_t)(((uint64_t)span * (xcd + 1u)) >> 3);
	for (uint32_t g = lo_tile + local; g < hi_tile; g += n_local) {
		uint32_t lo = j0, hi = j1; // last bucket with tile_prefix[b] <= g (empty buckets repeat the prefix: take the last)
		while (hi - lo > 1u) {
			const uint32_t mid = (lo + hi) >> 1;
			if (tile_prefix[mid] <= g) lo = mid; else hi = mid;
		}
		const uint32_t b1 = lo;
		const uint64_t filled = P.cnt1[b1] < G.cap1 ? P.cnt1[b1] : G.cap1;
		const uint64_t first = (uint64_t)(g - tile_prefix[b1]) * kWL2Records;
		const ull2 *in = P.l1 + (uint64_t)b1 * G.cap1;
		const uint32_t t = fresh_tid();
		ull2 rec[8];
		uint32_t bkt[8];
#pragma unroll
		for (int u = 0; u < 8; u++) { // coalesced: consecutive lanes read consecutive records
			const uint64_t i = first + (uint64_t)u * kWL2Threads + t;
			const bool have = i < filled;
			rec[u] = have ? __builtin_nontemporal_load(in + i) : ull2{0ull, 0ull};
			bkt[u] = have ? ((uint32_t)(rec[u].y >> (6 + kWRegionBits)) & (G.n2 - 1u)) : 0xFFFFu;
		}
#pragma unroll
		for (int j = 0; j < kBpt; j++) L.hist[kBpt * t + j] = 0u;
		lds_barrier();
#pragma unroll
		for (int u = 0; u < 8; u++) bkt[u] = (bkt[u] << 16) | ((bkt[u] != 0xFFFFu) ? atomicAdd(&L.hist[bkt[u]], 1u) : 0u);
		lds_barrier();
		uint32_t my_gbase[kBpt];
		scatter_reserve_scan(L, G.n2, P.cnt2 + (uint64_t)(b1 - j0) * G.n2, my_gbase);
#pragma unroll
		for (int u = 0; u < 8; u++)
			if ((bkt[u] >> 16) < (uint32_t)MAXB) L.stage[L.lbase[bkt[u] >> 16] + (bkt[u] & 0xFFFFu)] = rec[u];
#pragma unroll
		for (int j = 0; j < kBpt; j++) L.desc[kBpt * t + j] = my_gbase[j];
		lds_barrier();
		{
			// the final bucket of a staged record can be recomputed from the record: walk the sorted stage linearly, one
			// record per lane, every lane busy
			const uint32_t total = L.lbase[G.n2 - 1u] + L.hist[G.n2 - 1u];
			ull2 *out = P.l2 + (uint64_t)(b1 - j0) * G.n2 * G.cap2;
#pragma unroll
			for (int u = 0; u < 8; u++) {
				const uint32_t p = (uint32_t)u * kWL2Threads + t;
				if (p >= total) continue;
				const ull2 rcd = L.stage[p];
				const uint32_t b = (uint32_t)(rcd.y >> (6 + kWRegionBits)) & (G.n2 - 1u);
				const uint64_t off = (uint64_t)L.desc[b] + (p - L.lbase[b]);
				if (off < G.cap2) {
					out[(uint64_t)b * G.cap2 + off] = rcd;
				} else {
					wide_push_overflow(P, wide_record_key(rcd, b1, G), (uint32_t)(rcd.y >> 3) & 7u, (uint32_t)rcd.y & 7u, ctr);
				}
			}
		}
		lds_barrier(); // stage / hist are reused by the next tile; the global stores keep draining
	}
}

// ---- build ----------------------------------------------------------------------------------------
struct WBuildLds {
	unsigned long long ident[kWRegionSlots + kWSpillSlots]; // (record.y >> 6) + 1 = (q, slot_rel) + 1, 0 = empty
	unsigned long long hi1[kWRegionSlots + kWSpillSlots];   // key.hi + 1, published right after the claim of ident; 0 = not yet
	unsigned long long links[kWRegionSlots + kWSpillSlots];
	unsigned long long red[kWBuildThreads / 64];
	uint32_t next_region;
};

// Persistent workgroups pull regions from a cursor.  A slot is claimed with a CAS on `ident`; the winner then
// publishes hi + 1.  A record whose ident matches compares the high word too (two keys may share hash128): a lane
// that finds it unpublished looks again -- the winner's store follows its CAS in program order, so lanes of one
// wave never wait on each other, and another wave publishes without waiting for anybody.
__global__ __launch_bounds__(kWBuildThreads) void k_wide_build_regions(WPartGeom G, WPartStore P, WNode *__restrict__ table, Counters *__restrict__ ctr,
                                                                     unsigned int *__restrict__ cursor, uint32_t first_region, uint32_t n_regions)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	WBuildLds &L = *reinterpret_cast<WBuildLds *>(lds_raw);
	const uint32_t t = fresh_tid();
	constexpr uint32_t kNone = 0xFFFFFFFFu;
	constexpr uint32_t kAll = kWRegionSlots + kWSpillSlots;
	for (uint32_t i = t; i < kAll; i += kWBuildThreads) {
		L.ident[i] = 0ull;
		L.hi1[i] = 0ull;
		L.links[i] = 0ull;
	}
	uint32_t n_new = 0, n_conf = 0;
	for (;;) {
		if (t == 0) {
			const unsigned int kx = atomicAdd(cursor, 1u);
			L.next_region = kx < n_regions ? first_region + kx : kNone;
		}
		lds_barrier(); // also: the image is empty
		const uint32_t f = __builtin_amdgcn_readfirstlane(L.next_region);
		if (f == kNone) break;
		const uint32_t b1 = f >> (G.r - kWRegionBits);
		const uint64_t region_slot0 = (uint64_t)f << kWRegionBits;
		const uint32_t region_len = (uint32_t)((G.size - region_slot0 < (uint64_t)kWRegionSlots) ? G.size - region_slot0 : kWRegionSlots);
		const uint32_t fl = f - first_region; // the level-2 store holds this chunk's final buckets only
		const uint32_t filled = (uint32_t)(P.cnt2[fl] < G.cap2 ? P.cnt2[fl] : G.cap2);
		const ull2 *in = P.l2 + (uint64_t)fl * G.cap2;
		for (uint32_t base = 0; base < filled; base += kWBuildThreads) {
			const uint32_t i = base + t;
			const bool live = i < filled;
			const ull2 rec = live ? __builtin_nontemporal_load(in + i) : ull2{0ull, 0ull};
			const unsigned long long id = (rec.y >> 6) + 1ull, want_hi1 = rec.x + 1ull;
			const uint32_t lb = (uint32_t)(rec.y >> 3) & 7u, rb = (uint32_t)rec.y & 7u;
			const uint32_t home = (uint32_t)(rec.y >> 6) & (kWRegionSlots - 1u);
			uint32_t idx = home;
			bool probing = live, lost = false;
			while (probing) {
				unsigned long long cur = __hip_atomic_load(&L.ident[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				bool mine = false;
				if (cur == 0ull) {
					const unsigned long long prev = atomicCAS(&L.ident[idx], 0ull, id);
					if (prev == 0ull) {
						__hip_atomic_store(&L.hi1[idx], want_hi1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						mine = true;
						n_new += idx < region_len ? 1u : 0u; // spilled nodes are counted when they are merged
					}
					cur = prev == 0ull ? id : prev;
				}
				bool hit = mine, step = cur != id;
				if (cur == id && !mine) {
					const unsigned long long h1 = __hip_atomic_load(&L.hi1[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					hit = h1 == want_hi1;
					step = h1 != 0ull && !hit; // 0: its owner is about to publish the high word -- look at this slot again
				}
				idx += step ? 1u : 0u;
				n_conf += step ? 1u : 0u;
				lost = idx >= kAll; // region + spill area completely full
				probing = !hit && !lost;
			}
			// saturating +1 on the observed neighbour bytes, both dwords at once (see k_build_regions)
			const uint32_t dl = (lb != 4u) ? (1u << (24u - 8u * lb)) : 0u, dr = (rb != 4u) ? (1u << (24u - 8u * rb)) : 0u;
			bool pending = live && !lost;
			unsigned long long old = pending ? L.links[idx] : 0ull;
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
			if (live && lost) wide_push_overflow(P, wide_record_key(rec, b1, G), lb, rb, ctr);
		}
		lds_barrier();
		// emit the region: slot i of the table <- LDS slot i, the low word recomputed from (q, slot) and the high word;
		// the image is cleared on the way
		for (uint32_t i = t; i < kAll; i += kWBuildThreads) {
			const unsigned long long id = L.ident[i];
			WNode nd{0ull, 0ull, 0ull, 0ull};
			if (id) {
				const uint64_t v = id - 1ull;
				const uint64_t slot = ((uint64_t)b1 << G.r) | (v & ((1ull << G.r) - 1ull));
				const uint64_t hi = L.hi1[i] - 1ull;
				const uint64_t x = hash_code_inverse((v >> G.r) * G.size + slot);
				nd.hi1 = hi + 1ull;
				nd.lo = hi ? (x ^ hash_code(hi)) : x;
				nd.links = L.links[i];
				L.ident[i] = 0ull;
				L.hi1[i] = 0ull;
				L.links[i] = 0ull;
			}
			if (i < region_len) {
				ull2 *dst = reinterpret_cast<ull2 *>(&table[region_slot0 + i]);
				dst[0] = ull2{nd.hi1, nd.lo};
				dst[1] = ull2{nd.links, 0ull};
			} else if (id) { // probed past the region end: merged after all regions exist
				const unsigned long long j = atomicAdd(&P.ovf_n[1], 1ull);
				if (j < P.spill_cap) {
					dbgk_node32 o;
					o.kmer_hi = nd.hi1 - 1ull;
					o.kmer_lo = nd.lo;
					o.l_link = (uint32_t)nd.links;
					o.r_link = (uint32_t)(nd.links >> 32);
					o.reserved = 0;
					P.spill[j] = o;
				} else {
					atomicOr(&ctr->error, 2u);
				}
			}
		}
		// (the barrier at the top of the loop separates this clearing from the next region's inserts)
	}
	const unsigned long long a = block_sum_n<kWBuildThreads>(n_new, L.red);
	const unsigned long long b = block_sum_n<kWBuildThreads>(n_conf, L.red);
	if (t == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
	}
}

// bucket-overflow observations {hi, lo, lb, rb} through the atomic path; the count is known only on the device
__global__ __launch_bounds__(kBlock) void k_wide_insert_obs(const dbgk_node32 *__restrict__ in, const unsigned long long *__restrict__ n_ptr, uint64_t cap,
                                                            WTable T, Counters *__restrict__ ctr)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint64_t n = *n_ptr < cap ? *n_ptr : cap;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
		const dbgk_node32 nd = in[i];
		wide_insert(T, Key128{nd.kmer_hi, nd.kmer_lo}, nd.l_link, nd.r_link, ctr, n_new, n_conf, full);
	}
	const unsigned long long a = block_sum(n_new, red);
	const unsigned long long b = block_sum(n_conf, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
	}
	if (full) atomicOr(&ctr->error, 1u);
}

// spill nodes: like k_wide_merge_nodes with the count read on the device
__global__ __launch_bounds__(kBlock) void k_wide_merge_spill(const dbgk_node32 *__restrict__ in, const unsigned long long *__restrict__ n_ptr, uint64_t cap,
                                                             WTable T, Counters *__restrict__ ctr)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint64_t n = *n_ptr < cap ? *n_ptr : cap;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
		const dbgk_node32 nd = in[i];
		const Key128 key{nd.kmer_hi, nd.kmer_lo};
		uint64_t guess;
		const uint64_t s = wide_find_or_claim(T, key, guess, n_new, n_conf); // lo != 0: records never hold other keys
		if (s == ~0ull) { full = true; continue; }
		links_cas_merge(&T.nodes[s].links, guess, (uint64_t)nd.l_link | ((uint64_t)nd.r_link << 32));
	}
	const unsigned long long a = block_sum(n_new, red);
	const unsigned long long b = block_sum(n_conf, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
	}
	if (full) atomicOr(&ctr->error, 1u);
}

} // namespace dbgk
