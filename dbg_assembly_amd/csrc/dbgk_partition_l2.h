// dbgk_partition_l2.h -- part of the PARTITION engine (dbgk_partition.h includes it, in this order: _l1, _l2, _build): level 2 -- the tile plan and k_scatter_l2: every level-1 bucket into its final buckets
#pragma once

namespace dbgk {

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

} // namespace dbgk
