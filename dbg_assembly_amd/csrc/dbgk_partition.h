// dbgk_partition.h -- PARTITION engine: make the table traffic sequential.
//
// The DIRECT engine pays one random 64-byte sector read plus one 64-byte atomic write-back per
// k-mer occurrence.  Here every occurrence becomes an 8-byte RECORD that is
// radix-partitioned by the table slot it will finally live in, and each 4096-slot REGION of the
// reference-layout table (slot = hash_code(key) % size, DBGgraph.cpp:167) is then built inside LDS
// by one workgroup and written out once with coalesced 16-byte stores:
//
//   k_extract_scatter   reads -> records, scattered into n1 level-1 buckets (slot >> r)
//   k_scatter_l2        every level-1 bucket -> n2 = 2^(r-12) final buckets (slot >> 12)
//   k_build_regions     one workgroup per final bucket: LDS open addressing on the region's own
//                       slots (ds_cmpst claims, LDS CAS saturating counters), emit the region
//   k_insert_triples    the few records that found no room in their bucket, through the
//                       global-atomic path; region spill-over nodes go through k_merge_nodes
//
// RECORD (64 bit):  [ q | slot_rel : r bits | lb : 3 | rb : 3 ]   with hash_code(key) = q*size + slot,
// slot_rel = slot & (2^r - 1).  hash_code is a bijection on u64 (inverse below), so the key itself
// is not stored: it is recomputed from (q, slot) once per DISTINCT node when the region is emitted.
#pragma once

#include "dbgk_kernels.h"

namespace dbgk {

constexpr int kRegionBits = 12;                 // 4096 slots = 64 KiB of nodes per region
constexpr int kRegionSlots = 1 << kRegionBits;
constexpr int kSpillSlots = 128;                // LDS slots past the region end: probe overflow
constexpr int kTileThreads = 1024;              // scatter kernels: 16 waves, one workgroup per CU
constexpr int kTileRecords = kTileThreads * 16; // 16384 records staged in LDS (128 KiB)
constexpr int kMaxBuckets = 1024;               // per-level fan-out limit (LDS histogram size)
constexpr int kBuildThreads = 512;

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
	uint32_t n2;         // 2^(r - 12)                  <= kMaxBuckets
	uint32_t n_final;    // ceil(size / 4096)
	uint64_t cap1;       // records per level-1 bucket
	uint64_t cap2;       // records per final bucket
};

struct PartStore {
	uint64_t *l1;                 // [n1 * cap1]
	uint64_t *l2;                 // [n_final_padded * cap2], final bucket f = slot >> 12
	uint32_t *cnt1;               // [n1]       records appended (may exceed cap1: excess went to ovf)
	uint32_t *cnt2;               // [n1 * n2]
	Node *ovf;                    // overflow triples {key, lb | rb << 8}
	Node *spill;                  // nodes that probed past the end of their region
	unsigned long long *ovf_n;    // [0] = overflow triples, [1] = spill nodes
	uint64_t ovf_cap, spill_cap;
};

__device__ __forceinline__ uint64_t make_record(uint64_t q, uint64_t slot, uint32_t r, uint32_t lb, uint32_t rb)
{
	return (q << (r + 6)) | ((slot & ((1ull << r) - 1ull)) << 6) | ((uint64_t)lb << 3) | (uint64_t)rb;
}

__device__ __forceinline__ uint64_t record_key(uint64_t rec, uint32_t b1, const PartGeom &G)
{
	const uint64_t v = rec >> 6;
	const uint64_t slot = ((uint64_t)b1 << G.r) | (v & ((1ull << G.r) - 1ull));
	return hash_code_inverse((v >> G.r) * G.size + slot);
}

__device__ __forceinline__ void push_overflow(const PartStore &P, uint64_t key, uint32_t lb, uint32_t rb, Counters *ctr)
{
	const unsigned long long i = atomicAdd(&P.ovf_n[0], 1ull);
	if (i < P.ovf_cap) {
		P.ovf[i].kmer = key;
		P.ovf[i].links = (uint64_t)lb | ((uint64_t)rb << 8);
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

// ---- workgroup-wide bucket scatter of up to 16 records per thread ------------------------------
struct ScatterLds {
	uint64_t stage[kTileRecords];
	uint32_t hist[kMaxBuckets];
	uint32_t lbase[kMaxBuckets];
	uint32_t gbase[kMaxBuckets];
	uint32_t wave_tot[kTileThreads / 64];
};

// exclusive prefix sum of hist[0..kMaxBuckets) into lbase, one entry per thread (1024 threads)
__device__ __forceinline__ void scan_hist(ScatterLds &L)
{
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const uint32_t v = L.hist[t];
	uint32_t inc = v;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t n = __shfl_up(inc, off, 64);
		if (lane >= off) inc += n;
	}
	if (lane == 63) L.wave_tot[wave] = inc;
	lds_barrier();
	uint32_t before = 0;
	for (int w = 0; w < wave; w++) before += L.wave_tot[w];
	L.lbase[t] = before + inc - v;
}

// rec[u] valid iff bkt[u] != 0xFFFF.  Appends every record to bucket bkt[u] of `out` (bucket b
// occupies out[(bucket_index0 + b) * cap ..]), reserving space with one global atomic per non-empty
// bucket per tile; records beyond a bucket's capacity are decoded and pushed to the overflow list.
template <int PER_THREAD>
__device__ __forceinline__ void scatter_tile(ScatterLds &L, const uint64_t (&rec)[PER_THREAD], const uint32_t (&bkt)[PER_THREAD],
                                             uint32_t n_buckets, uint32_t *__restrict__ cnt, uint64_t *__restrict__ out,
                                             uint64_t cap, uint32_t b1_of_bucket0, bool bucket_is_b1, const PartGeom &G,
                                             const PartStore &P, Counters *ctr)
{
	const int t = threadIdx.x;
	L.hist[t] = 0;
	lds_barrier();
	uint32_t rank[PER_THREAD];
#pragma unroll
	for (int u = 0; u < PER_THREAD; u++) rank[u] = (bkt[u] != 0xFFFFu) ? atomicAdd(&L.hist[bkt[u]], 1u) : 0u;
	lds_barrier();
	// reserve space in the global buckets: issued now, consumed only at copy-out, so the atomic's
	// round trip overlaps the scan and the staging writes
	const uint32_t my_count = L.hist[t];
	uint32_t my_gbase = 0;
	if ((uint32_t)t < n_buckets && my_count) my_gbase = atomicAdd(&cnt[t], my_count);
	scan_hist(L);
	lds_barrier();
#pragma unroll
	for (int u = 0; u < PER_THREAD; u++)
		if (bkt[u] != 0xFFFFu) L.stage[L.lbase[bkt[u]] + rank[u]] = rec[u];
	L.gbase[t] = my_gbase;
	lds_barrier();
	// copy-out: each wave takes buckets wave, wave+16, ...; a run is written with contiguous 8-byte lanes
	const int lane = t & 63, wave = t >> 6;
	for (uint32_t b = wave; b < n_buckets; b += kTileThreads / 64) {
		const uint32_t n = L.hist[b];
		if (n == 0) continue;
		const uint32_t src = L.lbase[b];
		const uint64_t dst = L.gbase[b];
		uint64_t *o = out + (uint64_t)b * cap;
		for (uint32_t i = lane; i < n; i += 64) {
			const uint64_t rcd = L.stage[src + i];
			if (dst + i < cap) {
				o[dst + i] = rcd;
			} else {
				const uint32_t b1 = bucket_is_b1 ? b : b1_of_bucket0;
				push_overflow(P, record_key(rcd, b1, G), (uint32_t)(rcd >> 3) & 7u, (uint32_t)rcd & 7u, ctr);
			}
		}
	}
	lds_barrier(); // stage / hist are reused by the next tile; the global stores keep draining
}

// ---- level 1: extraction fused with the first scatter ------------------------------------------
template <bool HAS_DEAD>
__global__ __launch_bounds__(kTileThreads) void k_extract_scatter(ReadBatch rb, PartGeom G, PartStore P, Counters *__restrict__ ctr)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	ScatterLds &L = *reinterpret_cast<ScatterLds *>(lds_raw);
	const uint64_t n_chunks = (rb.n_bases + 15u) >> 4;
	const uint64_t n_tiles = (n_chunks + kTileThreads - 1) / kTileThreads;
	unsigned long long *polyA = &ctr->polyA_links;

	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint64_t chunk = tile * kTileThreads + threadIdx.x;
		uint64_t rec[16];
		uint32_t bkt[16];
		if (chunk < n_chunks) {
			LaneWindow w = load_lane_window<HAS_DEAD>(rb, chunk);
#pragma unroll
			for (uint32_t i = 0; i < 16; i++) {
				const Triple tr = next_triple<HAS_DEAD>(w, i, rb.k, rb.n_bases);
				bkt[i] = 0xFFFFu;
				rec[i] = 0;
				if (tr.valid) {
					if (tr.key == 0ull) {
						links_cas_observe(polyA, *reinterpret_cast<volatile unsigned long long *>(polyA), tr.lb, tr.rb);
					} else {
						uint64_t q;
						const uint64_t slot = fast_divmod(hash_code(tr.key), G.magic, q);
						rec[i] = make_record(q, slot, G.r, tr.lb, tr.rb);
						bkt[i] = (uint32_t)(slot >> G.r);
					}
				}
			}
		} else {
#pragma unroll
			for (uint32_t i = 0; i < 16; i++) { bkt[i] = 0xFFFFu; rec[i] = 0; }
		}
		scatter_tile<16>(L, rec, bkt, G.n1, P.cnt1, P.l1, G.cap1, 0u, true, G, P, ctr);
	}
}

// ---- level 2: split every level-1 bucket into its n2 final buckets -----------------------------
// k_plan_l2 (one workgroup): tile_prefix[b1] = number of 16384-record tiles in buckets < b1.
__global__ __launch_bounds__(kMaxBuckets) void k_plan_l2(PartGeom G, PartStore P, uint32_t *__restrict__ tile_prefix)
{
	__shared__ uint32_t tot[kMaxBuckets / 64];
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	uint32_t v = 0;
	if ((uint32_t)t < G.n1) {
		const uint64_t filled = P.cnt1[t] < G.cap1 ? P.cnt1[t] : G.cap1;
		v = (uint32_t)((filled + kTileRecords - 1) / kTileRecords);
	}
	uint32_t inc = v;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t n = __shfl_up(inc, off, 64);
		if (lane >= off) inc += n;
	}
	if (lane == 63) tot[wave] = inc;
	__syncthreads();
	uint32_t before = 0;
	for (int w = 0; w < wave; w++) before += tot[w];
	if ((uint32_t)t < G.n1) tile_prefix[t] = before + inc - v;
	if ((uint32_t)t == G.n1 - 1) tile_prefix[G.n1] = before + inc;
}

// Persistent workgroups (one per CU) walk the flattened tile list; the records of tile i+1 are
// loaded into registers before tile i is scattered, so HBM reads, the LDS work and the (undrained)
// stores of consecutive tiles overlap.
__device__ __forceinline__ void l2_load_tile(const PartGeom &G, const PartStore &P, const uint32_t *__restrict__ tile_prefix,
                                             uint32_t g, uint32_t n_tiles, uint64_t (&rec)[16], uint32_t &b1_out)
{
	b1_out = 0;
#pragma unroll
	for (int u = 0; u < 16; u++) rec[u] = ~0ull;
	if (g >= n_tiles) return;
	uint32_t lo = 0, hi = G.n1; // last b1 with tile_prefix[b1] <= g
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (tile_prefix[mid] <= g) lo = mid; else hi = mid;
	}
	const uint32_t b1 = lo;
	b1_out = b1;
	const uint64_t filled = P.cnt1[b1] < G.cap1 ? P.cnt1[b1] : G.cap1;
	const uint64_t first = (uint64_t)(g - tile_prefix[b1]) * kTileRecords;
	const uint64_t *in = P.l1 + (uint64_t)b1 * G.cap1;
#pragma unroll
	for (int u = 0; u < 16; u++) { // coalesced: consecutive lanes read consecutive records
		const uint64_t i = first + (uint64_t)u * kTileThreads + threadIdx.x;
		if (i < filled) rec[u] = __builtin_nontemporal_load(in + i);
	}
}

__global__ __launch_bounds__(kTileThreads) void k_scatter_l2(PartGeom G, PartStore P, const uint32_t *__restrict__ tile_prefix,
                                                             Counters *__restrict__ ctr)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	ScatterLds &L = *reinterpret_cast<ScatterLds *>(lds_raw);
	const uint32_t n_tiles = tile_prefix[G.n1];
	uint64_t nxt[16];
	uint32_t nxt_b1;
	l2_load_tile(G, P, tile_prefix, blockIdx.x, n_tiles, nxt, nxt_b1);
	for (uint32_t g = blockIdx.x; g < n_tiles; g += gridDim.x) {
		uint64_t rec[16];
		uint32_t bkt[16];
		const uint32_t b1 = nxt_b1;
#pragma unroll
		for (int u = 0; u < 16; u++) {
			rec[u] = nxt[u];
			// an all-ones word is never a record: the neighbour fields only take the values 0..4
			bkt[u] = (rec[u] == ~0ull) ? 0xFFFFu : ((uint32_t)(rec[u] >> (6 + kRegionBits)) & (G.n2 - 1u));
		}
		l2_load_tile(G, P, tile_prefix, g + gridDim.x, n_tiles, nxt, nxt_b1); // in flight during the scatter below
		scatter_tile<16>(L, rec, bkt, G.n2, P.cnt2 + (uint64_t)b1 * G.n2, P.l2 + (uint64_t)b1 * G.n2 * G.cap2, G.cap2, b1, false, G, P, ctr);
	}
}

// ---- build: one workgroup per 4096-slot region ---------------------------------------------------
struct BuildLds {
	unsigned long long ident[kRegionSlots + kSpillSlots]; // (record >> 6) + 1, 0 = empty
	unsigned long long links[kRegionSlots + kSpillSlots];
	unsigned long long red[kBuildThreads / 64];
};

__global__ __launch_bounds__(kBuildThreads) void k_build_regions(PartGeom G, PartStore P, Node *__restrict__ table,
                                                                  Counters *__restrict__ ctr)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	BuildLds &L = *reinterpret_cast<BuildLds *>(lds_raw);
	const uint32_t f = blockIdx.x;                       // final bucket == region index == slot >> 12
	const uint32_t b1 = f >> (G.r - kRegionBits);
	const uint64_t region_base = (uint64_t)f << kRegionBits;
	const uint32_t region_len = (uint32_t)((G.size - region_base < (uint64_t)kRegionSlots) ? G.size - region_base : kRegionSlots);
	const int t = threadIdx.x;
	const uint64_t filled = P.cnt2[f] < G.cap2 ? P.cnt2[f] : G.cap2;
	const uint64_t *in = P.l2 + (uint64_t)f * G.cap2;

	for (int i = t; i < kRegionSlots + kSpillSlots; i += kBuildThreads) {
		L.ident[i] = 0ull;
		L.links[i] = 0ull;
	}
	lds_barrier();

	unsigned long long n_new = 0, n_conf = 0;
	constexpr int kBatch = 8; // records per thread loaded together before any LDS work
	for (uint64_t base = 0; base < filled; base += (uint64_t)kBatch * kBuildThreads) {
		uint64_t recs[kBatch];
#pragma unroll
		for (int u = 0; u < kBatch; u++) {
			const uint64_t i = base + (uint64_t)u * kBuildThreads + t;
			recs[u] = (i < filled) ? __builtin_nontemporal_load(in + i) : ~0ull;
		}
#pragma unroll
		for (int u = 0; u < kBatch; u++) {
			const uint64_t rec = recs[u];
			if (rec == ~0ull) continue;
			const unsigned long long id = (rec >> 6) + 1ull;
			const uint32_t lb = (uint32_t)(rec >> 3) & 7u, rb = (uint32_t)rec & 7u;
			uint32_t idx = (uint32_t)(rec >> 6) & (kRegionSlots - 1u);
			bool placed = false;
			while (idx < (uint32_t)(kRegionSlots + kSpillSlots)) {
				unsigned long long cur = L.ident[idx];
				if (cur == 0ull) {
					cur = atomicCAS(&L.ident[idx], 0ull, id);
					if (cur == 0ull) {
						if (idx < region_len) n_new++; // spilled nodes are counted when they are merged
						placed = true;
						break;
					}
				}
				if (cur == id) { placed = true; break; }
				n_conf++;
				idx++;
			}
			if (placed) {
				unsigned long long old = L.links[idx];
				for (;;) {
					const unsigned long long upd = links_observe(old, lb, rb);
					if (upd == old) break;
					const unsigned long long prev = atomicCAS(&L.links[idx], old, upd);
					if (prev == old) break;
					old = prev;
				}
			} else {
				push_overflow(P, record_key(rec, b1, G), lb, rb, ctr); // region + spill area completely full
			}
		}
	}
	lds_barrier();

	// emit the region: slot i of the table <- LDS slot i (key recomputed from (q, home slot))
	for (uint32_t i = t; i < region_len; i += kBuildThreads) {
		const unsigned long long id = L.ident[i];
		uint64_t key = 0ull, links = 0ull;
		if (id) {
			const uint64_t v = id - 1ull;
			const uint64_t slot = ((uint64_t)b1 << G.r) | (v & ((1ull << G.r) - 1ull));
			key = hash_code_inverse((v >> G.r) * G.size + slot);
			links = L.links[i];
		}
		*reinterpret_cast<uint4 *>(&table[region_base + i]) =
		    make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)links, (uint32_t)(links >> 32));
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
	}
	const unsigned long long a = block_sum_n<kBuildThreads>(n_new, L.red);
	const unsigned long long b = block_sum_n<kBuildThreads>(n_conf, L.red);
	if (t == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
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

} // namespace dbgk
