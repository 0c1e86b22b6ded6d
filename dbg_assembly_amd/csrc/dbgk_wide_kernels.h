// dbgk_wide_kernels.h -- WIDE engine: k-mers of up to 63 bases, 128-bit keys, 32-byte nodes (BASELINE cfg5).
//
// Semantics: include/dbgk_wide.h (the reference's rules carried to 128 bits; the reference itself stops at
// k = 31, so k > 32 is "parity unpinned"; for k <= 32 every rule reduces to the reference's).  This is the
// fused extract + global-atomic insert form (like the DIRECT engine); a partitioned form is the next step.
//
// Device node (32 bytes, 32-byte aligned):  { hi + 1 | lo | links | 0 }
//   * a slot is claimed with ONE 64-bit CAS on `lo` (0 = empty), then `hi + 1` is published with an atomic store
//     (0 = not yet published: a reader that matched `lo` re-reads until it is there -- the winner's store is
//     issued in the same loop iteration as its CAS, so lanes of one wave cannot wait on each other forever);
//   * every word of the protocol is read and written with agent-scope atomics: the per-XCD L2s are not coherent
//     and no ordering between the two words is needed (each is final once non-zero);
//   * keys whose low word is 0 (the k-mer ends in 32 A's) cannot use that rule: they live in a small side table
//     claimed on `hi + 1`; key 0 itself (poly-A / poly-T) is the side node of DBGgraph.cpp:153-164 as everywhere;
//   * the two link words are bumped with one 64-bit CAS loop, bytes saturating at 255 (links_cas_observe).
#pragma once

#include "dbgk_kernels.h"
#include "dbgk_wide.h"

namespace dbgk {

using dbgk_wide::Key128;

struct alignas(32) WNode {
	unsigned long long hi1;   // key.hi + 1, 0 = not published yet
	unsigned long long lo;    // key.lo, 0 = empty slot
	unsigned long long links; // l_link | r_link << 32
	unsigned long long pad;
};

constexpr uint32_t kWideSideSlots = 4096; // keys with lo == 0 (and hi != 0)

struct WTable {
	WNode *nodes;
	uint64_t size;
	ModMagic magic;
	WNode *side; // [kWideSideSlots], claimed on hi1
};

__device__ __forceinline__ unsigned long long wload(const unsigned long long *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wstore(unsigned long long *p, unsigned long long v)
{
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// slot of `key` (lo != 0) in the main table, claiming it if absent; ~0 = table full
__device__ __forceinline__ uint64_t wide_find_or_claim(const WTable &T, Key128 key, uint64_t &links_guess, unsigned long long &n_new,
                                                       unsigned long long &n_conf)
{
	uint64_t slot = fast_mod(dbgk_wide::hash128(key), T.magic);
	uint64_t steps = 0, spins = 0;
	while (steps <= T.size) {
		WNode *nd = &T.nodes[slot];
		unsigned long long w0 = wload(&nd->lo);
		if (w0 == 0ull) {
			const unsigned long long prev = atomicCAS(&nd->lo, 0ull, (unsigned long long)key.lo);
			if (prev == 0ull) {
				wstore(&nd->hi1, key.hi + 1ull);
				n_new++;
				links_guess = 0ull;
				return slot;
			}
			w0 = prev;
		}
		if (w0 == key.lo) {
			const unsigned long long h1 = wload(&nd->hi1);
			if (h1 == 0ull && ++spins < (1ull << 24)) continue; // its owner is about to publish the high word: look again
			if (h1 == key.hi + 1ull) {
				links_guess = wload(&nd->links);
				return slot;
			}
		}
		n_conf++;
		steps++;
		slot = (slot + 1 == T.size) ? 0 : slot + 1;
	}
	return ~0ull;
}

// side table for keys with lo == 0: linear probing on hi + 1 from hash_code(hi) % kWideSideSlots
__device__ __forceinline__ uint64_t wide_side_find_or_claim(const WTable &T, Key128 key, uint64_t &links_guess, unsigned long long &n_new)
{
	uint32_t slot = (uint32_t)(dbgk_wide::hash_code64(key.hi) % kWideSideSlots);
	for (uint32_t steps = 0; steps < kWideSideSlots; steps++) {
		WNode *nd = &T.side[slot];
		unsigned long long h1 = wload(&nd->hi1);
		if (h1 == 0ull) {
			const unsigned long long prev = atomicCAS(&nd->hi1, 0ull, (unsigned long long)(key.hi + 1ull));
			if (prev == 0ull) {
				n_new++;
				links_guess = 0ull;
				return slot;
			}
			h1 = prev;
		}
		if (h1 == key.hi + 1ull) {
			links_guess = wload(&nd->links);
			return slot;
		}
		slot = (slot + 1u == kWideSideSlots) ? 0u : slot + 1u;
	}
	return ~0ull;
}

__device__ __forceinline__ void wide_insert(const WTable &T, Key128 key, uint32_t lb, uint32_t rb, Counters *ctr, unsigned long long &n_new,
                                            unsigned long long &n_conf, bool &full)
{
	if (dbgk_wide::is_zero(key)) { // poly-A / poly-T side node (DBGgraph.cpp:153-164)
		links_cas_observe(&ctr->polyA_links, *reinterpret_cast<volatile unsigned long long *>(&ctr->polyA_links), lb, rb);
		return;
	}
	uint64_t guess;
	if (key.lo == 0ull) {
		const uint64_t s = wide_side_find_or_claim(T, key, guess, n_new);
		if (s == ~0ull) { full = true; return; }
		links_cas_observe(&T.side[s].links, guess, lb, rb);
		return;
	}
	const uint64_t s = wide_find_or_claim(T, key, guess, n_new, n_conf);
	if (s == ~0ull) { full = true; return; }
	links_cas_observe(&T.nodes[s].links, guess, lb, rb);
}

// 128 bits of a position bitmap starting at p0 (p0 % 16 == 0); bits beyond the batch read as 0
__device__ __forceinline__ void load_bits128(const uint32_t *__restrict__ bits, uint64_t p0, uint64_t n_bases, uint64_t &b0, uint64_t &b1)
{
	b0 = load_bits64(bits, p0);
	b1 = (p0 + 64u < n_bases) ? load_bits64(bits, p0 + 64u) : 0ull;
}

// thread_parseBlock + thread_updatekmers for 128-bit keys: one lane owns 16 consecutive base positions
template <bool HAS_DEAD>
__global__ __launch_bounds__(kBlock) void k_wide_extract_insert(ReadBatch rb, WTable T, Counters *__restrict__ ctr)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint32_t k = (uint32_t)rb.k; // 1..63
	const uint64_t n_chunks = (rb.n_bases + 15u) >> 4;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t chunk = (uint64_t)blockIdx.x * kBlock + threadIdx.x; chunk < n_chunks; chunk += stride) {
		const uint64_t p0 = chunk * 16u;
		// 80 bases from p0 (window of the last position + its right neighbour: 15 + 63 + 1), MSB first
		uint64_t A = ((uint64_t)load_packed_chunk(rb, chunk) << 32) | load_packed_chunk(rb, chunk + 1);
		uint64_t B = ((uint64_t)load_packed_chunk(rb, chunk + 2) << 32) | load_packed_chunk(rb, chunk + 3);
		uint64_t C = (uint64_t)load_packed_chunk(rb, chunk + 4) << 32;
		uint32_t prev = chunk ? base_code(rb, p0 - 1) : 0u;
		uint64_t S0, S1, D0 = 0, D1 = 0;
		load_bits128(rb.start_bits, p0, rb.n_bases, S0, S1);
		if (HAS_DEAD) load_bits128(rb.dead_bits, p0, rb.n_bases, D0, D1);
#pragma unroll 1
		for (uint32_t i = 0; i < 16u; i++) {
			const uint64_t p = p0 + i;
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
			const uint64_t inner_mask = (k > 1u) ? ((1ull << (k - 1u)) - 1ull) : 0ull; // k - 1 <= 62 bits
			bool valid = (p + k <= rb.n_bases) && (((S0 >> 1) | (S1 << 63)) & inner_mask) == 0ull;
			bool has_left = p > 0 && !(S0 & 1ull);
			const uint64_t bit_k = k < 64u ? ((S0 >> k) & 1ull) : (S1 & 1ull);
			bool has_right = (p + k < rb.n_bases) && !bit_k;
			if (HAS_DEAD) {
				const uint64_t km = (1ull << k) - 1ull; // k <= 63
				valid = valid && (D0 & km) == 0ull;
				has_right = has_right && !((D0 >> k) & 1ull);
			}
			if (valid) {
				const dbgk_wide::Observation o = dbgk_wide::canonical(fwd, (int)k, has_left ? prev : 4u, has_right ? right : 4u);
				wide_insert(T, o.key, o.lb, o.rb, ctr, n_new, n_conf, full);
			}
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
		}
	}
	const unsigned long long a = block_sum(n_new, red);
	const unsigned long long b = block_sum(n_conf, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
	}
	if (full) atomicOr(&ctr->error, 1u);
}

// ---- results -------------------------------------------------------------------------------------
// occupied slots of the main table and of the side table -> dense array of host-layout nodes (any order)
__global__ __launch_bounds__(kBlock) void k_wide_compact(const WNode *__restrict__ nodes, uint64_t size, const WNode *__restrict__ side,
                                                         dbgk_node32 *__restrict__ out, unsigned long long *__restrict__ cursor, uint64_t cap)
{
	const uint64_t total = size + kWideSideSlots;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
		const WNode nd = i < size ? nodes[i] : side[i - size];
		const bool occ = i < size ? nd.lo != 0ull : nd.hi1 != 0ull;
		if (!occ) continue;
		const unsigned long long j = atomicAdd(cursor, 1ull);
		if (j >= cap) continue;
		dbgk_node32 o;
		o.kmer_hi = nd.hi1 - 1ull;
		o.kmer_lo = nd.lo;
		o.l_link = (uint32_t)nd.links;
		o.r_link = (uint32_t)(nd.links >> 32);
		o.reserved = 0;
		out[j] = o;
	}
}

// host-layout image of the main table (slot i -> out[i], empty slots all-zero) + nul_flag bytes (MSB first, kmerSet.cpp:53)
__global__ __launch_bounds__(kBlock) void k_wide_image(const WNode *__restrict__ nodes, uint64_t size, dbgk_node32 *__restrict__ out,
                                                       uint8_t *__restrict__ flags)
{
	const uint64_t n_bytes = size / 8 + 1;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_bytes; b += stride) {
		uint32_t byte = 0;
		for (uint32_t j = 0; j < 8; j++) {
			const uint64_t i = b * 8 + j;
			if (i >= size) break;
			const WNode nd = nodes[i];
			dbgk_node32 o = {0, 0, 0, 0, 0};
			if (nd.lo != 0ull) {
				byte |= 0x80u >> j;
				o.kmer_hi = nd.hi1 - 1ull;
				o.kmer_lo = nd.lo;
				o.l_link = (uint32_t)nd.links;
				o.r_link = (uint32_t)(nd.links >> 32);
			}
			out[i] = o;
		}
		flags[b] = (uint8_t)byte;
	}
}

// out[0] = digest (sum of node digests), out[1] = occupied slots
__global__ __launch_bounds__(kBlock) void k_wide_digest(const WNode *__restrict__ nodes, uint64_t size, const WNode *__restrict__ side,
                                                        unsigned long long *__restrict__ out)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long sum = 0, cnt = 0;
	const uint64_t total = size + kWideSideSlots;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
		const WNode nd = i < size ? nodes[i] : side[i - size];
		const bool occ = i < size ? nd.lo != 0ull : nd.hi1 != 0ull;
		if (!occ) continue;
		sum += dbgk_wide::node_digest(Key128{nd.hi1 - 1ull, nd.lo}, nd.links);
		cnt++;
	}
	const unsigned long long a = block_sum(sum, red);
	const unsigned long long b = block_sum(cnt, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&out[0], a);
		if (b) atomicAdd(&out[1], b);
	}
}

// first pass of calculate_kmer_links (contig.cpp:119-181) on the wide table; layout of `out` as k_link_stats
__global__ __launch_bounds__(kBlock) void k_wide_link_stats(const WNode *__restrict__ nodes, uint64_t size, const WNode *__restrict__ side, int cutoff,
                                                            uint64_t polyA_links, int with_key0, unsigned long long *__restrict__ out)
{
	__shared__ unsigned int hist[256];
	__shared__ unsigned long long red[kBlock / 64];
	hist[threadIdx.x] = 0; // kBlock == 256
	__syncthreads();
	unsigned long long cls[5] = {0, 0, 0, 0, 0};
	const uint64_t total = size + kWideSideSlots;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
		const WNode nd = i < size ? nodes[i] : side[i - size];
		const bool occ = i < size ? nd.lo != 0ull : nd.hi1 != 0ull;
		if (occ) link_classes(nd.links, cutoff, hist, cls);
	}
	if (with_key0 && blockIdx.x == 0 && threadIdx.x == 0) link_classes(polyA_links, cutoff, hist, cls); // the key-0 node (of a sharded table: shard 0 only)
	__syncthreads();
	if (hist[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
#pragma unroll
	for (int c = 0; c < 5; c++) {
		const unsigned long long s = block_sum(cls[c], red);
		if (threadIdx.x == 0 && s) atomicAdd(&out[256 + c], s);
	}
}

// ---- several GPUs: nodes grouped by owner = (hash128(key) >> 32) % n_parts, merged by the owner ----------------
// (the flow of dbgk_partition_export / dbgk_merge_nodes for 32-byte nodes: every GPU builds the graph of its share
// of the reads, ships each node to its owner, the owner adds the counters up -- exact for any split of the input
// because min(255, min(255,a) + min(255,b)) == min(255, a+b))
__device__ __forceinline__ uint32_t wide_owner_of(Key128 key, uint32_t n_parts)
{
	return (uint32_t)((dbgk_wide::hash128(key) >> 32) % n_parts);
}

// counts != null: count pass (counts[p] += nodes owned by p); else scatter pass: cursors[p] starts at the part's base
// offset, every block reserves one range per part per sweep
__global__ __launch_bounds__(kBlock) void k_wide_partition(const WNode *__restrict__ nodes, uint64_t size, const WNode *__restrict__ side,
                                                           uint32_t n_parts, unsigned long long *__restrict__ counts,
                                                           unsigned long long *__restrict__ cursors, dbgk_node32 *__restrict__ out, uint64_t capacity)
{
	__shared__ unsigned int local[kMaxParts];
	__shared__ unsigned long long base[kMaxParts];
	const uint64_t total = size + kWideSideSlots;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	const uint64_t n_iter = (total + stride - 1) / stride;
	for (uint64_t it = 0; it < n_iter; it++) {
		if (threadIdx.x < kMaxParts) local[threadIdx.x] = 0;
		__syncthreads();
		const uint64_t i = it * stride + (uint64_t)blockIdx.x * kBlock + threadIdx.x;
		WNode nd = {0, 0, 0, 0};
		if (i < total) nd = i < size ? nodes[i] : side[i - size];
		const bool occ = i < total && (i < size ? nd.lo != 0ull : nd.hi1 != 0ull);
		uint32_t part = 0, rank = 0;
		if (occ) {
			part = wide_owner_of(Key128{nd.hi1 - 1ull, nd.lo}, n_parts);
			rank = atomicAdd(&local[part], 1u);
		}
		__syncthreads();
		if (threadIdx.x < n_parts && local[threadIdx.x]) {
			if (counts) atomicAdd(&counts[threadIdx.x], (unsigned long long)local[threadIdx.x]);
			else base[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], (unsigned long long)local[threadIdx.x]);
		}
		__syncthreads();
		if (occ && !counts) {
			const uint64_t dst = base[part] + rank;
			if (dst < capacity) out[dst] = dbgk_node32{nd.hi1 - 1ull, nd.lo, (uint32_t)nd.links, (uint32_t)(nd.links >> 32), 0};
		}
		__syncthreads();
	}
}

// insert-if-absent + per-byte saturating add of aggregated nodes; a key-0 node is folded into the side node
__global__ __launch_bounds__(kBlock) void k_wide_merge_nodes(const dbgk_node32 *__restrict__ in, uint64_t n, WTable T, Counters *__restrict__ ctr)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
		const dbgk_node32 nd = in[i];
		const Key128 key{nd.kmer_hi, nd.kmer_lo};
		const uint64_t add = (uint64_t)nd.l_link | ((uint64_t)nd.r_link << 32);
		if (dbgk_wide::is_zero(key)) { // another GPU's key-0 node (an all-zero record adds nothing)
			links_cas_merge(&ctr->polyA_links, 0ull, add);
			continue;
		}
		uint64_t guess;
		if (key.lo == 0ull) {
			const uint64_t s = wide_side_find_or_claim(T, key, guess, n_new);
			if (s == ~0ull) { full = true; continue; }
			links_cas_merge(&T.side[s].links, guess, add);
			continue;
		}
		const uint64_t s = wide_find_or_claim(T, key, guess, n_new, n_conf);
		if (s == ~0ull) { full = true; continue; }
		links_cas_merge(&T.nodes[s].links, guess, add);
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
