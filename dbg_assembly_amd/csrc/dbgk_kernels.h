// dbgk_kernels.h -- the HIP kernels of the k-mer graph construction path (gfx950 only).
//
// Data layout in HBM:
//   bases       ASCII read bases back to back, no separators (what dbgk_push_reads* receives), OR
//   packed      the same bases as 2-bit codes, 16 per dword, base 0 in bits 31..30 (what dbgk_push_reads_packed*
//               receives; exactly what pack16_ascii makes of 16 ASCII bases, so every kernel consumes either form)
//   start_bits  1 bit per base position, LSB-first in dwords: set where a read starts
//   dead_bits   1 bit per base position: set on bases beyond maxReadLen of their read (only
//               allocated/used when a batch contains such reads)
//   table       Node[size], the reference's open-addressed KmerSet array: slot of a key is
//               hash_code(key) % size, linear probing, key 0 = empty (kmerSet.h:70-75,
//               DBGgraph.cpp:167-205)
//
// Work decomposition of the extraction: one lane owns 16 consecutive base positions (one 16-byte
// ASCII chunk -> one packed dword) and slides the 2-bit window over them in registers; a wave
// covers 1 KiB of contiguous read bytes per step, so the ASCII loads are fully coalesced
// 16-byte-per-lane loads.  Read boundaries come from the bitmap, so no lane ever searches the
// offsets array.
#pragma once

#include "dbgk_device.h"

namespace dbgk {

constexpr int kBlock = 256;          // 4 waves
constexpr int kPosPerLane = 16;      // one uint4 of ASCII per lane
constexpr int kGroup = 4;            // positions resolved together (independent table loads in flight)

// device-side counters of one handle (zeroed by reset)
struct Counters {
	unsigned long long n_new;          // keys claimed (distinct non-zero keys)
	unsigned long long n_conflict;     // probe steps
	unsigned long long total_reads;
	unsigned long long total_kmers;    // Kmer_total_num semantics (untrimmed)
	unsigned long long stored_kmers;   // windows extracted
	unsigned long long polyA_links;    // key-0 node, l_link | r_link << 32
	unsigned int       error;          // bit 0: table full
	// per batch, cleared with ONE memset before k_mark (any_dead .. len_max are contiguous):
	unsigned int       any_dead;       // some read longer than maxReadLen in the current batch
	unsigned long long len_min_inv;    // ~(shortest read length) of the batch (inverted so that 0 is the neutral start value)
	unsigned long long len_max;        // longest read length of the batch
	unsigned long long polyA_slot;     // where the key-0 node was placed for export (or ~0)
	// KFREQ, direct blocks: the table summary kept up to date by whatever writes the table (block build, overflow / heavy-hitter
	// application, key 0) -- wrapping sums of signed changes
	unsigned long long kf_nonzero;     // counters that are not 0 (distinct canonical k-mers)
	unsigned long long kf_sum;         // sum of all counters (saturated values)
	// input bytes that are none of ACGTNacgtn (read as 'A', dbgk_device.h): other_seen is raised by whichever extraction kernel
	// meets one, k_count_other_bytes then counts the batch's (only then: the kernel leaves at once while the flag is clear)
	unsigned long long other_bytes;
	unsigned int       other_seen;
	unsigned int       pad0;
};

struct TableRef {
	Node *nodes;
	uint64_t size;
	ModMagic magic;
};

struct ReadBatch {
	const char *bases;         // ASCII, or null when the batch came 2-bit packed
	uint64_t n_bases;
	const uint32_t *start_bits;
	const uint32_t *dead_bits; // may be null
	int k;
	const uint32_t *packed;    // 2-bit codes, 16 bases per dword (base 0 in bits 31..30), (n_bases + 15) / 16 words; or null
	unsigned int *other_seen;  // &Counters::other_seen
};

// the packed word of bases [16 * chunk, 16 * chunk + 16) of a PACKED batch; bases past the end read as 'A' like the ASCII tail
__device__ __forceinline__ uint32_t packed_word(const ReadBatch &rb, uint64_t chunk)
{
	const uint64_t off = chunk * 16u;
	if (off + 16u <= rb.n_bases) return rb.packed[chunk];
	if (off >= rb.n_bases) return 0u;
	return rb.packed[chunk] & ~(0xFFFFFFFFu >> (2u * (uint32_t)(rb.n_bases - off)));
}
// 2-bit code of base p (p < n_bases)
__device__ __forceinline__ uint32_t base_code(const ReadBatch &rb, uint64_t p)
{
	if (rb.packed) return (rb.packed[p >> 4] >> (30u - 2u * (uint32_t)(p & 15u))) & 3u;
	return code_ascii((uint32_t)(uint8_t)rb.bases[p], rb.other_seen);
}

// ---------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	return v; // valid in lane 0
}

// sum over the block; result valid in thread 0.  `lds` holds kBlock/64 entries per reduced value.
__device__ __forceinline__ unsigned long long block_sum(unsigned long long v, unsigned long long *lds)
{
	v = wave_sum(v);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	__syncthreads();
	if (lane == 0) lds[wave] = v;
	__syncthreads();
	unsigned long long s = 0;
	if (threadIdx.x == 0) {
#pragma unroll
		for (int w = 0; w < kBlock / 64; w++) s += lds[w];
	}
	return s;
}

template <int THREADS>
__device__ __forceinline__ unsigned long long block_sum_n(unsigned long long v, unsigned long long *lds)
{
	v = wave_sum(v);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	__syncthreads();
	if (lane == 0) lds[wave] = v;
	__syncthreads();
	unsigned long long s = 0;
	if (threadIdx.x == 0) {
#pragma unroll
		for (int w = 0; w < THREADS / 64; w++) s += lds[w];
	}
	return s;
}

// ---------------------------------------------------------------------------------------------
// k_mark: read boundaries + the reference's totals, one thread per read
//   start bit at offsets[r]; dead bits on [offsets[r]+maxReadLen, offsets[r+1]) (DBGgraph.cpp:63);
//   Kmer_total_num += len-K+1 for len >= K (untrimmed, :101); Total_reads_num += 1 (:274)
// ---------------------------------------------------------------------------------------------
// start_bits == null: statistics only (totals, any_dead, shortest / longest read) -- all a batch of equal-length reads
// needs, its level-1 kernel finds the read boundaries by arithmetic; with_stats == 0: bitmaps only (the second call for
// a device batch that turned out to need them)
__global__ __launch_bounds__(kBlock) void k_mark(const uint64_t *__restrict__ offsets, uint64_t n_reads,
                                                 uint64_t n_bases, int k, int max_read_len,
                                                 uint32_t *__restrict__ start_bits, uint32_t *__restrict__ dead_bits,
                                                 Counters *__restrict__ ctr, int with_stats = 1)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long tot = 0, stored = 0, len_lo = ~0ull, len_hi = 0ull;
	unsigned int dead_seen = 0;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n_reads; r += stride) {
		const uint64_t s = offsets[r], e = offsets[r + 1];
		const uint64_t len = e - s;
		len_lo = len < len_lo ? len : len_lo;
		len_hi = len > len_hi ? len : len_hi;
		if (start_bits && s < n_bases) atomicOr(&start_bits[s >> 5], 1u << (s & 31u));
		if (with_stats && len >= (uint64_t)k) {
			tot += len - (uint64_t)k + 1;
			const uint64_t rl = len > (uint64_t)max_read_len ? (uint64_t)max_read_len : len;
			if (rl >= (uint64_t)k) stored += rl - (uint64_t)k + 1;
		}
		if (len > (uint64_t)max_read_len) {
			dead_seen = 1;
			if (dead_bits) {
				for (uint64_t p = s + (uint64_t)max_read_len; p < e;) { // rare path: long reads only
					const uint32_t bit = (uint32_t)(p & 31u);
					const uint64_t span = (e - p < 32u - bit) ? e - p : 32u - bit;
					const uint32_t mask = (span == 32u) ? 0xFFFFFFFFu : (((1u << span) - 1u) << bit);
					atomicOr(&dead_bits[p >> 5], mask);
					p += span;
				}
			}
		}
	}
	if (!with_stats) return; // (wave-uniform: a kernel argument)
	unsigned long long a = block_sum(tot, red);
	unsigned long long b = block_sum(stored, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&ctr->total_kmers, a);
		if (b) atomicAdd(&ctr->stored_kmers, b);
	}
	if (dead_seen) atomicOr(&ctr->any_dead, 1u);
	// shortest / longest read of the batch: equal => the partition engine's equal-length level-1 kernel applies.
	// Reduced per BLOCK before the two atomics: thousands of same-address atomicMax (one pair per wave) serialise
	// at the memory side and were most of this kernel's time.
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const unsigned long long a2 = __shfl_down(len_lo, off, 64), b2 = __shfl_down(len_hi, off, 64);
		len_lo = a2 < len_lo ? a2 : len_lo;
		len_hi = b2 > len_hi ? b2 : len_hi;
	}
	__shared__ unsigned long long lo_w[kBlock / 64], hi_w[kBlock / 64];
	if ((threadIdx.x & 63) == 0) {
		lo_w[threadIdx.x >> 6] = len_lo;
		hi_w[threadIdx.x >> 6] = len_hi;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
#pragma unroll
		for (int w = 1; w < kBlock / 64; w++) {
			len_lo = lo_w[w] < len_lo ? lo_w[w] : len_lo;
			len_hi = hi_w[w] > len_hi ? hi_w[w] : len_hi;
		}
		if (len_hi >= len_lo) {
			atomicMax(&ctr->len_min_inv, ~len_lo);
			atomicMax(&ctr->len_max, len_hi);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// window extraction for the 16 positions owned by one lane
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t load_packed_chunk(const ReadBatch &rb, uint64_t chunk)
{
	if (rb.packed) return packed_word(rb, chunk); // (wave-uniform)
	const char *__restrict__ bases = rb.bases;
	const uint64_t n_bases = rb.n_bases;
	const uint64_t off = chunk * 16u;
	if (off + 16u <= n_bases) {
		return pack16_ascii(*reinterpret_cast<const uint4 *>(bases + off), rb.other_seen);
	}
	if (off >= n_bases) return 0u;
	uint32_t w[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u}; // tail chunk: pad with 'A'
	for (uint64_t i = 0; off + i < n_bases; i++) {
		const uint32_t c = (uint8_t)bases[off + i];
		w[i >> 2] = (w[i >> 2] & ~(0xFFu << ((i & 3u) * 8u))) | (c << ((i & 3u) * 8u));
	}
	return pack16_ascii(make_uint4(w[0], w[1], w[2], w[3]), rb.other_seen);
}

// 64 bits of a position bitmap starting at bit position p0 (p0 % 16 == 0)
__device__ __forceinline__ uint64_t load_bits64(const uint32_t *__restrict__ bits, uint64_t p0)
{
	const uint64_t wi = p0 >> 5;
	const uint32_t sh = (uint32_t)(p0 & 31u);
	const uint64_t lo = (uint64_t)bits[wi] | ((uint64_t)bits[wi + 1] << 32);
	if (sh == 0) return lo;
	return (lo >> sh) | ((uint64_t)bits[wi + 2] << (64u - sh));
}

struct LaneWindow {
	uint64_t hi, lo;     // 2-bit codes of bases p0.. (MSB first), 48 loaded
	uint64_t S, D;       // start / dead bits, bit t <-> position p0+t
	uint32_t prev_code;  // code of base p0-1
	uint64_t p0;
};

template <bool HAS_DEAD>
__device__ __forceinline__ LaneWindow load_lane_window(const ReadBatch &rb, uint64_t chunk)
{
	LaneWindow w;
	w.p0 = chunk * 16u;
	const uint32_t w0 = load_packed_chunk(rb, chunk);
	const uint32_t w1 = load_packed_chunk(rb, chunk + 1);
	const uint32_t w2 = load_packed_chunk(rb, chunk + 2);
	w.hi = ((uint64_t)w0 << 32) | w1;
	w.lo = (uint64_t)w2 << 32;
	w.prev_code = 0;
	if (chunk > 0) w.prev_code = base_code(rb, w.p0 - 1);
	w.S = load_bits64(rb.start_bits, w.p0);
	w.D = HAS_DEAD ? load_bits64(rb.dead_bits, w.p0) : 0ull;
	return w;
}

struct Triple {
	uint64_t key;
	uint32_t lb, rb;
	bool valid;
	bool strict_fwd; // forward k-mer strictly smaller than its reverse complement
};

// k-mer starting at the window's current first base, then slide by one base
// (DBGgraph.cpp:64-98 restated per position; N is A; tie kbit == rc -> forward :80)
template <bool HAS_DEAD>
__device__ __forceinline__ Triple next_triple(LaneWindow &w, uint32_t i, int k, uint64_t n_bases)
{
	Triple t;
	const uint64_t p = w.p0 + i;
	const uint64_t kbit = w.hi >> (64 - 2 * k);
	const uint32_t right = (k < 32) ? (uint32_t)(w.hi >> (62 - 2 * k)) & 3u : (uint32_t)(w.lo >> 62);
	const uint32_t left = w.prev_code;
	const uint64_t inner = (k > 1) ? ((w.S >> 1) & ((1ull << (k - 1)) - 1ull)) : 0ull;
	bool valid = (p + (uint64_t)k <= n_bases) && inner == 0ull;
	bool has_left = (p > 0) && !(w.S & 1ull);
	bool has_right = (p + (uint64_t)k < n_bases) && !((w.S >> k) & 1ull);
	if (HAS_DEAD) {
		const uint64_t km = (k < 64) ? ((1ull << k) - 1ull) : ~0ull;
		valid = valid && (w.D & km) == 0ull;
		has_right = has_right && !((w.D >> k) & 1ull);
	}
	const uint64_t rc = revcomp_kbit(kbit, k);
	t.strict_fwd = kbit < rc;
	if (kbit <= rc) {
		t.key = kbit;
		t.lb = has_left ? left : 4u;
		t.rb = has_right ? right : 4u;
	} else {
		t.key = rc;
		t.rb = has_left ? 3u - left : 4u;
		t.lb = has_right ? 3u - right : 4u;
	}
	t.valid = valid;
	// slide
	w.prev_code = (uint32_t)(w.hi >> 62);
	w.hi = (w.hi << 2) | (w.lo >> 62);
	w.lo <<= 2;
	w.S >>= 1;
	if (HAS_DEAD) w.D >>= 1;
	return t;
}

// ---------------------------------------------------------------------------------------------
// insertion into the global table (thread_updatekmers, DBGgraph.cpp:153-205, for many writers per
// key): claim with a 64-bit CAS on the key word, count with a 64-bit CAS loop on the link word so
// that every byte saturates exactly at 255.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void links_cas_observe(unsigned long long *addr, uint64_t guess, uint32_t lb, uint32_t rb)
{
	uint64_t old = guess;
	for (;;) {
		const uint64_t upd = links_observe(old, lb, rb);
		if (upd == old) return; // nothing to add (no neighbour, or both counters already 255: counters only grow)
		const uint64_t prev = atomicCAS(addr, (unsigned long long)old, (unsigned long long)upd);
		if (prev == old) return;
		old = prev;
	}
}

__device__ __forceinline__ void links_cas_merge(unsigned long long *addr, uint64_t guess, uint64_t add)
{
	uint64_t old = guess;
	for (;;) {
		const uint64_t upd = links_sat_add(old, add);
		if (upd == old) return;
		const uint64_t prev = atomicCAS(addr, (unsigned long long)old, (unsigned long long)upd);
		if (prev == old) return;
		old = prev;
	}
}

// Finds or claims the slot of `key` (key != 0) starting at `slot`, with `first` the node value
// already loaded from that slot.  Returns the slot index, or ~0 when the table is full.
// n_new / n_conf are per-thread tallies.
__device__ __forceinline__ uint64_t find_or_claim(const TableRef &T, uint64_t key, uint64_t slot, Node first,
                                                  uint64_t &links_guess, unsigned long long &n_new,
                                                  unsigned long long &n_conf)
{
	Node cur = first;
	for (uint64_t steps = 0; steps <= T.size; steps++) {
		uint64_t seen = cur.kmer;
		if (seen == 0ull) {
			seen = atomicCAS(reinterpret_cast<unsigned long long *>(&T.nodes[slot].kmer), 0ull, (unsigned long long)key);
			if (seen == 0ull) {
				n_new++;
				links_guess = 0ull;
				return slot;
			}
			cur.links = 0ull; // somebody else just claimed it; its links are about to grow from 0
		}
		if (seen == key) {
			links_guess = cur.links;
			return slot;
		}
		n_conf++;
		slot = (slot + 1 == T.size) ? 0 : slot + 1;
		const uint4 v = *reinterpret_cast<const uint4 *>(&T.nodes[slot]);
		cur.kmer = ((uint64_t)v.y << 32) | v.x;
		cur.links = ((uint64_t)v.w << 32) | v.z;
	}
	return ~0ull;
}

// ---------------------------------------------------------------------------------------------
// k_extract_insert: DIRECT engine -- fused thread_parseBlock + thread_updatekmers
// ---------------------------------------------------------------------------------------------
// TRACK: additionally keep, per slot, the smallest global base position at which the slot's key was
// seen (first_pos[slot], atomicMin).  Distinct keys replayed in that order through the reference's
// sequential insert reproduce its `-t 1` slot layout.
template <bool HAS_DEAD, bool TRACK = false>
__global__ __launch_bounds__(kBlock) void k_extract_insert(ReadBatch rb, TableRef T, Counters *__restrict__ ctr,
                                                           unsigned long long *__restrict__ first_pos = nullptr, uint64_t pos_base = 0)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint64_t n_chunks = (rb.n_bases + 15u) >> 4;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	unsigned long long *polyA = &ctr->polyA_links;

	for (uint64_t chunk = (uint64_t)blockIdx.x * kBlock + threadIdx.x; chunk < n_chunks; chunk += stride) {
		LaneWindow w = load_lane_window<HAS_DEAD>(rb, chunk);
#pragma unroll 1
		for (uint32_t g = 0; g < kPosPerLane; g += kGroup) {
			Triple t[kGroup];
			uint64_t slot[kGroup];
			Node first[kGroup];
#pragma unroll
			for (int u = 0; u < kGroup; u++) {
				t[u] = next_triple<HAS_DEAD>(w, g + u, rb.k, rb.n_bases);
				slot[u] = fast_mod(hash_code(t[u].key), T.magic);
			}
#pragma unroll
			for (int u = 0; u < kGroup; u++) { // independent loads, all in flight together
				if (t[u].valid && t[u].key != 0ull) {
					const uint4 v = *reinterpret_cast<const uint4 *>(&T.nodes[slot[u]]);
					first[u].kmer = ((uint64_t)v.y << 32) | v.x;
					first[u].links = ((uint64_t)v.w << 32) | v.z;
				}
			}
#pragma unroll
			for (int u = 0; u < kGroup; u++) {
				if (!t[u].valid) continue;
				if (t[u].key == 0ull) { // poly-A / poly-T side node (DBGgraph.cpp:153-164)
					links_cas_observe(polyA, *reinterpret_cast<volatile unsigned long long *>(polyA), t[u].lb, t[u].rb);
					continue;
				}
				uint64_t guess;
				const uint64_t s = find_or_claim(T, t[u].key, slot[u], first[u], guess, n_new, n_conf);
				if (s == ~0ull) { full = true; continue; }
				links_cas_observe(reinterpret_cast<unsigned long long *>(&T.nodes[s].links), guess, t[u].lb, t[u].rb);
				if (TRACK) {
					const unsigned long long pos = pos_base + chunk * 16u + g + (uint32_t)u;
					if (first_pos[s] > pos) atomicMin(&first_pos[s], pos); // plain read first: positions mostly arrive in rising order
				}
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

// ---------------------------------------------------------------------------------------------
// KFREQ engine: the direct-addressed k-mer frequency table that the
// reference's correct_error module loads (correct_error/main.cpp:161-220 8-bit format,
// main_parallel_senior.cpp:334-408 1-bit format).  counts[v] = min(255, occurrences of the
// canonical k-mer v), v < 4^k.  Same extraction as the graph path (N counts as A, tie -> forward).
// ---------------------------------------------------------------------------------------------
template <bool HAS_DEAD>
__global__ __launch_bounds__(kBlock) void k_extract_count(ReadBatch rb, uint32_t *__restrict__ count_words)
{
	const uint64_t n_chunks = (rb.n_bases + 15u) >> 4;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t chunk = (uint64_t)blockIdx.x * kBlock + threadIdx.x; chunk < n_chunks; chunk += stride) {
		LaneWindow w = load_lane_window<HAS_DEAD>(rb, chunk);
#pragma unroll 4
		for (uint32_t i = 0; i < kPosPerLane; i++) {
			const Triple t = next_triple<HAS_DEAD>(w, i, rb.k, rb.n_bases);
			if (!t.valid) continue;
			uint32_t *word = count_words + (t.key >> 2);
			const uint32_t sh = (uint32_t)(t.key & 3u) * 8u; // little-endian byte t.key of the table
			uint32_t old = *word;
			for (;;) { // saturating byte increment inside its dword
				if (((old >> sh) & 0xFFu) == 0xFFu) break;
				const uint32_t prev = atomicCAS(word, old, old + (1u << sh));
				if (prev == old) break;
				old = prev;
			}
		}
	}
}

// saturating add into byte v of the count table (32-bit CAS on its dword)
// (ctr != null: the table summary in *ctr follows the change)
__device__ __forceinline__ void kf_sat_add(uint32_t *__restrict__ count_words, uint64_t v, uint32_t add, Counters *ctr = nullptr)
{
	uint32_t *word = count_words + (v >> 2);
	const uint32_t sh = (uint32_t)(v & 3u) * 8u;
	uint32_t old = *word;
	for (;;) {
		const uint32_t cur = (old >> sh) & 0xFFu;
		const uint32_t nxt = cur + add > 255u ? 255u : cur + add;
		if (nxt == cur) break;
		const uint32_t prev = atomicCAS(word, old, (old & ~(0xFFu << sh)) | (nxt << sh));
		if (prev == old) {
			if (ctr) {
				if (cur == 0u) atomicAdd(&ctr->kf_nonzero, 1ull);
				atomicAdd(&ctr->kf_sum, (unsigned long long)(nxt - cur));
			}
			break;
		}
		old = prev;
	}
}

// KFREQ through the PARTITION engine: nodes that left their region's LDS image (is_triple = 0: {key,
// links}, the count is the A counter of l_link) and bucket-overflow observations (is_triple = 1: one
// occurrence each) are added to the byte table after all regions have been emitted
__global__ __launch_bounds__(kBlock) void k_kf_apply(const Node *__restrict__ in, const unsigned long long *__restrict__ n_ptr, uint64_t cap,
                                                     int is_triple, uint32_t *__restrict__ count_words, Counters *ctr)
{
	const uint64_t n = *n_ptr < cap ? *n_ptr : cap;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
		kf_sat_add(count_words, in[i].kmer, is_triple ? 1u : ((uint32_t)in[i].links >> 24), ctr);
}

// the same for the side table that aggregates the surplus of heavy hitters (empty slots are skipped)
__global__ __launch_bounds__(kBlock) void k_kf_apply_table(const Node *__restrict__ in, uint64_t n, uint32_t *__restrict__ count_words, Counters *ctr)
{
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
		if (in[i].kmer != 0ull) kf_sat_add(count_words, in[i].kmer, (uint32_t)in[i].links >> 24, ctr);
}

// key 0 (poly-A / poly-T) never enters the record stream: its occurrences are counted in the A counter
// of the side word
__global__ void k_kf_key0(Counters *__restrict__ ctr, uint8_t *__restrict__ counts, int track)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		const uint32_t c = (uint32_t)ctr->polyA_links >> 24, was = counts[0];
		counts[0] = (uint8_t)c;
		if (track) { // (the side word holds the count of the whole job so far: the byte is replaced, not added to)
			ctr->kf_nonzero += (unsigned long long)(c != 0u) - (unsigned long long)(was != 0u);
			ctr->kf_sum += (unsigned long long)c - (unsigned long long)was;
		}
	}
}

// bit table of the 1-bit format: bit (128 >> (v % 8)) of byte v / 8 is set when counts[v] > cutoff
// (bitAll, correct_error/seqKmer.cpp:34); one thread per output byte
__global__ __launch_bounds__(kBlock) void k_counts_to_bits(const uint8_t *__restrict__ counts, uint64_t first_byte, uint64_t n_bytes,
                                                           uint32_t cutoff, uint8_t *__restrict__ bits)
{
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_bytes; b += stride) {
		const uint64_t c8 = *reinterpret_cast<const uint64_t *>(counts + (first_byte + b) * 8u);
		uint32_t byte = 0;
#pragma unroll
		for (uint32_t j = 0; j < 8; j++)
			if (((uint32_t)(c8 >> (8u * j)) & 0xFFu) > cutoff) byte |= 0x80u >> j;
		bits[b] = (uint8_t)byte;
	}
}

// out[0] = number of non-zero counters, out[1] = sum of counters (saturated values)
__global__ __launch_bounds__(kBlock) void k_counts_summary(const uint8_t *__restrict__ counts, uint64_t n, unsigned long long *__restrict__ out)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long nz = 0, sum = 0;
	const uint64_t n8 = n >> 3;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n8; i += stride) {
		const uint64_t c8 = reinterpret_cast<const uint64_t *>(counts)[i];
#pragma unroll
		for (uint32_t j = 0; j < 8; j++) {
			const uint32_t c = (uint32_t)(c8 >> (8u * j)) & 0xFFu;
			nz += c != 0u;
			sum += c;
		}
	}
	const unsigned long long a = block_sum(nz, red);
	const unsigned long long b = block_sum(sum, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&out[0], a);
		if (b) atomicAdd(&out[1], b);
	}
}

// counts[i] = min(255, counts[i] + other[i]) over n bytes (n a multiple of 16, both 16-byte aligned): the
// combination rule of two partial frequency tables, exact because min(255, min(255,a) + min(255,b)) = min(255, a+b)
// (SURVEY 8(e)).  Four counters per 32-bit word, byte-wise saturating add without unpacking.
__device__ __forceinline__ uint32_t sat_add_u8x4(uint32_t a, uint32_t b)
{
	const uint32_t low = (a & 0x7F7F7F7Fu) + (b & 0x7F7F7F7Fu);         // 7-bit sums, carries stay inside their byte
	const uint32_t carry = ((a & b) | ((a | b) & low)) & 0x80808080u;   // carry out of bit 7 of every byte
	const uint32_t sum = low ^ ((a ^ b) & 0x80808080u);
	return sum | ((carry >> 7) * 0xFFu);
}

__global__ __launch_bounds__(kBlock) void k_counts_merge(uint8_t *__restrict__ counts, const uint8_t *__restrict__ other, uint64_t n)
{
	const uint64_t n16 = n >> 4;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	uint4 *dst = reinterpret_cast<uint4 *>(counts);
	const uint4 *src = reinterpret_cast<const uint4 *>(other);
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) {
		const uint4 b = src[i];
		if ((b.x | b.y | b.z | b.w) == 0u) continue; // most of a sparse table
		uint4 a = dst[i];
		a.x = sat_add_u8x4(a.x, b.x);
		a.y = sat_add_u8x4(a.y, b.y);
		a.z = sat_add_u8x4(a.z, b.z);
		a.w = sat_add_u8x4(a.w, b.w);
		dst[i] = a;
	}
}

// ---------------------------------------------------------------------------------------------
// SEEDIDX engine: the seed index of the link_scaffold module,
// chop_contig_to_kmerset (link_scaffold/map_func.cpp:119-173) over add_kmerset
// (link_scaffold/kmerSet.cpp:168-210).  Every k-mer of every contig, windows never span an
// upper-case 'N' (scaffold_to_contig :303-324; 'n' counts as A like every other use of alphabet[]);
// canonical by STRICT '<' (:160: a palindrome is stored with direct = 0); payload of a key = contig
// index, position and strand of its FIRST occurrence in (contig, position) order, freq = 1 only
// while the key has been seen once.  "First" is made order-free by keeping the minimum
// (contig, position): payload word while building = id << 32 | (pos + 1) << 2 | direct << 1 | dup,
// 0 = never written.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t seed_merge(uint64_t old, uint64_t mine)
{
	if (old == 0ull) return mine;
	return (((old >> 2) <= (mine >> 2)) ? old : mine) | 1ull; // earlier occurrence wins, key no longer unique
}

__device__ __forceinline__ void seed_cas_merge(unsigned long long *addr, uint64_t guess, uint64_t mine)
{
	uint64_t old = guess;
	for (;;) {
		const uint64_t upd = seed_merge(old, mine);
		if (upd == old) return;
		const uint64_t prev = atomicCAS(addr, (unsigned long long)old, (unsigned long long)upd);
		if (prev == old) return;
		old = prev;
	}
}

// dead bit on every upper-case 'N' (one thread per 32 positions)
__global__ __launch_bounds__(kBlock) void k_mark_n(const char *__restrict__ bases, uint64_t n_bases, uint32_t *__restrict__ dead_bits)
{
	const uint64_t n_words = (n_bases + 31u) >> 5;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t w = (uint64_t)blockIdx.x * kBlock + threadIdx.x; w < n_words; w += stride) {
		uint32_t mask = 0;
		for (uint32_t j = 0; j < 32u; j++) {
			const uint64_t p = w * 32u + j;
			if (p < n_bases && bases[p] == 'N') mask |= 1u << j;
		}
		if (mask) atomicOr(&dead_bits[w], mask);
	}
}

__global__ __launch_bounds__(kBlock) void k_seed_insert(ReadBatch rb, const uint64_t *__restrict__ offsets, uint64_t n_contigs,
                                                        uint64_t id_base, TableRef T, Counters *__restrict__ ctr)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint64_t n_chunks = (rb.n_bases + 15u) >> 4;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t chunk = (uint64_t)blockIdx.x * kBlock + threadIdx.x; chunk < n_chunks; chunk += stride) {
		LaneWindow w = load_lane_window<true>(rb, chunk);
		const uint64_t p0 = chunk * 16u;
		uint64_t lo = 0, hi = n_contigs; // offsets[lo] <= p0 < offsets[hi]; ties (empty contigs) resolve to the last one
		while (hi - lo > 1) {
			const uint64_t mid = (lo + hi) >> 1;
			if (offsets[mid] <= p0) lo = mid; else hi = mid;
		}
		uint64_t cid = lo, cur_off = offsets[lo], next_off = offsets[lo + 1];
		for (uint32_t i = 0; i < kPosPerLane; i++) {
			const Triple t = next_triple<true>(w, i, rb.k, rb.n_bases);
			const uint64_t p = p0 + i;
			while (p >= next_off && cid + 1 < n_contigs) {
				cid++;
				cur_off = next_off;
				next_off = offsets[cid + 1];
			}
			if (!t.valid) continue;
			const uint64_t mine = ((id_base + cid) << 32) | ((p - cur_off + 1ull) << 2) | (t.strict_fwd ? 2ull : 0ull);
			if (t.key == 0ull) {
				seed_cas_merge(&ctr->polyA_links, *reinterpret_cast<volatile unsigned long long *>(&ctr->polyA_links), mine);
				continue;
			}
			const uint64_t slot = fast_mod(hash_code(t.key), T.magic);
			const uint4 v = *reinterpret_cast<const uint4 *>(&T.nodes[slot]);
			Node first;
			first.kmer = ((uint64_t)v.y << 32) | v.x;
			first.links = ((uint64_t)v.w << 32) | v.z;
			uint64_t guess;
			const uint64_t s = find_or_claim(T, t.key, slot, first, guess, n_new, n_conf);
			if (s == ~0ull) { full = true; continue; }
			seed_cas_merge(reinterpret_cast<unsigned long long *>(&T.nodes[s].links), guess, mine);
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

// build-time payload -> the reference's bit-field word {id:32, pos:30, freq:1, direct:1}
// (link_scaffold/kmerSet.h:54-61, first field in the low bits)
__host__ __device__ __forceinline__ uint64_t seed_payload_out(uint64_t w)
{
	if (w == 0ull) return 0ull;
	const uint64_t id = w >> 32, pos = ((w >> 2) & 0x3FFFFFFFull) - 1ull, direct = (w >> 1) & 1ull, dup = w & 1ull;
	return id | (pos << 32) | ((dup ^ 1ull) << 62) | (direct << 63);
}

__global__ __launch_bounds__(kBlock) void k_seed_convert(Node *__restrict__ nodes, uint64_t size)
{
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < size; i += stride) nodes[i].links = seed_payload_out(nodes[i].links);
}

// place the key-0 node (only if key 0 was seen) on its probe chain; the slot keeps kmer == 0
__global__ void k_seed_place_key0(TableRef T, Counters *ctr)
{
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	ctr->polyA_slot = ~0ull;
	if (ctr->polyA_links == 0ull) return;
	uint64_t slot = fast_mod(hash_code(0ull), T.magic);
	for (uint64_t steps = 0; steps < T.size; steps++) {
		if (T.nodes[slot].kmer == 0ull) {
			T.nodes[slot].links = seed_payload_out(ctr->polyA_links);
			ctr->polyA_slot = slot;
			return;
		}
		slot = (slot + 1 == T.size) ? 0 : slot + 1;
	}
	atomicOr(&ctr->error, 1u);
}

// ---------------------------------------------------------------------------------------------
// k_extract_store: phase A alone, position-indexed outputs (parity of the extraction)
// ---------------------------------------------------------------------------------------------
template <bool HAS_DEAD>
__global__ __launch_bounds__(kBlock) void k_extract_store(ReadBatch rb, uint64_t *__restrict__ kmer,
                                                          uint8_t *__restrict__ left, uint8_t *__restrict__ right,
                                                          uint8_t *__restrict__ valid)
{
	const uint64_t n_chunks = (rb.n_bases + 15u) >> 4;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t chunk = (uint64_t)blockIdx.x * kBlock + threadIdx.x; chunk < n_chunks; chunk += stride) {
		LaneWindow w = load_lane_window<HAS_DEAD>(rb, chunk);
		for (uint32_t i = 0; i < kPosPerLane; i++) {
			const Triple t = next_triple<HAS_DEAD>(w, i, rb.k, rb.n_bases);
			const uint64_t p = chunk * 16u + i;
			if (p >= rb.n_bases) break;
			valid[p] = t.valid ? 1 : 0;
			kmer[p] = t.valid ? t.key : 0ull;
			left[p] = t.valid ? (uint8_t)t.lb : 4;
			right[p] = t.valid ? (uint8_t)t.rb : 4;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// table-wide passes (one thread per slot, grid-stride)
// ---------------------------------------------------------------------------------------------

// merge pre-aggregated nodes: insert-if-absent + per-byte saturating add (multi-GPU merge, rehash)
// only_if_gt / than: the list is scanned only if *only_if_gt > than (the heavy-hitter side table: in use once the overflow list is full)
__global__ __launch_bounds__(kBlock) void k_merge_nodes(const Node *__restrict__ in, uint64_t n, TableRef T,
                                                        Counters *__restrict__ ctr, const unsigned long long *__restrict__ only_if_gt = nullptr,
                                                        unsigned long long than = 0ull)
{
	if (only_if_gt && *only_if_gt <= than) return; // (kernel-uniform)
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
		const uint4 v = *reinterpret_cast<const uint4 *>(&in[i]);
		const uint64_t key = ((uint64_t)v.y << 32) | v.x;
		const uint64_t add = ((uint64_t)v.w << 32) | v.z;
		if (key == 0ull) {
			// the key-0 node of another shard; an all-zero record is padding and adds nothing
			links_cas_merge(&ctr->polyA_links, 0ull, add);
			continue;
		}
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

// re-seat every node of `src` into `dst` (different size): keys are unique, so the claim is the
// only atomic needed (device counterpart of enlarge_kmerset_parallel, kmerSet.cpp:132-189)
__global__ __launch_bounds__(kBlock) void k_rehash(const Node *__restrict__ src, uint64_t src_size, TableRef dst,
                                                   Counters *__restrict__ ctr, const unsigned long long *__restrict__ src_first = nullptr,
                                                   unsigned long long *__restrict__ dst_first = nullptr)
{
	bool full = false;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < src_size; i += stride) {
		const uint4 v = *reinterpret_cast<const uint4 *>(&src[i]);
		const uint64_t key = ((uint64_t)v.y << 32) | v.x;
		if (key == 0ull) continue;
		const uint64_t links = ((uint64_t)v.w << 32) | v.z;
		uint64_t slot = fast_mod(hash_code(key), dst.magic);
		bool placed = false;
		for (uint64_t steps = 0; steps <= dst.size; steps++) {
			const unsigned long long seen =
			    atomicCAS(reinterpret_cast<unsigned long long *>(&dst.nodes[slot].kmer), 0ull, (unsigned long long)key);
			if (seen == 0ull) {
				dst.nodes[slot].links = links;
				if (src_first) dst_first[slot] = src_first[i]; // first-seen position travels with the node
				placed = true;
				break;
			}
			slot = (slot + 1 == dst.size) ? 0 : slot + 1;
		}
		if (!placed) full = true;
	}
	if (full) atomicOr(&ctr->error, 1u);
}

// place the key-0 node on key 0's probe chain for export (add_node_to_kmerset, kmerSet.cpp:253-273:
// first free slot from hash_code(0) % size).  Single thread.  The slot keeps kmer == 0; only its
// link word is written, and k_unplace_polyA undoes it so later merges still see an empty slot.
__global__ void k_place_polyA(TableRef T, Counters *ctr)
{
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	uint64_t slot = fast_mod(hash_code(0ull), T.magic);
	for (uint64_t steps = 0; steps < T.size; steps++) {
		if (T.nodes[slot].kmer == 0ull) {
			T.nodes[slot].links = ctr->polyA_links;
			ctr->polyA_slot = slot;
			return;
		}
		slot = (slot + 1 == T.size) ? 0 : slot + 1;
	}
	ctr->polyA_slot = ~0ull;
	atomicOr(&ctr->error, 1u);
}

__global__ void k_unplace_polyA(TableRef T, Counters *ctr)
{
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	if (ctr->polyA_slot != ~0ull) T.nodes[ctr->polyA_slot].links = 0ull;
	ctr->polyA_slot = ~0ull;
}

// nul_flag bitmap of the reference (bit i = byte i/8, mask 128 >> (i%8); kmerSet.cpp:53):
// one thread per flag BYTE (8 slots)
__global__ __launch_bounds__(kBlock) void k_build_flags(const Node *__restrict__ nodes, uint64_t size,
                                                        uint64_t polyA_slot, uint8_t *__restrict__ flags)
{
	const uint64_t n_bytes = size / 8 + 1;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_bytes; b += stride) {
		uint32_t byte = 0;
#pragma unroll
		for (uint32_t j = 0; j < 8; j++) {
			const uint64_t i = b * 8 + j;
			if (i < size && (nodes[i].kmer != 0ull || i == polyA_slot)) byte |= 0x80u >> j;
		}
		flags[b] = (uint8_t)byte;
	}
}

// compact the occupied slots into SoA (keys, links) for sorting; order arbitrary
__global__ __launch_bounds__(kBlock) void k_compact(const Node *__restrict__ nodes, uint64_t size,
                                                    uint64_t *__restrict__ keys, uint64_t *__restrict__ links,
                                                    unsigned long long *__restrict__ cursor, uint64_t capacity)
{
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	const uint64_t n_iter = (size + stride - 1) / stride;
	for (uint64_t it = 0; it < n_iter; it++) {
		const uint64_t i = it * stride + (uint64_t)blockIdx.x * kBlock + threadIdx.x;
		Node nd;
		nd.kmer = 0;
		nd.links = 0;
		if (i < size) {
			const uint4 v = *reinterpret_cast<const uint4 *>(&nodes[i]);
			nd.kmer = ((uint64_t)v.y << 32) | v.x;
			nd.links = ((uint64_t)v.w << 32) | v.z;
		}
		const bool occ = nd.kmer != 0ull;
		const unsigned long long ballot = __ballot(occ);
		const int lane = threadIdx.x & 63;
		unsigned long long base = 0;
		if (lane == 0 && ballot) base = atomicAdd(cursor, (unsigned long long)__popcll(ballot));
		base = __shfl(base, 0, 64);
		if (occ) {
			const uint64_t dst = base + (uint64_t)__popcll(ballot & ((1ull << lane) - 1ull));
			if (dst < capacity) {
				keys[dst] = nd.kmer;
				links[dst] = nd.links;
			}
		}
	}
}

// compact occupied slots as (first_pos, slot index) pairs for the ordered export
__global__ __launch_bounds__(kBlock) void k_compact_order(const Node *__restrict__ nodes, const unsigned long long *__restrict__ first_pos,
                                                          uint64_t size, uint64_t *__restrict__ keys_pos, uint64_t *__restrict__ vals_slot,
                                                          unsigned long long *__restrict__ cursor, uint64_t capacity)
{
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	const uint64_t n_iter = (size + stride - 1) / stride;
	for (uint64_t it = 0; it < n_iter; it++) {
		const uint64_t i = it * stride + (uint64_t)blockIdx.x * kBlock + threadIdx.x;
		const bool occ = i < size && nodes[i].kmer != 0ull;
		const unsigned long long ballot = __ballot(occ);
		const int lane = threadIdx.x & 63;
		unsigned long long base = 0;
		if (lane == 0 && ballot) base = atomicAdd(cursor, (unsigned long long)__popcll(ballot));
		base = __shfl(base, 0, 64);
		if (occ) {
			const uint64_t dst = base + (uint64_t)__popcll(ballot & ((1ull << lane) - 1ull));
			if (dst < capacity) {
				keys_pos[dst] = first_pos[i];
				vals_slot[dst] = i;
			}
		}
	}
}

// out[i] = {node at slot vals_slot[i]} , pos_out[i] = keys_pos[i]   (after the pairs were sorted by position)
__global__ __launch_bounds__(kBlock) void k_gather_nodes(const Node *__restrict__ nodes, const uint64_t *__restrict__ vals_slot, uint64_t n,
                                                         Node *__restrict__ out)
{
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = nodes[vals_slot[i]];
}

// order-independent digest + occupied count
__global__ __launch_bounds__(kBlock) void k_digest(const Node *__restrict__ nodes, uint64_t size,
                                                   unsigned long long *__restrict__ out /* [0]=digest [1]=count */)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long acc = 0, cnt = 0;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < size; i += stride) {
		const uint4 v = *reinterpret_cast<const uint4 *>(&nodes[i]);
		const uint64_t key = ((uint64_t)v.y << 32) | v.x;
		if (key != 0ull) {
			acc += node_digest(key, ((uint64_t)v.w << 32) | v.z);
			cnt++;
		}
	}
	const unsigned long long a = block_sum(acc, red);
	const unsigned long long c = block_sum(cnt, red);
	if (threadIdx.x == 0) {
		atomicAdd(&out[0], a);
		atomicAdd(&out[1], c);
	}
}

// calculate_kmer_links first pass (contig.cpp:119-181): DepthStat[256] over all 8 counters of
// every node and the node classes.  out: [0..255] depth_stat, [256] total, [257] deleted,
// [258] linear, [259] tip, [260] branch
__device__ __forceinline__ void link_classes(uint64_t links, int cutoff, unsigned int *hist, unsigned long long *cls)
{
	int ln = 0, rn = 0;
#pragma unroll
	for (int b = 0; b < 8; b++) {
		const unsigned int d = (unsigned int)(links >> (8 * b)) & 0xFFu;
		atomicAdd(&hist[d], 1u);
		if ((int)d > cutoff) {
			if (b < 4) ln++; else rn++;
		}
	}
	if (ln > 3) ln = 3;
	if (rn > 3) rn = 3;
	cls[0]++;
	if (ln == 0 && rn == 0) cls[1]++;
	if (ln == 1 && rn == 1) cls[2]++;
	if (ln + rn == 1) cls[3]++;
	if (ln > 1 || rn > 1) cls[4]++;
}

__global__ __launch_bounds__(kBlock) void k_link_stats(const Node *__restrict__ nodes, uint64_t size, int cutoff,
                                                       uint64_t polyA_links, int with_polyA, unsigned long long *__restrict__ out)
{
	__shared__ unsigned int hist[256];
	__shared__ unsigned long long red[kBlock / 64];
	hist[threadIdx.x] = 0; // kBlock == 256
	__syncthreads();
	unsigned long long cls[5] = {0, 0, 0, 0, 0};
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < size; i += stride) {
		const uint4 v = *reinterpret_cast<const uint4 *>(&nodes[i]);
		const uint64_t key = ((uint64_t)v.y << 32) | v.x;
		if (key != 0ull) link_classes(((uint64_t)v.w << 32) | v.z, cutoff, hist, cls);
	}
	if (with_polyA && blockIdx.x == 0 && threadIdx.x == 0) link_classes(polyA_links, cutoff, hist, cls); // the key-0 node
	__syncthreads();
	if (hist[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
#pragma unroll
	for (int c = 0; c < 5; c++) {
		const unsigned long long s = block_sum(cls[c], red);
		if (threadIdx.x == 0 && s) atomicAdd(&out[256 + c], s);
	}
}

// ---------------------------------------------------------------------------------------------
// The WHOLE first pass of the consumer, calculate_kmer_links (DBG_contig/contig.cpp:107-181), on the host-layout image of
// the table (the one dbgk_export_host_table is about to copy out): per slot the 2-byte KmerLink record (contig.h:31-42:
// l_link_num:2 | l_link_base:2 | r_link_num:2 | r_link_base:2 in the first byte, linear in bit 0 of the second), the
// del_flag bit of nodes without any link above the cutoff (MSB first like nul_flag, set_entity_delete kmerSet.h:161-164),
// DepthStat / class counts, and tip_nodes / branch_nodes as ASCENDING slot lists (the order of the reference's slot loop,
// :119).  One block owns kLinkChunk consecutive slots; PASS 0 counts its tips / branches, the host prefix-sums the block
// counts, PASS 1 writes the lists in order.
// ---------------------------------------------------------------------------------------------
constexpr int kLinkChunk = 4096; // slots per block: 16 sweeps of 256 threads

__device__ __forceinline__ uint32_t kmer_link_record(uint64_t links, int cutoff)
{
	// contig.cpp:129-163: a side's link number = counters above the cutoff (at most 3: a 2-bit field), its base = the FIRST
	// base with the largest such counter (strict <), 0 when there is none
	uint32_t rec = 0;
#pragma unroll
	for (int side = 0; side < 2; side++) {
		const uint32_t w = side ? (uint32_t)(links >> 32) : (uint32_t)links;
		int num = 0, best = 0, base = 0;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int d = (int)((w >> (24 - 8 * j)) & 0xFFu); // get_next_kmer_depth (kmerSet.cpp:341-344): A in bits 31..24
			if (d > cutoff) {
				if (num < 3) num++;
				if (best < d) { best = d; base = j; }
			}
		}
		rec |= ((uint32_t)num | ((uint32_t)base << 2)) << (4 * side);
	}
	if ((rec & 3u) == 1u && ((rec >> 4) & 3u) == 1u) rec |= 1u << 8; // linear: exactly one link on each side (:170-173)
	return rec;
}

template <int PASS>
__global__ __launch_bounds__(kBlock) void k_kmer_links(const Node *__restrict__ nodes, uint64_t size, const Counters *__restrict__ ctr, int cutoff,
                                                       uint16_t *__restrict__ klink, uint8_t *__restrict__ del_flag, unsigned long long *__restrict__ stats,
                                                       uint32_t *__restrict__ block_counts, const unsigned long long *__restrict__ block_base,
                                                       unsigned long long *__restrict__ tips, unsigned long long *__restrict__ branches)
{
	__shared__ unsigned int hist[256];
	__shared__ unsigned long long red[kBlock / 64];
	__shared__ unsigned int wave_cnt[2][kBlock / 64];
	__shared__ unsigned long long run[2];
	const uint64_t polyA_slot = ctr->polyA_slot; // where the key-0 node was placed for this export
	const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
	if (PASS == 0) hist[t] = 0; // kBlock == 256
	if (t == 0) {
		run[0] = PASS ? block_base[2 * blockIdx.x] : 0ull;
		run[1] = PASS ? block_base[2 * blockIdx.x + 1] : 0ull;
	}
	__syncthreads();
	unsigned long long cls[5] = {0, 0, 0, 0, 0};
	const uint64_t first = (uint64_t)blockIdx.x * kLinkChunk;
	for (int sweep = 0; sweep < kLinkChunk / kBlock; sweep++) {
		const uint64_t i = first + (uint64_t)sweep * kBlock + t;
		bool occ = false;
		uint32_t rec = 0;
		if (i < size) {
			const uint4 v = *reinterpret_cast<const uint4 *>(&nodes[i]);
			const uint64_t key = ((uint64_t)v.y << 32) | v.x, links = ((uint64_t)v.w << 32) | v.z;
			occ = key != 0ull || i == polyA_slot;
			if (occ) {
				rec = kmer_link_record(links, cutoff);
				if (PASS == 0) link_classes(links, cutoff, hist, cls);
			}
		}
		const uint32_t ln = rec & 3u, rn = (rec >> 4) & 3u;
		const bool del = occ && ln == 0u && rn == 0u, tip = occ && ln + rn == 1u, branch = occ && (ln > 1u || rn > 1u);
		if (PASS == 0) {
			if (i < size) klink[i] = (uint16_t)rec;
			// del_flag byte of 8 consecutive slots, bit of slot i = 128 >> (i % 8): lanes 8b .. 8b+7 of the wave's ballot
			const unsigned long long m = __ballot(del);
			if ((lane & 7u) == 0u && i < size) {
				const uint32_t bits = (uint32_t)(m >> lane) & 0xFFu;
				del_flag[i >> 3] = (uint8_t)(__brev(bits) >> 24);
			}
		}
		// ordered positions: threads in slot order inside a sweep, sweeps in order
		const unsigned long long mt = __ballot(tip), mb = __ballot(branch);
		const unsigned long long below = (1ull << lane) - 1ull;
		if (lane == 0u) {
			wave_cnt[0][wave] = (unsigned int)__popcll(mt);
			wave_cnt[1][wave] = (unsigned int)__popcll(mb);
		}
		__syncthreads();
		if (PASS == 1) {
			unsigned long long bt = run[0], bb = run[1];
			for (uint32_t w = 0; w < wave; w++) { bt += wave_cnt[0][w]; bb += wave_cnt[1][w]; }
			if (tip) tips[bt + (unsigned long long)__popcll(mt & below)] = i;
			if (branch) branches[bb + (unsigned long long)__popcll(mb & below)] = i;
		}
		__syncthreads();
		if (t == 0) {
			for (uint32_t w = 0; w < kBlock / 64; w++) { run[0] += wave_cnt[0][w]; run[1] += wave_cnt[1][w]; }
		}
		__syncthreads();
	}
	if (PASS == 0) {
		if (t == 0) {
			block_counts[2 * blockIdx.x] = (uint32_t)run[0];
			block_counts[2 * blockIdx.x + 1] = (uint32_t)run[1];
		}
		if (hist[t]) atomicAdd(&stats[t], (unsigned long long)hist[t]);
#pragma unroll
		for (int c = 0; c < 5; c++) {
			const unsigned long long s = block_sum(cls[c], red);
			if (t == 0 && s) atomicAdd(&stats[256 + c], s);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// multi-GPU: nodes grouped by owner = (hash_code(key) >> 32) % n_parts; key 0 -> part 0
// ---------------------------------------------------------------------------------------------
constexpr int kMaxParts = 64;

__device__ __forceinline__ uint32_t owner_of(uint64_t key, uint32_t n_parts)
{
	return (uint32_t)((hash_code(key) >> 32) % n_parts);
}

__global__ __launch_bounds__(kBlock) void k_partition_count(const Node *__restrict__ nodes, uint64_t size,
                                                            uint32_t n_parts, unsigned long long *__restrict__ counts)
{
	__shared__ unsigned int local[kMaxParts];
	if (threadIdx.x < kMaxParts) local[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < size; i += stride) {
		const uint64_t key = nodes[i].kmer;
		if (key != 0ull) atomicAdd(&local[owner_of(key, n_parts)], 1u);
	}
	__syncthreads();
	if (threadIdx.x < n_parts && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)local[threadIdx.x]);
}

// cursors[p] starts at the part's base offset; every block reserves one range per part per sweep
__global__ __launch_bounds__(kBlock) void k_partition_scatter(const Node *__restrict__ nodes, uint64_t size,
                                                              uint32_t n_parts, unsigned long long *__restrict__ cursors,
                                                              Node *__restrict__ out, uint64_t capacity)
{
	__shared__ unsigned int local[kMaxParts];
	__shared__ unsigned long long base[kMaxParts];
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	const uint64_t n_iter = (size + stride - 1) / stride;
	for (uint64_t it = 0; it < n_iter; it++) {
		if (threadIdx.x < kMaxParts) local[threadIdx.x] = 0;
		__syncthreads();
		const uint64_t i = it * stride + (uint64_t)blockIdx.x * kBlock + threadIdx.x;
		uint4 v = make_uint4(0, 0, 0, 0);
		if (i < size) v = *reinterpret_cast<const uint4 *>(&nodes[i]);
		const uint64_t key = ((uint64_t)v.y << 32) | v.x;
		uint32_t part = 0, rank = 0;
		if (key != 0ull) {
			part = owner_of(key, n_parts);
			rank = atomicAdd(&local[part], 1u);
		}
		__syncthreads();
		if (threadIdx.x < n_parts && local[threadIdx.x])
			base[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], (unsigned long long)local[threadIdx.x]);
		__syncthreads();
		if (key != 0ull) {
			const uint64_t dst = base[part] + rank;
			if (dst < capacity) *reinterpret_cast<uint4 *>(&out[dst]) = v;
		}
		__syncthreads();
	}
}

// ---------------------------------------------------------------------------------------------
// synthetic reads straight into HBM (include/dbgk_synth.h); one thread per 16 output bytes
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_synth_reads(dbgk_synth_params P, uint64_t first_read, uint64_t n_reads,
                                                        char *__restrict__ bases, uint64_t *__restrict__ offsets)
{
	const uint64_t L = P.read_len;
	const uint64_t total = n_reads * L;
	const uint64_t n_chunks = (total + 15u) >> 4;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c < n_chunks; c += stride) {
		uint32_t w[4] = {0, 0, 0, 0};
		uint64_t cur_read = ~0ull, start = 0;
		uint32_t strand = 0;
		for (uint32_t b = 0; b < 16; b++) {
			const uint64_t g = c * 16u + b;
			if (g >= total) break;
			const uint64_t r = g / L;
			const uint32_t j = (uint32_t)(g - r * L);
			if (r != cur_read) {
				cur_read = r;
				dbgk_synth_read_origin(&P, first_read + r, &start, &strand);
			}
			const uint32_t ch = (uint8_t)dbgk_synth_read_base_at(&P, first_read + r, j, start, strand);
			w[b >> 2] |= ch << ((b & 3u) * 8u);
		}
		if (c * 16u + 16u <= total) {
			*reinterpret_cast<uint4 *>(bases + c * 16u) = make_uint4(w[0], w[1], w[2], w[3]);
		} else {
			for (uint32_t b = 0; c * 16u + b < total; b++) bases[c * 16u + b] = (char)(w[b >> 2] >> ((b & 3u) * 8u));
		}
	}
	for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r <= n_reads; r += stride) offsets[r] = r * L;
}

// ---------------------------------------------------------------------------------------------
// bytes outside ACGTNacgtn (dbgk_device.h: read as 'A'): counted per batch, but only once an extraction kernel has met one
// (Counters::other_seen) -- every workgroup looks at the flag first and leaves, so a clean input pays one empty launch
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t count_other4(uint32_t w)
{
	const uint32_t d = other4_ascii(w);
	return (uint32_t)__builtin_popcount((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u);
}

__global__ __launch_bounds__(kBlock) void k_count_other_bytes(const char *__restrict__ bases, uint64_t n_bases, Counters *__restrict__ ctr)
{
	if (*reinterpret_cast<volatile unsigned int *>(&ctr->other_seen) == 0u) return; // (kernel-uniform)
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n = 0;
	const uint64_t n_vec = n_bases >> 4, stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_vec; i += stride) {
		const uint4 v = reinterpret_cast<const uint4 *>(bases)[i];
		n += count_other4(v.x) + count_other4(v.y) + count_other4(v.z) + count_other4(v.w);
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
		for (uint64_t p = n_vec << 4; p < n_bases; p++) n += count_other4((uint32_t)(uint8_t)bases[p] | 0x41414100u);
	const unsigned long long tot = block_sum(n, red);
	if (threadIdx.x == 0 && tot) atomicAdd(&ctr->other_bytes, tot);
}

// The small control arrays of a step (bucket fill counts, list lengths, work cursors, the counters) zeroed by ONE launch: a
// dozen hipMemsetAsync calls cost ~8 us each on the stream and a few more on the host, per step and per flush.
struct ZeroList {
	void *p[8];
	uint32_t dwords[8];
};
__global__ __launch_bounds__(kBlock) void k_zero_list(ZeroList z, Counters *__restrict__ ctr /* may be null: also reset the counters */)
{
	const uint32_t stride = gridDim.x * kBlock, t0 = blockIdx.x * kBlock + threadIdx.x;
#pragma unroll 1
	for (int e = 0; e < 8; e++) {
		uint32_t *q = static_cast<uint32_t *>(z.p[e]);
		for (uint32_t i = t0; i < z.dwords[e]; i += stride) q[i] = 0u;
	}
	if (ctr) { // all zero, except polyA_slot = ~0 (no key-0 node placed)
		uint32_t *q = reinterpret_cast<uint32_t *>(ctr);
		const uint32_t lo = (uint32_t)(offsetof(Counters, polyA_slot) / 4u);
		for (uint32_t i = t0; i < (uint32_t)(sizeof(Counters) / 4u); i += stride) q[i] = (i == lo || i == lo + 1u) ? 0xFFFFFFFFu : 0u;
	}
}
// a buffer that is only ever written when a list overflowed (the heavy-hitter side table: used once the overflow list is full) is
// only then zeroed again -- 64 MiB per step otherwise
__global__ __launch_bounds__(kBlock) void k_zero_if_greater(uint4 *__restrict__ buf, uint64_t n_vec, const unsigned long long *__restrict__ value,
                                                            unsigned long long than)
{
	if (*value <= than) return; // (kernel-uniform)
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_vec; i += stride) buf[i] = make_uint4(0u, 0u, 0u, 0u);
}

// reads of ONE length, handed over without offsets (dbgk_push_reads_packed_uniform*): the totals k_mark would have summed up ...
__global__ void k_add_totals(Counters *__restrict__ ctr, unsigned long long total_kmers, unsigned long long stored_kmers)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		atomicAdd(&ctr->total_kmers, total_kmers);
		atomicAdd(&ctr->stored_kmers, stored_kmers);
	}
}
// ... and, for the kernels that do navigate by offsets (any engine but the PARTITION engine's equal-length forms), the offsets themselves
__global__ __launch_bounds__(kBlock) void k_iota_offsets(uint64_t *__restrict__ offsets, uint64_t n_reads, uint64_t read_len)
{
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r <= n_reads; r += stride) offsets[r] = r * read_len;
}

// ASCII -> 2-bit packed on the device (dbgk_pack_bases_device; the host twin is dbgk_pack_bases): one lane per 16 bases
__global__ __launch_bounds__(kBlock) void k_pack_bases(const char *__restrict__ bases, uint64_t n_bases, uint32_t *__restrict__ packed,
                                                      Counters *__restrict__ ctr)
{
	ReadBatch rb{bases, n_bases, nullptr, nullptr, 0, nullptr, &ctr->other_seen};
	const uint64_t n_chunks = (n_bases + 15u) >> 4, stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c < n_chunks; c += stride) packed[c] = load_packed_chunk(rb, c);
}

} // namespace dbgk
