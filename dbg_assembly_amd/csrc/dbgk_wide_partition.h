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
//   k_wide_scatter_l1[_uniform]  reads -> records, scattered into n1 <= 1024 level-1 buckets (slot >> r); 8192-record
//                         tiles; the _uniform form for batches of equal-length reads (no position straddles a read)
//   k_wide_l2_plan        tiles per level-1 bucket
//   k_wide_scatter_l2     a chunk of level-1 buckets -> n2 = 2^(r-11) final buckets each (slot >> 11); XCD-aware tile order
//   k_wide_build_regions  the regions of that chunk, one 2048-slot region at a time in LDS, emitted as 64 KiB of device nodes
//   afterwards            region spill-over nodes through k_wide_merge_spill, bucket-overflow observations through
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
constexpr int kWL2Threads = 512;                 // level 2: 4096-record tiles
constexpr int kWL2Records = kWL2Threads * 8;
constexpr int kWBuildThreads = 512;              // three workgroups per CU (51 KiB of LDS each)

struct WPartGeom {
	uint64_t size;       // slots of the (GLOBAL) table: slot = hash128(key) % size
	ModMagic magic;
	Div32Magic div;      // size < 2^32: exact 64/32 division (dbgk_device.h)
	uint32_t r;          // level-1 bucket = slot >> r
	uint32_t n1;         // ceil(size / 2^r): level-1 buckets of the whole table (any number, see passes)
	uint32_t n2;         // 2^(r - 11)       <= 2048
	uint32_t n_regions;  // ceil(size / 2048)
	uint32_t chunk_buckets; // level-1 buckets whose final buckets the level-2 store holds at a time (level 2 and the build
	                        // walk the level-1 buckets chunk by chunk: the level-2 store is 1/8 of the records, not all)
	uint64_t cap1, cap2; // records per level-1 / final bucket
	// SHARDS (several GPUs, dbgk_config.shard_count): rank d owns the level-1 buckets [d * B, (d + 1) * B) of the global table,
	// i.e. the slot range [slot_lo, slot_hi), and holds only that part of the table (the device analogue of the reference's
	// `kmer % threadNum` ownership, DBGgraph.cpp:148, as in the 64-bit engine).  n_ranks == 1: the whole table.
	// PASSES: the level-1 kernel fans out to at most 1024 buckets and the record stores must fit the device, so a big job reads
	// its input n_passes times (the way disk-based k-mer counters pass over their input once per set of partitions); pass p
	// keeps the records of the own-bucket indices [p * Bp, (p + 1) * Bp) of EVERY rank and drops the rest, level 2 and the
	// region build then complete those buckets.  n_passes == 1 and n_ranks == 1: level-1 store index == global bucket.
	uint32_t n_ranks, rank;
	uint32_t B;          // level-1 buckets per rank = ceil(n1 / n_ranks)
	uint32_t bmagic;     // ceil(2^32 / B): (b * bmagic) >> 32 == b / B for b < 65536
	uint32_t b_lo, nb_own; // this rank's buckets [b_lo, b_lo + nb_own)
	uint32_t n_passes, pass;
	uint32_t Bp;         // own-bucket indices per pass = ceil(B / n_passes)
	uint32_t pass_j0;    // pass * Bp
	uint32_t n_l1;       // n_ranks * Bp <= 1024: level-1 fan-out of a pass; store entry of (rank d, own index pass_j0 + jj) = d * Bp + jj
	uint64_t slot_lo, slot_hi;
};

// global level-1 bucket of store entry e of the current pass
__device__ __forceinline__ uint32_t wide_entry_bucket(uint32_t e, const WPartGeom &G)
{
	if (G.n_l1 == G.n1) return e; // one rank, one pass
	const uint32_t d = e / G.Bp;
	return d * G.B + G.pass_j0 + (e - d * G.Bp);
}

typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));

struct WPartStore {
	ull2 *l1;                     // [n_l1][cap1]: what this rank extracted in the current pass, by destination rank and own-bucket index
	uint32_t *cnt1;               // [n_l1] records appended (may exceed cap1: the excess went to ovf)
	const ull2 *inbox;            // [n_l1][cap1]: entry s * Bp + jj = rank s's records for MY bucket pass_j0 + jj (n_ranks == 1: == l1)
	const uint32_t *inbox_cnt;    // [n_l1]
	dbgk_node32 *outgoing;        // shards: nodes whose probe ran off the end of this shard, for the next rank
	unsigned long long *outgoing_n;
	uint64_t outgoing_cap;
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
struct WL1Lds : WScatterLds<kWL1Threads, 8, 1024, true> {
	uint16_t bucket_of[kWL1Threads * 8]; // level-1 bucket of every staged record: the copy-out walks the stage linearly
};
template <int MAXB> using WL2Lds = WScatterLds<kWL2Threads, 8, MAXB, true>;

// ---- level 1 --------------------------------------------------------------------------------------
// One lane owns 16 consecutive base positions, handled as two tiles of 8.  As in the 64-bit engine's level 1
// (dbgk_partition.h: decode_chunk16 / l1_positions) the forward and the reverse-complement k-mer ROLL by one base per
// position (DBGgraph.cpp:71-73), the entering / left-neighbour bases of the 16 positions sit in two packed words, and
// the three per-position predicates (window inside one read, has a left / right neighbour) are computed once per lane
// as 16-bit masks with log-step sliding ORs over the 128-bit boundary bitmaps.  A lane loads only ITS 16 bases; the
// four following packed words come from the next lanes by shuffle and from four halo chunks at the wave's end.
typedef unsigned __int128 u128;

// bit i of the result = OR of bits i .. i+w-1 of x (w wave-uniform, 0..63), for the low 16 result bits
__device__ __forceinline__ uint32_t sliding_or128_lo16(u128 x, uint32_t w)
{
	u128 res = 0, cur = x;
	uint32_t done = 0;
#pragma unroll
	for (uint32_t j = 0; j < 6; j++) {
		if (w & (1u << j)) {
			res |= cur >> done;
			done += 1u << j;
		}
		cur |= cur >> (1u << j);
	}
	return (uint32_t)res & 0xFFFFu;
}

struct WChunk16 {
	Key128 fwd, rc;      // forward / reverse-complement k-mer of the window at position 0
	uint32_t nb;         // bases k .. k+15: the base entering at position i+1 = right neighbour of position i
	uint32_t lw;         // bases -1 .. 14: left neighbour of position i
	uint32_t valid, has_l, has_r; // 16-bit masks, bit i <-> position p0 + i
};

template <bool HAS_DEAD>
__device__ __forceinline__ WChunk16 wide_decode_chunk16(const ReadBatch &rb, uint64_t chunk, uint64_t n_chunks)
{
	WChunk16 c;
	const uint32_t k = (uint32_t)rb.k;
	const uint32_t lane = fresh_tid() & 63u;
	const bool live = chunk < n_chunks;
	const uint64_t p0 = chunk * 16u;
	const uint64_t wave_chunk0 = chunk - lane;
	const uint32_t w0 = live ? load_packed_chunk(rb, chunk) : 0u;
	const uint32_t hw = (lane < 4u) ? load_packed_chunk(rb, wave_chunk0 + 64u + lane) : 0u; // (0 beyond the end)
	const uint32_t h0 = __builtin_amdgcn_readlane(hw, 0), h1 = __builtin_amdgcn_readlane(hw, 1), h2 = __builtin_amdgcn_readlane(hw, 2),
	               h3 = __builtin_amdgcn_readlane(hw, 3);
	uint32_t w1 = __shfl_down(w0, 1, 64), w2 = __shfl_down(w0, 2, 64), w3 = __shfl_down(w0, 3, 64), w4 = __shfl_down(w0, 4, 64);
	if (lane == 63u) { w1 = h0; w2 = h1; w3 = h2; w4 = h3; }
	if (lane == 62u) { w2 = h0; w3 = h1; w4 = h2; }
	if (lane == 61u) { w3 = h0; w4 = h1; }
	if (lane == 60u) w4 = h0;
	uint32_t prev = __shfl_up(w0, 1, 64) & 3u; // last base of the previous lane's chunk
	uint32_t pb = 0u;
	if (lane == 0u && live && chunk > 0u) pb = base_code(rb, p0 - 1u);
	pb = __builtin_amdgcn_readlane(pb, 0);
	if (lane == 0u) prev = pb;
	if (chunk == 0u) prev = 0u;
	// 80 bases from p0, MSB first
	const uint64_t A = ((uint64_t)w0 << 32) | w1, B = ((uint64_t)w2 << 32) | w3, C = (uint64_t)w4 << 32;
	if (2u * k <= 64u) {
		c.fwd.hi = 0ull;
		c.fwd.lo = A >> (64u - 2u * k);
	} else {
		const uint32_t sh = 128u - 2u * k; // 2..62
		c.fwd.hi = A >> sh;
		c.fwd.lo = (B >> sh) | (A << (64u - sh));
	}
	c.rc = dbgk_wide::revcomp(c.fwd, (int)k);
	// bases k .. k+15 of the 80: the 32 bits at bit offset 2k of the 160-bit string A:B:C
	{
		const uint32_t off = 2u * k; // 2..126
		const uint64_t top = (uint64_t)(((((u128)A << 64) | B) << off) >> 64);   // bits off .. off+63 of A:B (zeros shifted in)
		const uint64_t tail = off > 64u ? (C >> (128u - off)) : 0ull;           // ... continued by C where the 64 bits reach past B
		c.nb = (uint32_t)((top | tail) >> 32);
	}
	c.lw = (prev << 30) | (w0 >> 2);
	uint64_t S0 = 0, S1 = 0, D0 = 0, D1 = 0;
	if (live) {
		load_bits128(rb.start_bits, p0, rb.n_bases, S0, S1);
		if (HAS_DEAD) load_bits128(rb.dead_bits, p0, rb.n_bases, D0, D1);
	}
	const u128 S = ((u128)S1 << 64) | S0;
	// window i is inside one read iff no read starts at positions i+1 .. i+k-1 (and no dead position in i .. i+k-1)
	uint32_t bad = sliding_or128_lo16(S >> 1, k - 1u);
	uint32_t no_r = (uint32_t)(S >> k) & 0xFFFFu;
	if (HAS_DEAD) {
		const u128 D = ((u128)D1 << 64) | D0;
		bad |= sliding_or128_lo16(D, k);
		no_r |= (uint32_t)(D >> k) & 0xFFFFu;
	}
	const uint64_t room = (live && rb.n_bases > p0) ? rb.n_bases - p0 : 0ull;
	const uint32_t nv = room >= k ? (uint32_t)(room - k + 1u < 16u ? room - k + 1u : 16u) : 0u;
	const uint32_t nr = room > k ? (uint32_t)(room - k < 16u ? room - k : 16u) : 0u;
	c.valid = ~bad & ((1u << nv) - 1u);
	c.has_r = ~no_r & ((1u << nr) - 1u);
	c.has_l = ~(uint32_t)S0 & 0xFFFFu & (p0 ? 0xFFFFu : 0xFFFEu);
	return c;
}

// One tile of level 1: the first 8 positions of every lane's chunk state (bits 0..7 of its masks, the top halves of
// nb / lw) -> records parked, ranked, sorted by bucket in LDS and copied out.  The chunk state is rolled forward by
// 8 positions.  WIDE_D: how hash / size is computed -- 0: size < 2^31, 1: size < 2^32, 2: any size (see l1_positions)
struct WideL1Consts {
	uint32_t rmask, top;
	uint64_t mask_hi, mask_lo;
};

// One position of a lane's chunk: the canonical k-mer's record (key.hi | q, slot bits, neighbour codes) and its store entry
// (1024 + lane: none); the chunk state is rolled to the next position.  rev / zero_lo: reverse strand taken, canonical key.lo == 0.
struct WidePos {
	uint64_t key_hi, w;
	uint32_t b;
	bool rev, zero_lo;
};
template <int WIDE_D>
__device__ __forceinline__ WidePos wide_l1_position(const WPartGeom &G, WChunk16 &c, const WideL1Consts &K, uint32_t i, uint32_t tid)
{
	WidePos o;
	const uint32_t sh = 30u - 2u * i;
	const uint32_t left = (c.lw >> sh) & 3u, right = (c.nb >> sh) & 3u;
	// canonical pick: tie -> forward (DBGgraph.cpp:80); reverse strand: (comp(right), comp(left)) (:82-97).  Both neighbours are
	// assumed to exist; windows at a read's first / last position are patched by the caller.
	const bool rev = c.rc.hi < c.fwd.hi || (c.rc.hi == c.fwd.hi && c.rc.lo < c.fwd.lo);
	const Key128 key{rev ? c.rc.hi : c.fwd.hi, rev ? c.rc.lo : c.fwd.lo};
	const uint32_t links = rev ? (((3u - right) << 3) | (3u - left)) : ((left << 3) | right);
	o.rev = rev;
	const bool valid = (c.valid >> i) & 1u;
	o.zero_lo = valid && key.lo == 0ull; // rare: handled by the caller through the atomic path
	uint64_t q;
	const uint64_t hv = hash_code(key.hi ? (key.lo ^ hash_code(key.hi)) : key.lo);
	uint32_t slot_lo, bucket;
	if (WIDE_D == 2) {
		const uint64_t s64 = fast_divmod(hv, G.magic, q);
		slot_lo = (uint32_t)s64;
		bucket = (uint32_t)(s64 >> G.r);
	} else {
		slot_lo = WIDE_D ? divmod_u64_u32(hv, G.div, q) : divmod_magic_small(hv, G.magic.m, (uint32_t)G.magic.d, q);
		bucket = slot_lo >> G.r;
	}
	o.key_hi = key.hi;
	o.w = (q << (G.r + 6u)) | ((uint64_t)(slot_lo & K.rmask) << 6) | links;
	// store entry of the bucket: (owner rank, own-bucket index inside this pass's window); other passes' buckets are dropped
	uint32_t entry = bucket;
	bool keep = true;
	if (G.n_l1 != G.n1) {
		const uint32_t d = (uint32_t)(((uint64_t)bucket * G.bmagic) >> 32);
		const uint32_t jj = bucket - d * G.B - G.pass_j0;
		keep = jj < G.Bp;
		entry = d * G.Bp + jj;
	}
	o.b = (valid && key.lo != 0ull && keep) ? entry : 1024u + (tid & 63u);
	// roll to the next position
	c.fwd.hi = ((c.fwd.hi << 2) | (c.fwd.lo >> 62)) & K.mask_hi;
	c.fwd.lo = ((c.fwd.lo << 2) | right) & K.mask_lo;
	c.rc.lo = (c.rc.lo >> 2) | (c.rc.hi << 62);
	c.rc.hi >>= 2;
	const uint64_t comp = (uint64_t)(3u - right);
	if (K.top >= 64u) c.rc.hi |= comp << (K.top - 64u); else c.rc.lo |= comp << K.top;
	return o;
}

template <int WIDE_D>
__device__ __forceinline__ void wide_l1_tile(WL1Lds &L, const WPartGeom &G, const WPartStore &P, const WTable &T, Counters *ctr, WChunk16 &c,
                                             const WideL1Consts &K, unsigned long long &n_new, unsigned long long &n_conf, bool &full)
{
	const uint32_t tid = fresh_tid();
	uint32_t bkt[8];
	uint32_t rev_mask = 0, zero_lo = 0;
#pragma unroll
	for (uint32_t i = 0; i < 8u; i++) {
		const WidePos o = wide_l1_position<WIDE_D>(G, c, K, i, tid);
		rev_mask |= (o.rev ? 1u : 0u) << i;
		zero_lo |= (o.zero_lo ? 1u : 0u) << i;
		L.stage[i * kWL1Threads + tid] = ull2{o.key_hi, o.w};
		bkt[i] = (o.b << 16) | atomicAdd(&L.hist[o.b], 1u);
	}
	// windows without a left / right neighbour: that side's code becomes 4 = none; keys with lo == 0 go through the
	// atomic path with their final codes (their parked record sits in a dummy bin and is never copied out)
	const uint32_t no_l = ~c.has_l & 0xFFu, no_r = ~c.has_r & 0xFFu, v8 = c.valid & 0xFFu;
	for (uint32_t fix = ((no_l | no_r) & v8) | zero_lo; fix; fix &= fix - 1u) {
		const uint32_t i = (uint32_t)__builtin_ctz(fix);
		const bool fwd_strand = !((rev_mask >> i) & 1u), nl = (no_l >> i) & 1u, nr = (no_r >> i) & 1u;
		ull2 rec = L.stage[i * kWL1Threads + tid];
		uint32_t lb = ((uint32_t)rec.y >> 3) & 7u, rbb = (uint32_t)rec.y & 7u;
		if (fwd_strand ? nl : nr) lb = 4u;
		if (fwd_strand ? nr : nl) rbb = 4u;
		rec.y = (rec.y & ~63ull) | (lb << 3) | rbb;
		if ((zero_lo >> i) & 1u) {
			if (G.pass == 0u) wide_insert(T, Key128{rec.x, 0ull}, lb, rbb, ctr, n_new, n_conf, full); // (once per job: the input is read once per pass)
		} else {
			L.stage[i * kWL1Threads + tid] = rec;
		}
	}
	c.lw <<= 16;
	c.nb <<= 16;
	c.valid >>= 8;
	c.has_l >>= 8;
	c.has_r >>= 8;
	lds_barrier(); // histogram complete, parked records visible
	ull2 rec[8];
#pragma unroll
	for (uint32_t i = 0; i < 8u; i++) rec[i] = L.stage[i * kWL1Threads + fresh_tid()];
	uint32_t my_gbase[1];
	scatter_reserve_scan(L, G.n_l1, P.cnt1, my_gbase);
#pragma unroll
	for (uint32_t i = 0; i < 8u; i++)
		if ((bkt[i] >> 16) < 1024u) {
			const uint32_t at = L.lbase[bkt[i] >> 16] + (bkt[i] & 0xFFFFu);
			L.stage[at] = rec[i];
			L.bucket_of[at] = (uint16_t)(bkt[i] >> 16);
		}
	L.desc[fresh_tid()] = my_gbase[0];
	lds_barrier();
	// copy-out: the sorted stage is walked linearly, one 16-byte record per lane and every lane busy -- with several
	// hundred level-1 buckets a tile holds only a handful of records per bucket, and the memory pipe charges a store
	// instruction the same whatever its lane count (dbgk_partition.h)
	{
		const uint32_t t = fresh_tid();
		const uint32_t total = L.lbase[G.n_l1 - 1u] + L.hist[G.n_l1 - 1u];
#pragma unroll
		for (uint32_t u = 0; u < 8u; u++) {
			const uint32_t p = u * kWL1Threads + t;
			if (p >= total) continue;
			const ull2 rcd = L.stage[p];
			const uint32_t b = L.bucket_of[p];
			const uint64_t off = (uint64_t)L.desc[b] + (p - L.lbase[b]);
			if (off < G.cap1) {
				P.l1[(uint64_t)b * G.cap1 + off] = rcd;
			} else { // the bucket is full
				wide_push_overflow(P, wide_record_key(rcd, wide_entry_bucket(b, G), G), (uint32_t)(rcd.y >> 3) & 7u, (uint32_t)rcd.y & 7u, ctr);
			}
		}
	}
	lds_barrier(); // the next tile parks its records in the stage buffer again
}

__device__ __forceinline__ WideL1Consts wide_l1_consts(uint32_t k, const WPartGeom &G)
{
	WideL1Consts K;
	K.rmask = (1u << G.r) - 1u;
	K.top = 2u * k - 2u; // bit position of a window's first base
	K.mask_hi = k > 32u ? ((1ull << (2u * k - 64u)) - 1ull) : 0ull;
	K.mask_lo = k >= 32u ? ~0ull : ((1ull << (2u * k)) - 1ull);
	return K;
}

__device__ __forceinline__ void wide_l1_finish(unsigned long long n_new, unsigned long long n_conf, bool full, Counters *ctr)
{
	__shared__ unsigned long long red[kWL1Threads / 64];
	const unsigned long long a = block_sum_n<kWL1Threads>(n_new, red);
	const unsigned long long b = block_sum_n<kWL1Threads>(n_conf, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
	}
	if (full) atomicOr(&ctr->error, 1u);
}

// the general form: lanes own flat base positions, 16 each, as two tiles
template <bool HAS_DEAD, int WIDE_D>
__global__ __launch_bounds__(kWL1Threads) void k_wide_scatter_l1(ReadBatch rb, WPartGeom G, WPartStore P, WTable T, Counters *__restrict__ ctr)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	WL1Lds &L = *reinterpret_cast<WL1Lds *>(lds_raw);
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint64_t n_chunks = (rb.n_bases + 15u) >> 4;
	const WideL1Consts K = wide_l1_consts((uint32_t)rb.k, G);
	for (uint64_t c0 = (uint64_t)blockIdx.x * kWL1Threads; c0 < n_chunks; c0 += (uint64_t)gridDim.x * kWL1Threads) {
		WChunk16 c = wide_decode_chunk16<HAS_DEAD>(rb, c0 + fresh_tid(), n_chunks);
#pragma unroll 1
		for (uint32_t half = 0; half < 2u; half++) {
			const uint32_t tid = fresh_tid();
			L.hist[tid] = 0u; // kBpt == 1
			if (tid < 64u) L.hist[1024u + tid] = 0u;
			lds_barrier();
			wide_l1_tile<WIDE_D>(L, G, P, T, ctr, c, K, n_new, n_conf, full);
		}
	}
	wide_l1_finish(n_new, n_conf, full, ctr);
}

// Batches of EQUAL-LENGTH reads (no read longer than maxReadLen): lanes are mapped to chunks of 8 VALID windows -- read
// lane / Q, chunk lane % Q, Q = ceil(W / 8), W = L - k + 1 -- instead of flat base positions, so no position a lane
// hashes straddles a read boundary (62 of 150 do at k = 63) and no boundary bitmap is needed.  As in
// k_extract_scatter_uniform the tile's byte range is packed cooperatively into LDS (it borrows the bucket-tag array,
// which is only used from the staging on) and every lane funnels its 80 bases out of seven packed words.
struct WUniformGeom {
	uint32_t L;          // length of every read of the batch, k <= L <= maxReadLen
	uint32_t W;          // windows of a read = L - k + 1
	uint32_t Q;          // lanes per read = ceil(W / 8)
	uint32_t qmagic;     // ceil(2^22 / Q): (x * qmagic) >> 22 == x / Q for x < 2048 + Q
	uint64_t n_lanes;    // n_reads * Q
};
constexpr uint32_t kWPkWords = kWL1Threads * 8 * 2 / 4; // the bucket-tag array seen as 32-bit words (4096)

// PIPE (round 5, as the 64-bit engine's regular tiles): the copy-out of a tile runs inside the position loop of the next one -- the
// records of a tile stay in registers until they are staged in sorted order, the stage buffer, the bucket tags and the descriptors
// keep the tile before while the next one is ranked, and the tile's packed words get an array of their own (kWPipePkWords: what is
// left of the 160 KiB beside the stage buffer -- the host takes this form when a tile's byte range fits, wide_uniform_mode).
struct WL1PipeLds : WL1Lds {
	static constexpr int kPkWords = (160 * 1024 - 128 - (int)sizeof(WL1Lds)) / 4; // (128 bytes: the static array of wide_l1_finish)
	uint32_t pk[kPkWords];
};
constexpr uint32_t kWPipePkWords = WL1PipeLds::kPkWords;

template <int WIDE_D, bool PIPE = false>
__global__ __launch_bounds__(kWL1Threads) void k_wide_scatter_l1_uniform(ReadBatch rb, WUniformGeom U, WPartGeom G, WPartStore P, WTable T,
                                                                        Counters *__restrict__ ctr)
{
	using LdsT = typename std::conditional<PIPE, WL1PipeLds, WL1Lds>::type;
	extern __shared__ __align__(16) unsigned char lds_raw[];
	LdsT &L = *reinterpret_cast<LdsT *>(lds_raw);
	uint32_t *pk;
	if constexpr (PIPE) pk = L.pk;
	else pk = reinterpret_cast<uint32_t *>(L.bucket_of);
	constexpr uint32_t kPk = PIPE ? kWPipePkWords : kWPkWords;
	unsigned long long n_new = 0, n_conf = 0;
	bool full = false;
	const uint32_t k = (uint32_t)rb.k;
	const WideL1Consts K = wide_l1_consts(k, G);
	const uint64_t n_tiles = (U.n_lanes + kWL1Threads - 1) / kWL1Threads;
	// a tile's byte range: from one base before its first lane's first window to the end of its last lane's windows
	struct Range {
		uint64_t r0, B0;
		uint32_t c0, n_blocks;
	};
	auto range_of = [&](uint64_t tile) {
		Range R{0, 0, 0, 0};
		if (tile >= n_tiles) return R;
		const uint64_t lane0 = tile * kWL1Threads;
		R.r0 = lane0 / U.Q;
		R.c0 = (uint32_t)(lane0 - R.r0 * U.Q);
		const uint64_t p_first = R.r0 * U.L + 8u * R.c0;
		R.B0 = (p_first ? p_first - 1u : 0u) & ~15ull;
		const uint64_t lane_last = min(lane0 + kWL1Threads, U.n_lanes) - 1u;
		const uint32_t xl = R.c0 + (uint32_t)(lane_last - lane0);
		const uint32_t drl = (xl * U.qmagic) >> 22;
		uint64_t end = (R.r0 + drl) * U.L + 8u * (xl - drl * U.Q) + 8u + k + 2u;
		end = min(end, (rb.n_bases + 15u) & ~15ull);
		R.n_blocks = end > R.B0 ? min((uint32_t)((end - R.B0 + 15u) >> 4), kPk - 8u) : 0u;
		return R;
	};
	// the first two 16-byte blocks of a lane (block tid, block tid + 1024) are requested one tile ahead
	auto fetch = [&](const Range &R, uint4 &a, uint4 &b) {
		const uint32_t t = fresh_tid();
		a = b = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
		if (rb.packed) { // (wave-uniform) a 2-bit packed batch: .x holds the packed word
			if (t < R.n_blocks) a.x = packed_word(rb, (R.B0 >> 4) + t);
			if (t + kWL1Threads < R.n_blocks) b.x = packed_word(rb, (R.B0 >> 4) + t + kWL1Threads);
		} else {
			if (t < R.n_blocks) a = load_ascii16(rb.bases, rb.n_bases, (R.B0 >> 4) + t);
			if (t + kWL1Threads < R.n_blocks) b = load_ascii16(rb.bases, rb.n_bases, (R.B0 >> 4) + t + kWL1Threads);
		}
	};
	// the tile's packed words into LDS (+ 8 words of 'A' padding: a lane reads seven words from its first)
	auto fill_pk = [&](const Range &R, const uint4 &ra, const uint4 &rb2) {
		const uint32_t tid = fresh_tid(), n_blocks = R.n_blocks;
		if (rb.packed) {
			if (tid < n_blocks + 8u) pk[tid] = tid < n_blocks ? ra.x : 0u;
			if (tid + kWL1Threads < n_blocks + 8u) pk[tid + kWL1Threads] = tid + kWL1Threads < n_blocks ? rb2.x : 0u;
			for (uint32_t b = tid + 2u * kWL1Threads; b < n_blocks + 8u; b += kWL1Threads) // long reads: the rest of the range, loaded here
				pk[b] = b < n_blocks ? packed_word(rb, (R.B0 >> 4) + b) : 0u;
		} else {
			if (tid < n_blocks + 8u) pk[tid] = tid < n_blocks ? pack16_ascii(ra, rb.other_seen) : 0u;
			if (tid + kWL1Threads < n_blocks + 8u) pk[tid + kWL1Threads] = tid + kWL1Threads < n_blocks ? pack16_ascii(rb2, rb.other_seen) : 0u;
			for (uint32_t b = tid + 2u * kWL1Threads; b < n_blocks + 8u; b += kWL1Threads) // long reads: the rest of the range, loaded here
				pk[b] = b < n_blocks ? pack16_ascii(load_ascii16(rb.bases, rb.n_bases, (R.B0 >> 4) + b), rb.other_seen) : 0u;
		}
	};
	// a lane's chunk of 8 windows out of the packed words
	auto decode = [&](const Range &R, uint64_t lane0) {
		const uint32_t tid = fresh_tid();
		WChunk16 c;
		const uint32_t x = R.c0 + tid;
		const uint32_t dr = (x * U.qmagic) >> 22;
		const uint32_t cc = x - dr * U.Q;
		const uint32_t first_w = 8u * cc;   // index of the lane's first window inside its read
		const bool live = lane0 + tid < U.n_lanes && first_w < U.W;
		const uint64_t p = (R.r0 + dr) * U.L + first_w; // flat position of the lane's first window
		const uint64_t s0 = p ? p - 1u : 0u;           // the packed stream starts one base earlier (left neighbour)
		const uint32_t rel = live ? (uint32_t)(s0 - R.B0) : 0u;
		const uint32_t d = rel >> 4, sh = 2u * (rel & 15u);
		const uint32_t x0 = pk[d], x1 = pk[d + 1], x2 = pk[d + 2], x3 = pk[d + 3], x4 = pk[d + 4], x5 = pk[d + 5], x6 = pk[d + 6];
		const uint32_t X0 = funnel_left(x0, x1, sh), X1 = funnel_left(x1, x2, sh), X2 = funnel_left(x2, x3, sh), X3 = funnel_left(x3, x4, sh),
		               X4 = funnel_left(x4, x5, sh), X5 = funnel_left(x5, x6, sh);
		// stream Y starts at position p (X starts at p - 1 unless p == 0)
		const uint32_t adv = p ? 2u : 0u;
		const uint32_t Y0 = funnel_left(X0, X1, adv), Y1 = funnel_left(X1, X2, adv), Y2 = funnel_left(X2, X3, adv), Y3 = funnel_left(X3, X4, adv),
		               Y4 = funnel_left(X4, X5, adv);
		c.lw = p ? X0 : (X0 >> 2); // bases p-1 .. p+14 (the base before position 0 does not exist: has_l excludes it)
		const uint64_t A = ((uint64_t)Y0 << 32) | Y1, B = ((uint64_t)Y2 << 32) | Y3, C = (uint64_t)Y4 << 32;
		if (2u * k <= 64u) {
			c.fwd.hi = 0ull;
			c.fwd.lo = A >> (64u - 2u * k);
		} else {
			const uint32_t s2 = 128u - 2u * k; // 2..62
			c.fwd.hi = A >> s2;
			c.fwd.lo = (B >> s2) | (A << (64u - s2));
		}
		c.rc = dbgk_wide::revcomp(c.fwd, (int)k);
		{
			const uint32_t off = 2u * k; // 2..126
			const uint64_t top = (uint64_t)(((((u128)A << 64) | B) << off) >> 64);
			const uint64_t tail = off > 64u ? (C >> (128u - off)) : 0ull;
			c.nb = (uint32_t)((top | tail) >> 32);
		}
		const uint32_t nv = live ? min(8u, U.W - first_w) : 0u;
		const uint32_t nr = (live && first_w + 1u < U.W) ? min(8u, U.W - 1u - first_w) : 0u;
		c.valid = (1u << nv) - 1u;
		c.has_r = (1u << nr) - 1u;          // the read's last window has no right neighbour
		c.has_l = cc ? 0xFFu : 0xFEu;       // its first window no left one
		return c;
	};
	Range R = range_of(blockIdx.x);
	uint4 ra, rb2;
	fetch(R, ra, rb2);
	if constexpr (PIPE) {
		const uint32_t tid = fresh_tid();
		uint32_t total_prev = 0u; // records of the tile before, sorted in the stage buffer
		// element u of the tile before: staged record u * 1024 + t goes to its bucket; a full bucket's records are left for after the positions
		auto copy_elem = [&](uint32_t u, uint32_t &slow) {
			const uint32_t p = u * kWL1Threads + tid;
			if (p >= total_prev) return;
			const ull2 rcd = L.stage[p];
			const uint32_t b = L.bucket_of[p];
			const uint64_t off = (uint64_t)(uint32_t)(L.desc[b] + p); // desc = reserved place - first staged index
			if (off < G.cap1) P.l1[(uint64_t)b * G.cap1 + off] = rcd;
			else slow |= 1u << u;
		};
		auto copy_slow = [&](uint32_t slow) { // the bucket is full
			for (; slow; slow &= slow - 1u) {
				const uint32_t p = (uint32_t)__builtin_ctz(slow) * kWL1Threads + tid;
				const ull2 rcd = L.stage[p];
				const uint32_t b = L.bucket_of[p];
				wide_push_overflow(P, wide_record_key(rcd, wide_entry_bucket(b, G), G), (uint32_t)(rcd.y >> 3) & 7u, (uint32_t)rcd.y & 7u, ctr);
			}
		};
		if (blockIdx.x < n_tiles) {
			fill_pk(R, ra, rb2);
			L.hist[tid] = 0u; // kBpt == 1
			if (tid < 64u) L.hist[1024u + tid] = 0u;
			lds_barrier();
		}
		for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
			WChunk16 c = decode(R, tile * kWL1Threads);
			ull2 rec[8];
			uint32_t bkt[8];
			uint32_t rev_mask = 0, zero_lo = 0, slow = 0;
#pragma unroll
			for (uint32_t i = 0; i < 8u; i++) {
				copy_elem(i, slow);
				const WidePos o = wide_l1_position<WIDE_D>(G, c, K, i, tid);
				uint32_t rbit = o.rev ? 1u : 0u, zbit = o.zero_lo ? 1u : 0u;
				rev_mask |= rbit << i;
				zero_lo |= zbit << i;
				// (built HERE and kept: left to itself the compiler sinks the packing below the loop and keeps its inputs alive instead)
				uint32_t w0 = (uint32_t)o.w, w1 = (uint32_t)(o.w >> 32);
				asm volatile("" : "+v"(w0), "+v"(w1), "+v"(rev_mask), "+v"(zero_lo));
				rec[i] = ull2{o.key_hi, ((uint64_t)w1 << 32) | w0};
				if (i > 0u) asm volatile("" : "+v"(bkt[i - 1u]));
				bkt[i] = (o.b << 16) | atomicAdd(&L.hist[o.b], 1u);
			}
			if (slow) copy_slow(slow);
			// windows without a left / right neighbour: that side's code becomes 4 = none; keys with lo == 0 go through the atomic path
			// with their final codes (their record has no bucket)
			const uint32_t no_l = ~c.has_l & 0xFFu, no_r = ~c.has_r & 0xFFu, v8 = c.valid & 0xFFu;
			if (const uint32_t fix = ((no_l | no_r) & v8) | zero_lo) {
#pragma unroll
				for (uint32_t i = 0; i < 8u; i++) {
					if (!((fix >> i) & 1u)) continue;
					const bool fwd_strand = !((rev_mask >> i) & 1u), nl = (no_l >> i) & 1u, nr = (no_r >> i) & 1u;
					uint32_t lb = ((uint32_t)rec[i].y >> 3) & 7u, rbb = (uint32_t)rec[i].y & 7u;
					if (fwd_strand ? nl : nr) lb = 4u;
					if (fwd_strand ? nr : nl) rbb = 4u;
					rec[i].y = (rec[i].y & ~63ull) | (lb << 3) | rbb;
				}
				for (uint32_t z = zero_lo; z; z &= z - 1u) { // (rare: one inlined insert, the record picked by a chain of selects)
					const uint32_t i = (uint32_t)__builtin_ctz(z);
					ull2 r = rec[0];
#pragma unroll
					for (uint32_t j = 1; j < 8u; j++) r = (i == j) ? rec[j] : r;
					if (G.pass == 0u) wide_insert(T, Key128{r.x, 0ull}, ((uint32_t)r.y >> 3) & 7u, (uint32_t)r.y & 7u, ctr, n_new, n_conf, full); // (once per job: the input is read once per pass)
				}
			}
			// the next tile's blocks: requested behind the positions (the eight registers of an ASCII batch's two blocks are not there to
			// be had during them), in flight across the barrier, the scan and the staging
			const Range Rn = range_of(tile + gridDim.x);
			fetch(Rn, ra, rb2);
			lds_barrier(); // (C) every rank taken; the stage buffer, the tags and the descriptors of the tile before read out; the packed words decoded
			// thread b: reserve bucket b's run, scan, clear the histogram entry for the next tile
			const uint32_t c_t = L.hist[tid];
			L.hist[tid] = 0u;
			if (tid < 64u) L.hist[1024u + tid] = 0u; // (the bins of the positions without a record)
			const uint32_t g_t = (tid < G.n_l1 && c_t) ? atomicAdd(&P.cnt1[tid], c_t) : 0u;
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
			for (uint32_t w = 0; w < kWL1Threads / 64; w++) {
				const uint32_t wt = L.wave_tot[w];
				run += (w < (tid >> 6)) ? wt : 0u;
				all += wt;
			}
			L.lbase[tid] = run;
			lds_barrier(); // every bucket's first staged index is known; `all` = the tile's records with a bucket
#pragma unroll
			for (uint32_t i = 0; i < 8u; i++)
				if ((bkt[i] >> 16) < 1024u) {
					const uint32_t at = L.lbase[bkt[i] >> 16] + (bkt[i] & 0xFFFFu);
					L.stage[at] = rec[i];
					L.bucket_of[at] = (uint16_t)(bkt[i] >> 16);
				}
			L.desc[tid] = g_t - run;
			fill_pk(Rn, ra, rb2);
			lds_barrier(); // (E) the tile is staged, the next one's packed words are in place
			total_prev = all;
			R = Rn;
		}
		{ // the workgroup's last tile
			uint32_t slow = 0;
#pragma unroll 1
			for (uint32_t u = 0; u < 8u; u++) copy_elem(u, slow);
			if (slow) copy_slow(slow);
		}
		wide_l1_finish(n_new, n_conf, full, ctr);
		return;
	}
	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint32_t tid = fresh_tid();
		fill_pk(R, ra, rb2);
		const Range Rc = R;
		R = range_of(tile + gridDim.x);
		fetch(R, ra, rb2); // in flight during this tile
		L.hist[tid] = 0u;
		if (tid < 64u) L.hist[1024u + tid] = 0u;
		lds_barrier();
		WChunk16 c = decode(Rc, tile * kWL1Threads);
		wide_l1_tile<WIDE_D>(L, G, P, T, ctr, c, K, n_new, n_conf, full);
	}
	wide_l1_finish(n_new, n_conf, full, ctr);
}

// ---- level 2 --------------------------------------------------------------------------------------
// The inbox entries of the current pass flattened OWN-BUCKET-MAJOR, f = jj * n_ranks + s (records of rank s for my bucket
// pass_j0 + jj): a range of own buckets is a contiguous range of tiles.  tile_prefix[f] = tiles (kWL2Records records each) of
// the flat entries before f, tile_prefix[n_l1] = all.
__device__ __forceinline__ uint32_t wide_flat_entry(const WPartGeom &G, uint32_t f, uint32_t &jj)
{
	jj = f / G.n_ranks;
	return (f - jj * G.n_ranks) * G.Bp + jj;
}

__global__ __launch_bounds__(1024) void k_wide_l2_plan(WPartGeom G, const uint32_t *__restrict__ inbox_cnt, uint32_t *__restrict__ tile_prefix)
{
	__shared__ uint32_t wave_tot[16];
	const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
	uint64_t filled = 0;
	if (t < G.n_l1) {
		uint32_t jj;
		const uint32_t e = wide_flat_entry(G, t, jj);
		if (G.pass_j0 + jj < G.nb_own) filled = inbox_cnt[e] < G.cap1 ? inbox_cnt[e] : G.cap1; // (the last rank may own fewer buckets)
	}
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
	if (t < G.n_l1) tile_prefix[t] = run;
	if (t == G.n_l1 - 1u) tile_prefix[G.n_l1] = run + tiles;
}

template <int MAXB>
__global__ __launch_bounds__(kWL2Threads) void k_wide_scatter_l2(WPartGeom G, WPartStore P, const uint32_t *__restrict__ tile_prefix, Counters *__restrict__ ctr,
                                                                uint32_t j0, uint32_t j1) // the own buckets pass_j0 + [j0, j1) of this chunk
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	WL2Lds<MAXB> &L = *reinterpret_cast<WL2Lds<MAXB> *>(lds_raw);
	constexpr int kBpt = WL2Lds<MAXB>::kBpt;
	const uint32_t f0 = j0 * G.n_ranks, f1 = j1 * G.n_ranks; // flat entries of the chunk
	const uint32_t first_tile = tile_prefix[f0], span = tile_prefix[f1] - first_tile;
	// XCD-aware tile order (see k_scatter_l2): XCD x takes the x-th eighth of the tile range, so that the appends to a
	// final bucket come through ONE L2 and merge into full lines there
	const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3, n_local = gridDim.x >> 3; // gridDim.x is a multiple of 8
	const uint32_t lo_tile = first_tile + (uint32_t)(((uint64_t)span * xcd) >> 3), hi_tile = first_tile + (uint32_t)(((uint64_t)span * (xcd + 1u)) >> 3);
	for (uint32_t g = lo_tile + local; g < hi_tile; g += n_local) {
		uint32_t lo = f0, hi = f1; // last flat entry with tile_prefix[f] <= g (empty entries repeat the prefix: take the last)
		while (hi - lo > 1u) {
			const uint32_t mid = (lo + hi) >> 1;
			if (tile_prefix[mid] <= g) lo = mid; else hi = mid;
		}
		uint32_t jj;
		const uint32_t e = wide_flat_entry(G, lo, jj);
		const uint32_t b1 = G.b_lo + G.pass_j0 + jj; // global level-1 bucket
		const uint64_t filled = P.inbox_cnt[e] < G.cap1 ? P.inbox_cnt[e] : G.cap1;
		const uint64_t first = (uint64_t)(g - tile_prefix[lo]) * kWL2Records;
		const ull2 *in = P.inbox + (uint64_t)e * G.cap1;
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
		scatter_reserve_scan(L, G.n2, P.cnt2 + (uint64_t)(jj - j0) * G.n2, my_gbase);
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
			ull2 *out = P.l2 + (uint64_t)(jj - j0) * G.n2 * G.cap2;
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
	uint32_t redo;                 // FAST build: a counter of the current region passed 255 (or the region is full): the exact pass rebuilds it
};

// FAST (round 5, as k_build_regions<FAST> of the 64-bit engine): the two neighbour counters of a record are bumped with ONE
// ds_add_rtn_u64 instead of a saturating compare-swap loop.  A plain add cannot saturate, so the returned bytes are folded into a
// per-thread maximum; a region in which a byte that already held 255 was bumped (or whose image is full) emits NOTHING -- no slots,
// no spill nodes, no counts, no overflow observations -- and is appended to `redo`; the exact form of this kernel (FAST = false,
// FROM_LIST = true) rebuilds the listed regions from their records right after the chunk's fast launch (the level-2 store still
// holds them).  Exact for any input; a region pays twice only when one of its k-mers has a neighbour seen more than 255 times.
struct WRedoList {
	uint32_t *list;        // region indices inside the chunk
	unsigned int *n;       // appended so far
	uint32_t cap;
};

// Persistent workgroups pull regions from a cursor.  A slot is claimed with a CAS on `ident`; the winner then
// publishes hi + 1.  A record whose ident matches compares the high word too (two keys may share hash128): a lane
// that finds it unpublished looks again -- the winner's store follows its CAS in program order, so lanes of one
// wave never wait on each other, and another wave publishes without waiting for anybody.
// `table` holds this handle's slot range [slot_lo, slot_hi); the launch covers the final buckets of the own buckets
// pass_j0 + [j0, j0 + n_regions / n2) whose records level 2 has just put into the level-2 store.
template <bool FAST, bool FROM_LIST>
__global__ __launch_bounds__(kWBuildThreads) void k_wide_build_regions(WPartGeom G, WPartStore P, WNode *__restrict__ table, Counters *__restrict__ ctr,
                                                                     unsigned int *__restrict__ cursor, uint32_t j0, uint32_t n_regions, WRedoList redo)
{
	static_assert(!(FAST && FROM_LIST), "the exact pass is what the list is for");
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
	if (t == 0) L.redo = 0u;
	uint32_t n_new = 0, n_conf = 0;
	for (;;) {
		if (t == 0) {
			const unsigned int kx = atomicAdd(cursor, 1u);
			if (FROM_LIST) {
				const unsigned int n_list = *redo.n < redo.cap ? *redo.n : redo.cap; // complete: the fast launch has finished
				L.next_region = kx < n_list ? redo.list[kx] : kNone;
			} else {
				L.next_region = kx < n_regions ? kx : kNone;
			}
		}
		lds_barrier(); // also: the image is empty, the flag is clear
		const uint32_t fl = __builtin_amdgcn_readfirstlane(L.next_region); // final bucket inside the chunk's level-2 store
		if (fl == kNone) break;
		const uint32_t b1 = G.b_lo + G.pass_j0 + j0 + (fl >> (G.r - kWRegionBits)); // global level-1 bucket
		const uint64_t region_slot0 = ((uint64_t)b1 << G.r) + ((uint64_t)(fl & (G.n2 - 1u)) << kWRegionBits); // global slot of the region's first entry
		// (the table -- or this shard -- may end inside, or before, the last bucket's regions)
		const uint32_t region_len = region_slot0 >= G.slot_hi ? 0u : (uint32_t)((G.slot_hi - region_slot0 < (uint64_t)kWRegionSlots) ? G.slot_hi - region_slot0 : kWRegionSlots);
		const uint32_t filled = (uint32_t)(P.cnt2[fl] < G.cap2 ? P.cnt2[fl] : G.cap2);
		const ull2 *in = P.l2 + (uint64_t)fl * G.cap2;
		uint32_t n_new_r = 0, n_conf_r = 0, sat = 0; // of this region: committed only when the region is emitted
		bool flag = false;
		// two records per thread in flight: the pair of the NEXT round is requested before this round's records are probed (the
		// kernel's read rate is bytes in flight / latency: one 16-byte load per thread and round kept it at 2.8 TB/s)
		// (every lane loads, past the end the bucket's last record again: a load under a branch would make the compiler wait for ALL
		// loads in flight at the next use -- `live` decides what is used)
		const uint32_t last = filled ? filled - 1u : 0u;
		auto load_rec = [&](uint32_t i) { return __builtin_nontemporal_load(in + min(i, last)); };
		auto round = [&](const ull2 &rec, uint32_t base) { // one record per thread: records base .. base + kWBuildThreads - 1
			const uint32_t i = base + t;
			const bool live = i < filled;
			const unsigned long long id = (rec.y >> 6) + 1ull, want_hi1 = rec.x + 1ull;
			const uint32_t lb = (uint32_t)(rec.y >> 3) & 7u, rb = (uint32_t)rec.y & 7u;
			const uint32_t home = (uint32_t)(rec.y >> 6) & (kWRegionSlots - 1u);
			uint32_t idx = home;
			bool probing = live, lost = false;
			while (probing) {
				// one compare-swap per probe, whatever the slot holds: it claims an empty slot or tells whose it is (no load in front of it)
				const unsigned long long prev = atomicCAS(&L.ident[idx], 0ull, id);
				const bool mine = prev == 0ull;
				if (mine) {
					__hip_atomic_store(&L.hi1[idx], want_hi1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					n_new_r += idx < region_len ? 1u : 0u; // spilled nodes are counted when they are merged
				}
				const unsigned long long cur = mine ? id : prev;
				bool hit = mine, step = cur != id;
				if (cur == id && !mine) {
					const unsigned long long h1 = __hip_atomic_load(&L.hi1[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					hit = h1 == want_hi1;
					step = h1 != 0ull && !hit; // 0: its owner is about to publish the high word -- look at this slot again
				}
				idx += step ? 1u : 0u;
				n_conf_r += step ? 1u : 0u;
				lost = idx >= kAll; // region + spill area completely full
				probing = !hit && !lost;
			}
			// saturating +1 on the observed neighbour bytes, both dwords at once (see k_build_regions)
			const uint32_t dl = (lb != 4u) ? (1u << (24u - 8u * lb)) : 0u, dr = (rb != 4u) ? (1u << (24u - 8u * rb)) : 0u;
			bool pending = live && !lost;
			if constexpr (FAST) {
				if (pending) {
					const unsigned long long was = atomicAdd(&L.links[idx], ((unsigned long long)dr << 32) | dl);
					// the bumped byte moved to the top of a word: >= 0xFF000000 says it already held 255 (a side without neighbour shifts its word out)
					const uint32_t sh_l = 8u * lb, sh_r = 8u * rb;
					const uint32_t bl = (uint32_t)((uint64_t)(uint32_t)was << sh_l), br = (uint32_t)((uint64_t)(uint32_t)(was >> 32) << sh_r);
					sat = max(sat, max(bl, br));
				}
				flag = flag || (live && lost); // the exact pass sends it to the overflow list
				pending = false;
			}
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
			if (!FAST && live && lost) wide_push_overflow(P, wide_record_key(rec, b1, G), lb, rb, ctr);
		};
		// (two rounds per trip, each with a register pair of its own: nothing is copied, so nothing waits for the load behind it)
		ull2 ra = load_rec(t), rb2 = load_rec(kWBuildThreads + t);
		for (uint32_t base = 0; base < filled; base += 2u * kWBuildThreads) {
			round(ra, base);
			ra = load_rec(base + 2u * kWBuildThreads + t);
			if (base + kWBuildThreads < filled) round(rb2, base + kWBuildThreads); // (workgroup-uniform)
			rb2 = load_rec(base + 3u * kWBuildThreads + t);
		}
		bool redo_region = false;
		if constexpr (FAST) {
			if (flag || sat >= 0xFF000000u) L.redo = 1u;
			lds_barrier();
			redo_region = __builtin_amdgcn_readfirstlane(L.redo) != 0u;
		} else {
			lds_barrier();
		}
		if (!redo_region) {
			n_new += n_new_r;
			n_conf += n_conf_r;
		} else {
			if (t == 0) {
				const unsigned int j = atomicAdd(redo.n, 1u);
				if (j < redo.cap) redo.list[j] = fl; else atomicOr(&ctr->error, 2u);
			}
			for (uint32_t i = t; i < kAll; i += kWBuildThreads) { // nothing of this region is emitted: the exact pass writes it
				L.ident[i] = 0ull;
				L.hi1[i] = 0ull;
				L.links[i] = 0ull;
			}
			lds_barrier();            // (everybody has read the flag: that happened before its clearing loop, i.e. before this barrier)
			if (t == 0) L.redo = 0u;  // visible behind the barrier at the top of the loop
			continue;
		}
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
				ull2 *dst = reinterpret_cast<ull2 *>(&table[region_slot0 - G.slot_lo + i]);
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

// ---- shards: probing confined to the handle's slot range -------------------------------------------------------------
// wide_find_or_claim (dbgk_wide_kernels.h) on the shard-local array, without wrap-around: the local index, or ~0 when the
// probe reaches slot_hi (the node then continues at the first slot of the next shard)
__device__ __forceinline__ uint64_t wide_find_or_claim_range(WNode *nodes, uint64_t slot_lo, uint64_t slot_hi, Key128 key, uint64_t slot,
                                                             uint64_t &links_guess, unsigned long long &n_new, unsigned long long &n_conf)
{
	uint64_t spins = 0;
	while (slot < slot_hi) {
		WNode *nd = &nodes[slot - slot_lo];
		unsigned long long w0 = wload(&nd->lo);
		if (w0 == 0ull) {
			const unsigned long long prev = atomicCAS(&nd->lo, 0ull, (unsigned long long)key.lo);
			if (prev == 0ull) {
				wstore(&nd->hi1, key.hi + 1ull);
				n_new++;
				links_guess = 0ull;
				return slot - slot_lo;
			}
			w0 = prev;
		}
		if (w0 == key.lo) {
			const unsigned long long h1 = wload(&nd->hi1);
			if (h1 == 0ull && ++spins < (1ull << 24)) continue; // its owner is about to publish the high word: look again
			if (h1 == key.hi + 1ull) {
				links_guess = wload(&nd->links);
				return slot - slot_lo;
			}
		}
		n_conf++;
		slot++;
	}
	return ~0ull;
}

// Merge nodes (is_obs == 0: {hi, lo, l_link, r_link}) or single observations (is_obs == 1: {hi, lo, lb, rb}) into this shard.
// An entry whose home slot lies in another shard is skipped unless start_foreign_at_lo (nodes handed over by the previous
// shard continue their probe at this shard's first slot).  Nodes whose probe runs off the end of the shard are appended to
// P.outgoing for the next rank.  A key-0 node is folded into the handle's side node; keys with a zero low word never come
// this way (they are no records: side table, gathered onto rank 0 separately).
__global__ __launch_bounds__(kBlock) void k_wide_merge_sharded(const dbgk_node32 *__restrict__ in, const unsigned long long *__restrict__ n_ptr, uint64_t n_direct,
                                                               uint64_t cap, int is_obs, int start_foreign_at_lo, WPartGeom G, WPartStore P,
                                                               WNode *__restrict__ table, Counters *__restrict__ ctr)
{
	__shared__ unsigned long long red[kBlock / 64];
	unsigned long long n_new = 0, n_conf = 0;
	uint64_t n = n_ptr ? (uint64_t)*n_ptr : n_direct;
	if (n > cap) n = cap;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
		const dbgk_node32 nd = in[i];
		const Key128 key{nd.kmer_hi, nd.kmer_lo};
		const uint64_t add = is_obs ? dbgk_wide::observe(0ull, nd.l_link, nd.r_link) : ((uint64_t)nd.l_link | ((uint64_t)nd.r_link << 32));
		if (dbgk_wide::is_zero(key)) {
			links_cas_merge(&ctr->polyA_links, 0ull, add);
			continue;
		}
		if (key.lo == 0ull) continue;
		const uint64_t home = fast_mod(dbgk_wide::hash128(key), G.magic);
		const bool mine = home >= G.slot_lo && home < G.slot_hi;
		if (!mine && !start_foreign_at_lo) continue;
		uint64_t guess;
		const uint64_t idx = wide_find_or_claim_range(table, G.slot_lo, G.slot_hi, key, mine ? home : G.slot_lo, guess, n_new, n_conf);
		if (idx == ~0ull) {
			const unsigned long long j = atomicAdd(P.outgoing_n, 1ull);
			if (j < P.outgoing_cap) {
				dbgk_node32 o = nd;
				o.l_link = (uint32_t)add;
				o.r_link = (uint32_t)(add >> 32);
				P.outgoing[j] = o;
			} else {
				atomicOr(&ctr->error, 2u);
			}
			continue;
		}
		links_cas_merge(&table[idx].links, guess, add);
	}
	const unsigned long long a = block_sum(n_new, red);
	const unsigned long long b = block_sum(n_conf, red);
	if (threadIdx.x == 0) {
		if (a) atomicAdd(&ctr->n_new, a);
		if (b) atomicAdd(&ctr->n_conflict, b);
	}
}

// the side table (keys with a zero low word) and the key-0 links as a dense list of host-layout nodes: out[0] = the key-0 node
// (always), then the occupied side slots; *n_out = entries written
__global__ __launch_bounds__(kBlock) void k_wide_side_export(const WNode *__restrict__ side, const Counters *__restrict__ ctr, dbgk_node32 *__restrict__ out,
                                                             unsigned long long *__restrict__ n_out, uint64_t cap)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		const unsigned long long links = ctr->polyA_links;
		out[0] = dbgk_node32{0ull, 0ull, (uint32_t)links, (uint32_t)(links >> 32), 0ull};
	}
	for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < kWideSideSlots; i += gridDim.x * kBlock) {
		const WNode nd = side[i];
		if (nd.hi1 == 0ull) continue;
		const unsigned long long j = 1ull + atomicAdd(n_out, 1ull);
		if (j < cap) out[j] = dbgk_node32{nd.hi1 - 1ull, 0ull, (uint32_t)nd.links, (uint32_t)(nd.links >> 32), 0ull};
	}
}

} // namespace dbgk
