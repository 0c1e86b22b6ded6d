// dbgk_device.h -- device-side building blocks shared by all kernels (gfx950 only).
//
// Reference semantics (paths relative to /root/reference/):
//   alphabet / seq2bit           DBG_contig/seqKmer.cpp:9-19,34-41
//   get_rev_com_kbit             DBG_contig/seqKmer.cpp:89-97
//   hash_code                    DBG_contig/kmerSet.h:105-116
//   BitAddVal / link layout      DBG_contig/kmerSet.cpp:56, :341-344
//   canonical pick + neighbours  DBG_contig/DBGgraph.cpp:76-89
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dbgk {

// ---- 64-bit modulo / divide by the run-time table size ------------------------------------
// slot = hash_code(key) % size with size a run-time "prime" (DBGgraph.cpp:167).  A hardware 64-bit
// divide is ~40 VALU ops with a long dependent chain; with m = floor(2^64/d) the quotient estimate
// mulhi(h, m) is at most 1 too small (proof in DESIGN.md), so one multiply-high, one multiply-low
// and a conditional subtract give exact q and r.
struct ModMagic {
	uint64_t d;
	uint64_t m;
};

static inline ModMagic make_mod_magic(uint64_t d)
{
	ModMagic g;
	g.d = d;
	g.m = (uint64_t)((((unsigned __int128)1) << 64) / d); // d >= 2
	return g;
}

__device__ __forceinline__ uint64_t fast_divmod(uint64_t h, const ModMagic g, uint64_t &q_out)
{
	uint64_t q = __umul64hi(h, g.m);
	uint64_t r = h - q * g.d;
	if (r >= g.d) { r -= g.d; q++; }
	if (r >= g.d) { r -= g.d; q++; } // never taken; keeps the result exact even if the bound were off by one
	q_out = q;
	return r;
}

__device__ __forceinline__ uint64_t fast_mod(uint64_t h, const ModMagic g)
{
	uint64_t q;
	return fast_divmod(h, g, q);
}

// ---- exact 64-bit / 32-bit division by a run-time invariant divisor ----------------------------
// For tables below 2^32 slots (always true for the PARTITION engine) the divisor fits one dword and
// hash / size, hash % size are two steps of the Moller-Granlund "2-by-1" division with a precomputed
// reciprocal (Improved division by invariant integers, IEEE TC 2011, Alg. 4): one 32x32->64
// multiply and one 32-bit multiply per step instead of the ~11 quarter-rate multiplies of the
// generic 64-bit path.
struct Div32Magic {
	uint32_t dn;   // divisor << s (normalised: top bit set)
	uint32_t v;    // floor((2^64 - 1) / dn) - 2^32
	uint32_t s;    // clz(divisor)
	uint32_t d;    // the divisor
};

static inline Div32Magic make_div32_magic(uint32_t d)
{
	Div32Magic g;
	g.d = d;
	g.s = (uint32_t)__builtin_clz(d);
	g.dn = d << g.s;
	g.v = (uint32_t)((~0ull / g.dn) - (1ull << 32));
	return g;
}

// (u1:u0) / dn with u1 < dn, dn normalised
__device__ __forceinline__ uint32_t div_2by1(uint32_t u1, uint32_t u0, const Div32Magic g, uint32_t &rem)
{
	uint64_t p = (uint64_t)g.v * u1;
	p += ((uint64_t)u1 << 32) | u0;
	uint32_t q1 = (uint32_t)(p >> 32) + 1u;
	const uint32_t q0 = (uint32_t)p;
	uint32_t r = u0 - q1 * g.dn;
	if (r > q0) { q1--; r += g.dn; }
	if (r >= g.dn) { q1++; r -= g.dn; }
	rem = r;
	return q1;
}

// h = q * d + r, d < 2^32; returns r, q in q_out (q < 2^64 / d)
__device__ __forceinline__ uint32_t divmod_u64_u32(uint64_t h, const Div32Magic g, uint64_t &q_out)
{
	const uint32_t a = (uint32_t)(h >> 32), b = (uint32_t)h;
	uint32_t u2, u1, u0;
	if (g.s) { // wave-uniform
		u2 = a >> (32u - g.s);
		u1 = (a << g.s) | (b >> (32u - g.s));
		u0 = b << g.s;
	} else {
		u2 = 0u; u1 = a; u0 = b;
	}
	uint32_t r1, r2;
	const uint32_t qh = div_2by1(u2, u1, g, r1);
	const uint32_t ql = div_2by1(r1, u0, g, r2);
	q_out = ((uint64_t)qh << 32) | ql;
	return r2 >> g.s;
}

// h = q * d + r for d < 2^31 with m = floor(2^64 / d) (ModMagic): mulhi(h, m) is q or q - 1, so the
// estimate's remainder is < 2d < 2^32 and 32-bit wrap-around arithmetic gives it exactly.  Integer
// multiplies issue at full rate on gfx950 (profiles/ubench/alu_rate.hip), which makes this ~13
// instructions against ~35 for the two 2-by-1 steps above.
__device__ __forceinline__ uint32_t divmod_magic_small(uint64_t h, uint64_t m, uint32_t d, uint64_t &q_out)
{
	const uint32_t h0 = (uint32_t)h, h1 = (uint32_t)(h >> 32), m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
	const uint64_t t1 = (uint64_t)h1 * m0 + __umulhi(h0, m0);
	const uint64_t t2 = (uint64_t)h0 * m1 + (uint32_t)t1;
	uint64_t q = (uint64_t)h1 * m1 + (t1 >> 32) + (t2 >> 32);
	uint32_t r = h0 - (uint32_t)q * d;
	const bool fix = r >= d;
	r = fix ? r - d : r;
	q += fix ? 1u : 0u;
	q_out = q;
	return r;
}

// ---- hash_code (kmerSet.h:105-116) and its inverse ------------------------------------------
__host__ __device__ __forceinline__ uint64_t hash_code(uint64_t k)
{
	k += ~(k << 32);
	k ^= (k >> 22);
	k += ~(k << 13);
	k ^= (k >> 8);
#if defined(__HIP_DEVICE_COMPILE__)
	asm("v_lshl_add_u64 %0, %1, 3, %1" : "=v"(k) : "v"(k)); // k += k << 3 in one instruction (the compiler emits a 64-bit multiply by 9)
#else
	k += (k << 3);
#endif
	k ^= (k >> 15);
	k += ~(k << 27);
	k ^= (k >> 31);
	return k;
}

// splitmix64 finaliser (digest only; not part of the reference)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x)
{
	uint64_t z = x + 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

__host__ __device__ __forceinline__ uint64_t node_digest(uint64_t kmer, uint64_t links_lr)
{
	// links_lr as stored in memory (l_link low word, r_link high word); digest is defined on
	// (l_link << 32) | r_link
	uint64_t v = (links_lr << 32) | (links_lr >> 32);
	return mix64(kmer ^ mix64(v));
}

// ---- 2-bit codec ----------------------------------------------------------------------------
// A,a,N,n -> 0; C,c -> 1; G,g -> 2; T,t -> 3 (seqKmer.cpp:9-19).  For those ten letters the code is
// ((c >> 1) ^ (c >> 2)) & 3.  EVERY OTHER BYTE (IUPAC codes, '-', '*', bytes >= 128; undefined behaviour in the
// reference, whose alphabet[] maps them to 4 and then indexes KmerRCOrVal[4] out of bounds, DBGgraph.cpp:71-73)
// IS READ AS 'A', like N, and counted (dbgk_stats.other_bytes) -- in every engine, the host packer
// (dbgk_pack_bases), the oracle and the checkers.  The kernels get there in two steps: the formula packs, a
// five-instruction detector says whether any of the four bytes is outside the ten letters, and only then (rare)
// the word is repacked byte by byte.
// 4 ASCII bases (one little-endian dword, first base in the low byte) -> 8 bits, first base in bits 7..6.
__device__ __forceinline__ uint32_t pack4_formula(uint32_t w)
{
	const uint32_t x = ((w >> 1) ^ (w >> 2)) & 0x03030303u;
	return (x * 0x40100401u) >> 24;                       // gather the four 2-bit codes MSB-first
}

// non-zero iff some byte of w is none of ACGTNacgtn: bits 3..1 of a letter -- A 0, C 1, T 2, G 3, N 7 -- select the
// upper-case letter they would have to belong to (v_perm_b32 as an 8-entry byte table, 0xFF where no letter lives);
// the byte with bit 5 cleared either is that letter or it is not one of the ten
__device__ __forceinline__ uint32_t other4_ascii(uint32_t w)
{
	const uint32_t expect = __builtin_amdgcn_perm(0x4EFFFFFFu, 0x47544341u, (w >> 1) & 0x07070707u);
	return (w & 0xDFDFDFDFu) ^ expect;
}

// the exact form: a byte outside the ten letters becomes 'A' (code 0)
__device__ __forceinline__ uint32_t pack4_exact(uint32_t w)
{
	uint32_t out = 0;
#pragma unroll 1
	for (uint32_t i = 0; i < 4; i++) {
		const uint32_t c = (w >> (8u * i)) & 0xFFu, u = c & 0xDFu;
		const bool letter = u == 0x41u || u == 0x43u || u == 0x47u || u == 0x54u || u == 0x4Eu;
		const uint32_t code = letter ? ((c >> 1) ^ (c >> 2)) & 3u : 0u;
		out |= code << (6u - 2u * i);
	}
	return out;
}

// `other_seen` (Counters::other_seen, through ReadBatch) is raised when a byte outside the ten letters is met; the bytes
// are then COUNTED by k_count_other_bytes, once per base (tiles overlap, so the extraction kernels cannot count)
__device__ __forceinline__ uint32_t pack4_ascii(uint32_t w, unsigned int *other_seen)
{
	uint32_t p = pack4_formula(w);
	if (__builtin_expect(other4_ascii(w) != 0u, 0)) {
		p = pack4_exact(w);
		atomicOr(other_seen, 1u);
	}
	return p;
}

// one ASCII base -> its code
__device__ __forceinline__ uint32_t code_ascii(uint32_t c, unsigned int *other_seen)
{
	return pack4_ascii((c & 0xFFu) | 0x41414100u, other_seen) >> 6;
}

// 16 ASCII bases -> one dword of 2-bit codes, base 0 in bits 31..30
__device__ __forceinline__ uint32_t pack16_ascii(const uint4 v, unsigned int *other_seen)
{
	const uint32_t bad = other4_ascii(v.x) | other4_ascii(v.y) | other4_ascii(v.z) | other4_ascii(v.w);
	uint32_t p = (pack4_formula(v.x) << 24) | (pack4_formula(v.y) << 16) | (pack4_formula(v.z) << 8) | pack4_formula(v.w);
	if (__builtin_expect(bad != 0u, 0)) {
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
		p = 0u;
#pragma unroll 1
		for (uint32_t j = 0; j < 4; j++) p = (p << 8) | pack4_exact(w[j]);
		atomicOr(other_seen, 1u);
	}
	return p;
}

// reverse complement of a 2-bit packed k-mer (seqKmer.cpp:89-97): complement = ~, reverse the 32
// groups with one v_bfrev per half plus a swap of the two bits inside each group
__device__ __forceinline__ uint64_t revcomp_kbit(uint64_t kbit, int k)
{
	uint64_t x = __brevll(~kbit);
	x = ((x & 0x5555555555555555ULL) << 1) | ((x >> 1) & 0x5555555555555555ULL);
	return x >> (64 - 2 * k);
}

// ---- link words -------------------------------------------------------------------------------
// A node's two link words are handled as ONE little-endian 64-bit word at byte offset 8 of the
// node: l_link = low dword, r_link = high dword.  Counter of base b (0..3) sits at bits
// (3-b)*8 of its dword (kmerSet.cpp:56, :341-344).

// one observation: +1 on the left counter of base lb and the right counter of base rb (4 = none),
// each saturating at 255 (DBGgraph.cpp:188-194)
__device__ __forceinline__ uint64_t links_observe(uint64_t links, uint32_t lb, uint32_t rb)
{
	if (lb != 4u) {
		uint32_t sh = (3u - lb) * 8u;
		if (((links >> sh) & 0xFFu) != 0xFFu) links += 1ULL << sh;
	}
	if (rb != 4u) {
		uint32_t sh = 32u + (3u - rb) * 8u;
		if (((links >> sh) & 0xFFu) != 0xFFu) links += 1ULL << sh;
	}
	return links;
}

// per-byte saturating add of eight packed counters: min(255, a+b) in every byte
__host__ __device__ __forceinline__ uint64_t links_sat_add(uint64_t a, uint64_t b)
{
	const uint64_t H = 0x8080808080808080ULL;
	uint64_t lo = (a & ~H) + (b & ~H);          // 7-bit sums, carry lands in bit 7 of each byte
	uint64_t s = lo ^ ((a ^ b) & H);            // wrapped byte sums
	uint64_t carry = ((a & b) | ((a | b) & lo)) & H;
	return s | ((carry >> 7) * 0xFFULL);
}

// ---- the graph node ---------------------------------------------------------------------------
struct alignas(16) Node {
	uint64_t kmer;
	uint64_t links; // l_link | r_link << 32
};

} // namespace dbgk
