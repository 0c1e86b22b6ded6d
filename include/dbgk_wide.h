/*
 * dbgk_wide.h -- definition of the WIDE key path: k-mers of 33..63 bases (128-bit keys), BASELINE cfg5.
 *
 * The reference stops at k = 31 (`uint64_t kmer`, DBG_contig/kmerSet.h:71; "max 31", main.cpp:100), so nothing
 * in it says what a 63-mer graph node looks like: PARITY UNPINNED for k > 32.  This header is the definition
 * this build adopts -- the reference's rules carried over to 128 bits -- written once, host/device neutral, and
 * used by BOTH the HIP kernels (dbg_assembly_amd/csrc/dbgk_wide_kernels.h) and the CPU restatement
 * (oracle/wide_oracle.cpp).  For k <= 32 the high word of every key is 0 and every rule below reduces to the
 * reference's 64-bit one, so the wide path instantiated at k <= 31 must reproduce the pinned oracle bit for bit
 * (tests/test_wide.py) -- that is what anchors it.
 *
 *   key        2 bits per base, first base in the most significant used bits (seq2bit, seqKmer.cpp:34-41),
 *              as a 128-bit number {hi, lo}
 *   canonical  min(forward, reverse complement) as 128-bit numbers, tie -> forward (DBGgraph.cpp:80)
 *   neighbours left/right base codes exactly as DBGgraph.cpp:82-97 (complemented and swapped on the reverse
 *              strand, 4 = none)
 *   node       32 bytes: {kmer_hi, kmer_lo, l_link, r_link, reserved = 0}; link words as in KmerNode
 *              (four saturating bytes, A in bits 31..24, kmerSet.cpp:56)
 *   hash       hash_code(lo) when hi == 0 (the reference's slot for every k <= 32), else
 *              hash_code(lo ^ hash_code(hi)); slot = hash % table size, linear probing (DBGgraph.cpp:167-205)
 *   key 0      (poly-A / poly-T) kept aside and appended last (DBGgraph.cpp:153-164, :418)
 *   digest     sum over nodes of mix64((lo ^ (hi ? mix64(hi) : 0)) ^ mix64(l_link << 32 | r_link)): equals
 *              dbgk_digest's value whenever every hi is 0
 */
#ifndef DBGK_WIDE_H_
#define DBGK_WIDE_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define DBGK_WIDE_HD __host__ __device__ inline
#else
#define DBGK_WIDE_HD inline
#endif

#ifdef __cplusplus
extern "C" {
#endif
typedef struct dbgk_node32 {
	uint64_t kmer_hi;
	uint64_t kmer_lo;
	uint32_t l_link;
	uint32_t r_link;
	uint64_t reserved;
} dbgk_node32;
#ifdef __cplusplus
}

namespace dbgk_wide {

struct Key128 {
	uint64_t hi, lo;
};

/* alphabet[] of DBG_contig/seqKmer.cpp:9-19: A a N n -> 0, C c -> 1, G g -> 2, T t -> 3 (other bytes are outside
 * the input contract) */
DBGK_WIDE_HD uint32_t base_code(unsigned char c)
{
	switch (c) {
		case 'C': case 'c': return 1u;
		case 'G': case 'g': return 2u;
		case 'T': case 't': return 3u;
		default: return 0u;
	}
}

/* hash_code, DBG_contig/kmerSet.h:105-116 */
DBGK_WIDE_HD uint64_t hash_code64(uint64_t k)
{
	k += ~(k << 32);
	k ^= (k >> 22);
	k += ~(k << 13);
	k ^= (k >> 8);
	k += (k << 3);
	k ^= (k >> 15);
	k += ~(k << 27);
	k ^= (k >> 31);
	return k;
}

DBGK_WIDE_HD uint64_t hash128(Key128 k) { return hash_code64(k.hi ? (k.lo ^ hash_code64(k.hi)) : k.lo); }

/* complement every base of a 64-bit word of 32 bases and reverse their order (get_rev_com_kbit's core,
 * seqKmer.cpp:89-97, without the final shift) */
DBGK_WIDE_HD uint64_t rc_word(uint64_t x)
{
	x = ~x;
	x = ((x & 0x3333333333333333ULL) << 2) | ((x & 0xCCCCCCCCCCCCCCCCULL) >> 2);
	x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((x & 0xF0F0F0F0F0F0F0F0ULL) >> 4);
	x = ((x & 0x00FF00FF00FF00FFULL) << 8) | ((x & 0xFF00FF00FF00FF00ULL) >> 8);
	x = ((x & 0x0000FFFF0000FFFFULL) << 16) | ((x & 0xFFFF0000FFFF0000ULL) >> 16);
	return (x << 32) | (x >> 32);
}

/* reverse complement of a k-base key, 1 <= k <= 64 */
DBGK_WIDE_HD Key128 revcomp(Key128 x, int k)
{
	Key128 r;
	r.hi = rc_word(x.lo); /* as a 64-base word: the halves swap */
	r.lo = rc_word(x.hi);
	const int sh = 128 - 2 * k; /* bring the k bases down to the least significant end */
	Key128 o;
	if (sh == 0) return r;
	if (sh >= 64) {
		o.hi = 0;
		o.lo = r.hi >> (sh - 64);
	} else {
		o.hi = r.hi >> sh;
		o.lo = (r.lo >> sh) | (r.hi << (64 - sh));
	}
	return o;
}

DBGK_WIDE_HD bool less_equal(Key128 a, Key128 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo <= b.lo); }
DBGK_WIDE_HD bool is_zero(Key128 a) { return (a.hi | a.lo) == 0; }

struct Observation {
	Key128 key;
	uint32_t lb, rb; /* 0..3, 4 = none */
};

/* canonical pick + neighbour bases, DBGgraph.cpp:76-97.  left / right: codes of the bases before / after the
 * window, 4 where the read (after trimming to maxReadLen) has none */
DBGK_WIDE_HD Observation canonical(Key128 fwd, int k, uint32_t left, uint32_t right)
{
	const Key128 rc = revcomp(fwd, k);
	Observation o;
	if (less_equal(fwd, rc)) {
		o.key = fwd;
		o.lb = left;
		o.rb = right;
	} else {
		o.key = rc;
		o.rb = left == 4u ? 4u : 3u - left;
		o.lb = right == 4u ? 4u : 3u - right;
	}
	return o;
}

/* one observation on a node's two link words held as l_link | r_link << 32: +1 on the counter of lb in l_link
 * and of rb in r_link, each byte saturating at 255 (DBGgraph.cpp:188-194; BitAddVal, kmerSet.cpp:56) */
DBGK_WIDE_HD uint64_t observe(uint64_t links, uint32_t lb, uint32_t rb)
{
	if (lb != 4u) {
		const uint32_t sh = (3u - lb) * 8u;
		if (((links >> sh) & 0xFFu) != 0xFFu) links += 1ULL << sh;
	}
	if (rb != 4u) {
		const uint32_t sh = 32u + (3u - rb) * 8u;
		if (((links >> sh) & 0xFFu) != 0xFFu) links += 1ULL << sh;
	}
	return links;
}

DBGK_WIDE_HD uint64_t mix64(uint64_t x) /* splitmix64 finaliser (digest only) */
{
	uint64_t z = x + 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

/* links_lr as stored: l_link low dword, r_link high dword */
DBGK_WIDE_HD uint64_t node_digest(Key128 key, uint64_t links_lr)
{
	const uint64_t v = (links_lr << 32) | (links_lr >> 32);
	return mix64((key.lo ^ (key.hi ? mix64(key.hi) : 0ULL)) ^ mix64(v));
}

} /* namespace dbgk_wide */
#endif /* __cplusplus */
#endif /* DBGK_WIDE_H_ */
