/*
 * dbgk_synth.h -- deterministic synthetic short-read generator.
 *
 * Not part of the reference: DBG_assembly ships no read simulator (test/00.raw_reads/work.sh:1-6
 * calls the external tool `pirs`).  This is the bench/test workload generator.  It is COUNTER
 * BASED: every base of every read is a pure function of (seed, read index, position), so the
 * host (plain C), the HIP kernel and any rank of a multi-GPU job produce bit-identical reads
 * without materialising the genome or sharing RNG state.
 *
 *   genome  : G bases, i.i.d. uniform over ACGT        base(p)  = 2 bits of mix(genome_seed, p/32)
 *   read i  : start  = floor(u * (G-L+1)), u = mix(read_seed, 2i)   (uniform start)
 *             strand = mix(read_seed, 2i+1) & 1                     (1 = reverse complement)
 *   base j  : substitution with probability sub_thr / 2^32 (uniform over the 3 other bases),
 *             then 'N' with probability n_thr / 2^24, both drawn from mix(err_seed, i*1024+j)
 *
 * Usable from C11, C++ and HIP device code (define DBGK_HD before including for __host__ __device__).
 */
#ifndef DBGK_SYNTH_H_
#define DBGK_SYNTH_H_

#include <stdint.h>

#ifndef DBGK_HD
#define DBGK_HD
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dbgk_synth_params {
	uint64_t genome_len;   /* G, must be >= read_len                                   */
	uint32_t read_len;     /* L (<= 1024)                                               */
	uint32_t sub_thr;      /* substitution probability * 2^32  (0.5 % -> 21474836)      */
	uint32_t n_thr;        /* 'N' probability * 2^24           (0.01 % -> 1678)         */
	uint32_t reserved;
	uint64_t genome_seed;
	uint64_t read_seed;
	uint64_t err_seed;
} dbgk_synth_params;

/* splitmix64 finaliser */
static inline DBGK_HD uint64_t dbgk_mix64(uint64_t x)
{
	uint64_t z = x + 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

static inline DBGK_HD uint64_t dbgk_mulhi64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __umul64hi(a, b);
#else
	return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

/* 2-bit code (A=0,C=1,G=2,T=3) of genome position p */
static inline DBGK_HD uint32_t dbgk_synth_genome_base(const dbgk_synth_params *P, uint64_t p)
{
	uint64_t w = dbgk_mix64(P->genome_seed ^ dbgk_mix64(p >> 5));
	return (uint32_t)(w >> ((p & 31u) * 2u)) & 3u;
}

/* start position and strand (1 = reverse complement) of read i */
static inline DBGK_HD void dbgk_synth_read_origin(const dbgk_synth_params *P, uint64_t i,
                                                  uint64_t *start, uint32_t *strand)
{
	uint64_t u = dbgk_mix64(P->read_seed ^ dbgk_mix64(2u * i));
	*start = dbgk_mulhi64(u, P->genome_len - (uint64_t)P->read_len + 1u);
	*strand = (uint32_t)dbgk_mix64(P->read_seed ^ dbgk_mix64(2u * i + 1u)) & 1u;
}

/* ASCII character of base j of read i, given the read's origin */
static inline DBGK_HD char dbgk_synth_read_base_at(const dbgk_synth_params *P, uint64_t i, uint32_t j,
                                                   uint64_t start, uint32_t strand)
{
	const uint64_t L = P->read_len;
	uint32_t b = strand ? 3u - dbgk_synth_genome_base(P, start + (L - 1u - j))
	                    : dbgk_synth_genome_base(P, start + j);
	uint64_t e = dbgk_mix64(P->err_seed ^ dbgk_mix64(i * 1024u + j));
	if ((uint32_t)e < P->sub_thr) {
		b = (b + 1u + ((uint32_t)((e >> 32) & 0xFFu) % 3u)) & 3u;
	}
	if ((uint32_t)(e >> 40) < P->n_thr) return 'N';
	return (char)(0x54474341u >> (8u * b)); /* "ACGT"[b] without a memory table */
}

/* ASCII character of base j of read i */
static inline DBGK_HD char dbgk_synth_read_base(const dbgk_synth_params *P, uint64_t i, uint32_t j)
{
	uint64_t start;
	uint32_t strand;
	dbgk_synth_read_origin(P, i, &start, &strand);
	return dbgk_synth_read_base_at(P, i, j, start, strand);
}

/* host helper: fill `out` (n_reads * read_len bytes, no separators) with reads first..first+n_reads-1 */
static inline void dbgk_synth_fill_host(const dbgk_synth_params *P, uint64_t first, uint64_t n_reads, char *out)
{
	for (uint64_t i = 0; i < n_reads; i++) {
		uint64_t start;
		uint32_t strand;
		dbgk_synth_read_origin(P, first + i, &start, &strand);
		for (uint32_t j = 0; j < P->read_len; j++) {
			out[i * P->read_len + j] = dbgk_synth_read_base_at(P, first + i, j, start, strand);
		}
	}
}

#ifdef __cplusplus
}
#endif
#endif /* DBGK_SYNTH_H_ */
