/*
 * dbg_oracle.c -- TEST INFRASTRUCTURE ONLY (see dbg_oracle.h for the rules and the pinning).
 *
 * Plain-C CPU restatement of DBG_contig's k-mer codec, hash set and graph builder.  Written from
 * the behaviour of the reference, not from its text; every routine names the reference lines it
 * follows (paths relative to /root/reference/).
 */
#define _GNU_SOURCE
#include "dbg_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------------------------------- */
/* codec                                                                                       */
/* ------------------------------------------------------------------------------------------- */

/* DBG_contig/seqKmer.cpp:9-19 -- A,a,N,n -> 0; C,c -> 1; G,g -> 2; T,t -> 3; everything else 4 */
const signed char orc_alphabet[128] = {
	4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
	4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
	/* @ A B C D E F G H I J K L M N O */
	4,0,4,1,4,4,4,2,4,4,4,4,4,4,0,4,
	/* P Q R S T ... */
	4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
	4,0,4,1,4,4,4,2,4,4,4,4,4,4,0,4,
	4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
};

/* The reference indexes alphabet[] with a plain char (seqKmer.cpp:38, DBGgraph.cpp:71): bytes >= 128 index it out of
 * bounds, and every byte that maps to 4 (anything but ACGTNacgtn) then indexes KmerRCOrVal[4] out of bounds (DBGgraph.cpp:73) --
 * undefined behaviour, so there is nothing to restate.  THIS BUILD'S RULE, applied by every GPU engine, the host packer
 * (dbgk_pack_bases) and this oracle alike: such a byte is read as 'A' (as N already is, seqKmer.cpp:15,17) and counted
 * (orc_count_other_bytes here, dbgk_stats.other_bytes there).  For ACGTNacgtn nothing changes: the goldens made by the real
 * reference still pin every line below. */
static inline int code_of(char c)
{
	const unsigned char u = (unsigned char)c;
	const int v = u < 128 ? orc_alphabet[u] : 4;
	return v == 4 ? 0 : v;
}

uint64_t orc_count_other_bytes(const char *seq, uint64_t n)
{
	uint64_t other = 0;
	for (uint64_t i = 0; i < n; i++) {
		const unsigned char u = (unsigned char)seq[i];
		other += (u >= 128 || orc_alphabet[u] == 4) ? 1u : 0u;
	}
	return other;
}

uint64_t orc_seq2bit(const char *seq, int n)
{
	/* seqKmer.cpp:34-41: MSB-first packing, 2 bits per base */
	uint64_t v = 0;
	for (int i = 0; i < n; i++) v = (v << 2) | (uint64_t)code_of(seq[i]);
	return v;
}

void orc_bit2seq(uint64_t kbit, int k, char *out)
{
	/* seqKmer.cpp:45-52 */
	static const char letters[4] = {'A', 'C', 'G', 'T'};
	for (int i = 0; i < k; i++) out[i] = letters[(kbit >> ((k - 1 - i) * 2)) & 3u];
	out[k] = 0;
}

uint64_t orc_rev_com_kbit(uint64_t kbit, int k)
{
	/* seqKmer.cpp:89-97: complement = bitwise not; reverse the 32 two-bit groups by a
	 * butterfly of swaps (2,4,8,16,32 bits); drop the 64-2k low garbage bits. */
	uint64_t x = ~kbit;
	static const uint64_t m[5] = {0x3333333333333333ULL, 0x0F0F0F0F0F0F0F0FULL, 0x00FF00FF00FF00FFULL,
	                              0x0000FFFF0000FFFFULL, 0x00000000FFFFFFFFULL};
	for (int s = 0; s < 5; s++) {
		int sh = 2 << s;
		x = ((x & m[s]) << sh) | ((x & ~m[s]) >> sh);
	}
	return x >> (64 - 2 * k);
}

uint64_t orc_pow_integer(int base, int exponent)
{
	/* seqKmer.cpp:130-136: wrapping u64 product; pow_integer(2,64) == 0 */
	uint64_t r = 1;
	for (int i = 0; i < exponent; i++) r *= (uint64_t)(int64_t)base;
	return r;
}

/* ------------------------------------------------------------------------------------------- */
/* hash set                                                                                    */
/* ------------------------------------------------------------------------------------------- */

uint64_t orc_hash_code(uint64_t k)
{
	/* kmerSet.h:105-116 (Thomas Wang / "Jenkins" 64-bit integer mix) */
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

int orc_is_prime(uint64_t num)
{
	/* kmerSet.cpp:72-81.  The bound is a FLOAT sqrt truncated to integer and the loop uses a
	 * strict '<', so squares of primes (9, 25, 49, ...) and some other composites pass.  In the
	 * reference `sqrt((float)num)` resolves to std::sqrt(float) -> sqrtf. */
	if (num < 4) return 1;
	if ((num & 1u) == 0) return 0;
	uint64_t bound = (uint64_t)sqrtf((float)num);
	for (uint64_t i = 3; i < bound; i += 2) {
		if (num % i == 0) return 0;
	}
	return 1;
}

uint64_t orc_find_next_prime(uint64_t num)
{
	/* kmerSet.cpp:85-95 */
	if ((num & 1u) == 0) num++;
	while (!orc_is_prime(num)) num += 2;
	return num;
}

uint8_t orc_get_next_kmer_depth(uint32_t link, uint8_t base)
{
	/* kmerSet.cpp:341-344: A is the most significant byte */
	return (uint8_t)((link >> ((3 - base) * 8)) & 0xFFu);
}

/* flag bit i lives in byte i/8 under mask 128>>(i%8) (kmerSet.cpp:53, kmerSet.h:144-169) */
static inline uint8_t flag_mask(uint64_t idx) { return (uint8_t)(0x80u >> (idx & 7u)); }
static inline int flag_get(const uint8_t *f, uint64_t idx) { return (f[idx >> 3] & flag_mask(idx)) != 0; }
static inline void flag_set(uint8_t *f, uint64_t idx) { f[idx >> 3] |= flag_mask(idx); }

int orc_is_entity_null(const uint8_t *flag, uint64_t idx) { return !flag_get(flag, idx); }
int orc_is_entity_delete(const uint8_t *flag, uint64_t idx) { return flag_get(flag, idx); }

static const uint32_t link_add[4] = {0x1000000u, 0x10000u, 0x100u, 0x1u}; /* BitAddVal kmerSet.cpp:56 */

/* max = (uint64_t)(size * load_factor): the product is evaluated in FLOAT (kmerSet.cpp:114,145) */
static uint64_t max_cutoff(uint64_t size, float lf) { return (uint64_t)((float)size * lf); }

orc_kmerset *orc_kmerset_init(uint64_t init_size, float load_factor)
{
	/* kmerSet.cpp:98-127 */
	orc_kmerset *s = (orc_kmerset *)calloc(1, sizeof(*s));
	if (!s) return NULL;
	init_size = (init_size < 3) ? 3 : orc_find_next_prime(init_size);
	if (load_factor <= 0) load_factor = 0.25f;
	else if (load_factor >= 1) load_factor = 0.75f;
	s->e_size = (uint32_t)sizeof(orc_node);
	s->size = init_size;
	s->load_factor = load_factor;
	s->max = max_cutoff(s->size, load_factor);
	s->array = (orc_node *)calloc(s->size, sizeof(orc_node));
	s->nul_flag = (uint8_t *)calloc(s->size / 8 + 1, 1);
	s->del_flag = (uint8_t *)calloc(s->size / 8 + 1, 1);
	if (!s->array || !s->nul_flag || !s->del_flag) { orc_kmerset_free(s); return NULL; }
	return s;
}

void orc_kmerset_free(orc_kmerset *s)
{
	if (!s) return;
	free(s->array);
	free(s->nul_flag);
	free(s->del_flag);
	free(s);
}

void orc_kmerset_enlarge(orc_kmerset *s, uint64_t num)
{
	/* kmerSet.cpp:132-189.  New size: next "prime" after doubling, repeated until
	 * new_size*load_factor (float) >= count+num (compared in float).  The table is grown in
	 * place and every live old entry is re-seated with a displacement chain: an entry lifted from
	 * slot i is dropped on the first free slot of its new probe sequence; if an as-yet-unmoved old
	 * entry sits there it is lifted in turn.  Slot order of the scan (0..old_size-1) matters for
	 * the resulting layout, so it is kept. */
	const uint64_t old_size = s->size;
	uint64_t new_size = s->size;
	do {
		new_size = orc_find_next_prime(new_size * 2);
	} while ((float)new_size * s->load_factor < (float)(s->count + num));

	s->size = new_size;
	s->array = (orc_node *)realloc(s->array, new_size * sizeof(orc_node));
	memset(s->array + old_size, 0, (new_size - old_size) * sizeof(orc_node));
	s->max = max_cutoff(new_size, s->load_factor);

	uint8_t *old_nul = s->nul_flag;
	uint8_t *old_del = s->del_flag; /* doubles as "already lifted" marker during the pass */
	s->nul_flag = (uint8_t *)calloc(new_size / 8 + 1, 1);
	s->del_flag = (uint8_t *)calloc(new_size / 8 + 1, 1);

	for (uint64_t i = 0; i < old_size; i++) {
		if (!flag_get(old_nul, i) || flag_get(old_del, i)) continue;
		orc_node carry = s->array[i];
		memset(&s->array[i], 0, sizeof(orc_node));
		flag_set(old_del, i);
		for (;;) {
			uint64_t hc = orc_hash_code(carry.kmer) % new_size;
			while (flag_get(s->nul_flag, hc)) hc = (hc + 1) % new_size;
			flag_set(s->nul_flag, hc);
			if (hc < old_size && flag_get(old_nul, hc) && !flag_get(old_del, hc)) {
				orc_node lifted = s->array[hc];
				s->array[hc] = carry;
				carry = lifted;
				flag_set(old_del, hc);
			} else {
				s->array[hc] = carry;
				break;
			}
		}
	}
	free(old_nul);
	free(old_del);
}

int orc_kmerset_add_node(orc_kmerset *s, const orc_node *e)
{
	/* kmerSet.cpp:253-273: first slot with a clear null-flag on the key's probe sequence */
	uint64_t hc = orc_hash_code(e->kmer) % s->size;
	for (;;) {
		if (!flag_get(s->nul_flag, hc)) {
			s->array[hc] = *e;
			flag_set(s->nul_flag, hc);
			s->count++;
			return 1;
		}
		s->count_conflict++;
		hc = (hc + 1 == s->size) ? 0 : hc + 1;
	}
}

uint64_t orc_kmerset_exist(const orc_kmerset *s, uint64_t kmer)
{
	/* kmerSet.cpp:280-302 */
	uint64_t hc = orc_hash_code(kmer) % s->size;
	for (;;) {
		if (!flag_get(s->nul_flag, hc)) return s->size;
		if (s->array[hc].kmer == kmer) return flag_get(s->del_flag, hc) ? s->size : hc;
		hc = (hc + 1 == s->size) ? 0 : hc + 1;
	}
}

static int node_cmp(const void *a, const void *b)
{
	uint64_t x = ((const orc_node *)a)->kmer, y = ((const orc_node *)b)->kmer;
	return (x > y) - (x < y);
}

uint64_t orc_kmerset_dump_sorted(const orc_kmerset *s, orc_node *out)
{
	uint64_t n = 0;
	for (uint64_t i = 0; i < s->size; i++) {
		if (flag_get(s->nul_flag, i)) out[n++] = s->array[i];
	}
	qsort(out, n, sizeof(orc_node), node_cmp);
	return n;
}

static inline uint64_t mix64(uint64_t x)
{
	uint64_t z = x + 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

uint64_t orc_nodes_digest(const orc_node *nodes, uint64_t n)
{
	uint64_t acc = 0;
	for (uint64_t i = 0; i < n; i++) {
		uint64_t links = ((uint64_t)nodes[i].l_link << 32) | nodes[i].r_link;
		acc += mix64(nodes[i].kmer ^ mix64(links));
	}
	return acc;
}

int orc_check_host_table(const orc_node *array, const uint8_t *nul_flag, uint64_t size, uint64_t expect_count)
{
	uint64_t n = 0;
	for (uint64_t i = 0; i < size; i++) {
		if (!flag_get(nul_flag, i)) {
			if (array[i].kmer | array[i].l_link | array[i].r_link) return -2;
			continue;
		}
		n++;
		/* walk from the home slot: must reach i without meeting a null slot, and must not meet
		 * the same key earlier (duplicate) */
		uint64_t hc = orc_hash_code(array[i].kmer) % size;
		while (hc != i) {
			if (!flag_get(nul_flag, hc)) return -1;
			if (array[hc].kmer == array[i].kmer) return -4;
			hc = (hc + 1 == size) ? 0 : hc + 1;
		}
	}
	return (n == expect_count) ? 0 : -3;
}

/* ------------------------------------------------------------------------------------------- */
/* phase A                                                                                     */
/* ------------------------------------------------------------------------------------------- */

int orc_parse_read(const char *seq, int len, int k, int max_read_len,
                   uint64_t *kmers, uint8_t *left, uint8_t *right)
{
	/* DBGgraph.cpp:51-98.  readlen = min(len, maxReadLen) (:63); first window by seq2bit +
	 * full reverse complement (:66-69), later windows by the rolling update (:71-73) with
	 * KmerHeadMaskVal = 2^(2k)-1 and KmerRCOrVal[b] = (3-b) << (2k-2) (:371-376).
	 * Canonical pick and neighbour bases :80-89 (tie -> forward). */
	if (len < k) return 0;
	const int readlen = len > max_read_len ? max_read_len : len;
	const uint64_t head_mask = orc_pow_integer(2, 2 * k) - 1;
	uint64_t fwd = 0, rev = 0;
	int n = 0;
	for (int j = 0; j + k <= readlen; j++) {
		if (j == 0) {
			fwd = orc_seq2bit(seq, k);
			rev = orc_rev_com_kbit(fwd, k);
		} else {
			uint64_t b = (uint64_t)code_of(seq[j + k - 1]);
			fwd = ((fwd << 2) | b) & head_mask;
			rev = (rev >> 2) | ((3 - b) << (2 * k - 2));
		}
		const int has_l = j > 0, has_r = j < readlen - k;
		uint8_t lb = 4, rb = 4;
		uint64_t key;
		if (fwd <= rev) {
			key = fwd;
			if (has_l) lb = (uint8_t)code_of(seq[j - 1]);
			if (has_r) rb = (uint8_t)code_of(seq[j + k]);
		} else {
			key = rev;
			if (has_l) rb = (uint8_t)(3 - code_of(seq[j - 1]));
			if (has_r) lb = (uint8_t)(3 - code_of(seq[j + k]));
		}
		kmers[n] = key;
		left[n] = lb;
		right[n] = rb;
		n++;
	}
	return n;
}

/* ------------------------------------------------------------------------------------------- */
/* graph builder                                                                               */
/* ------------------------------------------------------------------------------------------- */

struct orc_graph {
	orc_graph_params p;
	orc_kmerset *set;
	orc_node poly_a;               /* PolyA side node, DBGgraph.cpp:399-402 */
	uint64_t total_reads;
	uint64_t total_kmers;
	uint64_t double_times;
	int kmer_num_in_read;          /* KmerNumInRead :389 */
	/* block staging (StoreKmer / StoreLeftBase / StoreRightBase :394-396) */
	uint64_t *store_kmer;
	uint8_t *store_left, *store_right;
	const char **blk_seq;          /* RawReads of the current block */
	int *blk_len;
	int blk_n;
};

void orc_graph_default_params(orc_graph_params *p)
{
	/* DBGgraph.cpp:10-21 */
	p->kmer_size = 31;
	p->max_read_len = 250;
	p->thread_num = 10;
	p->init_hash_size = 1.0;
	p->max_double_hash_times = 10;
	p->hash_load_factor = 0.7f;
	p->buffer_num = 10000;
}

orc_graph *orc_graph_create(const orc_graph_params *p)
{
	orc_graph *g = (orc_graph *)calloc(1, sizeof(*g));
	if (!g) return NULL;
	g->p = *p;
	if (g->p.thread_num < 1) g->p.thread_num = 1;
	g->set = orc_kmerset_init((uint64_t)(p->init_hash_size * 1000000000), p->hash_load_factor); /* :381 */
	g->kmer_num_in_read = p->max_read_len - p->kmer_size + 1;
	size_t slots = (size_t)p->buffer_num * (size_t)(g->kmer_num_in_read > 0 ? g->kmer_num_in_read : 1);
	g->store_kmer = (uint64_t *)malloc(slots * sizeof(uint64_t));
	g->store_left = (uint8_t *)malloc(slots);
	g->store_right = (uint8_t *)malloc(slots);
	g->blk_seq = (const char **)malloc((size_t)p->buffer_num * sizeof(char *));
	g->blk_len = (int *)malloc((size_t)p->buffer_num * sizeof(int));
	if (!g->set || !g->store_kmer || !g->store_left || !g->store_right || !g->blk_seq || !g->blk_len) {
		orc_graph_destroy(g);
		return NULL;
	}
	return g;
}

void orc_graph_destroy(orc_graph *g)
{
	if (!g) return;
	orc_kmerset_free(g->set);
	free(g->store_kmer);
	free(g->store_left);
	free(g->store_right);
	free((void *)g->blk_seq);
	free(g->blk_len);
	free(g);
}

orc_kmerset *orc_graph_kmerset(orc_graph *g) { return g->set; }
uint64_t orc_graph_total_reads(const orc_graph *g) { return g->total_reads; }
uint64_t orc_graph_total_kmers(const orc_graph *g) { return g->total_kmers; }
uint64_t orc_graph_double_times(const orc_graph *g) { return g->double_times; }

/* saturating per-byte add of one neighbour observation (DBGgraph.cpp:188-194 / :155-161) */
static inline void link_bump(uint32_t *link, uint8_t base)
{
	if (base != 4 && orc_get_next_kmer_depth(*link, base) < 255) *link += link_add[base];
}

/* one stored triple into the table: DBGgraph.cpp:153-205 */
static void insert_triple(orc_graph *g, uint64_t kmer, uint8_t lb, uint8_t rb, int threaded)
{
	if (kmer == 0) { /* :153-164 poly-A / poly-T goes to the side node */
		link_bump(&g->poly_a.l_link, lb);
		link_bump(&g->poly_a.r_link, rb);
		return;
	}
	orc_kmerset *s = g->set;
	uint64_t hc = orc_hash_code(kmer) % s->size;
	for (;;) {
		orc_node *e = &s->array[hc];
		int claimed = 0;
		if (e->kmer == 0) {
			if (threaded) {
				uint64_t expect = 0;
				claimed = __atomic_compare_exchange_n(&e->kmer, &expect, kmer, 0, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
			} else {
				e->kmer = kmer;
				claimed = 1;
			}
		}
		if (claimed) { /* :174-182 first sight SETS the links */
			e->l_link = (lb != 4) ? link_add[lb] : 0;
			e->r_link = (rb != 4) ? link_add[rb] : 0;
			if (threaded) {
				__atomic_fetch_or(&s->nul_flag[hc >> 3], flag_mask(hc), __ATOMIC_SEQ_CST);
				__atomic_fetch_add(&s->count, 1, __ATOMIC_SEQ_CST);
			} else {
				flag_set(s->nul_flag, hc);
				s->count++;
			}
			return;
		}
		if (e->kmer == kmer) { /* :185-196 */
			link_bump(&e->l_link, lb);
			link_bump(&e->r_link, rb);
			return;
		}
		if (threaded) __atomic_fetch_add(&s->count_conflict, 1, __ATOMIC_SEQ_CST);
		else s->count_conflict++;
		hc = (hc + 1 == s->size) ? 0 : hc + 1;
	}
}

typedef struct worker_arg {
	orc_graph *g;
	int tid;
	uint64_t kmers_added; /* phase A: sum of (len-k+1) for this thread's reads */
} worker_arg;

/* thread_parseBlock, DBGgraph.cpp:38-120: thread t takes reads t, t+T, ... */
static void *phase_a_worker(void *vp)
{
	worker_arg *a = (worker_arg *)vp;
	orc_graph *g = a->g;
	const int T = g->p.thread_num, k = g->p.kmer_size;
	for (int i = a->tid; i < g->blk_n; i += T) {
		int len = g->blk_len[i];
		if (len < k) continue;
		size_t base = (size_t)i * (size_t)g->kmer_num_in_read;
		orc_parse_read(g->blk_seq[i], len, k, g->p.max_read_len,
		               g->store_kmer + base, g->store_left + base, g->store_right + base);
		a->kmers_added += (uint64_t)(len - k + 1); /* :101 untrimmed length */
	}
	return NULL;
}

/* thread_updatekmers, DBGgraph.cpp:126-213: every thread scans the whole block and keeps the
 * triples whose kmer % T is its id */
static void *phase_b_worker(void *vp)
{
	worker_arg *a = (worker_arg *)vp;
	orc_graph *g = a->g;
	const int T = g->p.thread_num, k = g->p.kmer_size;
	for (int i = 0; i < g->blk_n; i++) {
		int len = g->blk_len[i];
		int readlen = len > g->p.max_read_len ? g->p.max_read_len : len;
		size_t base = (size_t)i * (size_t)g->kmer_num_in_read;
		for (int j = 0; j < readlen - k + 1; j++) {
			uint64_t kmer = g->store_kmer[base + j];
			if (T > 1 && kmer % (uint64_t)T != (uint64_t)a->tid) continue;
			insert_triple(g, kmer, g->store_left[base + j], g->store_right[base + j], T > 1);
		}
	}
	return NULL;
}

static void run_phase(orc_graph *g, void *(*fn)(void *), uint64_t *sum_out)
{
	const int T = g->p.thread_num;
	worker_arg *args = (worker_arg *)calloc((size_t)T, sizeof(worker_arg));
	if (T == 1) {
		args[0].g = g;
		fn(&args[0]);
	} else {
		/* the reference creates and joins T threads per phase per block (:233-239,:306-318) */
		pthread_t *th = (pthread_t *)malloc((size_t)T * sizeof(pthread_t));
		for (int t = 0; t < T; t++) {
			args[t].g = g;
			args[t].tid = t;
			pthread_create(&th[t], NULL, fn, &args[t]);
		}
		for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
		free(th);
	}
	if (sum_out) {
		for (int t = 0; t < T; t++) *sum_out += args[t].kmers_added;
	}
	free(args);
}

/* process the block currently staged in blk_seq/blk_len; returns 1 when the file must be
 * abandoned (:346-350), 0 otherwise.  `full` = block holds BufferNum reads. */
static int process_block(orc_graph *g, int full)
{
	g->total_reads += (uint64_t)g->blk_n;                      /* :274 */
	run_phase(g, phase_a_worker, &g->total_kmers);
	run_phase(g, phase_b_worker, NULL);
	if (!full) return 0;                                        /* :329-331 */
	if (g->set->count > g->set->max) {                          /* :337 */
		if (g->double_times < g->p.max_double_hash_times) {
			orc_kmerset_enlarge(g->set, 1);                     /* :341 */
			g->double_times++;
		} else {
			return 1;
		}
	}
	return 0;
}

int orc_graph_add_file_mem(orc_graph *g, const char *bases, const uint64_t *offsets, uint64_t n_reads)
{
	/* parse_one_reads_file, DBGgraph.cpp:226-353, with the reads already split.  The reference
	 * ends a file on the first block shorter than BufferNum -- which is an EMPTY block when the
	 * file holds an exact multiple of BufferNum reads, so the last full block still gets its
	 * enlarge check. */
	const uint64_t B = (uint64_t)g->p.buffer_num;
	uint64_t next = 0;
	for (;;) {
		uint64_t n = n_reads - next < B ? n_reads - next : B;
		for (uint64_t i = 0; i < n; i++) {
			g->blk_seq[i] = bases + offsets[next + i];
			g->blk_len[i] = (int)(offsets[next + i + 1] - offsets[next + i]);
		}
		g->blk_n = (int)n;
		next += n;
		int full = (n == B);
		if (process_block(g, full)) return 1;
		if (!full) return 0;
	}
}

/* growable line reader over zlib (transparent for plain files) */
typedef struct line_reader {
	gzFile fp;
	char *buf;
	size_t cap;
} line_reader;

/* returns 1 and the line (without '\n') in lr->buf / *len, or 0 at EOF with nothing read */
static int next_line(line_reader *lr, size_t *len)
{
	size_t n = 0;
	for (;;) {
		if (lr->cap - n < 2) {
			lr->cap = lr->cap ? lr->cap * 2 : 4096;
			lr->buf = (char *)realloc(lr->buf, lr->cap);
		}
		if (!gzgets(lr->fp, lr->buf + n, (int)(lr->cap - n))) {
			if (n == 0) { lr->buf[0] = 0; *len = 0; return 0; }
			break;
		}
		n += strlen(lr->buf + n);
		if (n && lr->buf[n - 1] == '\n') { n--; break; }
	}
	lr->buf[n] = 0;
	*len = n;
	return 1;
}

int64_t orc_read_sequences(const char *path, int format, char **bases_out, uint64_t **offsets_out)
{
	/* record detection of DBGgraph.cpp:244-272: a line whose first character is '@' (format 1)
	 * or '>' (otherwise) starts a record, the NEXT line is the sequence; format 1 then skips two
	 * more lines.  Other lines are ignored.  A header at EOF yields an empty sequence. */
	line_reader lr = {gzopen(path, "rb"), NULL, 0};
	if (!lr.fp) return -1;
	const char marker = (format == 1) ? '@' : '>';
	size_t bcap = 1 << 20, blen = 0, ocap = 1 << 12;
	char *bases = (char *)malloc(bcap);
	uint64_t *offs = (uint64_t *)malloc(ocap * sizeof(uint64_t));
	int64_t n = 0;
	size_t len;
	offs[0] = 0;
	while (next_line(&lr, &len)) {
		if (lr.buf[0] != marker) continue;
		next_line(&lr, &len); /* sequence (empty at EOF) */
		if (blen + len + 1 > bcap) {
			while (blen + len + 1 > bcap) bcap *= 2;
			bases = (char *)realloc(bases, bcap);
		}
		memcpy(bases + blen, lr.buf, len);
		blen += len;
		if ((size_t)n + 2 > ocap) {
			ocap *= 2;
			offs = (uint64_t *)realloc(offs, ocap * sizeof(uint64_t));
		}
		offs[++n] = blen;
		if (format == 1) {
			size_t dummy;
			next_line(&lr, &dummy);
			next_line(&lr, &dummy);
		}
	}
	gzclose(lr.fp);
	free(lr.buf);
	*bases_out = bases;
	*offsets_out = offs;
	return n;
}

int orc_graph_add_file(orc_graph *g, const char *path, int format)
{
	char *bases = NULL;
	uint64_t *offs = NULL;
	int64_t n = orc_read_sequences(path, format, &bases, &offs);
	if (n < 0) return 0; /* reference: open failure is silent, zero reads (gzstream) */
	int rc = orc_graph_add_file_mem(g, bases, offs, (uint64_t)n);
	free(bases);
	free(offs);
	return rc;
}

void orc_graph_finish(orc_graph *g)
{
	/* DBGgraph.cpp:418: the key-0 node is always appended, even with both links zero */
	orc_kmerset_add_node(g->set, &g->poly_a);
}

/* ------------------------------------------------------------------------------------------- */
/* consumer first pass                                                                         */
/* ------------------------------------------------------------------------------------------- */

void orc_calc_link_stats(const orc_node *nodes, uint64_t n, int cutoff, orc_link_stats *out)
{
	/* contig.cpp:119-181 over the node list (order-independent quantities only): per node, the 4
	 * left and 4 right byte counters go into DepthStat; a side's link number is the count of
	 * counters > KmerFreqCutoff, capped at 3 (2-bit field, contig.h:32-34). */
	memset(out, 0, sizeof(*out));
	for (uint64_t i = 0; i < n; i++) {
		int ln = 0, rn = 0;
		for (uint8_t b = 0; b < 4; b++) {
			int d = orc_get_next_kmer_depth(nodes[i].l_link, b);
			out->depth_stat[d]++;
			if (d > cutoff && ln < 3) ln++;
		}
		for (uint8_t b = 0; b < 4; b++) {
			int d = orc_get_next_kmer_depth(nodes[i].r_link, b);
			out->depth_stat[d]++;
			if (d > cutoff && rn < 3) rn++;
		}
		out->total_nodes++;
		if (ln == 0 && rn == 0) out->deleted_lowfreq++;
		if (ln == 1 && rn == 1) out->linear_nodes++;
		if (ln + rn == 1) out->tip_nodes++;
		if (ln > 1 || rn > 1) out->branch_nodes++;
	}
}
