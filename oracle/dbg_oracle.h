/*
 * dbg_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, CPU restatement of the reference's k-mer / de Bruijn graph hot path
 * (fanagislab/DBG_assembly, DBG_contig/{seqKmer,kmerSet,DBGgraph}.cpp).  It exists to CHECK the
 * HIP implementation; nothing in the product (dbg_assembly_amd/, include/dbgk.h consumers) may
 * include, link or call it.  Allowed users: tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg.
 *
 * PARITY PINNING: every function below is validated in tests/test_oracle_vs_golden.py against
 * golden vectors produced by the REAL reference compiled from /root/reference (oracle/_ref,
 * recipe oracle/Makefile, generator tests/golden/make_golden.py), and -- where /root/reference is
 * present -- directly against oracle/_ref/ref_dbg on fresh random inputs
 * (tests/test_oracle_vs_ref_live.py).  The reference's own repository holds no numeric fixture
 * for this path.
 *
 * All citations are relative to /root/reference/.
 */
#ifndef DBG_ORACLE_H_
#define DBG_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- seqKmer codec (DBG_contig/seqKmer.cpp) ---------------------------------------------- */
extern const signed char orc_alphabet[128];                  /* seqKmer.cpp:9-19   */
uint64_t orc_count_other_bytes(const char *seq, uint64_t n); /* bytes that are none of ACGTNacgtn: read as 'A' (this build's rule, dbg_oracle.c) */
uint64_t orc_seq2bit(const char *seq, int n);                 /* seqKmer.cpp:34-41  */
void     orc_bit2seq(uint64_t kbit, int k, char *out);        /* seqKmer.cpp:45-52 (out: k+1 bytes) */
uint64_t orc_rev_com_kbit(uint64_t kbit, int k);              /* seqKmer.cpp:89-97  */
uint64_t orc_pow_integer(int base, int exponent);             /* seqKmer.cpp:130-136 */

/* ---- kmerSet (DBG_contig/kmerSet.{h,cpp}) -------------------------------------------------- */
typedef struct orc_node {                                     /* KmerNode, kmerSet.h:70-75 */
	uint64_t kmer;
	uint32_t l_link;
	uint32_t r_link;
} orc_node;

typedef struct orc_kmerset {                                  /* KmerSet, kmerSet.h:88-99 */
	uint32_t e_size;
	uint64_t size;
	uint64_t count;
	uint64_t count_conflict;
	uint64_t max;
	float    load_factor;
	uint64_t iter_ptr;
	orc_node *array;
	uint8_t  *nul_flag;
	uint8_t  *del_flag;
} orc_kmerset;

uint64_t orc_hash_code(uint64_t kmer);                        /* kmerSet.h:105-116 */
int      orc_is_prime(uint64_t num);                          /* kmerSet.cpp:72-81 */
uint64_t orc_find_next_prime(uint64_t num);                   /* kmerSet.cpp:85-95 */
uint8_t  orc_get_next_kmer_depth(uint32_t link, uint8_t base);/* kmerSet.cpp:341-344 */
int      orc_is_entity_null(const uint8_t *flag, uint64_t idx);   /* kmerSet.h:144-147 */
int      orc_is_entity_delete(const uint8_t *flag, uint64_t idx); /* kmerSet.h:159-162 */

orc_kmerset *orc_kmerset_init(uint64_t init_size, float load_factor);     /* kmerSet.cpp:98-127 */
void     orc_kmerset_enlarge(orc_kmerset *set, uint64_t num);              /* kmerSet.cpp:132-189 */
int      orc_kmerset_add_node(orc_kmerset *set, const orc_node *e);        /* kmerSet.cpp:253-273 */
uint64_t orc_kmerset_exist(const orc_kmerset *set, uint64_t kmer);         /* kmerSet.cpp:280-302 */
void     orc_kmerset_free(orc_kmerset *set);                               /* kmerSet.cpp:62-68 */

/* canonical dump: all non-null slots sorted by kmer; `out` must hold set->count nodes; returns n */
uint64_t orc_kmerset_dump_sorted(const orc_kmerset *set, orc_node *out);
/* order-independent 64-bit digest of a node multiset (sum of per-node mixes, wrapping) */
uint64_t orc_nodes_digest(const orc_node *nodes, uint64_t n);
/* checks invariants (2)-(5) of SURVEY.md section 8(b) on any host-layout table; 0 = ok, else a
 * negative code: -1 unreachable key, -2 non-zero unused slot, -3 count mismatch, -4 duplicate key */
int      orc_check_host_table(const orc_node *array, const uint8_t *nul_flag, uint64_t size,
                              uint64_t expect_count);

/* ---- phase A: one read -> (kmer, left, right) triples (DBGgraph.cpp:38-120) --------------- */
/* returns the number of triples written (0 if len < k); arrays need max_read_len-k+1 entries   */
int orc_parse_read(const char *seq, int len, int k, int max_read_len,
                   uint64_t *kmers, uint8_t *left, uint8_t *right);

/* ---- the graph builder (DBGgraph.cpp:126-430) ------------------------------------------------ */
typedef struct orc_graph orc_graph;

typedef struct orc_graph_params {      /* extern globals of DBGgraph.h:25-36, defaults DBGgraph.cpp:10-21 */
	int      kmer_size;                /* KmerSize           (31)   */
	int      max_read_len;             /* maxReadLen         (250)  */
	int      thread_num;               /* threadNum          (10)   */
	double   init_hash_size;           /* initHashSize, in units of 1e9 slots (1.0) */
	uint64_t max_double_hash_times;    /* maxDoubleHashTimes (10)   */
	float    hash_load_factor;         /* hashLoadFactor     (0.7)  */
	int      buffer_num;               /* BufferNum          (10000) */
} orc_graph_params;

void       orc_graph_default_params(orc_graph_params *p);
orc_graph *orc_graph_create(const orc_graph_params *p);        /* build_debruijn_graph :364-403 */
/* one "file" worth of reads held in memory: bases = concatenated sequences, offsets[n+1].
 * Runs the reference's block loop (blocks of BufferNum reads, enlarge check after every FULL
 * block, last short block ends the file) -- parse_one_reads_file :217-359.  Returns 0, or 1 if
 * the reference's "Memory reach the maximum allowed" branch (:346-350) dropped the rest.        */
int        orc_graph_add_file_mem(orc_graph *g, const char *bases, const uint64_t *offsets, uint64_t n_reads);
/* same, reading a (optionally gzip'ed) one-line FASTQ (format 1) / FASTA (format 2) file        */
int        orc_graph_add_file(orc_graph *g, const char *path, int format);
void       orc_graph_finish(orc_graph *g);                     /* add PolyA node :418 */
orc_kmerset *orc_graph_kmerset(orc_graph *g);
uint64_t   orc_graph_total_reads(const orc_graph *g);           /* Total_reads_num */
uint64_t   orc_graph_total_kmers(const orc_graph *g);           /* Kmer_total_num  */
uint64_t   orc_graph_double_times(const orc_graph *g);          /* doubleHashTimes */
void       orc_graph_destroy(orc_graph *g);

/* ---- consumer first pass (DBG_contig/contig.cpp:107-181) ------------------------------------ */
typedef struct orc_link_stats {
	int64_t depth_stat[256];           /* DepthStat, contig.cpp:112-115,130,151 */
	int64_t total_nodes;               /* total_kmer_speceis_num  */
	int64_t deleted_lowfreq;           /* deleted_lowFreq_kmer_num */
	int64_t linear_nodes;              /* linear_kmer_node_num     */
	int64_t tip_nodes;                 /* tip_nodes.size()         */
	int64_t branch_nodes;              /* branch_nodes.size()      */
} orc_link_stats;
void orc_calc_link_stats(const orc_node *nodes, uint64_t n, int kmer_freq_cutoff, orc_link_stats *out);

/* one-line-FASTA/FASTQ record splitter used by orc_graph_add_file, exposed for tests:
 * appends sequences to bases[] / offsets[] (malloc-ed, caller frees); returns number of records */
int64_t orc_read_sequences(const char *path, int format, char **bases, uint64_t **offsets);

#ifdef __cplusplus
}
#endif
#endif /* DBG_ORACLE_H_ */
