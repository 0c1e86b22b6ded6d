/*
 * dbgk.h -- C ABI of the MI355X (gfx950) k-mer / de Bruijn graph construction layer.
 *
 * This is the drop-in boundary underneath DBG_assembly's build_debruijn_graph()
 * (reference: DBG_contig/DBGgraph.h:66, DBG_contig/DBGgraph.cpp:364-430).  The reference has no
 * FFI of its own (single C++ process); the entry points below are what
 * its hot path decomposes into when the compute moves to the GPU.  Each one names the reference
 * routine it replaces.  All functions are extern "C", take plain pointers and sizes, return an
 * int status (0 = DBGK_OK, < 0 = error, never hang on a full table) and may be called from one
 * host thread per handle.  One handle = one GPU = one HIP stream.
 *
 * There is NO CPU fallback behind this interface: every entry point fails with DBGK_ERR_HIP when
 * no gfx950 device is usable.
 *
 * Paths are relative to /root/reference/.
 */
#ifndef DBGK_H_
#define DBGK_H_

#include <stddef.h>
#include <stdint.h>
#include "dbgk_synth.h"
#include "dbgk_wide.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DBGK_ABI_VERSION 6

/* status codes */
#define DBGK_OK               0
#define DBGK_ERR_ARG         -1   /* bad argument / unsupported parameter (k < 1 or k > 32, ...)    */
#define DBGK_ERR_HIP         -2   /* HIP runtime error or no gfx950 device; see dbgk_last_error()    */
#define DBGK_ERR_TABLE_FULL  -3   /* an insert found no free slot: the reference would spin forever
                                     here (DBGgraph.cpp:170-205); create a larger handle            */
#define DBGK_ERR_STATE       -4   /* call order violated (push after finalize, export before, ...)  */
#define DBGK_ERR_NOMEM       -5   /* device or host allocation failed                               */
#define DBGK_ERR_CAPACITY    -6   /* caller-provided output buffer too small                        */

/* the 16-byte graph node: bit-identical to KmerNode (DBG_contig/kmerSet.h:70-75).  Each link word
 * holds four saturating 8-bit counters, A in bits 31..24 ... T in bits 7..0 (kmerSet.cpp:56).      */
typedef struct dbgk_node {
	uint64_t kmer;
	uint32_t l_link;
	uint32_t r_link;
} dbgk_node;

typedef struct dbgk_handle dbgk_handle;

/* engine selection for the insert path */
#define DBGK_ENGINE_AUTO       0
#define DBGK_ENGINE_DIRECT     1  /* fused extract + 64-bit-atomic insert into the global table      */
#define DBGK_ENGINE_PARTITION  2  /* extract -> radix-partitioned records -> LDS-built table regions */
#define DBGK_ENGINE_SEEDIDX    4  /* link_scaffold's seed index: every k-mer of the pushed CONTIGS -> (contig
                                     index, position, strand) of its first occurrence + uniqueness flag   */
#define DBGK_ENGINE_WIDE       5  /* k-mers of up to 63 bases: 128-bit keys, 32-byte nodes (include/dbgk_wide.h; the reference
                                     stops at k = 31, so this path is this build's own definition: parity unpinned for
                                     k > 32, identical to the reference's rules for k <= 32).  expected_kmers > 0 and
                                     2^26 <= table_slots < 2^34: 16-byte records are radix-partitioned and the table is built
                                     region by region in LDS (2.7x faster; 16 bytes per occurrence of a pass + an eighth again
                                     must fit the device; with one shard and one pass what does not fit the record store joins
                                     through the atomic kernels after the build); otherwise fused extract + atomic insert.
                                     shard_count >= 1: slot-range shards of ONE table like the 64-bit engine (dbgk_shard_*);
                                     big tables / small devices: several passes over the input (dbgk_wide_begin_pass).
                                     Results through dbgk_wide_export_*, dbgk_digest, dbgk_link_stats_device            */
#define DBGK_ENGINE_KFREQ      3  /* no graph: direct-addressed 4^k table of saturating 8-bit counts of
                                     canonical k-mers (the correct_error module's frequency table);
                                     table_slots is ignored, k <= 18.  expected_kmers > 0: the occurrences
                                     are partitioned and aggregated in LDS (2.6x faster; 8 bytes per
                                     occurrence must fit the device), 0: atomics on the table, any size */

typedef struct dbgk_config {
	int32_t  kmer_size;        /* KmerSize   (DBGgraph.cpp:10), 1..32 (1..63 with DBGK_ENGINE_WIDE)     */
	int32_t  max_read_len;     /* maxReadLen (DBGgraph.cpp:11): longer reads are trimmed           */
	uint64_t table_slots;      /* number of 16-byte slots of the device table == kset->size the
	                              host wants (a "prime" from find_next_prime, kmerSet.cpp:85-95);
	                              slot of a key = hash_code(key) % table_slots (DBGgraph.cpp:167) */
	int32_t  device_id;        /* HIP device ordinal                                               */
	int32_t  engine;           /* DBGK_ENGINE_*                                                    */
	uint64_t max_batch_bases;  /* capacity of the internal staging buffers used by
	                              dbgk_push_reads (host buffers); 0 = default (256 MiB)            */
	uint64_t expected_kmers;   /* PARTITION engine: k-mer occurrences the record store is sized for
	                              (8 bytes each, twice).  Input of unknown size streams through it:
	                              a push that would not fit is preceded by a flush (dbgk_flush), which
	                              merges the stored records into the table.  A job that fits is built
	                              in one go at finalize (fastest).  0 = derive from table_slots      */
	uint32_t shard_count;      /* > 1: this handle is one of shard_count handles (one per GPU) that
	                              together hold ONE table of table_slots slots, each a contiguous
	                              slot range; 0 = unsharded (1 = a single shard that still follows
	                              the exchange protocol, useful for testing it on one GPU)        */
	uint32_t shard_index;      /* 0 .. shard_count-1                                               */
	uint64_t flags;            /* DBGK_FLAG_*                                                      */
	uint64_t n_passes;         /* DBGK_ENGINE_WIDE through records: read the input (at least) this many times, every
	                              pass keeping the records of a part of the level-1 buckets (dbgk_wide_begin_pass);
	                              the geometry may need more (a pass fans out to at most 1024 level-1 buckets, counted
	                              over all shards): ask dbgk_wide_pass_info.  0 on an unsharded handle = the plain
	                              create / push / finalize flow, never the pass protocol: a table one pass cannot
	                              cover is then built by the atomic kernels.  0 on a sharded handle = as few as needed */
	uint64_t reserved[1];
} dbgk_config;

/* DIRECT engine only: remember for every key the position (in pushed bases, over all pushes) of its
 * first occurrence, so that dbgk_export_first_seen_order can list the nodes in the order in which
 * the reference's single-threaded path first inserts them (DBGgraph.cpp:139-205 at -t 1).  Costs one
 * extra 64-bit atomic per k-mer and 8 bytes per slot.                                            */
#define DBGK_FLAG_TRACK_FIRST_SEEN 1ull
/* allocate the two pinned staging buffers (max_batch_bases each + offsets) and their device twins inside dbgk_create instead of on the
 * first dbgk_push_reads / dbgk_push_acquire: page-locking a few hundred MB costs tens of milliseconds, which a host that creates the
 * handle on a thread beside its first file read hides this way (host/DBGgraph.cpp)                                               */
#define DBGK_FLAG_PREALLOC_STAGING 2ull

/* totals after dbgk_finalize (the globals the reference prints, DBGgraph.cpp:410-411, and the
 * KmerSet counters of kmerSet.cpp:331-338) */
typedef struct dbgk_stats {
	uint64_t total_reads;      /* Total_reads_num: every record pushed, also too-short ones (:274) */
	uint64_t total_kmers;      /* Kmer_total_num : sum of (len-K+1) over reads with len >= K,
	                              UNtrimmed length (:101)                                          */
	uint64_t stored_kmers;     /* windows actually extracted = sum max(0,min(len,maxReadLen)-K+1);
	                              the unit of BASELINE.json's metric                               */
	uint64_t count;            /* kset->count: distinct keys INCLUDING the key-0 node (:418)       */
	uint64_t count_conflict;   /* probe steps taken (layout dependent, informational)              */
	uint64_t table_slots;
	uint32_t polyA_l_link;     /* links of the key-0 (poly-A / poly-T) node (:153-164)             */
	uint32_t polyA_r_link;
	uint64_t other_bytes;      /* sequence bytes that are none of ACGTNacgtn.  The reference maps them to
	                              4 and then reads out of bounds (seqKmer.cpp:9-19, DBGgraph.cpp:71-73);
	                              here EVERY engine reads them as 'A', like N, and counts them (ABI 5) */
} dbgk_stats;

/* first pass of the consumer (DBG_contig/contig.cpp:107-181) computed on the device table */
typedef struct dbgk_link_stats {
	int64_t depth_stat[256];   /* DepthStat: histogram of all 8 counters of every node            */
	int64_t total_nodes;
	int64_t deleted_lowfreq;   /* nodes with no counter > cutoff on either side                    */
	int64_t linear_nodes;      /* exactly one on each side                                         */
	int64_t tip_nodes;         /* l+r == 1                                                         */
	int64_t branch_nodes;      /* l > 1 or r > 1                                                   */
} dbgk_link_stats;

/* per-phase device timings of the most recent push/finalize, from HIP events on the handle's
 * stream (milliseconds; 0 when the phase did not run) */
typedef struct dbgk_timings {
	float mark_ms;             /* read-boundary bitmap + totals                                    */
	float insert_ms;           /* DIRECT: fused extract+insert kernel; PARTITION: extract+scatter  */
	float partition_ms;        /* PARTITION: second-level scatter                                  */
	float build_ms;            /* PARTITION: LDS region build + emit                               */
	float fixup_ms;            /* PARTITION: overflow records through the direct path              */
	float finalize_ms;         /* key-0 node, flags, count reduce                                  */
	uint64_t insert_launches;  /* number of launches accumulated into insert_ms                    */
	float l2_build_wall_ms;    /* PARTITION: wall time of the second-level scatter and the region build
	                              together; they run concurrently on two streams, so partition_ms and
	                              build_ms (sums of their launches) overlap and add up to more than this */
	uint32_t partition_launches; /* launches accumulated into partition_ms (= into build_ms)         */
	uint32_t uniform_launches;   /* of insert_launches: batches of equal-length reads, which take the
	                                PARTITION engine's k_extract_scatter_uniform instead of k_extract_scatter */
	uint32_t prefix_launches;    /* of insert_launches: batches of mixed-length reads through k_extract_scatter_prefix (every read
	                                exactly the lanes its windows need; dbgk_partition.h)                                          */
	uint64_t reserved[1];
} dbgk_timings;

/* ---- life cycle ------------------------------------------------------------------------------ */

/* replaces init_kmerset_parallel + the staging allocations of build_debruijn_graph
 * (kmerSet.cpp:98-127, DBGgraph.cpp:381-402): allocates and zeroes the device table.            */
int dbgk_create(const dbgk_config *cfg, dbgk_handle **out);
int dbgk_destroy(dbgk_handle *h);
/* back to the state right after dbgk_create (table zeroed, totals cleared) without reallocating */
int dbgk_reset(dbgk_handle *h);

/* ---- the hot path ------------------------------------------------------------------------------ */

/* replaces one block iteration of parse_one_reads_file: thread_parseBlock + thread_updatekmers
 * (DBGgraph.cpp:38-120,126-213) for `n_reads` reads.  bases = the sequences back to back with no
 * separators (ASCII; ACGT in either case, N / n counts as A like seqKmer.cpp:9-19; any other byte -- IUPAC codes,
 * '-', '*', bytes >= 128: undefined behaviour in the reference -- is read as A as well and counted in
 * dbgk_stats.other_bytes), offsets[n_reads+1] = start of each read in `bases`, offsets[0] == 0.
 * HOST buffers; the call copies them through pinned double buffers and returns once the batch is
 * queued (asynchronous w.r.t. the device).  If `bases` is page-locked memory the GPU can read (hipHostMalloc,
 * hipHostRegister: detected with hipPointerGetAttributes) the sequences are copied host-to-device straight out of
 * it -- no staging copy, which is what bounds the pageable path -- and the call returns when the last of those
 * copies has run.  Either way the buffers may be reused on return.                                */
int dbgk_push_reads(dbgk_handle *h, const char *bases, const uint64_t *offsets, uint64_t n_reads);

/* the same without the copy: dbgk_push_acquire returns the handle's next pinned staging buffers -- room for cap_bases sequence
 * bytes and cap_reads + 1 offsets (offsets[0] = 0); it waits until the batch that used them last has left for the device -- the
 * caller fills them (a parser writes its reads there directly) and dbgk_push_commit(n_reads) queues the batch.  One acquire
 * per commit; dbgk_push_reads may be mixed in between batches.                                                        */
int dbgk_push_acquire(dbgk_handle *h, char **bases, uint64_t **offsets, uint64_t *cap_bases, uint64_t *cap_reads);
int dbgk_push_commit(dbgk_handle *h, uint64_t n_reads);

/* ---- 2-bit packed reads ------------------------------------------------------------------------
 * The reference's alphabet is two bits wide by definition (alphabet[], seqKmer.cpp:9-19: A a N n -> 0, C c -> 1, G g -> 2,
 * T t -> 3), so a reader can hand the reads over packed -- a quarter of the bytes over PCIe, and the level-1 kernels skip
 * their ASCII decode.  FORMAT: the bases of all reads back to back (no separators, like `bases` above), 16 per 32-bit word,
 * base i in bits 31 - 2 * (i % 16) .. 30 - 2 * (i % 16) of word i / 16 (first base in the top bits, as seq2bit packs a
 * k-mer, seqKmer.cpp:34-41); offsets[] count BASES, offsets[0] may be any base position of `packed`.
 * dbgk_pack_bases: the packer (host, any thread; AVX2 when the CPU has it): n_bases ASCII bytes -> bits of `packed` starting at
 *   base position first_base.  Bytes outside ACGTNacgtn become A and are counted: *other_bytes += their number.  Words that a
 *   call covers only partly (its first / last) are OR-ed into atomically, so several threads may pack neighbouring ranges of
 *   one buffer -- such boundary words must be zero beforehand.  dbgk_unpack_bases is its inverse (upper-case letters).
 * dbgk_push_reads_packed: dbgk_push_reads for such a buffer in HOST memory (page-locked memory is read by the copy engine
 *   directly); other_bytes = what the packer counted for these reads, added to dbgk_stats.other_bytes.
 * dbgk_push_commit_packed: commit for a batch that was written PACKED into the buffers of dbgk_push_acquire (the `bases`
 *   buffer taken as uint32_t words, offsets[0] = 0; a batch still holds at most cap_bases bases).
 * dbgk_push_reads_packed_device: the packed words and the offsets are in device memory (d_packed 16-byte aligned, readable
 *   through word (n_bases + 15) / 16 - 1); dbgk_pack_bases_device makes such a buffer from ASCII bases in device memory.
 * Every engine takes packed batches except DBGK_ENGINE_SEEDIDX, whose windows are cut at 'N' (DBGK_ERR_ARG).             */
int dbgk_pack_bases(const char *bases, uint64_t n_bases, uint32_t *packed, uint64_t first_base, uint64_t *other_bytes);
/* the same for n_reads sequences that lie anywhere in memory (a parser's view of a file window), packed back to back from base
 * position first_base on -- one call per reader thread and share of a batch; only the first and the last word of the call's range are
 * OR-ed into (zero beforehand where a neighbour shares them)                                                                   */
typedef struct dbgk_read_ref {
	const char *seq;
	uint32_t    len;
} dbgk_read_ref;
int dbgk_pack_reads(const dbgk_read_ref *reads, uint64_t n_reads, uint32_t *packed, uint64_t first_base, uint64_t *other_bytes);
int dbgk_unpack_bases(const uint32_t *packed, uint64_t first_base, uint64_t n_bases, char *bases);
int dbgk_push_reads_packed(dbgk_handle *h, const uint32_t *packed, const uint64_t *offsets, uint64_t n_reads, uint64_t other_bytes);
int dbgk_push_commit_packed(dbgk_handle *h, uint64_t n_reads, uint64_t other_bytes);
int dbgk_push_reads_packed_device(dbgk_handle *h, const uint32_t *d_packed, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases);
int dbgk_pack_bases_device(dbgk_handle *h, const char *d_bases, uint64_t n_bases, uint32_t *d_packed);
/* reads of ONE length (what a sequencer writes, before anything trims them): n_reads reads of read_len bases each, back to back in the
 * packed format.  No offsets travel (at 150 bases they are a sixth of the packed bytes) and no statistics pass runs in front of
 * level 1; engines that navigate by offsets get them made on the device.  Otherwise like dbgk_push_reads_packed[_device].      */
int dbgk_push_reads_packed_uniform(dbgk_handle *h, const uint32_t *packed, uint64_t n_reads, uint32_t read_len, uint64_t other_bytes);
int dbgk_push_reads_packed_uniform_device(dbgk_handle *h, const uint32_t *d_packed, uint64_t n_reads, uint32_t read_len);

/* same, for reads already resident in device memory of the handle's GPU (both pointers 16-byte
 * aligned, readable through the end of the last read).  Nothing is copied; the buffers must stay
 * valid until the next dbgk_sync/dbgk_finalize.                                                  */
int dbgk_push_reads_device(dbgk_handle *h, const char *d_bases, const uint64_t *d_offsets,
                           uint64_t n_reads, uint64_t n_bases);

/* replaces the tail of build_debruijn_graph (DBGgraph.cpp:418, add_node_to_kmerset
 * kmerSet.cpp:253-273): completes all queued work, appends the key-0 node, reduces the counters.
 * Returns DBGK_ERR_TABLE_FULL if any insert ran out of slots.                                    */
int dbgk_finalize(dbgk_handle *h, dbgk_stats *out);

/* wait for all queued work of the handle */
int dbgk_sync(dbgk_handle *h);

/* replaces enlarge_kmerset_parallel (kmerSet.cpp:132-189) for the DEVICE table: allocates a table
 * of new_slots, re-seats every node, frees the old one.  Content (the node multiset) is unchanged.
 * Allowed between pushes (it synchronises first; a PARTITION handle flushes its records first and
 * re-plans its bucket geometry for the new size, which must again be in the engine's range).      */
int dbgk_resize_table(dbgk_handle *h, uint64_t new_slots);

/* PARTITION engine, input of unknown size (the block loop of parse_one_reads_file, DBGgraph.cpp:226-356,
 * never knows how much is still to come): turn the records pushed so far into table nodes NOW.  Regions
 * that already hold nodes of an earlier flush are loaded back into LDS, the new records inserted, the
 * region written out again; the record store is empty afterwards and dbgk_refresh_stats is exact (between
 * flushes its counts exclude what still sits in the store).  Pushes do this on their own when the store
 * is full.  No-op for the other engines (their table is always current).                           */
int dbgk_flush(dbgk_handle *h);
/* PARTITION engine: upper bound of the k-mer occurrences waiting in the record store, and what the
 * store was sized for (0, 0 for the other engines)                                                 */
int dbgk_store_room(dbgk_handle *h, uint64_t *pending_kmers, uint64_t *capacity_kmers);

/* ---- results ----------------------------------------------------------------------------------- */

/* fills a host KmerSet: `array` (host_size nodes) and `nul_flag` (host_size/8+1 bytes, bit i =
 * byte i/8 mask 128>>(i%8), kmerSet.cpp:53) such that every key is reachable by linear probing
 * from hash_code(key) % host_size without crossing a clear flag (exist_kmerset kmerSet.cpp:280-302),
 * unused slots are all-zero, and the key-0 node sits on key 0's probe chain.  host_size may differ
 * from table_slots (the table is then re-seated on the device first: the GPU counterpart of
 * enlarge_kmerset_parallel, kmerSet.cpp:132-189).  Every slot of `array` and every byte of `nul_flag`
 * is written (the buffers need not be zeroed; ordinary pageable memory).  Tables of 256 MiB and more
 * cross the link as their occupied nodes + the occupancy bits and are laid out at their slots by host
 * threads (DBGK_EXPORT_THREADS, default 12), through the handle's idle page-locked batch buffers
 * where it has them: same bytes, a third of the traffic at the reference's load factors.           */
int dbgk_export_host_table(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag);

/* dbgk_export_host_table + the WHOLE first pass of the consumer, calculate_kmer_links (DBG_contig/contig.cpp:107-181), computed
 * on the device for exactly the table that is handed over, so that the consumer can skip its serial scan of it:
 *   klink[host_size]      the 2-byte KmerLink record of every slot (contig.h:31-42): byte 0 = l_link_num | l_link_base << 2 |
 *                         r_link_num << 4 | r_link_base << 6, byte 1 bit 0 = linear; 0 for empty slots (the consumer zeroes klink,
 *                         contig.cpp:60).  link number = counters > kmer_freq_cutoff, at most 3; base = first base with the
 *                         largest such counter (contig.cpp:129-163)
 *   del_flag[host_size/8+1]  bit set (128 >> (i % 8) of byte i / 8, kmerSet.h:161-164) for nodes with no link above the cutoff
 *                         (contig.cpp:165-168); all other bits 0
 *   tip_nodes / branch_nodes  slots with l_link_num + r_link_num == 1 / with a side of more than one link, in ASCENDING slot
 *                         order -- the order of the reference's loop (contig.cpp:119,173-178); either may be NULL (counts only);
 *                         *n_tips / *n_branches always return the counts; DBGK_ERR_CAPACITY when a list does not fit
 *   stats                 DepthStat[256] and the five class counts (optional)
 * Unsharded handles.  PARITY UNPINNED: contig.cpp needs Boost headers, absent here, so this is checked against this build's
 * restatement of those lines (tests), not against the compiled reference.                                                       */
int dbgk_export_host_table_links(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag, int32_t kmer_freq_cutoff,
                                 uint16_t *klink, uint8_t *del_flag, uint64_t *tip_nodes, uint64_t tip_capacity, uint64_t *n_tips,
                                 uint64_t *branch_nodes, uint64_t branch_capacity, uint64_t *n_branches, dbgk_link_stats *stats);

/* canonical dump: all nodes sorted by kmer (the parity artefact of SURVEY.md section 8(a)).
 * `capacity` = number of nodes `out` can hold (>= stats.count).                                  */
int dbgk_export_sorted(dbgk_handle *h, dbgk_node *out, uint64_t capacity, uint64_t *n_out);

/* all non-zero-key nodes sorted by the position of their first occurrence (needs
 * DBGK_FLAG_TRACK_FIRST_SEEN): out[i] and first_pos[i] (index into the concatenation of everything
 * pushed) for i < *n_out.  Replaying out[] through the reference's sequential insert and enlarge
 * schedule reproduces its -t 1 slot layout (host side: DBGK_LAYOUT=ref).   */
int dbgk_export_first_seen_order(dbgk_handle *h, dbgk_node *out, uint64_t *first_pos, uint64_t capacity, uint64_t *n_out);

/* order-independent digest of the node multiset: sum over nodes of
 * mix64(kmer ^ mix64((l_link << 32) | r_link)) mod 2^64, mix64 = splitmix64 finaliser             */
int dbgk_digest(dbgk_handle *h, uint64_t *digest);

/* calculate_kmer_links first pass (contig.cpp:119-181) on the device table                      */
int dbgk_link_stats_device(dbgk_handle *h, int32_t kmer_freq_cutoff, dbgk_link_stats *out);

/* ---- WIDE engine (k <= 63, 128-bit keys): results as 32-byte nodes, include/dbgk_wide.h ------------
 * canonical dump sorted by (kmer_hi, kmer_lo), the key-0 node first; `capacity` >= stats.count             */
int dbgk_wide_export_sorted(dbgk_handle *h, dbgk_node32 *out, uint64_t capacity, uint64_t *n_out);
/* host-layout table of host_size == table_slots nodes + nul_flag (bit i = byte i/8, mask 128 >> (i%8)): every key
 * reachable by linear probing from hash128(key) % host_size without crossing a clear flag, unused slots all-zero,
 * the key-0 node on key 0's chain -- the invariants of SURVEY 8(b) with the 128-bit hash                      */
int dbgk_wide_export_host_table(dbgk_handle *h, uint64_t host_size, dbgk_node32 *array, uint8_t *nul_flag);
/* several GPUs: every GPU builds the graph of its share of the reads (reads shard by record), then ships each node to
 * its owner, (hash128(key) >> 32) % n_parts -- the device analogue of the reference's `kmer % threadNum` ownership
 * (DBGgraph.cpp:148) -- which adds the counters up (per-byte saturating, exact for any split of the input).
 * dbgk_wide_partition_export: counts[p] = nodes owned by part p (this handle's key-0 node counts for part 0); with
 * d_nodes != NULL the nodes are written grouped by owner into that DEVICE buffer (part p at sum(counts[0..p))).
 * dbgk_wide_merge_nodes: insert-if-absent / saturating add of n nodes held in device memory of the handle's GPU;
 * allowed before and after dbgk_finalize (dbgk_refresh_stats recounts).                                        */
int dbgk_wide_partition_export(dbgk_handle *h, uint32_t n_parts, dbgk_node32 *d_nodes, uint64_t capacity, uint64_t *counts);
int dbgk_wide_merge_nodes(dbgk_handle *h, const dbgk_node32 *d_nodes, uint64_t n);
/* WIDE through records, SHARDS and PASSES.  shard_count = N: N handles (one per GPU) hold ONE table of table_slots slots, rank d
 * the slot range of its level-1 buckets -- the device analogue of the reference's `kmer % threadNum` ownership
 * (DBGgraph.cpp:148), as for 64-bit keys; the protocol is dbgk_shard_buffers / _mark_exchanged / _plan / _build_range /
 * _outgoing / _overflow / _merge below, with 16-byte records and 32-byte list entries (dbgk_node32: nodes {hi, lo, l_link, r_link},
 * observations {hi, lo, lb, rb}).  Keys whose low word is 0 and the key-0 node live outside the table (a 4096-slot side table per
 * handle): after dbgk_finalize they are gathered onto shard 0 -- dbgk_shard_side_export on every other shard (device list, the
 * key-0 node first), dbgk_wide_merge_nodes of that list on shard 0, dbgk_shard_side_clear on the others.
 * PASSES: the level-1 kernel fans out to at most 1024 buckets (counted over all shards) and a pass's records must fit the device,
 * so a big job reads its input n_passes times, the way disk-based k-mer counters pass over their input once per set of
 * partitions: for p in 0 .. n_passes-1 { dbgk_wide_begin_pass(p); push ALL reads; [exchange the buffers of dbgk_shard_buffers,
 * dbgk_shard_mark_exchanged;] dbgk_wide_end_pass } then dbgk_finalize.  Pass p keeps the records of own-bucket indices
 * [p * Bp, (p+1) * Bp) of every shard and completes those regions of the table; totals count the input once.  A handle with one
 * shard and one pass needs none of these calls.                                                                           */
int dbgk_wide_pass_info(dbgk_handle *h, uint32_t *n_passes, uint32_t *passes_done);
int dbgk_wide_begin_pass(dbgk_handle *h, uint32_t pass);
int dbgk_wide_end_pass(dbgk_handle *h);
int dbgk_shard_side_export(dbgk_handle *h, dbgk_node32 **d_nodes, uint64_t *n);
int dbgk_shard_side_clear(dbgk_handle *h);

/* ---- KFREQ engine: the k-mer frequency table of the correct_error module --------
 * The reference only CONSUMES this table (its producer, `kmerfreq`, is not part of the repository):
 * 8-bit format = 4^k saturating counts indexed by k-mer value, read by
 * correct_error/main.cpp:161-220; 1-bit format = bitmap, bit 128 >> (v % 8) of byte v / 8, read by
 * correct_error/main_parallel_senior.cpp:334-408 which mirrors bit v to its reverse complement only
 * when v <= rc(v) -- so counts live on the canonical (smaller) k-mer.  push_reads / finalize work as
 * for the graph engines; stats.count = number of distinct canonical k-mers.                       */
int dbgk_kfreq_export_counts(dbgk_handle *h, uint64_t first_kmer, uint64_t n, uint8_t *host_out);
/* n_bytes bytes of the bit table starting at byte first_byte: bit set when count > cutoff          */
int dbgk_kfreq_export_bits(dbgk_handle *h, uint32_t cutoff, uint64_t first_byte, uint64_t n_bytes, uint8_t *host_out);
/* Partial tables of several GPUs: counts[first_kmer, first_kmer + n) of a finalized handle
 * += n counters held in device memory of the handle's GPU, per-byte saturating -- exact, because
 * min(255, min(255,a) + min(255,b)) = min(255, a+b).  first_kmer, n and the address: multiples of 16.
 * dbgk_kfreq_device_counts: the handle's 4^k counters in device memory (what a peer sends).          */
int dbgk_kfreq_merge_counts(dbgk_handle *h, const uint8_t *d_counts, uint64_t first_kmer, uint64_t n);
int dbgk_kfreq_device_counts(dbgk_handle *h, uint8_t **d_counts, uint64_t *n);

/* ---- SEEDIDX engine: the contig k-mer index of the link_scaffold module ----------
 * chop_contig_to_kmerset (link_scaffold/map_func.cpp:119-173): push the contig sequences with
 * dbgk_push_reads (contig index = order of pushing; windows never span an upper-case 'N'), finalize,
 * export.  Nodes come out as the reference's 16-byte KmerNode of THAT module
 * (link_scaffold/kmerSet.h:54-61): kmer, then one 64-bit word {id:32, pos:30, freq:1, direct:1}
 * (low bits first) -- held in dbgk_node as l_link = low dword, r_link = high dword.  Key 0 is an
 * ordinary key here (the reference tests emptiness with nul_flag, not kmer == 0).
 * dbgk_seed_export_host_table fills a table every key of which is reachable from
 * hash_code(key) % host_size without crossing a clear nul_flag bit (exist_kmerset,
 * link_scaffold/kmerSet.cpp:216-238).                                                              */
int dbgk_seed_export_sorted(dbgk_handle *h, dbgk_node *out, uint64_t capacity, uint64_t *n_out);
int dbgk_seed_export_host_table(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag);

/* ---- phase A alone (parity of the extraction kernel) ------------------------------------------ */

/* thread_parseBlock only (DBGgraph.cpp:38-120) on host buffers: for every base position p of
 * `bases` writes valid[p] (1 if a k-mer window starts at p inside its read after trimming), and
 * for valid positions kmer[p], left[p], right[p] exactly as StoreKmer/StoreLeftBase/StoreRightBase
 * would hold them for (read i, j = p - offsets[i]).  All four outputs are HOST arrays of
 * offsets[n_reads] entries.                                                                      */
int dbgk_extract_kmers(dbgk_handle *h, const char *bases, const uint64_t *offsets, uint64_t n_reads,
                       uint64_t *kmer, uint8_t *left, uint8_t *right, uint8_t *valid);

/* ---- multi-GPU building blocks (reads shard by record, keys are owned by hash) ---------------- */

/* owner of a key among n_parts ranks: (hash_code(key) >> 32) % n_parts -- the device analogue of
 * the reference's `kmer % threadNum` ownership (DBGgraph.cpp:148).  After dbgk_finalize:
 * counts[p] = number of nodes of this handle's table owned by part p (host array, n_parts).      */
int dbgk_partition_counts(dbgk_handle *h, uint32_t n_parts, uint64_t *counts);
/* writes the nodes grouped by owner into the DEVICE buffer d_nodes (capacity nodes): part p
 * occupies [sum(counts[0..p)), +counts[p]).  The key-0 node is owned by part 0.                  */
int dbgk_partition_export(dbgk_handle *h, uint32_t n_parts, dbgk_node *d_nodes, uint64_t capacity);
/* merges n already-aggregated nodes (DEVICE buffer) into the table: insert-if-absent, otherwise
 * per-byte saturating add min(255, a+b) -- exact for any split of the input because
 * min(255, min(255,a)+min(255,b)) == min(255,a+b).  Allowed before and after dbgk_finalize; a
 * key-0 node in the input is folded into the handle's key-0 node.                                */
int dbgk_merge_nodes(dbgk_handle *h, const dbgk_node *d_nodes, uint64_t n);
/* recount after merges (finalize semantics without re-appending the key-0 node twice); also
 * usable between pushes to watch the load of the table                                          */
int dbgk_refresh_stats(dbgk_handle *h, dbgk_stats *out);
/* single-process multi-GPU: copy n nodes from a device buffer of src's GPU into a device buffer
 * of dst's GPU (peer copy over xGMI), synchronous                                               */
int dbgk_copy_nodes_peer(dbgk_handle *dst, dbgk_node *d_dst, dbgk_handle *src, const dbgk_node *d_src, uint64_t n);

/* ---- sharded table: reads shard by record, k-mers are owned by SLOT RANGE ---------------------
 * With shard_count = N every handle extracts its own reads into level-1 buckets of the GLOBAL table
 * (slot >> r); rank d owns the bucket range [d*B, (d+1)*B).  All N handles of a job must be created
 * with the same kmer_size, table_slots and expected_kmers (they fix the bucket geometry and the
 * capacity of the exchanged buffers; compare chunk_bytes across ranks if in doubt).  Between the pushes and dbgk_finalize
 * the caller moves chunk d of every rank's send buffer to rank d (an all-to-all: RCCL
 * ncclSend/ncclRecv, torch all_to_all_single, or peer copies), the bucket fill counts likewise,
 * then calls dbgk_shard_mark_exchanged.  finalize builds only the handle's slot range; the few
 * nodes whose probe runs off the end of a shard (dbgk_shard_outgoing) are handed to the next rank
 * (dbgk_shard_merge(..., from_previous_shard = 1)), bucket-overflow records (dbgk_shard_overflow,
 * normally none) are offered to every rank, which keeps its own.  Exact for the same reason the
 * reference's per-thread ownership is (DBGgraph.cpp:148): a key has exactly one home slot.       */
typedef struct dbgk_shard_info {
	uint32_t n_ranks, rank;
	uint64_t table_slots_global;
	uint64_t slot_lo, slot_hi;     /* this handle's slot range                                     */
	void    *d_send;               /* n_ranks chunks of chunk_bytes: chunk d goes to rank d        */
	void    *d_recv;               /* n_ranks chunks of chunk_bytes: chunk s comes from rank s     */
	uint64_t chunk_bytes;
	void    *d_send_cnt;           /* n_ranks chunks of cnt_chunk_bytes (u32 fill counts)          */
	void    *d_recv_cnt;
	uint64_t cnt_chunk_bytes;
	/* a chunk is buckets_per_rank level-1 buckets of bucket_bytes each (fill counts: cnt_bucket_bytes);
	 * this handle owns the first own_buckets of its range (the last rank may own fewer)          */
	uint32_t buckets_per_rank, own_buckets;
	uint64_t bucket_bytes, cnt_bucket_bytes;
} dbgk_shard_info;

/* The geometry a PARTITION handle of these parameters would get -- level-1 bucket width, fan-outs, this shard's slot range and
 * buckets, the bytes of its table shard and record stores -- WITHOUT touching a device (ABI 6): what a multi-GPU launcher plans
 * with before any rank has created its handle (bench.py --plan-only, tests/test_multigpu_gloo.py).  shard_count 0 = one
 * unsharded handle.  DBGK_ERR_ARG when the engine cannot take the table (2^26 <= table_slots < 2^34).                        */
typedef struct dbgk_plan_info {
	uint64_t table_slots;          /* the (global) table                                                       */
	uint32_t r;                    /* level-1 bucket = slot >> r                                               */
	uint32_t level1_buckets;       /* of the global table                                                      */
	uint32_t final_per_level1;     /* 4096-slot regions per level-1 bucket = level-2 fan-out                   */
	uint32_t three_level;          /* 1: level 2 runs as two passes (tables of 2^33 slots and more)            */
	uint32_t buckets_per_rank, own_buckets, first_bucket;
	uint32_t reserved;
	uint64_t slot_lo, slot_hi;     /* this shard's slot range                                                  */
	uint64_t records_per_level1_bucket, records_per_final_bucket; /* capacities                               */
	uint64_t table_bytes, level1_store_bytes, inbox_bytes, final_store_bytes; /* device memory of this shard   */
} dbgk_plan_info;
int dbgk_plan_partition(uint64_t table_slots, uint64_t expected_kmers, uint32_t shard_count, uint32_t shard_index, dbgk_plan_info *out);

int dbgk_shard_buffers(dbgk_handle *h, dbgk_shard_info *out);
int dbgk_shard_mark_exchanged(dbgk_handle *h);
/* Exchange and build IN PIECES, so that the transfer of later buckets overlaps the build of earlier
 * ones: once the fill counts have been exchanged call dbgk_shard_plan; whenever the records of the
 * own buckets [j0, j1) have arrived from every rank (bucket j of chunk s of d_recv) call
 * dbgk_shard_build_range(j0, j1) -- ranges in ascending order, each bucket once; the kernels are
 * queued on the handle's streams and the call returns.  dbgk_shard_mark_exchanged + dbgk_finalize
 * then build whatever is left and complete the step.                                            */
int dbgk_shard_plan(dbgk_handle *h);
int dbgk_shard_build_range(dbgk_handle *h, uint32_t j0, uint32_t j1);
/* after dbgk_finalize: device list of nodes that left this shard / of overflow observations
 * {kmer, lb | rb << 8}; the lists stay valid until the next dbgk_reset                            */
int dbgk_shard_outgoing(dbgk_handle *h, dbgk_node **d_nodes, uint64_t *n);
int dbgk_shard_overflow(dbgk_handle *h, dbgk_node **d_triples, uint64_t *n);
/* once the overflow list is full, further observations are aggregated in a side table of n_slots nodes
 * (empty slots are all-zero); n_slots = 0 when it was not needed.  Like the overflow list it may hold keys
 * of any shard: offer it to every rank with dbgk_shard_merge(..., is_triple = 0, from_previous_shard = 0)   */
int dbgk_shard_heavy(dbgk_handle *h, dbgk_node **d_table, uint64_t *n_slots);
/* merge nodes (is_triple = 0) or observations (is_triple = 1) held in device memory of this GPU:
 * entries whose home slot lies in another shard are ignored unless from_previous_shard is set,
 * in which case they continue their probe at this shard's first slot                            */
int dbgk_shard_merge(dbgk_handle *h, const dbgk_node *d_nodes, uint64_t n, int is_triple, int from_previous_shard);
/* fold another handle's key-0 node into this one (per-byte saturating add)                      */
int dbgk_add_polyA(dbgk_handle *h, uint32_t l_link, uint32_t r_link);
int dbgk_memcpy_d2d(dbgk_handle *h, void *d_dst, const void *d_src, size_t bytes);

/* ---- several GPUs inside one process (the C++ host layer with DBGK_GPUS=N) ------------------------
 * A communicator owns n sharded handles of ONE table of cfg->table_slots slots (shard i on devices[i]; the
 * same device may appear more than once -- n shards on one GPU, which is how the tests exercise the path).
 * cfg->expected_kmers sizes the record store of EACH handle; engine, shard_count and shard_index of cfg are
 * set by the call.  Reads shard by record (dbgk_comm_push_reads deals batches round robin), k-mers are owned
 * by slot range -- the device analogue of the reference's `kmer % threadNum` ownership
 * (DBG_contig/DBGgraph.cpp:148).  flush / finalize move every level-1 record bucket to its owner with peer
 * copies over xGMI (hipMemcpyPeerAsync, in pieces that overlap the region build of the previous piece), build
 * each shard's slot range and hand over the few stragglers (overflow observations, nodes that probed past the
 * end of a shard).  Same protocol as the one-process-per-GPU flow over RCCL (dbg_assembly_amd/multigpu.py).
 * One host thread per communicator.                                                                 */
typedef struct dbgk_comm dbgk_comm;
int dbgk_comm_create(const dbgk_config *cfg, const int32_t *devices, uint32_t n, dbgk_comm **out);
int dbgk_comm_destroy(dbgk_comm *c);
uint32_t dbgk_comm_size(const dbgk_comm *c);
dbgk_handle *dbgk_comm_handle(dbgk_comm *c, uint32_t i);
int dbgk_comm_push_reads(dbgk_comm *c, const char *bases, const uint64_t *offsets, uint64_t n_reads);
int dbgk_comm_push_reads_packed(dbgk_comm *c, const uint32_t *packed, const uint64_t *offsets, uint64_t n_reads, uint64_t other_bytes);
/* records -> table on every shard now (dbgk_flush for a communicator); a push does it when a store is full */
int dbgk_comm_flush(dbgk_comm *c);
/* enlarge_kmerset_parallel (kmerSet.cpp:132-189) for the table of a communicator: flushes, creates shards of a table of new_slots
 * slots, re-seats every node into the shard that owns its new home slot, frees the old shards.  The node multiset and the totals
 * are unchanged; handles obtained from dbgk_comm_handle before the call are gone.  Graph communicators only.                  */
int dbgk_comm_resize(dbgk_comm *c, uint64_t new_slots);
/* totals of the whole job (count includes the one key-0 node); exact after a flush                       */
int dbgk_comm_refresh_stats(dbgk_comm *c, dbgk_stats *out);
int dbgk_comm_finalize(dbgk_comm *c, dbgk_stats *out);
int dbgk_comm_digest(dbgk_comm *c, uint64_t *digest);
int dbgk_comm_link_stats(dbgk_comm *c, int32_t kmer_freq_cutoff, dbgk_link_stats *out);
/* the host KmerSet of the whole job (same contract as dbgk_export_host_table): the shards side by side when
 * host_size is the global table size, otherwise every node re-seated on the host                          */
int dbgk_comm_export_host_table(dbgk_comm *c, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag);
/* dbgk_export_host_table_links for a communicator: the host table of the whole job AND the consumer's first pass for it (klink,
 * del_flag, the ascending tip / branch slot lists, DepthStat: contig.cpp:107-181), computed on the device.  The shards' nodes are
 * first assembled in one table of host_size slots on the first member's device (host_size * 16 bytes must be free there).  Same
 * arguments, results and (un)pinned parity as dbgk_export_host_table_links.                                                  */
int dbgk_comm_export_host_table_links(dbgk_comm *c, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag, int32_t kmer_freq_cutoff,
                                      uint16_t *klink, uint8_t *del_flag, uint64_t *tip_nodes, uint64_t tip_capacity, uint64_t *n_tips,
                                      uint64_t *branch_nodes, uint64_t branch_capacity, uint64_t *n_branches, dbgk_link_stats *stats);
/* cfg->engine == DBGK_ENGINE_WIDE (k <= 63, expected_kmers > 0 = what EACH member extracts): n slot-range shards of one table of
 * 32-byte nodes; the reads stream through once (a geometry that needs several passes is refused: drive the handles yourself with
 * dbgk_wide_begin_pass), the record stores are built at dbgk_comm_finalize.  Results of the whole job:                          */
int dbgk_comm_wide_export_sorted(dbgk_comm *c, dbgk_node32 *out, uint64_t capacity, uint64_t *n_out);
int dbgk_comm_wide_export_host_table(dbgk_comm *c, uint64_t host_size, dbgk_node32 *array, uint8_t *nul_flag);
/* cfg->engine == DBGK_ENGINE_KFREQ: a communicator of n whole frequency tables.  Every member counts the reads
 * dealt to it; dbgk_comm_finalize makes member d the owner of an n-th of the k-mer values and adds the other
 * members' slices of that range to its own (peer copies in chunks, overlapped with the saturating add).  The
 * two exports read every range from its owner; stats.count = distinct canonical k-mers of the whole job.     */
int dbgk_comm_kfreq_export_counts(dbgk_comm *c, uint64_t first_kmer, uint64_t n, uint8_t *host_out);
int dbgk_comm_kfreq_export_bits(dbgk_comm *c, uint32_t cutoff, uint64_t first_byte, uint64_t n_bytes, uint8_t *host_out);

/* ---- utilities --------------------------------------------------------------------------------- */

/* synthetic reads (include/dbgk_synth.h) generated straight into device memory: d_bases gets
 * n_reads * read_len bytes, d_offsets n_reads+1 entries                                          */
int dbgk_synth_reads_device(dbgk_handle *h, const dbgk_synth_params *p, uint64_t first_read,
                            uint64_t n_reads, char *d_bases, uint64_t *d_offsets);

int dbgk_device_malloc(dbgk_handle *h, size_t bytes, void **d_ptr);
int dbgk_device_free(dbgk_handle *h, void *d_ptr);
int dbgk_memcpy_d2h(dbgk_handle *h, void *dst, const void *d_src, size_t bytes);
int dbgk_memcpy_h2d(dbgk_handle *h, void *d_dst, const void *src, size_t bytes);

int dbgk_get_timings(dbgk_handle *h, dbgk_timings *out);
int dbgk_reset_timings(dbgk_handle *h);
/* HIP stream of the handle as an opaque pointer (hipStream_t), for callers that time with events */
void *dbgk_stream(dbgk_handle *h);

/* the measured-HBM denominator of the roofline (GB/s, bytes read + bytes written): the best of the runtime's device-to-device
 * memcpy and this library's own streaming copy kernels (16 bytes per lane, default and non-temporal policy, 1 and 2 workgroups
 * per CU) over two buffers of `bytes` each (use >= 2 GiB: far beyond the 256 MiB Infinity Cache), `iters` copies per variant   */
int dbgk_measure_copy_bandwidth(dbgk_handle *h, size_t bytes, int iters, double *gbps);
/* the same, and the runtime's hipMemcpyDtoD figure alone next to the best (ABI 6): bench.py reports both denominators */
int dbgk_measure_copy_bandwidth2(dbgk_handle *h, size_t bytes, int iters, double *gbps, double *runtime_memcpy_gbps);
/* random 64-byte gather over a buffer of `bytes` (the practical random-access ceiling of the engines
 * that touch one random node per k-mer occurrence): n_accesses sectors fetched, GB/s and G sectors/s             */
int dbgk_measure_gather_bandwidth(dbgk_handle *h, size_t bytes, uint64_t n_accesses, double *gbps, double *gaccesses_per_s);

int dbgk_device_count(void);
int dbgk_abi_version(void);
const char *dbgk_strerror(int status);
/* text of the most recent HIP failure seen by this thread ("" if none) */
const char *dbgk_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DBGK_H_ */
