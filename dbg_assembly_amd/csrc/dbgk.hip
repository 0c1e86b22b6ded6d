// dbgk.hip -- host side of the C ABI declared in include/dbgk.h: device memory, stream, staging,
// kernel launches.  gfx950 only; no CPU fallback.
#define DBGK_HD __host__ __device__
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <chrono>
#include <emmintrin.h>
#include <string>
#include <thread>
#include <vector>

#include "dbgk.h"
#include "dbgk_env.h"
#include "dbgk_kernels.h"
#include "dbgk_partition.h"
#include "dbgk_wide_kernels.h"
#include "dbgk_wide_partition.h"

// dbgk_sort.hip
extern "C" int dbgk_internal_sort_pairs(uint64_t *d_keys, uint64_t *d_vals, uint64_t n, hipStream_t stream);

using namespace dbgk;

static_assert(sizeof(dbgk_node) == 16 && sizeof(Node) == 16, "node layout must match KmerNode (kmerSet.h:70-75)");

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int hip_fail(hipError_t e, const char *what, int line)
{
	char buf[512];
	snprintf(buf, sizeof buf, "%s failed at dbgk.hip:%d: %s", what, line, hipGetErrorString(e));
	g_last_error = buf;
	return DBGK_ERR_HIP;
}

#define HIPCHK(expr)                                                   \
	do {                                                               \
		hipError_t e__ = (expr);                                       \
		if (e__ != hipSuccess) return hip_fail(e__, #expr, __LINE__);  \
	} while (0)

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t kMaxBuildLaunches = 4096; // build ranges per step (one work cursor each)
constexpr uint64_t kHeavyHitterSlots = 4194301; // side table for the surplus of heavy hitters (prime, 64 MiB)

enum Phase { PH_MARK = 0, PH_INSERT, PH_PARTITION, PH_BUILD, PH_FIXUP, PH_FINALIZE, PH_L2_BUILD_WALL, PH_COUNT };

struct TimedSpan {
	hipEvent_t a, b;
	int phase;
};

struct StageSlot {
	char *h_bases = nullptr;
	uint64_t *h_offsets = nullptr;
	char *d_bases = nullptr;
	uint64_t *d_offsets = nullptr;
	uint32_t *d_start = nullptr;
	uint32_t *d_dead = nullptr;
	hipEvent_t done = nullptr;      // the kernels that read the device buffers have run
	hipEvent_t copied = nullptr;    // the host-to-device copies of the batch have run (copy stream)
	bool busy = false;
	bool acquired = false;          // handed out by dbgk_push_acquire and not committed yet
};

} // namespace

struct dbgk_handle {
	dbgk_config cfg;
	int device = 0;
	int n_cu = 256;
	int grid = 2048;
	hipStream_t stream = nullptr;

	Node *table = nullptr;
	uint64_t size = 0;            // hash modulus: slot = hash_code(key) % size (the GLOBAL table when sharded)
	uint64_t tslots = 0;          // slots held by `table` (== size unless sharded)
	ModMagic magic;

	Counters *d_ctr = nullptr;
	Counters *h_ctr = nullptr; // pinned

	// host-buffer staging
	StageSlot slots[2];
	uint64_t cap_bases = 0, cap_reads = 0;
	int next_slot = 0;

	// bitmaps for the device-pointer path
	uint32_t *dev_start = nullptr, *dev_dead = nullptr;
	uint64_t dev_bits_words = 0;
	// level 1 for reads of any lengths (k_extract_scatter_prefix): the per-read lane prefix of the current batch, and the
	// 2-bit form of a batch that came as ASCII.  One set per handle: batches follow each other on one stream.
	ReadLanes *pf_ent = nullptr;
	uint32_t *pf_tile_first = nullptr;
	PrefixTile *pf_tiles = nullptr;
	unsigned long long *pf_bsum = nullptr;
	PrefixTotals *pf_tot = nullptr;
	uint64_t pf_cap_reads = 0, pf_cap_tiles = 0;
	uint32_t *pf_packed = nullptr;
	uint64_t pf_packed_words = 0;
	uint64_t *uni_offsets = nullptr; // offsets made on the device for a batch that came without (dbgk_push_reads_packed_uniform*) and needs them
	uint64_t uni_cap = 0;
	uint32_t prefix_launches = 0;

	bool finalized = false;
	uint64_t total_reads = 0;
	uint64_t host_other_bytes = 0; // bytes outside ACGTNacgtn that a host packer met (dbgk_push_reads_packed / dbgk_push_commit_packed)

	// first-seen tracking (DBGK_FLAG_TRACK_FIRST_SEEN, DIRECT engine)
	bool track = false;
	unsigned long long *first_pos = nullptr; // [tslots]
	uint64_t pos_base = 0;                   // bases pushed so far

	bool wide = false;            // WIDE engine: 128-bit keys, 32-byte nodes (dbgk_wide_kernels.h)
	WNode *wnodes = nullptr;      // [size]
	WNode *wside = nullptr;       // [kWideSideSlots]
	// WIDE through radix-partitioned records (dbgk_wide_partition.h): expected_kmers > 0 and a feasible geometry
	bool wpart = false;
	bool wbuilt = false;          // the records have been turned into the table: later batches use the atomic kernels
	bool wzero_pending = false;   // the main table is stale (reset without memset): the build overwrites every slot
	WPartGeom wgeom;
	WPartStore wstore;
	uint32_t *w_tile_prefix = nullptr;
	unsigned int *w_cursor = nullptr;
	// shards and passes of the wide record path (WPartGeom): `wmulti` handles follow the strict protocol
	// begin_pass -> pushes -> [exchange] -> end_pass, ..., finalize; nothing streams through the atomic kernels
	bool wmulti = false;          // sharded and / or several passes
	bool wpass_open = false;      // a pass has been begun and not ended
	bool wplanned = false;        // level-2 tile plan of the current pass made
	uint32_t wnext = 0;           // own-bucket indices [0, wnext) of the current pass have been built
	uint32_t wpasses_done = 0;
	ull2 *winbox = nullptr;       // sharded: [n_l1][cap1] records of my buckets from every rank
	uint32_t *winbox_cnt = nullptr;
	dbgk_node32 *w_side_out = nullptr;      // [kWideSideSlots + 1] dbgk_shard_side_export
	unsigned long long *w_side_n = nullptr;
	unsigned long long wsaved_totals[2] = {0, 0}; // total_kmers, stored_kmers before a repeated pass over the input
	uint64_t wsaved_reads = 0;
	unsigned long long wsaved_other = 0;          // ... and Counters::other_bytes
	uint64_t wsaved_host_other = 0;
	uint32_t shard_rank = 0;      // shard_index of a sharded handle (any engine)
	bool seed = false;            // SEEDIDX engine: node payload = first occurrence + uniqueness
	// KFREQ engine: counts[4^k] instead of a node table
	bool kfreq = false;
	bool kf_blocks = false;       // KFREQ through the PARTITION engine in its direct-block form (geom.kf == 2, dbgk_partition.h)
	uint8_t *counts = nullptr;
	uint64_t n_counts = 0;
	uint64_t kf_distinct = 0, kf_sum = 0;

	// PARTITION engine
	bool part = false;            // records are partitioned at push time, table built at finalize
	bool part_built = false;      // finalize already turned the records into the table
	bool part_planned = false;    // level-2 tile plan made for this step (part_plan)
	uint32_t next_bucket = 0;     // own level-1 buckets [0, next_bucket) have been handed to level 2 + build
	uint32_t chunks_used = 0;     // chunk events consumed this step
	unsigned int *region_cursor = nullptr; // [kMaxBuildLaunches] work cursors of the persistent build launches
	uint32_t cursors_used = 0;
	TimedSpan wall_span;
	bool zero_pending = false;    // table content is stale and must be zeroed before a direct-path write
	bool incr = false;            // the table holds nodes of an earlier flush: the next region build loads them back (INCR)
	uint64_t pending_kmers = 0;   // upper bound of the k-mer occurrences sitting in the record store
	uint64_t store_capacity = 0;  // occurrences the record store is sized for (expected_kmers)
	PartGeom geom;
	PartStore store;
	uint32_t *tile_prefix = nullptr; // [n_ranks * B + 1] level-2 tile plan
	uint32_t *l2_done = nullptr;     // EARLY level 2 (early_l2): records per level-1 bucket that level 2 has taken already; null = not in use
	uint64_t l2_seen_kmers = 0;      // pending_kmers when the last early level-2 round was launched
	// THREE-LEVEL partition for tables whose level-1 buckets hold 4096 regions (2^33 slots and more): level 2
	// runs as two passes of the same kernel -- MID: every level-1 bucket into fan_mid = n2 / 64 mid buckets (store `mid`,
	// laid out like the level-1 store of a table with r - log2(fan_mid)), FINAL: every mid bucket into its 64 final buckets,
	// written exactly where the two-level form puts them, so the build is the same.  One more pass over the records instead
	// of ~3500 reservation atomics per 8192-record tile.
	bool three = false;
	uint32_t fan_mid = 0;
	PartGeom g_mid, g_fin;
	PartStore s_mid, s_fin;
	uint64_t *mid = nullptr;
	uint32_t *cnt_mid = nullptr;
	uint32_t *tile_prefix2 = nullptr; // [nb_own * fan_mid + 1]
	hipStream_t stream2 = nullptr;   // finalize: region build of bucket chunk c runs here while level 2 of chunk c+1 runs on `stream`
	std::vector<hipEvent_t> chunk_ev; // level 2 of chunk c finished
	hipEvent_t join_ev = nullptr;
	// sharding: the handle owns slots [geom.slot_lo, geom.slot_hi) of a GLOBAL table of `size` slots
	bool sharded = false;         // shard_count > 1
	bool exchanged = false;       // the caller has filled the inbox (all-to-all) for this step
	uint64_t *inbox = nullptr;    // [n_ranks][B][cap1]
	uint32_t *inbox_cnt = nullptr;

	// large device-to-host copies (the host KmerSet): slices through pinned buffers, host threads moving them on
	std::vector<void *> d2h_stage;
	std::vector<hipEvent_t> d2h_ev;
	hipEvent_t source_read = nullptr; // dbgk_push_reads from a pinned caller buffer: the last host-to-device copy out of it
	hipStream_t copy_stream = nullptr; // host-to-device copies of the batches: batch i+1 travels while the kernels of batch i run

	std::vector<TimedSpan> spans, free_spans;
	float phase_ms[PH_COUNT] = {0};
	uint64_t insert_launches = 0;
	uint32_t partition_launches = 0;
	uint32_t uniform_launches = 0;

	TableRef tref() const { return TableRef{table, size, magic}; }
	WTable wref() const { return WTable{wnodes, size, magic, wside}; }
};

static int use_device(dbgk_handle *h)
{
	HIPCHK(hipSetDevice(h->device));
	return DBGK_OK;
}

static int span_begin(dbgk_handle *h, int phase, TimedSpan &s, hipStream_t stream = nullptr)
{
	if (!h->free_spans.empty()) {
		s = h->free_spans.back();
		h->free_spans.pop_back();
	} else {
		HIPCHK(hipEventCreate(&s.a));
		HIPCHK(hipEventCreate(&s.b));
	}
	s.phase = phase;
	HIPCHK(hipEventRecord(s.a, stream ? stream : h->stream));
	return DBGK_OK;
}

static int span_end(dbgk_handle *h, TimedSpan &s, hipStream_t stream = nullptr)
{
	HIPCHK(hipEventRecord(s.b, stream ? stream : h->stream));
	h->spans.push_back(s);
	return DBGK_OK;
}

// fold finished spans into the per-phase totals.  The caller has synchronised `stream`; region builds
// of a ranged finalize may still be running on `stream2` (their spans are recorded there).
static int collect_spans(dbgk_handle *h)
{
	if (h->stream2) HIPCHK(hipStreamSynchronize(h->stream2));
	for (auto &s : h->spans) {
		float ms = 0.f;
		HIPCHK(hipEventElapsedTime(&ms, s.a, s.b));
		h->phase_ms[s.phase] += ms;
		if (s.phase == PH_INSERT) h->insert_launches++;
		if (s.phase == PH_PARTITION) h->partition_launches++;
		h->free_spans.push_back(s);
	}
	h->spans.clear();
	return DBGK_OK;
}

static inline uint64_t bitmap_words(uint64_t n_bases) { return (n_bases >> 5) + 4; }

// level-1 workgroups per CU: as many as fill the CU's wave slots, unless DBGK_L1_PER_CU says otherwise (experiments with
// 512-thread builds that leave LDS for a workgroup of another kernel)
static int l1_wgs_per_cu()
{
	static const int v = DBGK_EXPERIMENT_ENV("DBGK_L1_PER_CU") ? std::max(1, atoi(DBGK_EXPERIMENT_ENV("DBGK_L1_PER_CU"))) : 1024 / kL1Threads;
	return v;
}

static int grid_for(const dbgk_handle *h, uint64_t items)
{
	uint64_t blocks = (items + kBlock - 1) / kBlock;
	if (blocks < 1) blocks = 1;
	return (int)std::min<uint64_t>(blocks, (uint64_t)h->grid);
}

// ---------------------------------------------------------------------------------------------
// life cycle
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_abi_version(void) { return DBGK_ABI_VERSION; }

extern "C" int dbgk_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" const char *dbgk_last_error(void) { return g_last_error.c_str(); }

extern "C" const char *dbgk_strerror(int status)
{
	switch (status) {
		case DBGK_OK: return "ok";
		case DBGK_ERR_ARG: return "bad argument";
		case DBGK_ERR_HIP: return "HIP runtime error (see dbgk_last_error)";
		case DBGK_ERR_TABLE_FULL: return "k-mer table full";
		case DBGK_ERR_STATE: return "call order violated";
		case DBGK_ERR_NOMEM: return "out of memory";
		case DBGK_ERR_CAPACITY: return "output buffer too small";
		default: return "unknown status";
	}
}

// record stores of the PARTITION engine (re-allocated when the table is resized: the geometry changes)
static void free_partition_stores(dbgk_handle *h)
{
	for (void *p : {(void *)h->store.l1, (void *)h->store.l2, (void *)h->store.cnt1, (void *)h->store.cnt2, (void *)h->store.ovf,
	                (void *)h->store.spill, (void *)h->store.ovf_n, (void *)h->tile_prefix, (void *)h->store.hh, (void *)h->region_cursor, (void *)h->l2_done,
	                (void *)h->mid, (void *)h->cnt_mid, (void *)h->tile_prefix2,
	                (void *)h->inbox, (void *)h->inbox_cnt, (void *)h->store.outgoing, (void *)h->store.outgoing_n})
		if (p) (void)hipFree(p);
	memset(&h->store, 0, sizeof h->store);
	h->tile_prefix = nullptr;
	h->l2_done = nullptr;
	h->region_cursor = nullptr;
	h->inbox = nullptr;
	h->inbox_cnt = nullptr;
	h->mid = nullptr;
	h->cnt_mid = nullptr;
	h->tile_prefix2 = nullptr;
}

static void free_wide_partition(dbgk_handle *h)
{
	WPartStore &P = h->wstore;
	for (void *p : {(void *)P.l1, (void *)P.cnt1, (void *)P.l2, (void *)P.cnt2, (void *)P.ovf, (void *)P.spill, (void *)P.ovf_n, (void *)h->w_tile_prefix,
	                (void *)h->w_cursor, (void *)h->winbox, (void *)h->winbox_cnt, (void *)P.outgoing, (void *)P.outgoing_n, (void *)h->w_side_out,
	                (void *)h->w_side_n})
		if (p) (void)hipFree(p);
	memset(&P, 0, sizeof P);
	h->w_tile_prefix = nullptr;
	h->w_cursor = nullptr;
	h->winbox = nullptr;
	h->winbox_cnt = nullptr;
	h->w_side_out = nullptr;
	h->w_side_n = nullptr;
}

static void free_handle(dbgk_handle *h)
{
	if (!h) return;
	(void)hipSetDevice(h->device);
	if (h->stream2) (void)hipStreamSynchronize(h->stream2); // region builds of an unfinished ranged finalize
	if (h->stream) (void)hipStreamSynchronize(h->stream);
	for (void *q : {(void *)h->pf_ent, (void *)h->pf_tile_first, (void *)h->pf_tiles, (void *)h->pf_bsum, (void *)h->pf_tot, (void *)h->pf_packed, (void *)h->uni_offsets})
		if (q) (void)hipFree(q);
	for (void *p : h->d2h_stage)
		if (p) (void)hipHostFree(p);
	for (hipEvent_t e : h->d2h_ev)
		if (e) (void)hipEventDestroy(e);
	if (h->source_read) (void)hipEventDestroy(h->source_read);
	if (h->copy_stream) {
		(void)hipStreamSynchronize(h->copy_stream);
		(void)hipStreamDestroy(h->copy_stream);
	}
	for (auto &s : h->slots) {
		if (s.h_bases) (void)hipHostFree(s.h_bases);
		if (s.h_offsets) (void)hipHostFree(s.h_offsets);
		if (s.d_bases) (void)hipFree(s.d_bases);
		if (s.d_offsets) (void)hipFree(s.d_offsets);
		if (s.d_start) (void)hipFree(s.d_start);
		if (s.d_dead) (void)hipFree(s.d_dead);
		if (s.done) (void)hipEventDestroy(s.done);
		if (s.copied) (void)hipEventDestroy(s.copied);
	}
	for (auto &v : {&h->spans, &h->free_spans})
		for (auto &s : *v) {
			(void)hipEventDestroy(s.a);
			(void)hipEventDestroy(s.b);
		}
	if (h->dev_start) (void)hipFree(h->dev_start);
	if (h->dev_dead) (void)hipFree(h->dev_dead);
	if (h->part) {
		free_partition_stores(h);
		for (hipEvent_t e : h->chunk_ev) (void)hipEventDestroy(e);
		if (h->join_ev) (void)hipEventDestroy(h->join_ev);
		if (h->stream2) (void)hipStreamDestroy(h->stream2);
	}
	if (h->wpart) free_wide_partition(h);
	if (h->wnodes) (void)hipFree(h->wnodes);
	if (h->wside) (void)hipFree(h->wside);
	if (h->table) (void)hipFree(h->table);
	if (h->counts) (void)hipFree(h->counts);
	if (h->first_pos) (void)hipFree(h->first_pos);
	if (h->d_ctr) (void)hipFree(h->d_ctr);
	if (h->h_ctr) (void)hipHostFree(h->h_ctr);
	if (h->stream) (void)hipStreamDestroy(h->stream);
	delete h;
}

static int zero_table_now(dbgk_handle *h)
{
	HIPCHK(hipMemsetAsync(h->table, 0, h->tslots * sizeof(Node), h->stream)); // memset_parallel, kmerSet.cpp:358-386
	h->zero_pending = false;
	return DBGK_OK;
}

// empty the record stores of the PARTITION engine (the table and the counters stay)
static int clear_record_store(dbgk_handle *h, bool with_counters = false /* also reset the handle's counters (dbgk_reset) */)
{
	if (h->chunks_used > 0 && h->stream2) { // region builds on the second stream must not race the memsets
		HIPCHK(hipEventRecord(h->join_ev, h->stream2));
		HIPCHK(hipStreamWaitEvent(h->stream, h->join_ev, 0));
	}
	// the heavy-hitter side table is written only once the overflow list is full: zeroed again only then (read BEFORE ovf_n is cleared)
	if (h->store.hh)
		hipLaunchKernelGGL(k_zero_if_greater, dim3(h->n_cu * 4), dim3(kBlock), 0, h->stream, reinterpret_cast<uint4 *>(h->store.hh), h->store.hh_size,
		                   h->store.ovf_n, (unsigned long long)h->store.ovf_cap);
	ZeroList z{};
	int ne = 0;
	auto add = [&](void *q, size_t bytes) { z.p[ne] = q; z.dwords[ne] = (uint32_t)(bytes / 4); ne++; };
	add(h->store.cnt1, (size_t)h->geom.n_ranks * h->geom.B * h->geom.n_sub * 4);
	add(h->store.cnt2, (size_t)h->geom.nb_own * h->geom.n2 * 4);
	if (h->three) add(h->cnt_mid, (size_t)h->geom.nb_own * h->fan_mid * 4);
	add(h->store.ovf_n, 16);
	add(h->store.outgoing_n, 8);
	if (h->l2_done) add(h->l2_done, (size_t)h->geom.n_ranks * h->geom.B * h->geom.n_sub * 8); // done[] and upto[]
	add(h->region_cursor, (kMaxBuildLaunches + 2) * sizeof(unsigned int)); // one work cursor per build launch of the next build, + the exact pass's cursor and count
	hipLaunchKernelGGL(k_zero_list, dim3(64), dim3(kBlock), 0, h->stream, z, with_counters ? h->d_ctr : (Counters *)nullptr);
	HIPCHK(hipGetLastError());
	h->part_built = false;
	h->exchanged = false;
	h->part_planned = false;
	h->next_bucket = 0;
	h->chunks_used = 0;
	h->pending_kmers = 0;
	h->l2_seen_kmers = 0;
	return DBGK_OK;
}

// kernels with more than 64 KiB of dynamic LDS have to say so
#define DBGK_LDS_ATTR(KERNEL, BYTES) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)))

// the host side by engine / concern (one translation unit; the kernels live in dbgk_kernels.h, dbgk_partition.h, dbgk_wide_*.h)
#include "dbgk_host_wide.h"
#include "dbgk_host_partition.h"
#include "dbgk_host_create.h"
#include "dbgk_host_level1.h"
#include "dbgk_host_push.h"
#include "dbgk_host_build.h"
#include "dbgk_host_export.h"
#include "dbgk_host_engines.h"
#include "dbgk_host_misc.h"
#include "dbgk_comm.h"
