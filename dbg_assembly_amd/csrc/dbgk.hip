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

// ---- WIDE through radix-partitioned records (dbgk_wide_partition.h) ---------------------------------
// geometry: level-1 bucket = slot >> r, final bucket = slot >> 11 (one 2048-slot region), n2 = 2^(r - 11) <= 2048; the second
// half of a record holds q = hash / size, r slot bits, 6 neighbour bits.  The level-1 kernel fans out to at most 1024 store
// entries per pass: rank x own-bucket index inside the pass's window (WPartGeom).  One rank, one pass, n1 <= 1024: the
// round-2 form.  *err: the configuration ASKS for shards / passes and cannot have them.
static bool plan_wide_partition(dbgk_handle *h, bool *err)
{
	*err = false;
	const bool off = dbgk_hook("wide_direct") != nullptr; // always the atomic kernels
	const uint32_t n_ranks = h->cfg.shard_count > 1 ? h->cfg.shard_count : 1;
	const bool want_shard = h->cfg.shard_count >= 1;
	const uint32_t want_passes = (uint32_t)h->cfg.n_passes;
	auto refuse = [&](const char *why) {
		if (want_shard || want_passes > 1) {
			g_last_error = why;
			*err = true;
		}
		return false;
	};
	if (off || h->cfg.expected_kmers == 0) return refuse("a sharded / multi-pass WIDE handle needs expected_kmers > 0 (the record path)");
	if (h->size < (1ull << 26) || h->size >= (1ull << 34)) return refuse("the WIDE record path needs 2^26 <= table_slots < 2^34");
	if (want_shard && h->cfg.shard_index >= n_ranks) return refuse("shard_index >= shard_count");
	const uint64_t qmax = ~0ull / h->size;
	int qbits = 0;
	while (qbits < 64 && (qmax >> qbits)) qbits++;
	uint32_t r = 21;
	if (const char *e = DBGK_EXPERIMENT_ENV("DBGK_WIDE_R")) r = (uint32_t)std::max(kWRegionBits + 1, std::min(22, atoi(e)));
	if (!want_shard && want_passes <= 1)
		while (r < 22u && ((h->size + (1ull << r) - 1) >> r) > 1024ull) r++; // one pass if the fan-out allows it
	// passes: at least as many as keep the level-1 fan-out (ranks x buckets of a pass) within 1024
	auto passes_at = [&](uint32_t rr) {
		const uint64_t nn1 = (h->size + (1ull << rr) - 1) >> rr;
		const uint32_t BB = (uint32_t)((nn1 + n_ranks - 1) / n_ranks);
		uint32_t pp = std::max<uint32_t>(1u, want_passes);
		while ((uint64_t)n_ranks * ((BB + pp - 1) / pp) > 1024ull) pp++;
		return pp;
	};
	// every pass extracts the whole input again: beyond two passes the wider level-1 buckets of r = 22 (half the passes, level 2
	// fanning out 2048 ways) are the better trade
	if (!DBGK_EXPERIMENT_ENV("DBGK_WIDE_R") && r == 21u && passes_at(21u) > 2u && passes_at(22u) < passes_at(21u)) r = 22u;
	while (r > (uint32_t)kWRegionBits + 1u && qbits + (int)r + 6 > 64) r--;
	const uint64_t n1 = (h->size + (1ull << r) - 1) >> r;
	if ((1u << (r - kWRegionBits)) > 2048u || qbits + (int)r + 6 > 64 || n1 >= 65536ull) return refuse("no feasible wide record geometry for this table size");
	const uint32_t B = (uint32_t)((n1 + n_ranks - 1) / n_ranks);
	uint32_t n_passes = passes_at(r);
	if (n_passes > B) return refuse("shard_count too large for this table size");
	// Several passes are a PROTOCOL (begin_pass / push everything / end_pass, per pass): a caller who asked for neither shards nor
	// passes (n_passes == 0, the plain create / push / finalize flow) never gets it -- a table whose fan-out one pass cannot cover
	// is then built by the atomic kernels, as before the record path existed.  n_passes >= 1 says "I follow the protocol".
	if (n_passes > 1 && !want_shard && want_passes == 0) {
		// (not an error -- the handle works -- but a large performance step: say so where a caller can find it, once on stderr too)
		g_last_error = "WIDE handle: this table's level-1 fan-out needs several passes over the input and dbgk_config.n_passes is 0 -- "
		               "built by the atomic kernels; set n_passes (dbgk_wide_pass_info) for the record path";
		static std::atomic<bool> told{false};
		if (!told.exchange(true)) fprintf(stderr, "dbgk: %s\n", g_last_error.c_str());
		return false;
	}
	WPartGeom &G = h->wgeom;
	memset(&G, 0, sizeof G);
	G.size = h->size;
	G.magic = h->magic;
	if (h->size < (1ull << 32)) G.div = make_div32_magic((uint32_t)h->size);
	G.r = r;
	G.n1 = (uint32_t)n1;
	G.n2 = 1u << (r - kWRegionBits);
	G.n_regions = (uint32_t)((h->size + kWRegionSlots - 1) >> kWRegionBits);
	G.n_ranks = n_ranks;
	G.rank = want_shard ? h->cfg.shard_index : 0;
	G.B = B;
	G.bmagic = (uint32_t)(((1ull << 32) + B - 1) / B);
	G.b_lo = std::min(G.rank * B, G.n1);
	G.nb_own = std::min(B, G.n1 - G.b_lo);
	if (G.nb_own == 0) return refuse("shard_count too large for this table size");
	G.n_passes = n_passes;
	G.Bp = (B + n_passes - 1) / n_passes;
	G.pass = 0;
	G.pass_j0 = 0;
	G.n_l1 = n_ranks * G.Bp;
	G.slot_lo = (uint64_t)G.b_lo << r;
	G.slot_hi = std::min<uint64_t>(h->size, ((uint64_t)G.b_lo + G.nb_own) << r);
	// expected_kmers = occurrences THIS handle extracts (per pass over its input); a region receives the global density
	const double per_slot_mine = (double)h->cfg.expected_kmers / (double)h->size;
	G.cap1 = (uint64_t)(per_slot_mine * (double)(1ull << r) * 1.05) + 65536;
	G.cap2 = (uint64_t)(per_slot_mine * (double)n_ranks * (double)kWRegionSlots * 1.25) + 512;
	G.chunk_buckets = (G.Bp + 7u) / 8u;
	h->wmulti = want_shard || n_passes > 1;
	h->sharded = n_ranks > 1;
	h->shard_rank = G.rank;
	h->tslots = G.slot_hi - G.slot_lo;
	return true;
}

static int setup_wide_partition(dbgk_handle *h)
{
	const WPartGeom &G = h->wgeom;
	WPartStore &P = h->wstore;
	memset(&P, 0, sizeof P);
	P.ovf_cap = h->cfg.expected_kmers / 16 + (1ull << 20);
	P.spill_cap = (uint64_t)((h->tslots + kWRegionSlots - 1) >> kWRegionBits) * 8 + (1ull << 16);
	P.outgoing_cap = 1ull << 16;
	const size_t l1_bytes = (size_t)G.n_l1 * G.cap1 * 16, l2_bytes = (size_t)G.chunk_buckets * G.n2 * G.cap2 * 16;
	bool ok = hipMalloc(&P.l1, l1_bytes) == hipSuccess && hipMalloc(&P.l2, l2_bytes) == hipSuccess &&
	          hipMalloc(&P.cnt1, (size_t)G.n_l1 * 4) == hipSuccess && hipMalloc(&P.cnt2, (size_t)G.chunk_buckets * G.n2 * 4) == hipSuccess &&
	          hipMalloc(&P.ovf, P.ovf_cap * sizeof(dbgk_node32)) == hipSuccess && hipMalloc(&P.spill, P.spill_cap * sizeof(dbgk_node32)) == hipSuccess &&
	          hipMalloc(&P.ovf_n, 16) == hipSuccess && hipMalloc(&h->w_tile_prefix, ((size_t)G.n_l1 + 1) * 4) == hipSuccess &&
	          hipMalloc(&h->w_cursor, 4) == hipSuccess && hipMalloc(&P.outgoing, P.outgoing_cap * sizeof(dbgk_node32)) == hipSuccess &&
	          hipMalloc(&P.outgoing_n, 8) == hipSuccess && hipMalloc(&h->w_side_out, ((size_t)kWideSideSlots + 1) * sizeof(dbgk_node32)) == hipSuccess &&
	          hipMalloc(&h->w_side_n, 8) == hipSuccess;
	if (ok && h->sharded) ok = hipMalloc(&h->winbox, l1_bytes) == hipSuccess && hipMalloc(&h->winbox_cnt, (size_t)G.n_l1 * 4) == hipSuccess;
	if (!ok) {
		(void)hipGetLastError();
		g_last_error = "hipMalloc of the wide record stores failed";
		return DBGK_ERR_NOMEM;
	}
	P.inbox = h->sharded ? h->winbox : P.l1;
	P.inbox_cnt = h->sharded ? h->winbox_cnt : P.cnt1;
	HIPCHK(hipMemsetAsync(P.cnt1, 0, (size_t)G.n_l1 * 4, h->stream));
	HIPCHK(hipMemsetAsync(P.ovf_n, 0, 16, h->stream));
	HIPCHK(hipMemsetAsync(P.outgoing_n, 0, 8, h->stream));
	DBGK_LDS_ATTR((k_wide_scatter_l1<false, 0>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1<true, 0>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1<false, 1>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1<true, 1>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1<false, 2>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1<true, 2>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1_uniform<0>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1_uniform<1>), sizeof(WL1Lds));
	DBGK_LDS_ATTR((k_wide_scatter_l1_uniform<2>), sizeof(WL1Lds));
	DBGK_LDS_ATTR(k_wide_scatter_l2<1024>, sizeof(WL2Lds<1024>));
	DBGK_LDS_ATTR(k_wide_scatter_l2<2048>, sizeof(WL2Lds<2048>));
	DBGK_LDS_ATTR(k_wide_build_regions, sizeof(WBuildLds));
	h->store_capacity = h->cfg.expected_kmers;
	h->pending_kmers = 0;
	h->wpass_open = true; // pass 0 is open from the start
	return DBGK_OK;
}

// the main table of a WIDE handle that was reset without a memset (the region build rewrites every slot)
static int wide_ensure_zero(dbgk_handle *h)
{
	if (!h->wzero_pending) return DBGK_OK;
	HIPCHK(hipMemsetAsync(h->wnodes, 0, h->tslots * sizeof(WNode), h->stream));
	h->wzero_pending = false;
	return DBGK_OK;
}

// own-bucket indices of the current pass that exist on this rank (the last rank / pass may have fewer)
static uint32_t wide_pass_buckets(const dbgk_handle *h)
{
	const WPartGeom &G = h->wgeom;
	return G.pass_j0 >= G.nb_own ? 0u : std::min(G.Bp, G.nb_own - G.pass_j0);
}

static int wide_plan_pass(dbgk_handle *h)
{
	if (h->wplanned) return DBGK_OK;
	hipLaunchKernelGGL(k_wide_l2_plan, dim3(1), dim3(1024), 0, h->stream, h->wgeom, h->wstore.inbox_cnt, h->w_tile_prefix);
	HIPCHK(hipGetLastError());
	h->wplanned = true;
	h->wnext = 0;
	return DBGK_OK;
}

// level 2 + region build of the own-bucket indices [j0, j1) of the current pass, chunk by chunk of level-1 buckets
// through the (small) level-2 store
static int wide_build_range(dbgk_handle *h, uint32_t j0, uint32_t j1)
{
	const WPartGeom &G = h->wgeom;
	const WPartStore &P = h->wstore;
	TimedSpan sp;
	int rc = DBGK_OK;
	for (uint32_t c0 = j0; c0 < j1; c0 += G.chunk_buckets) {
		const uint32_t c1 = std::min(c0 + G.chunk_buckets, j1);
		rc = span_begin(h, PH_PARTITION, sp);
		if (rc) return rc;
		HIPCHK(hipMemsetAsync(P.cnt2, 0, (size_t)G.chunk_buckets * G.n2 * 4, h->stream));
		if (G.n2 > 1024u)
			hipLaunchKernelGGL(k_wide_scatter_l2<2048>, dim3(h->n_cu & ~7), dim3(kWL2Threads), sizeof(WL2Lds<2048>), h->stream, G, P, h->w_tile_prefix, h->d_ctr, c0, c1);
		else
			hipLaunchKernelGGL(k_wide_scatter_l2<1024>, dim3((h->n_cu * 2) & ~7), dim3(kWL2Threads), sizeof(WL2Lds<1024>), h->stream, G, P, h->w_tile_prefix, h->d_ctr, c0, c1);
		HIPCHK(hipGetLastError());
		rc = span_end(h, sp);
		if (rc) return rc;
		const uint32_t n_regions = (c1 - c0) * G.n2;
		rc = span_begin(h, PH_BUILD, sp);
		if (rc) return rc;
		HIPCHK(hipMemsetAsync(h->w_cursor, 0, 4, h->stream));
		hipLaunchKernelGGL(k_wide_build_regions, dim3(std::min<uint32_t>(n_regions, (uint32_t)h->n_cu * 3u)), dim3(kWBuildThreads), sizeof(WBuildLds), h->stream, G, P,
		                   h->wnodes, h->d_ctr, h->w_cursor, c0, n_regions);
		HIPCHK(hipGetLastError());
		rc = span_end(h, sp);
		if (rc) return rc;
	}
	h->wnext = j1;
	return DBGK_OK;
}

// the rest of the current pass: whatever the caller has not built by ranges
static int wide_end_pass(dbgk_handle *h)
{
	if (!h->wpass_open) return DBGK_OK;
	if (h->sharded && !h->exchanged) {
		g_last_error = "sharded WIDE handle: exchange the level-1 buckets of this pass (dbgk_shard_buffers) and call dbgk_shard_mark_exchanged first";
		return DBGK_ERR_STATE;
	}
	int rc = wide_plan_pass(h);
	if (rc) return rc;
	const uint32_t nb = wide_pass_buckets(h);
	if (h->wnext < nb) {
		rc = wide_build_range(h, h->wnext, nb);
		if (rc) return rc;
	}
	if (h->wgeom.pass > 0) { // the input was read again: its totals were counted in pass 0
		HIPCHK(hipMemcpyAsync(&h->d_ctr->total_kmers, h->wsaved_totals, 16, hipMemcpyHostToDevice, h->stream));
		HIPCHK(hipMemcpyAsync(&h->d_ctr->other_bytes, &h->wsaved_other, 8, hipMemcpyHostToDevice, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		h->total_reads = h->wsaved_reads;
		h->host_other_bytes = h->wsaved_host_other;
	}
	h->wpass_open = false;
	h->wpasses_done = h->wgeom.pass + 1;
	h->pending_kmers = 0;
	return DBGK_OK;
}

static int wide_begin_pass(dbgk_handle *h, uint32_t p)
{
	WPartGeom &G = h->wgeom;
	if (h->wpass_open || p != h->wpasses_done || p >= G.n_passes) {
		g_last_error = "dbgk_wide_begin_pass: passes run in order, each ended (dbgk_wide_end_pass) before the next begins";
		return DBGK_ERR_STATE;
	}
	G.pass = p;
	G.pass_j0 = p * G.Bp;
	HIPCHK(hipMemsetAsync(h->wstore.cnt1, 0, (size_t)G.n_l1 * 4, h->stream));
	if (p > 0) {
		HIPCHK(hipMemcpyAsync(h->wsaved_totals, &h->d_ctr->total_kmers, 16, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(&h->wsaved_other, &h->d_ctr->other_bytes, 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		h->wsaved_reads = h->total_reads;
		h->wsaved_host_other = h->host_other_bytes;
	}
	h->wpass_open = true;
	h->wplanned = false;
	h->exchanged = false;
	h->wnext = 0;
	h->pending_kmers = 0;
	return DBGK_OK;
}

// after the last pass: region spill-over nodes and bucket-overflow observations through the atomic kernels
static int wide_finish_records(dbgk_handle *h)
{
	const WPartGeom &G = h->wgeom;
	const WPartStore &P = h->wstore;
	TimedSpan sp;
	h->wzero_pending = false; // every slot has just been written
	int rc = span_begin(h, PH_FIXUP, sp);
	if (rc) return rc;
	if (h->wmulti && G.n_ranks > 1) {
		// spill nodes stay in the shard unless they run off its end (-> outgoing); overflow observations may belong to any
		// shard: the caller exchanges them (dbgk_shard_overflow)
		hipLaunchKernelGGL(k_wide_merge_sharded, dim3(h->n_cu), dim3(kBlock), 0, h->stream, P.spill, &P.ovf_n[1], (uint64_t)0, P.spill_cap, 0, 0, G, P, h->wnodes,
		                   h->d_ctr);
	} else {
		hipLaunchKernelGGL(k_wide_merge_spill, dim3(h->n_cu), dim3(kBlock), 0, h->stream, P.spill, &P.ovf_n[1], P.spill_cap, h->wref(), h->d_ctr);
		hipLaunchKernelGGL(k_wide_insert_obs, dim3(h->n_cu), dim3(kBlock), 0, h->stream, P.ovf, &P.ovf_n[0], P.ovf_cap, h->wref(), h->d_ctr);
	}
	HIPCHK(hipGetLastError());
	rc = span_end(h, sp);
	if (rc) return rc;
	h->wbuilt = true;
	h->pending_kmers = 0;
	return DBGK_OK;
}

// records -> table (one rank, one pass).  Afterwards the handle is an ordinary WIDE handle: whatever is pushed later goes
// through the atomic kernels onto the table built here.
static int wide_build_from_records(dbgk_handle *h)
{
	if (!h->wpart || h->wbuilt) return DBGK_OK;
	if (h->wmulti) {
		g_last_error = "the record store of a sharded / multi-pass WIDE handle is full: expected_kmers too small (or more passes needed)";
		return DBGK_ERR_CAPACITY;
	}
	int rc = wide_end_pass(h);
	if (rc) return rc;
	return wide_finish_records(h);
}

static int reset_state(dbgk_handle *h)
{
	bool counters_done = false;
	// (direct-block form: the build writes every block of the table, also the empty ones -- nothing to zero)
	if (h->kfreq && !(h->part && h->kf_blocks)) HIPCHK(hipMemsetAsync(h->counts, 0, h->n_counts, h->stream));
	if (h->wide) {
		if (h->wpart) { // the region build overwrites every slot of the main table: zero it only if something writes it before
			h->wzero_pending = true;
			h->wbuilt = false;
			h->pending_kmers = 0;
			h->wgeom.pass = 0;
			h->wgeom.pass_j0 = 0;
			h->wpass_open = true;
			h->wplanned = false;
			h->wnext = 0;
			h->wpasses_done = 0;
			h->exchanged = false;
			HIPCHK(hipMemsetAsync(h->wstore.cnt1, 0, (size_t)h->wgeom.n_l1 * 4, h->stream));
			HIPCHK(hipMemsetAsync(h->wstore.ovf_n, 0, 16, h->stream));
			HIPCHK(hipMemsetAsync(h->wstore.outgoing_n, 0, 8, h->stream));
		} else {
			HIPCHK(hipMemsetAsync(h->wnodes, 0, h->tslots * sizeof(WNode), h->stream));
		}
		HIPCHK(hipMemsetAsync(h->wside, 0, kWideSideSlots * sizeof(WNode), h->stream));
	} else if (h->kfreq && !h->part) {
	} else if (h->part) {
		// the region build of finalize overwrites every slot, so the 16 B/slot memset is only needed
		// if a direct-path write (merge) happens first
		h->zero_pending = true;
		h->incr = false;
		int rc = clear_record_store(h, true); // (one launch: the stores' control arrays and the counters)
		if (rc) return rc;
		counters_done = true;
	} else {
		int rc = zero_table_now(h);
		if (rc) return rc;
	}
	if (h->track) HIPCHK(hipMemsetAsync(h->first_pos, 0xFF, h->tslots * 8, h->stream));
	h->pos_base = 0;
	if (!counters_done) {
		HIPCHK(hipMemsetAsync(h->d_ctr, 0, sizeof(Counters), h->stream));
		HIPCHK(hipMemsetAsync(&h->d_ctr->polyA_slot, 0xFF, sizeof(unsigned long long), h->stream));
	}
	h->finalized = false;
	h->total_reads = 0;
	h->host_other_bytes = 0;
	return DBGK_OK;
}

// Decide whether the PARTITION engine is used and allocate its record stores.
//   geometry: level-1 bucket = slot >> r, r >= 20 chosen so that n1 = ceil(size / 2^r) <= 1024;
//   final bucket = slot >> 12 (one 4096-slot region); n2 = 2^(r-12) sub-buckets per level-1 bucket.
//   An 8-byte record must hold q = hash / size, r slot bits and 6 neighbour bits.
static int plan_partition(dbgk_handle *h)
{
	h->part = false;
	h->sharded = false;
	memset(&h->store, 0, sizeof h->store);
	memset(&h->geom, 0, sizeof h->geom);
	h->tslots = h->size;
	if (h->kfreq && h->kf_blocks) { // direct blocks: `size` = 4^k, slot = the key with its block index permuted (kf_slot_of_key)
		PartGeom &G = h->geom;
		const uint32_t bits = 2u * (uint32_t)h->cfg.kmer_size; // >= 26
		// level-1 bucket = slot >> r: 256 buckets where the table allows (the wave-per-bucket level-1 kernel), level 2 then
		// splits a bucket into its 2^(r - 16) <= 1024 blocks in one pass
		const uint32_t r = std::max(20u, std::min(26u, bits - 8u));
		G.size = h->size;
		G.magic = h->magic;
		if (h->size < (1ull << 32)) G.div = make_div32_magic((uint32_t)h->size);
		G.r = r;
		G.n1 = (uint32_t)(h->size >> r);
		G.n2 = 1u << (r - kKfBlockBits);
		G.n_final = (uint32_t)(h->size >> kKfBlockBits);
		G.n_ranks = 1;
		G.rank = 0;
		G.B = G.n1;
		G.n_sub = kSubStores;
		G.b_lo = 0;
		G.nb_own = G.n1;
		G.slot_lo = 0;
		G.slot_hi = h->size;
		G.n_regions_own = G.n_final;
		const double per_slot = (double)h->cfg.expected_kmers / (double)h->size;
		// a level-1 bucket sums 2^(r - 16) blocks of very different weight (canonical k-mers favour small key values): more slack
		// than the hashed form's 5 %; a single block may hold 2.2 times the average
		G.cap1 = (uint64_t)(per_slot * (double)(1ull << r) * 1.2 / (double)G.n_sub) + 65536 / G.n_sub + 8192;
		G.cap2 = ((uint64_t)(per_slot * (double)(1ull << kKfBlockBits) * 2.6) + 1024 + 3) & ~3ull; // (16-bit records, read four at a time)
		G.r_rec = r;
		G.l2_shift = kKfBlockBits - (uint32_t)kRegionBits; // level 2 splits by block, not by 4096-slot region
		G.kf = 2u;
		G.kf_mask = (uint32_t)((1ull << (bits - kKfBlockBits)) - 1ull);
		if (G.n1 > (uint32_t)kL1MaxB || G.n2 > (uint32_t)kMaxBuckets) return DBGK_OK; // (cannot happen for 13 <= k <= 18)
		h->shard_rank = 0;
		h->part = true;
		h->three = false;
		return DBGK_OK;
	}
	const uint32_t n_ranks = h->cfg.shard_count > 1 ? h->cfg.shard_count : 1;
	const int want = h->cfg.engine == DBGK_ENGINE_SEEDIDX ? DBGK_ENGINE_DIRECT // the seed index uses the plain table
	                 : h->cfg.engine == DBGK_ENGINE_KFREQ ? DBGK_ENGINE_AUTO   // KFREQ: only if the geometry is feasible
	                                                      : h->cfg.engine;
	const bool want_shard = h->cfg.shard_count >= 1; // shard_count == 1: one-rank sharded handle (same protocol, for testing)
	if (want_shard && h->cfg.shard_index >= n_ranks) return DBGK_ERR_ARG;
	if (!want_shard) {
		if (want == DBGK_ENGINE_DIRECT) return DBGK_OK;
		if (want == DBGK_ENGINE_AUTO && h->cfg.expected_kmers == 0) return DBGK_OK; // streaming use: total unknown
	}
	uint32_t r = 22; // measured on cfg2 (round 2, profiles/r02_r_sweep.txt): r = 20 / 21 / 22 -> 17.3 / 16.5 / 16.3 ms per step
	if (const char *e = DBGK_EXPERIMENT_ENV("DBGK_PART_R")) r = (uint32_t)std::max(20, std::min(24, atoi(e))); // tuning knob: level-1 bucket = slot >> r
	while (((h->size + (1ull << r) - 1) >> r) > (uint64_t)kL1MaxB) r++;
	const uint64_t qmax = ~0ull / h->size;
	int qbits = 0;
	while (qbits < 64 && (qmax >> qbits)) qbits++;
	while (r > 20 && qbits + (int)r + 6 > 64 && ((h->size + (1ull << (r - 1)) - 1) >> (r - 1)) <= (uint64_t)kL1MaxB) r--; // small tables: q needs the bits
	// level 1 fans out to <= 1024 buckets, level 2 to 2^(r-12) <= 4096 final buckets per level-1 bucket: 2^34 slots
	const bool feasible = (1u << (r - kRegionBits)) <= (uint32_t)kMaxBucketsL2 && (qbits + (int)r + 6) <= 64 && h->size >= (1ull << 26) &&
	                      h->size < (1ull << 34);
	if (!feasible || (want_shard && want == DBGK_ENGINE_DIRECT)) {
		if (want == DBGK_ENGINE_PARTITION || want_shard) {
			g_last_error = "PARTITION engine (and any sharded handle) needs 2^26 <= table_slots < 2^34";
			return DBGK_ERR_ARG;
		}
		return DBGK_OK;
	}
	for (uint64_t x : {0ull, 1ull, 0x0123456789ABCDEFull, ~0ull, 488296166657017542ull}) {
		if (hash_code_inverse(hash_code(x)) != x) {
			g_last_error = "hash_code_inverse self-check failed";
			return DBGK_ERR_STATE;
		}
	}
	PartGeom &G = h->geom;
	G.size = h->size;
	G.magic = h->magic;
	if (h->size < (1ull << 32)) G.div = make_div32_magic((uint32_t)h->size);
	G.r = r;
	G.n1 = (uint32_t)((h->size + (1ull << r) - 1) >> r);
	G.n2 = 1u << (r - kRegionBits);
	G.n_final = (uint32_t)((h->size + kRegionSlots - 1) >> kRegionBits);
	G.n_ranks = n_ranks;
	G.rank = want_shard ? h->cfg.shard_index : 0;
	h->shard_rank = G.rank;
	G.B = (G.n1 + n_ranks - 1) / n_ranks;
	G.n_sub = kSubStores;
	G.b_lo = std::min(G.rank * G.B, G.n1);
	G.nb_own = std::min(G.B, G.n1 - G.b_lo);
	G.slot_lo = (uint64_t)G.b_lo << r;
	G.slot_hi = std::min<uint64_t>(h->size, ((uint64_t)G.b_lo + G.nb_own) << r);
	G.n_regions_own = (uint32_t)((G.slot_hi - G.slot_lo + kRegionSlots - 1) >> kRegionBits);
	if (G.nb_own == 0 || (uint64_t)n_ranks * G.B * G.n_sub > (uint64_t)kMaxInboxEntries) {
		g_last_error = "shard_count too large for this table size";
		return DBGK_ERR_ARG;
	}
	// expected_kmers = occurrences THIS handle extracts; a region receives the global density
	const uint64_t expected = h->cfg.expected_kmers ? h->cfg.expected_kmers : h->size * 2 / n_ranks;
	const double per_slot = (double)expected / (double)h->size;
	G.cap1 = (uint64_t)(per_slot * (double)(1ull << r) * 1.05 / (double)G.n_sub) + 65536 / G.n_sub + (G.n_sub > 1 ? 8192 : 0); // per sub-store
	G.cap2 = (uint64_t)(per_slot * (double)n_ranks * (double)kRegionSlots * 1.15) + 512;
	G.cap1 = (G.cap1 + 15u) & ~15ull; // every bucket starts on a 128-byte line: the 16-byte record loads of level 2 and of the build are aligned
	G.cap2 = (G.cap2 + 15u) & ~15ull;
	G.r_rec = r;
	G.l2_shift = 0;
	h->tslots = G.slot_hi - G.slot_lo;
	h->sharded = want_shard;
	h->part = true;
	static const bool no_three = DBGK_EXPERIMENT_ENV("DBGK_THREE_LEVEL") && atoi(DBGK_EXPERIMENT_ENV("DBGK_THREE_LEVEL")) == 0; // measurements
	// measured with cfg2's 1.2 G records, level 2 alone: n2 = 4096 (8.6 G slots) 17.6 ms in one pass, 9.7 ms in two;
	// n2 = 2048 (5 G slots) 6.9 ms in one pass, 9.5 in two -- so only the 4096-way fan-out is split
	h->three = G.n2 > 2048u && !no_three;
	if (h->three) {
		h->fan_mid = G.n2 / 64u; // 64
		uint32_t lg = 0;
		while ((1u << lg) < h->fan_mid) lg++;
		const uint64_t cap_mid = ((uint64_t)(per_slot * (double)n_ranks * (double)(1ull << (r - lg)) * 1.08) + 8192 + 15u) & ~15ull;
		h->g_mid = G;
		h->g_mid.n2 = h->fan_mid;
		h->g_mid.l2_shift = 6;      // the low 6 bits of the final-bucket index are left to the final pass
		h->g_mid.cap2 = cap_mid;
		h->g_fin = G;
		h->g_fin.n_ranks = 1;       // its input is this handle's own mid store
		h->g_fin.rank = 0;
		h->g_fin.n_sub = 1;
		h->g_fin.r = r - lg;
		h->g_fin.n1 = G.n1 * h->fan_mid;
		h->g_fin.B = G.nb_own * h->fan_mid;
		h->g_fin.b_lo = G.b_lo * h->fan_mid;
		h->g_fin.nb_own = G.nb_own * h->fan_mid;
		h->g_fin.n2 = 64;
		h->g_fin.cap1 = cap_mid;
	}
	return DBGK_OK;
}

// allocate the record stores of the PARTITION engine (geometry already planned)
static int setup_partition(dbgk_handle *h)
{
	if (!h->part) return DBGK_OK;
	const PartGeom &G = h->geom;
	PartStore &P = h->store;
	const uint64_t expected = h->cfg.expected_kmers ? h->cfg.expected_kmers : h->size * 2 / G.n_ranks;
	// records that find their bucket full are kept as {key, lb, rb} triples and inserted through the global
	// path after the build: a key that occurs more often than a final bucket holds (cap2, ~1.15x the mean
	// bucket fill) sends its surplus here, so this bounds the share of occurrences that may belong to such
	// heavy hitters (high-copy repeats): 1/16 of the input + 1 M; beyond it finalize returns DBGK_ERR_CAPACITY
	P.ovf_cap = expected / 16 + (1ull << 20);
	P.spill_cap = (uint64_t)G.n_regions_own * 8 + (1ull << 16);
	P.outgoing_cap = 1ull << 16;
	const size_t n_entries = (size_t)G.n_ranks * G.B * G.n_sub;
	const size_t l1_bytes = n_entries * G.cap1 * 8, l2_bytes = (size_t)G.nb_own * G.n2 * G.cap2 * (G.kf == 2u ? 2 : 8); // (direct blocks: 16-bit records)
	// (+ 64 bytes: level 2 and the build load their records in pairs, and the second half of a bucket's last pair may lie behind the bucket)
	bool ok = hipMalloc(&P.l1, l1_bytes + 64) == hipSuccess && hipMalloc(&P.l2, l2_bytes + 64) == hipSuccess &&
	          hipMalloc(&P.cnt1, n_entries * 4) == hipSuccess && hipMalloc(&P.cnt2, (size_t)G.nb_own * G.n2 * 4) == hipSuccess &&
	          hipMalloc(&P.ovf, P.ovf_cap * sizeof(Node)) == hipSuccess && hipMalloc(&P.spill, P.spill_cap * sizeof(Node)) == hipSuccess &&
	          hipMalloc(&P.ovf_n, 16) == hipSuccess && hipMalloc(&h->tile_prefix, (n_entries + 1) * 4) == hipSuccess &&
	          hipMalloc(&h->region_cursor, ((size_t)kMaxBuildLaunches + 2 + G.n_regions_own) * sizeof(unsigned int)) == hipSuccess && // + redo cursor, count, list

	          hipMalloc(&P.outgoing, P.outgoing_cap * sizeof(Node)) == hipSuccess && hipMalloc(&P.outgoing_n, 8) == hipSuccess;
	if (ok && h->sharded)
		ok = hipMalloc(&h->inbox, l1_bytes + 64) == hipSuccess && hipMalloc(&h->inbox_cnt, n_entries * 4) == hipSuccess;
	if (!ok) {
		g_last_error = "hipMalloc of the PARTITION record stores failed";
		return DBGK_ERR_NOMEM;
	}
	// side table for the surplus of heavy hitters; on a sharded handle it may hold keys of any shard and is
	// offered to every rank after the build (dbgk_shard_heavy), like the overflow list
	if (hipMalloc(&P.hh, kHeavyHitterSlots * sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
	HIPCHK(hipMemsetAsync(P.hh, 0, kHeavyHitterSlots * sizeof(Node), h->stream)); // once: afterwards it is zeroed again only when it was used (clear_record_store)
	P.hh_size = kHeavyHitterSlots;
	P.hh_magic = make_mod_magic(kHeavyHitterSlots);
	P.inbox = h->sharded ? h->inbox : P.l1;
	P.inbox_cnt = h->sharded ? h->inbox_cnt : P.cnt1;
	// EARLY level 2 (early_l2): a handle that extracts into its own inbox scatters what earlier batches stored while the next
	// batch is on the link.  Not for shards (their inbox is filled by the exchange), the three-level form, 32-bit KFREQ records
	const bool no_early = dbgk_hook("early_l2") && atoi(dbgk_hook("early_l2")) == 0; // (read per handle: measurements, and the tests compare the two)
	P.l2_done = P.l2_upto = nullptr;
	if (!h->sharded && !h->three && G.kf != 2u && !no_early) {
		if (hipMalloc(&h->l2_done, 2 * n_entries * 4) != hipSuccess) return DBGK_ERR_NOMEM; // done[] and upto[]
		HIPCHK(hipMemsetAsync(h->l2_done, 0, 2 * n_entries * 4, h->stream));
		P.l2_done = h->l2_done;
		P.l2_upto = h->l2_done + n_entries;
	}
	h->l2_seen_kmers = 0;
	h->store_capacity = expected;
	if (h->three) {
		const size_t n_mid = (size_t)G.nb_own * h->fan_mid;
		if (hipMalloc(&h->mid, n_mid * h->g_mid.cap2 * 8 + 64) != hipSuccess || hipMalloc(&h->cnt_mid, n_mid * 4) != hipSuccess ||
		    hipMalloc(&h->tile_prefix2, (n_mid + 1) * 4) != hipSuccess) {
			g_last_error = "hipMalloc of the mid-level record store failed";
			return DBGK_ERR_NOMEM;
		}
		h->s_mid = P;           // reads the inbox like level 2, writes the mid store
		h->s_mid.l2 = h->mid;
		h->s_mid.cnt2 = h->cnt_mid;
		h->s_fin = P;           // reads the mid store, writes the final buckets where level 2 would
		h->s_fin.inbox = h->mid;
		h->s_fin.inbox_cnt = h->cnt_mid;
	}
#define DBGK_UNIFORM_ATTRS(W)                                                  \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 16, false>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 16, true>), sizeof(UniformLds));  \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 15, false>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 15, true>), sizeof(UniformLds));  \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 16, false, false, true>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 15, false, false, true>), sizeof(UniformLds)); \
	DBGK_LDS_ATTR((k_extract_scatter<true, 0, W>), sizeof(ScatterLds));              \
	DBGK_LDS_ATTR((k_extract_scatter<false, 0, W>), sizeof(ScatterLds))
#ifdef DBGK_EXPERIMENTS // timing experiments (DBGK_DEBUG_MODE, results are wrong): not in the product library
	DBGK_LDS_ATTR((k_extract_scatter_uniform<1, 0, 15, false>), sizeof(UniformLds));
	DBGK_LDS_ATTR((k_extract_scatter_uniform<2, 0, 15, false>), sizeof(UniformLds));
	DBGK_LDS_ATTR((k_extract_scatter_uniform<3, 0, 15, false>), sizeof(UniformLds));
	DBGK_LDS_ATTR((k_extract_scatter<false, 1>), sizeof(ScatterLds));
	DBGK_LDS_ATTR((k_extract_scatter<false, 2>), sizeof(ScatterLds));
	DBGK_LDS_ATTR((k_extract_scatter<false, 3>), sizeof(ScatterLds));
	DBGK_LDS_ATTR((k_scatter_l2<1>), sizeof(ScatterLdsL2));
	DBGK_LDS_ATTR((k_scatter_l2<2>), sizeof(ScatterLdsL2));
	DBGK_LDS_ATTR((k_scatter_l2<3>), sizeof(ScatterLdsL2));
#endif
	DBGK_UNIFORM_ATTRS(0);
	DBGK_UNIFORM_ATTRS(1);
	DBGK_UNIFORM_ATTRS(2);
#define DBGK_LIN_ATTRS(W)                                                                          \
	DBGK_LDS_ATTR((k_extract_scatter_lin<false, W>), sizeof(ScatterLdsLin<8>));                   \
	DBGK_LDS_ATTR((k_extract_scatter_lin<true, W>), sizeof(ScatterLdsLin<8>));                    \
	DBGK_LIN_ATTRS_U(W)
#define DBGK_LIN_ATTRS_U(W)                                                                          \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 8, false, true>), sizeof(UniformLdsLin<8>));   \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 8, true, true>), sizeof(UniformLdsLin<8>));    \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 12, false, true>), sizeof(UniformLdsLin<12>)); \
	DBGK_LDS_ATTR((k_extract_scatter_uniform<0, W, 12, true, true>), sizeof(UniformLdsLin<12>))
	DBGK_LIN_ATTRS(0);
	DBGK_LIN_ATTRS(1);
	DBGK_LIN_ATTRS(2);
#undef DBGK_LIN_ATTRS
#undef DBGK_LIN_ATTRS_U
#undef DBGK_UNIFORM_ATTRS
	DBGK_LDS_ATTR((k_scatter_l2<0>), sizeof(ScatterLdsL2));
	DBGK_LDS_ATTR((k_scatter_l2<0, kMaxBuckets, true>), sizeof(ScatterLdsL2));
	DBGK_LDS_ATTR((k_scatter_l2<0, 2048>), sizeof(ScatterLdsL2T<2048>));
	DBGK_LDS_ATTR((k_scatter_l2<0, 4096>), sizeof(ScatterLdsL2T<4096>));
#define DBGK_BUILD_ATTR(...) DBGK_LDS_ATTR((k_build_regions<__VA_ARGS__>), sizeof(BuildLds))
	DBGK_BUILD_ATTR(0, false, false, false); DBGK_BUILD_ATTR(0, true, false, false); DBGK_BUILD_ATTR(0, false, true, false); DBGK_BUILD_ATTR(0, true, true, false);
	DBGK_BUILD_ATTR(0, false, false, true);  DBGK_BUILD_ATTR(0, true, false, true);  DBGK_BUILD_ATTR(0, false, true, true);  DBGK_BUILD_ATTR(0, true, true, true);
	DBGK_BUILD_ATTR(0, false, false, false, true); DBGK_BUILD_ATTR(0, true, false, false, true); DBGK_BUILD_ATTR(0, false, true, false, true);
	DBGK_BUILD_ATTR(0, true, true, false, true);
#ifdef DBGK_EXPERIMENTS
	DBGK_BUILD_ATTR(1, false, false, false); DBGK_BUILD_ATTR(2, false, false, false); DBGK_BUILD_ATTR(3, false, false, false);
	DBGK_BUILD_ATTR(1, false, false, true);  DBGK_BUILD_ATTR(2, false, false, true);  DBGK_BUILD_ATTR(3, false, false, true);
#endif
#undef DBGK_BUILD_ATTR
	DBGK_LDS_ATTR((k_kf_build_blocks<false, true>), sizeof(KfBlockLds));
	DBGK_LDS_ATTR((k_kf_build_blocks<true, true>), sizeof(KfBlockLds));
	DBGK_LDS_ATTR((k_kf_build_blocks<false, false>), sizeof(KfBlockLds));
	DBGK_LDS_ATTR((k_kf_build_blocks<true, false>), sizeof(KfBlockLds));
	DBGK_LDS_ATTR((k_kf_build_blocks<false, false, true>), sizeof(KfBlockLds));
	DBGK_LDS_ATTR((k_kf_build_blocks<true, false, true>), sizeof(KfBlockLds));
	return DBGK_OK;
}

static int ensure_slot(dbgk_handle *h, StageSlot &s);

extern "C" int dbgk_create(const dbgk_config *cfg, dbgk_handle **out)
{
	if (!cfg || !out) return DBGK_ERR_ARG;
	*out = nullptr;
	const bool wide = cfg->engine == DBGK_ENGINE_WIDE;
	if (cfg->kmer_size < 1 || cfg->kmer_size > (wide ? 63 : 32)) return DBGK_ERR_ARG; // 64-bit keys: the reference's "max 31" (+32, main.cpp:100); WIDE: 128-bit keys
	if (wide && (cfg->flags & ~DBGK_FLAG_PREALLOC_STAGING)) return DBGK_ERR_ARG;
	if (cfg->max_read_len < cfg->kmer_size) return DBGK_ERR_ARG;
	const bool kfreq = cfg->engine == DBGK_ENGINE_KFREQ;
	if (kfreq && cfg->kmer_size > 18) return DBGK_ERR_ARG; // 4^18 bytes = 64 GiB
	if (!kfreq && cfg->table_slots < 3) return DBGK_ERR_ARG;
	const bool seed = cfg->engine == DBGK_ENGINE_SEEDIDX;
	if (cfg->engine != DBGK_ENGINE_AUTO && cfg->engine != DBGK_ENGINE_DIRECT && cfg->engine != DBGK_ENGINE_PARTITION && !kfreq && !seed && !wide)
		return DBGK_ERR_ARG;
	if (seed && (cfg->shard_count || (cfg->flags & DBGK_FLAG_TRACK_FIRST_SEEN))) return DBGK_ERR_ARG;

	// DBGK_TIMINGS: where the time of creating a handle goes (stderr, one line)
	static const bool lap_wanted = getenv("DBGK_TIMINGS") != nullptr;
	double laps[6] = {0, 0, 0, 0, 0, 0};
	auto clock_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double lap_t = clock_s();
	auto lap = [&](int i) { const double t = clock_s(); laps[i] += t - lap_t; lap_t = t; };
	int n_dev = 0;
	hipError_t e = hipGetDeviceCount(&n_dev);
	if (e != hipSuccess || n_dev <= 0) {
		g_last_error = "no HIP device visible (this library has no CPU fallback)";
		return DBGK_ERR_HIP;
	}
	if (cfg->device_id < 0 || cfg->device_id >= n_dev) return DBGK_ERR_ARG;
	lap(0);

	dbgk_handle *h = new (std::nothrow) dbgk_handle();
	if (!h) return DBGK_ERR_NOMEM;
	h->cfg = *cfg;
	h->device = cfg->device_id;
	h->kfreq = kfreq;
	h->seed = seed;
	h->wide = wide;
	if (seed) h->cfg.max_read_len = 0x7FFFFFFF; // contigs are never trimmed (the pos field bounds them, see push)
	h->size = kfreq ? 3 : cfg->table_slots;
	// KFREQ with a known input size runs through the PARTITION engine: occurrences are partitioned by
	// hash_code(key) % size like graph records and aggregated per key in LDS; `size` is only the modulus
	// (no node table exists), chosen so that the LDS regions stay about half empty even if every second
	// occurrence were a new key.  Without expected_kmers: direct atomics on the byte table.
	static const bool kf_direct = DBGK_EXPERIMENT_ENV("DBGK_KFREQ_DIRECT") != nullptr;
	const bool kf_part = kfreq && cfg->expected_kmers > 0 && cfg->shard_count == 0 && !kf_direct;
	// k >= 13 (a table of 2^26 bytes and more): the direct-block form -- regions ARE 64-KiB blocks of the table
	// (dbgk_partition.h, kf_slot_of_key); smaller k (or DBGK_KFREQ_HASHED=1, measurements): the hashed form
	const bool kf_hashed = dbgk_hook("kfreq_hashed") && atoi(dbgk_hook("kfreq_hashed"));
	h->kf_blocks = kf_part && cfg->kmer_size >= 13 && !kf_hashed;
	if (h->kf_blocks) {
		h->size = 1ull << (2 * cfg->kmer_size);
	} else if (kf_part) {
		const uint64_t want = std::max<uint64_t>(1ull << 26, cfg->expected_kmers / 2);
		h->size = std::min<uint64_t>(want, (1ull << 32) - (1ull << 23)) | 1ull;
	}
	h->magic = make_mod_magic(h->size);
	h->tslots = h->size;
	if ((!kfreq || kf_part) && !wide) {
		const int prc = plan_partition(h); // geometry first: a sharded handle holds only its slot range
		if (prc != DBGK_OK) {
			delete h;
			return prc;
		}
		if (kfreq) {
			if (!h->part) { // infeasible geometry: fall back to the direct table
				h->size = 3;
				h->magic = make_mod_magic(h->size);
			}
			if (!h->part) h->kf_blocks = false;
			h->geom.kf = h->part ? (h->kf_blocks ? 2u : 1u) : 0u;
			h->tslots = 0;
		}
	}

	auto fail = [&](int rc) {
		free_handle(h);
		return rc;
	};
	if (hipSetDevice(h->device) != hipSuccess) return fail(DBGK_ERR_HIP);
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) return fail(DBGK_ERR_HIP);
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		g_last_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
		return fail(DBGK_ERR_HIP);
	}
	h->n_cu = prop.multiProcessorCount;
	h->grid = h->n_cu * 8; // 8 x 256-thread blocks per CU = 32 waves/CU, the residency limit

	if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(DBGK_ERR_HIP);
	lap(1);
	if (kfreq) {
		h->n_counts = 1ull << (2 * cfg->kmer_size);
		if (h->n_counts < 64) h->n_counts = 64; // whole dwords / 8-byte groups for the scan kernels (k < 3)
		if (hipMalloc(&h->counts, h->n_counts) != hipSuccess) {
			g_last_error = "hipMalloc of the 4^k count table failed";
			return fail(DBGK_ERR_NOMEM);
		}
	} else if (wide) {
		bool werr = false;
		h->wpart = plan_wide_partition(h, &werr); // input size known: records first, the table region by region (a shard: its slot range only)
		if (werr) return fail(DBGK_ERR_ARG);
		if (hipMalloc(&h->wnodes, h->tslots * sizeof(WNode)) != hipSuccess || hipMalloc(&h->wside, kWideSideSlots * sizeof(WNode)) != hipSuccess) {
			g_last_error = "hipMalloc of the wide k-mer table failed";
			return fail(DBGK_ERR_NOMEM);
		}
		if (h->wpart) {
			const int wrc = setup_wide_partition(h);
			if (wrc != DBGK_OK) return fail(wrc);
		}
	} else if (hipMalloc(&h->table, h->tslots * sizeof(Node)) != hipSuccess) {
		g_last_error = "hipMalloc of the k-mer table failed";
		return fail(DBGK_ERR_NOMEM);
	}
	if (cfg->flags & DBGK_FLAG_TRACK_FIRST_SEEN) {
		if (h->part || h->kfreq) {
			g_last_error = "DBGK_FLAG_TRACK_FIRST_SEEN needs the DIRECT engine";
			return fail(DBGK_ERR_ARG);
		}
		if (hipMalloc(&h->first_pos, h->tslots * 8) != hipSuccess) return fail(DBGK_ERR_NOMEM);
		h->track = true;
	}
	if (hipMalloc(&h->d_ctr, sizeof(Counters)) != hipSuccess) return fail(DBGK_ERR_NOMEM);
	if (hipHostMalloc(&h->h_ctr, sizeof(Counters), hipHostMallocDefault) != hipSuccess) return fail(DBGK_ERR_NOMEM);
	h->cap_bases = cfg->max_batch_bases ? cfg->max_batch_bases : (256ull << 20);
	h->cap_reads = h->cap_bases / 16 + 1024;
	lap(2);
	{
		const int prc = setup_partition(h);
		if (prc != DBGK_OK) return fail(prc);
	}
	lap(3);
	int rc = reset_state(h);
	if (rc != DBGK_OK) return fail(rc);
	if (lap_wanted && hipStreamSynchronize(h->stream) != hipSuccess) return fail(DBGK_ERR_HIP);
	lap(4);
	if (cfg->flags & DBGK_FLAG_PREALLOC_STAGING) { // page-locking is most of it, and two threads lock two buffers in about the time of one
		int rc1 = DBGK_OK;
		std::thread second([&]() { rc1 = hipSetDevice(h->device) == hipSuccess ? ensure_slot(h, h->slots[1]) : DBGK_ERR_HIP; });
		rc = ensure_slot(h, h->slots[0]);
		second.join();
		if (rc != DBGK_OK || rc1 != DBGK_OK) return fail(rc != DBGK_OK ? rc : rc1);
	}
	if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(DBGK_ERR_HIP);
	lap(5);
	if (lap_wanted)
		fprintf(stderr, "dbgk_create (s): runtime start %.4f device+stream %.4f table %.4f record store %.4f first reset %.4f staging %.4f\n", laps[0], laps[1],
		        laps[2], laps[3], laps[4], laps[5]);
	*out = h;
	return DBGK_OK;
}

extern "C" int dbgk_destroy(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	free_handle(h);
	return DBGK_OK;
}

extern "C" int dbgk_reset(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(h->stream));
	rc = collect_spans(h);
	if (rc) return rc;
	return reset_state(h);
}

extern "C" int dbgk_sync(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(h->stream));
	for (auto &s : h->slots) s.busy = false;
	return collect_spans(h);
}

extern "C" void *dbgk_stream(dbgk_handle *h) { return h ? (void *)h->stream : nullptr; }

static int build_from_records(dbgk_handle *h);
static int plan_partition(dbgk_handle *h);
static int setup_partition(dbgk_handle *h);

// PARTITION engine, streaming use: records -> table now (regions that already hold nodes are loaded back
// into LDS first), record store emptied.
static int flush_records(dbgk_handle *h)
{
	if (h->wpart && !h->finalized) return h->pending_kmers ? wide_build_from_records(h) : DBGK_OK;
	if (!h->part || h->finalized) return DBGK_OK;
	if (h->sharded) {
		g_last_error = "dbgk_flush: a sharded handle is flushed by its communicator (dbgk_comm_flush)";
		return DBGK_ERR_STATE;
	}
	if (h->pending_kmers == 0) return DBGK_OK;
	int rc = build_from_records(h);
	if (rc) return rc;
	h->incr = true;
	return clear_record_store(h);
}

// the PARTITION engine's bucket geometry is a function of the table size: a resize re-seats the nodes
// (k_rehash, global atomics) and re-plans / re-allocates the record stores.  Pending records are flushed first.
static int resize_partition_table(dbgk_handle *h, uint64_t new_slots)
{
	if (h->kfreq || h->sharded) {
		g_last_error = "dbgk_resize_table: not available for KFREQ and sharded handles";
		return DBGK_ERR_STATE;
	}
	int rc = flush_records(h);
	if (rc) return rc;
	dbgk_handle probe_cfg;           // feasibility first: nothing is touched if the new size does not fit the engine
	probe_cfg.cfg = h->cfg;
	probe_cfg.size = new_slots;
	probe_cfg.magic = make_mod_magic(new_slots);
	rc = plan_partition(&probe_cfg);
	if (rc) return rc;
	if (!probe_cfg.part) {
		g_last_error = "dbgk_resize_table: the new size does not fit the PARTITION engine's geometry";
		return DBGK_ERR_ARG;
	}
	Node *fresh = nullptr;
	if (hipMalloc(&fresh, new_slots * sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
	const TableRef dst{fresh, new_slots, make_mod_magic(new_slots)};
	if (h->incr) { // the old table holds nodes
		hipError_t e = hipMemsetAsync(fresh, 0, new_slots * sizeof(Node), h->stream);
		if (e == hipSuccess) {
			hipLaunchKernelGGL(k_rehash, dim3(grid_for(h, h->size)), dim3(kBlock), 0, h->stream, h->table, h->size, dst, h->d_ctr,
			                   (const unsigned long long *)nullptr, (unsigned long long *)nullptr);
			e = hipGetLastError();
		}
		if (e == hipSuccess) e = hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
		if (e != hipSuccess) {
			(void)hipFree(fresh);
			return hip_fail(e, "resize_table(partition)", __LINE__);
		}
		if (h->h_ctr->error & 1u) {
			(void)hipFree(fresh);
			HIPCHK(hipMemsetAsync(&h->d_ctr->error, 0, sizeof(unsigned int), h->stream));
			HIPCHK(hipStreamSynchronize(h->stream));
			return DBGK_ERR_TABLE_FULL;
		}
	}
	HIPCHK(hipStreamSynchronize(h->stream));
	if (h->stream2) HIPCHK(hipStreamSynchronize(h->stream2));
	(void)hipFree(h->table);
	h->table = fresh;
	h->size = new_slots;
	h->magic = dst.magic;
	h->cfg.table_slots = new_slots;
	free_partition_stores(h);
	rc = plan_partition(h);
	if (rc == DBGK_OK && !h->part) rc = DBGK_ERR_STATE;
	if (rc == DBGK_OK) rc = setup_partition(h);
	if (rc) return rc;
	h->zero_pending = !h->incr; // nothing built yet: the first region build writes every slot
	return clear_record_store(h);
}

extern "C" int dbgk_resize_table(dbgk_handle *h, uint64_t new_slots)
{
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || new_slots < 3) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	rc = dbgk_sync(h);
	if (rc) return rc;
	if (new_slots == h->size) return DBGK_OK;
	if (h->part) return resize_partition_table(h, new_slots);
	Node *fresh = nullptr;
	unsigned long long *fresh_first = nullptr;
	if (hipMalloc(&fresh, new_slots * sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
	if (h->track) {
		if (hipMalloc(&fresh_first, new_slots * 8) != hipSuccess) {
			(void)hipFree(fresh);
			return DBGK_ERR_NOMEM;
		}
		if (hipMemsetAsync(fresh_first, 0xFF, new_slots * 8, h->stream) != hipSuccess) {
			(void)hipFree(fresh);
			(void)hipFree(fresh_first);
			return DBGK_ERR_HIP;
		}
	}
	TableRef dst{fresh, new_slots, make_mod_magic(new_slots)};
	hipError_t e = hipMemsetAsync(fresh, 0, new_slots * sizeof(Node), h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_rehash, dim3(grid_for(h, h->size)), dim3(kBlock), 0, h->stream, h->table, h->size, dst, h->d_ctr,
		                   (const unsigned long long *)h->first_pos, fresh_first);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) {
		(void)hipFree(fresh);
		if (fresh_first) (void)hipFree(fresh_first);
		return hip_fail(e, "resize_table", __LINE__);
	}
	if (h->h_ctr->error & 1u) { // new table too small for the existing nodes: keep the old one
		(void)hipFree(fresh);
		if (fresh_first) (void)hipFree(fresh_first);
		HIPCHK(hipMemsetAsync(&h->d_ctr->error, 0, sizeof(unsigned int), h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		return DBGK_ERR_TABLE_FULL;
	}
	(void)hipFree(h->table);
	if (h->track) {
		(void)hipFree(h->first_pos);
		h->first_pos = fresh_first;
	}
	h->table = fresh;
	h->size = new_slots;
	h->tslots = new_slots;
	h->magic = dst.magic;
	h->cfg.table_slots = new_slots;
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// the hot path
// ---------------------------------------------------------------------------------------------

// queue mark + insert for a batch that is already in device memory
// The lane-per-chunk-of-valid-windows level-1 kernel (k_extract_scatter_uniform) applies when nothing is
// trimmed (longest read <= maxReadLen), the longest read has at least 64 windows (a tile's byte range
// must fit its LDS image) and either
//   * every read has that length and they lie back to back from offset 0 (EQUAL), or
//   * lengths differ, but giving every read the lane count of the longest one still needs fewer lane
//     slots than the flat kernel has positions (RAGGED: reads mostly full length, some shorter).
// Returns 0 = flat kernel, 1 = equal, 2 = ragged; fills U.
// lin (out): the linear level-1 form (8 windows per lane, linear copy-out) is to be used -- many level-1 buckets
static int uniform_mode(const dbgk_handle *h, int64_t uniform_len, uint64_t len_max, uint64_t n_reads, uint64_t n_bases, int has_long,
                        UniformGeom &U, bool &c15, bool &lin, bool &lin12)
{
	lin = lin12 = false;
	static const bool off = DBGK_EXPERIMENT_ENV("DBGK_L1_FLAT") != nullptr; // force the general kernel
	static const int dbg_mode = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE")) : 0;
	if (off || has_long || n_reads == 0) return 0;
	if (dbg_mode && (uniform_len != 150 || h->cfg.kmer_size != 31 || h->geom.size >= (1ull << 31))) return 0; // debug builds: cfg2's shape only
	const uint64_t L = uniform_len > 0 ? (uint64_t)uniform_len : len_max, k = (uint64_t)h->cfg.kmer_size;
	if (L > (uint64_t)h->cfg.max_read_len || L < k + 63 || L >= (1ull << 24)) return 0;
	const uint32_t W = (uint32_t)(L - k + 1);
	// 16 or 15 windows per lane, whichever leaves fewer empty slots at the end of a full-length read (W = 120: 15 -> none)
	const uint32_t q16 = (W + 15u) / 16u, q15 = (W + 14u) / 15u;
	c15 = q15 * 15u - W < q16 * 16u - W;
	const uint64_t Q = c15 ? q15 : q16, C = c15 ? 15 : 16;
	if (Q >= 2048 || n_reads * Q >= (1ull << 32)) return 0;
	if (((uint64_t)kL1Threads / Q + 2) * L + 96 > (uint64_t)kPkWords * 16) return 0; // bytes a tile of kL1Threads lanes can touch
	U.L = (uint32_t)L;
	U.W = W;
	U.Q = (uint32_t)Q;
	U.qmagic = ((1u << 22) + U.Q - 1u) / U.Q;
	U.n_lanes = n_reads * Q;
	U.lq = 0;
	U.tile_blocks = 0; // 0: no regular tiles
	// (the regular form also assumes k >= 17 -- the rolls then only touch the high words --, reads that fill their lanes exactly,
	// and a graph handle: no KFREQ neighbour codes)
	if ((Q & (Q - 1)) == 0 && Q <= (uint64_t)kL1Threads && (((uint64_t)kL1Threads / Q) * L) % 16 == 0 &&
	    ((uint64_t)kL1Threads / Q) * L / 16 + 8 <= (uint64_t)kPkWords && k >= 17 && Q * C == (uint64_t)W && !h->kfreq) {
		while ((1ull << U.lq) < Q) U.lq++;
		U.tile_blocks = (uint32_t)(((uint64_t)kL1Threads / Q) * L / 16);
	}
	int mode;
	if (uniform_len > 0) mode = n_bases == n_reads * L ? 1 : 0;
	else mode = (double)(n_reads * Q * C) <= 0.93 * (double)n_bases ? 2 : 0; // mostly full-length reads: the ragged form
	if (mode == 0) return 0;
	// Many level-1 buckets (big tables; every rank of a multi-GPU job partitions by the GLOBAL table's buckets): the linear
	// form, 12 (or 8) windows per lane.  Measured on cfg2's reads, level 1 in ms, wave-per-bucket / linear with 8 / with 12
	// windows: n1 = 143: 5.57 / 6.76 / 6.05, 573: 8.13 / 7.57 / 6.81, 1023: 10.61 / 7.76 / 6.99 (287: 6.34 / 6.77 / -).
	// DBGK_L1_LINEAR=0/1 forces the choice.
	const int force = dbgk_hook("l1_linear") ? atoi(dbgk_hook("l1_linear")) : -1; // (read per batch: tests switch it)
	// 12 or 8 windows per lane, whichever leaves fewer empty slots at the end of a read (W = 120: both none -> 12)
	const uint64_t q12 = (W + 11u) / 12u, q8 = (W + 7u) / 8u;
	lin12 = q12 * 12u - W <= q8 * 8u - W;
	if (const char *e = DBGK_EXPERIMENT_ENV("DBGK_L1_LINEAR_C")) lin12 = atoi(e) == 12; // measurements
	const uint64_t QL = lin12 ? q12 : q8, CL = lin12 ? 12 : 8;
	bool fits = QL < 2048 && n_reads * QL < (1ull << 32) && ((uint64_t)kL1Threads / QL + 2) * L + 96 <= (uint64_t)kPkWords * 16;
	if (mode == 2) fits = fits && (double)(n_reads * QL * CL) <= 0.93 * (double)n_bases;
	if (fits && (force == 1 || (force < 0 && h->geom.n1 > 320u))) {
		lin = true;
		U.tile_blocks = 0;
		U.Q = (uint32_t)QL;
		U.qmagic = ((1u << 22) + U.Q - 1u) / U.Q;
		U.n_lanes = n_reads * QL;
	}
	return mode;
}

// WIDE record path: can this batch take the equal-length level-1 kernel (k_wide_scatter_l1_uniform)?
static int wide_uniform_mode(const dbgk_handle *h, int64_t uniform_len, uint64_t n_reads, uint64_t n_bases, int has_long, WUniformGeom &U)
{
	static const bool off = DBGK_EXPERIMENT_ENV("DBGK_L1_FLAT") != nullptr; // force the general kernel
	if (off || has_long || n_reads == 0 || uniform_len <= 0) return 0;
	const uint64_t L = (uint64_t)uniform_len, k = (uint64_t)h->cfg.kmer_size;
	if (L > (uint64_t)h->cfg.max_read_len || L < k || L >= (1ull << 24) || n_bases != n_reads * L) return 0;
	const uint32_t W = (uint32_t)(L - k + 1);
	const uint64_t Q = (W + 7u) / 8u;
	if (Q >= 2048 || n_reads * Q >= (1ull << 40)) return 0;
	if (((uint64_t)kWL1Threads / Q + 2) * L + 96 > (uint64_t)(kWPkWords - 8u) * 16) return 0; // bytes a tile of 1024 lanes can touch
	if ((double)(n_reads * Q * 8) > 0.93 * (double)n_bases) return 0; // (k small against L: the flat kernel wastes little)
	U.L = (uint32_t)L;
	U.W = W;
	U.Q = (uint32_t)Q;
	U.qmagic = ((1u << 22) + U.Q - 1u) / U.Q;
	U.n_lanes = n_reads * Q;
	return 1;
}

// scratch of the prefix form for a batch of n_reads reads / n_bases bases (grown as needed; the stream is idle when it grows)
static int ensure_prefix_scratch(dbgk_handle *h, uint64_t n_reads, uint64_t n_bases, bool need_packed)
{
	const uint64_t tiles = (n_bases / 15 + n_reads) / kL1Threads + 2; // a read of W windows has at most W / 15 + 1 lanes
	if (n_reads > h->pf_cap_reads || tiles > h->pf_cap_tiles) {
		HIPCHK(hipStreamSynchronize(h->stream));
		for (void *q : {(void *)h->pf_ent, (void *)h->pf_tile_first, (void *)h->pf_tiles, (void *)h->pf_bsum})
			if (q) (void)hipFree(q);
		h->pf_ent = nullptr; h->pf_tile_first = nullptr; h->pf_tiles = nullptr; h->pf_bsum = nullptr;
		h->pf_cap_reads = h->pf_cap_tiles = 0;
		const uint64_t cr = std::max(n_reads, h->cap_reads), ct = std::max(tiles, (h->cap_bases / 15 + h->cap_reads) / kL1Threads + 2);
		const uint64_t blocks = (cr + kPrefixBlock * kPrefixItems - 1) / (kPrefixBlock * kPrefixItems) + 1;
		if (hipMalloc(&h->pf_ent, cr * sizeof(ReadLanes)) != hipSuccess || hipMalloc(&h->pf_tile_first, (ct + 1) * 4) != hipSuccess ||
		    hipMalloc(&h->pf_tiles, ct * sizeof(PrefixTile)) != hipSuccess || hipMalloc(&h->pf_bsum, blocks * 8) != hipSuccess)
			return DBGK_ERR_NOMEM;
		h->pf_cap_reads = cr;
		h->pf_cap_tiles = ct;
	}
	if (!h->pf_tot && hipMalloc(&h->pf_tot, sizeof(PrefixTotals)) != hipSuccess) return DBGK_ERR_NOMEM;
	const uint64_t words = (n_bases + 15) / 16 + 16;
	if (need_packed && words > h->pf_packed_words) {
		HIPCHK(hipStreamSynchronize(h->stream));
		if (h->pf_packed) (void)hipFree(h->pf_packed);
		h->pf_packed = nullptr;
		h->pf_packed_words = 0;
		const uint64_t cw = std::max(words, h->cap_bases / 16 + 16);
		if (hipMalloc(&h->pf_packed, cw * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		h->pf_packed_words = cw;
	}
	return DBGK_OK;
}

static int early_l2(dbgk_handle *h);

static int launch_batch(dbgk_handle *h, const char *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                        uint64_t n_bases, uint32_t *d_start, uint32_t *d_dead, int has_long /* 0,1 or -1 = ask device */,
                        int64_t uniform_len = -1 /* every read this long; 0 = lengths differ; -1 = ask device */,
                        uint64_t len_max = 0 /* longest read of the batch (with uniform_len >= 0) */,
                        const uint32_t *d_packed = nullptr /* the batch as 2-bit codes instead of d_bases (then null) */)
{
	if (n_reads == 0) return DBGK_OK;
	if (h->seed && d_packed) return DBGK_ERR_ARG; // the seed index cuts its windows at 'N', which two bits cannot say
	if (h->seed) has_long = 1; // the dead bitmap carries the 'N' positions
	const uint64_t words = bitmap_words(n_bases);
	TimedSpan sp;
	int rc = early_l2(h); // level 2 of what the batches before this one stored: queued IN FRONT of this batch's level 1, i.e. it runs while
	if (rc) return rc;    // this batch is still on the link (the level-1 launch below waits for the copy, the level-2 round does not)
	rc = span_begin(h, PH_MARK, sp);
	if (rc) return rc;
	static_assert(offsetof(Counters, len_max) + sizeof(unsigned long long) - offsetof(Counters, any_dead) == 20, "per-batch fields are contiguous");
	if (d_offsets) HIPCHK(hipMemsetAsync(&h->d_ctr->any_dead, 0, 20, h->stream)); // any_dead, len_min_inv, len_max: what k_mark reports per batch
	// The read-boundary bitmaps are what the general kernels navigate by; the PARTITION engine's level-1 kernel for
	// (nearly) equal-length reads does without them, so for such a batch only the statistics are taken.  A device
	// batch tells its shape only after those statistics: the bitmaps follow in a second pass if they are needed.
	UniformGeom U{};
	bool c15 = false, lin8 = false, lin12 = false;
	int umode = -1; // not decided yet
	auto mark_bits = [&](int with_stats) -> int {
		HIPCHK(hipMemsetAsync(d_start, 0, words * 4, h->stream));
		if (has_long != 0) HIPCHK(hipMemsetAsync(d_dead, 0, words * 4, h->stream));
		hipLaunchKernelGGL(k_mark, dim3(grid_for(h, n_reads)), dim3(kBlock), 0, h->stream, d_offsets, n_reads, n_bases,
		                   h->cfg.kmer_size, h->cfg.max_read_len, d_start, has_long != 0 ? d_dead : nullptr, h->d_ctr, with_stats);
		return DBGK_OK;
	};
	const bool wrec = h->wide && h->wpart && !h->wbuilt; // WIDE handle that is still collecting records
	WUniformGeom WU{};
	auto decide_umode = [&]() {
		return wrec ? wide_uniform_mode(h, uniform_len, n_reads, n_bases, has_long, WU) : uniform_mode(h, uniform_len, len_max, n_reads, n_bases, has_long, U, c15, lin8, lin12);
	};
	// A batch of equal-length reads that came WITHOUT offsets (dbgk_push_reads_packed_uniform*): the equal-length level-1 forms of the
	// PARTITION / WIDE record engines never look at offsets -- the totals are added by a one-thread kernel and nothing else runs in
	// front of level 1; any other consumer gets the offsets made on the device first.
	const bool no_offsets = d_offsets == nullptr;
	auto make_offsets = [&]() -> int {
		if (n_reads + 1 > h->uni_cap) {
			HIPCHK(hipStreamSynchronize(h->stream));
			if (h->uni_offsets) (void)hipFree(h->uni_offsets);
			h->uni_offsets = nullptr;
			h->uni_cap = 0;
			const uint64_t cap = std::max(n_reads + 1, h->cap_reads + 1);
			if (hipMalloc(&h->uni_offsets, cap * 8) != hipSuccess) return DBGK_ERR_NOMEM;
			h->uni_cap = cap;
		}
		hipLaunchKernelGGL(k_iota_offsets, dim3(grid_for(h, n_reads + 1)), dim3(kBlock), 0, h->stream, h->uni_offsets, n_reads, (uint64_t)uniform_len);
		HIPCHK(hipMemsetAsync(&h->d_ctr->any_dead, 0, 20, h->stream));
		d_offsets = h->uni_offsets;
		return DBGK_OK;
	};
	const bool may_skip_bits = (h->part && !h->seed) || wrec;
	// The PREFIX form of level 1 (k_extract_scatter_prefix: every read exactly the lanes its windows need, reads of any lengths,
	// trimmed ones included) takes what would otherwise go through the flat kernel -- a fifth of whose positions straddle a
	// read boundary at 150 bases and k = 31 -- and the batches of the ragged form as well; not with many level-1 buckets (the
	// linear forms), not for reads of more than 4 M windows.  DBGK_L1_PREFIX=0 switches it off (ragged / flat as before).
	const int prefix_env = dbgk_hook("l1_prefix") ? atoi(dbgk_hook("l1_prefix")) : -1; // (read per batch: tests switch it)
	bool use_prefix = false;
	auto prefix_wanted = [&](int um) {
		static const bool flat_only = DBGK_EXPERIMENT_ENV("DBGK_L1_FLAT") != nullptr;
		static const bool dbg = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE") != nullptr;
		const int force_lin = dbgk_hook("l1_linear") ? atoi(dbgk_hook("l1_linear")) : -1;
		if (!h->part || h->seed || wrec || flat_only || dbg || prefix_env == 0) return false;
		if (force_lin == 1 || (force_lin < 0 && h->geom.n1 > 320u)) return false;
		if (len_max > (uint64_t)kPrefixMaxW || n_bases / 15 + n_reads >= (1ull << 32)) return false;
		return um == 0 || um == 2; // (measured on cfg2t, level 1 per step: prefix 5.32 ms, ragged 5.72, flat 6.05 + 0.14 of bitmaps: profiles/r04_cfg2t_level1_forms_ab.json)
	};
	if (may_skip_bits && has_long >= 0 && uniform_len >= 0) {
		umode = decide_umode();
		use_prefix = prefix_wanted(umode);
	}
	if (no_offsets && umode == 1) { // (umode 1 = equal lengths, nothing trimmed: every read has uniform_len - k + 1 windows)
		const unsigned long long w = uniform_len >= h->cfg.kmer_size ? (unsigned long long)(uniform_len - h->cfg.kmer_size + 1) * n_reads : 0ull;
		hipLaunchKernelGGL(k_add_totals, dim3(1), dim3(64), 0, h->stream, h->d_ctr, w, w);
	} else if (no_offsets) {
		rc = make_offsets();
		if (rc) return rc;
	}
	if (no_offsets && umode == 1) {
	} else if (may_skip_bits && (umode > 0 || umode < 0 || use_prefix)) {
		hipLaunchKernelGGL(k_mark, dim3(grid_for(h, n_reads)), dim3(kBlock), 0, h->stream, d_offsets, n_reads, n_bases, h->cfg.kmer_size,
		                   h->cfg.max_read_len, (uint32_t *)nullptr, (uint32_t *)nullptr, h->d_ctr, 1); // statistics only
	} else {
		rc = mark_bits(1);
		if (rc) return rc;
	}
	if (h->seed && n_bases)
		hipLaunchKernelGGL(k_mark_n, dim3(grid_for(h, (n_bases + 31) >> 5)), dim3(kBlock), 0, h->stream, d_bases, n_bases, d_dead);
	HIPCHK(hipGetLastError());
	if (has_long < 0 || (uniform_len < 0 && (h->part || wrec))) {
		HIPCHK(hipMemcpyAsync(&h->h_ctr->any_dead, &h->d_ctr->any_dead, 20, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		if (has_long < 0) has_long = h->h_ctr->any_dead ? 1 : 0;
		if (uniform_len < 0) {
			uniform_len = ~h->h_ctr->len_min_inv == h->h_ctr->len_max ? (int64_t)h->h_ctr->len_max : 0;
			len_max = h->h_ctr->len_max;
		}
	}
	if (may_skip_bits && umode < 0) {
		umode = decide_umode();
		use_prefix = prefix_wanted(umode);
		if (umode == 0 && !use_prefix) { // the general kernel after all: it needs the bitmaps
			rc = mark_bits(0);
			if (rc) return rc;
			HIPCHK(hipGetLastError());
		}
	}
	rc = span_end(h, sp);
	if (rc) return rc;
	const uint64_t id_base = h->total_reads; // contig index of the batch's first sequence (SEEDIDX)
	h->total_reads += n_reads;
	if (n_bases == 0) return DBGK_OK;

	ReadBatch rb{d_bases, n_bases, d_start, has_long ? d_dead : nullptr, h->cfg.kmer_size, d_packed, &h->d_ctr->other_seen};
	const uint64_t n_chunks = (n_bases + 15) >> 4;
	rc = span_begin(h, PH_INSERT, sp);
	if (rc) return rc;
	if (h->seed) {
		hipLaunchKernelGGL(k_seed_insert, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, d_offsets, n_reads, id_base, h->tref(), h->d_ctr);
	} else if (wrec && umode > 0) {
		h->uniform_launches++;
		const int grid = (int)std::min<uint64_t>((WU.n_lanes + kWL1Threads - 1) / kWL1Threads, (uint64_t)h->n_cu);
		const int wd = h->size >= (1ull << 32) ? 2 : (h->size >= (1ull << 31) ? 1 : 0); // how hash / size is computed
#define DBGK_LAUNCH_WIDE_L1U(WD)                                                                                                                    \
	hipLaunchKernelGGL((k_wide_scatter_l1_uniform<WD>), dim3(grid), dim3(kWL1Threads), sizeof(WL1Lds), h->stream, rb, WU, h->wgeom, h->wstore, h->wref(), \
	                   h->d_ctr)
		if (wd == 2) DBGK_LAUNCH_WIDE_L1U(2); else if (wd == 1) DBGK_LAUNCH_WIDE_L1U(1); else DBGK_LAUNCH_WIDE_L1U(0);
#undef DBGK_LAUNCH_WIDE_L1U
	} else if (wrec) {
		const int grid = (int)std::min<uint64_t>((n_chunks + kWL1Threads - 1) / kWL1Threads, (uint64_t)h->n_cu);
		const int wd = h->size >= (1ull << 32) ? 2 : (h->size >= (1ull << 31) ? 1 : 0); // how hash / size is computed
#define DBGK_LAUNCH_WIDE_L1(DEAD, WD)                                                                                                          \
	hipLaunchKernelGGL((k_wide_scatter_l1<DEAD, WD>), dim3(grid), dim3(kWL1Threads), sizeof(WL1Lds), h->stream, rb, h->wgeom, h->wstore, h->wref(), \
	                   h->d_ctr)
		if (has_long) {
			if (wd == 2) DBGK_LAUNCH_WIDE_L1(true, 2); else if (wd == 1) DBGK_LAUNCH_WIDE_L1(true, 1); else DBGK_LAUNCH_WIDE_L1(true, 0);
		} else {
			if (wd == 2) DBGK_LAUNCH_WIDE_L1(false, 2); else if (wd == 1) DBGK_LAUNCH_WIDE_L1(false, 1); else DBGK_LAUNCH_WIDE_L1(false, 0);
		}
#undef DBGK_LAUNCH_WIDE_L1
	} else if (h->wide) {
		rc = wide_ensure_zero(h);
		if (rc) return rc;
		if (has_long)
			hipLaunchKernelGGL(k_wide_extract_insert<true>, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->wref(), h->d_ctr);
		else
			hipLaunchKernelGGL(k_wide_extract_insert<false>, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->wref(), h->d_ctr);
	} else if (h->kfreq && !h->part) {
		if (has_long)
			hipLaunchKernelGGL(k_extract_count<true>, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, reinterpret_cast<uint32_t *>(h->counts));
		else
			hipLaunchKernelGGL(k_extract_count<false>, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, reinterpret_cast<uint32_t *>(h->counts));
	} else if (h->part && use_prefix) {
		// lane prefix of the batch (three short kernels over the offsets), the batch itself as 2-bit words, then level 1
		h->prefix_launches++;
		rc = ensure_prefix_scratch(h, n_reads, n_bases, d_packed == nullptr);
		if (rc) return rc;
		if (!d_packed) {
			hipLaunchKernelGGL(k_pack_bases, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, d_bases, n_bases, h->pf_packed, h->d_ctr);
			rb.packed = h->pf_packed;
		}
		const uint32_t W_max = (uint32_t)std::min<uint64_t>(len_max, (uint64_t)h->cfg.max_read_len) >= (uint32_t)h->cfg.kmer_size
		                           ? (uint32_t)std::min<uint64_t>(len_max, (uint64_t)h->cfg.max_read_len) - (uint32_t)h->cfg.kmer_size + 1u : 1u;
		const bool pc15 = (W_max + 14u) / 15u * 15u - W_max < (W_max + 15u) / 16u * 16u - W_max; // 15 or 16 windows per lane: fewer empty slots for a full-length read
		const uint32_t n_blocks = (uint32_t)((n_reads + kPrefixBlock * kPrefixItems - 1) / (kPrefixBlock * kPrefixItems));
		const uint32_t tiles_max = (uint32_t)std::min<uint64_t>((n_bases / 15 + n_reads) / kL1Threads + 1, h->pf_cap_tiles);
		const int wide = (h->geom.kf == 2u || h->geom.size >= (1ull << 32)) ? 2 : (h->geom.size >= (1ull << 31) ? 1 : 0);
		const int grid_p = (int)((uint64_t)h->n_cu * l1_wgs_per_cu());
#define DBGK_LAUNCH_PREFIX(WIDE, CC)                                                                                                              \
	do {                                                                                                                                          \
		hipLaunchKernelGGL((k_prefix_count<CC>), dim3(n_blocks), dim3(kPrefixBlock), 0, h->stream, d_offsets, n_reads, (uint32_t)h->cfg.kmer_size,   \
		                   (uint32_t)h->cfg.max_read_len, h->pf_bsum);                                                                            \
		hipLaunchKernelGGL(k_prefix_blocks, dim3(1), dim3(kPrefixBlock), 0, h->stream, h->pf_bsum, n_blocks, h->pf_tot);                          \
		hipLaunchKernelGGL((k_prefix_emit<CC>), dim3(n_blocks), dim3(kPrefixBlock), 0, h->stream, d_offsets, n_reads, (uint32_t)h->cfg.kmer_size,    \
		                   (uint32_t)h->cfg.max_read_len, h->pf_bsum, h->pf_ent, h->pf_tile_first, h->d_ctr);                                     \
		hipLaunchKernelGGL((k_prefix_tiles<CC>), dim3((tiles_max + 255) / 256), dim3(256), 0, h->stream, h->pf_ent, h->pf_tile_first, h->pf_tot,    \
		                   (uint32_t)h->cfg.kmer_size, n_bases, h->pf_tiles);                                                                     \
		if (h->cfg.kmer_size >= 17)                                                                                                                 \
			hipLaunchKernelGGL((k_extract_scatter_prefix<WIDE, CC, true>), dim3(grid_p), dim3(kL1Threads), sizeof(PrefixLds), h->stream, rb, h->pf_ent, \
			                   h->pf_tiles, h->pf_tot, h->geom, h->store, h->d_ctr);                                                              \
		else                                                                                                                                      \
			hipLaunchKernelGGL((k_extract_scatter_prefix<WIDE, CC, false>), dim3(grid_p), dim3(kL1Threads), sizeof(PrefixLds), h->stream, rb, h->pf_ent, \
			                   h->pf_tiles, h->pf_tot, h->geom, h->store, h->d_ctr);                                                              \
	} while (0)
		if (pc15) { if (wide == 2) DBGK_LAUNCH_PREFIX(2, 15); else if (wide == 1) DBGK_LAUNCH_PREFIX(1, 15); else DBGK_LAUNCH_PREFIX(0, 15); }
		else { if (wide == 2) DBGK_LAUNCH_PREFIX(2, 16); else if (wide == 1) DBGK_LAUNCH_PREFIX(1, 16); else DBGK_LAUNCH_PREFIX(0, 16); }
#undef DBGK_LAUNCH_PREFIX
	} else if (h->part && umode > 0) {
		h->uniform_launches++;
		const int wide = (h->geom.kf == 2u || h->geom.size >= (1ull << 32)) ? 2 : (h->geom.size >= (1ull << 31) ? 1 : 0); // (direct blocks: the 64-bit slot path)
		const bool ragged = umode == 2;
#define DBGK_LAUNCH_UNIFORM(WIDE, CC, RAG)                                                                                                   \
	hipLaunchKernelGGL((k_extract_scatter_uniform<0, WIDE, CC, RAG>), dim3(grid), dim3(kL1Threads), sizeof(UniformLds), h->stream, rb, U, \
	                   d_offsets, h->geom, h->store, h->d_ctr)
#define DBGK_LAUNCH_UNIFORM_W(WIDE)                                    \
	do {                                                               \
		if (c15 && ragged) DBGK_LAUNCH_UNIFORM(WIDE, 15, true);        \
		else if (c15) DBGK_LAUNCH_UNIFORM(WIDE, 15, false);            \
		else if (ragged) DBGK_LAUNCH_UNIFORM(WIDE, 16, true);          \
		else DBGK_LAUNCH_UNIFORM(WIDE, 16, false);                     \
	} while (0)
#define DBGK_LAUNCH_UNIFORM8(WIDE, CC, RAG)                                                                                                               \
	hipLaunchKernelGGL((k_extract_scatter_uniform<0, WIDE, CC, RAG, true>), dim3(grid), dim3(kL1Threads), sizeof(UniformLdsLin<CC>), h->stream, rb, U, \
	                   d_offsets, h->geom, h->store, h->d_ctr)
		static const int dbg_mode_u = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE")) : 0;
		static const bool no_reg = DBGK_EXPERIMENT_ENV("DBGK_L1_NO_REG") != nullptr; // A/B: the general form everywhere
		bool rest_only = false;
		if (!dbg_mode_u && !no_reg && umode == 1 && !lin8 && U.tile_blocks) {
			// regular tiles: kL1Threads / Q whole reads each, from a 16-byte boundary; the reads behind the last whole tile
			// (fewer than kL1Threads / Q) go through the general form below
			const uint64_t reads_per_tile = (uint64_t)kL1Threads / U.Q, full_tiles = n_reads / reads_per_tile;
			if (full_tiles) {
				UniformGeom UR = U;
				UR.n_lanes = full_tiles * kL1Threads;
				ReadBatch rr = rb;
				rr.n_bases = full_tiles * reads_per_tile * U.L;
				const int grid_r = (int)std::min<uint64_t>(full_tiles, (uint64_t)h->n_cu * l1_wgs_per_cu());
#define DBGK_LAUNCH_REG(WIDE, CC, PK)                                                                                                                  \
	hipLaunchKernelGGL((k_extract_scatter_uniform<0, WIDE, CC, false, false, true, PK>), dim3(grid_r), dim3(kL1Threads), sizeof(UniformLds), h->stream, rr, \
	                   UR, d_offsets, h->geom, h->store, h->d_ctr)
				if (d_packed) {
					if (c15) { if (wide == 2) DBGK_LAUNCH_REG(2, 15, true); else if (wide == 1) DBGK_LAUNCH_REG(1, 15, true); else DBGK_LAUNCH_REG(0, 15, true); }
					else { if (wide == 2) DBGK_LAUNCH_REG(2, 16, true); else if (wide == 1) DBGK_LAUNCH_REG(1, 16, true); else DBGK_LAUNCH_REG(0, 16, true); }
				} else if (c15) { if (wide == 2) DBGK_LAUNCH_REG(2, 15, false); else if (wide == 1) DBGK_LAUNCH_REG(1, 15, false); else DBGK_LAUNCH_REG(0, 15, false); }
				else { if (wide == 2) DBGK_LAUNCH_REG(2, 16, false); else if (wide == 1) DBGK_LAUNCH_REG(1, 16, false); else DBGK_LAUNCH_REG(0, 16, false); }
#undef DBGK_LAUNCH_REG
				const uint64_t done_reads = full_tiles * reads_per_tile;
				if (d_packed) rb.packed += done_reads * U.L / 16; // (a whole number of words: a tile is a multiple of 16 bases)
				else rb.bases += done_reads * U.L;
				rb.n_bases -= done_reads * U.L;
				U.n_lanes = (n_reads - done_reads) * U.Q;
				rest_only = U.n_lanes == 0;
			}
		}
		const uint64_t n_tiles_rest = (U.n_lanes + kL1Threads - 1) / kL1Threads;
		const int grid = (int)std::min<uint64_t>(std::max<uint64_t>(n_tiles_rest, 1), (uint64_t)h->n_cu * l1_wgs_per_cu());
		if (rest_only) {
		}
#ifdef DBGK_EXPERIMENTS
		else if (dbg_mode_u == 1)   // timing experiments on cfg2's shape (C = 15, equal lengths, size < 2^31): results are wrong
			hipLaunchKernelGGL((k_extract_scatter_uniform<1, 0, 15, false>), dim3(grid), dim3(kL1Threads), sizeof(UniformLds), h->stream, rb, U, d_offsets, h->geom, h->store, h->d_ctr);
		else if (dbg_mode_u == 2)
			hipLaunchKernelGGL((k_extract_scatter_uniform<2, 0, 15, false>), dim3(grid), dim3(kL1Threads), sizeof(UniformLds), h->stream, rb, U, d_offsets, h->geom, h->store, h->d_ctr);
		else if (dbg_mode_u == 3)
			hipLaunchKernelGGL((k_extract_scatter_uniform<3, 0, 15, false>), dim3(grid), dim3(kL1Threads), sizeof(UniformLds), h->stream, rb, U, d_offsets, h->geom, h->store, h->d_ctr);
#endif
		else if (lin8 && lin12 && ragged) {
			if (wide == 2) DBGK_LAUNCH_UNIFORM8(2, 12, true); else if (wide == 1) DBGK_LAUNCH_UNIFORM8(1, 12, true); else DBGK_LAUNCH_UNIFORM8(0, 12, true);
		} else if (lin8 && lin12) {
			if (wide == 2) DBGK_LAUNCH_UNIFORM8(2, 12, false); else if (wide == 1) DBGK_LAUNCH_UNIFORM8(1, 12, false); else DBGK_LAUNCH_UNIFORM8(0, 12, false);
		} else if (lin8 && ragged) {
			if (wide == 2) DBGK_LAUNCH_UNIFORM8(2, 8, true); else if (wide == 1) DBGK_LAUNCH_UNIFORM8(1, 8, true); else DBGK_LAUNCH_UNIFORM8(0, 8, true);
		} else if (lin8) {
			if (wide == 2) DBGK_LAUNCH_UNIFORM8(2, 8, false); else if (wide == 1) DBGK_LAUNCH_UNIFORM8(1, 8, false); else DBGK_LAUNCH_UNIFORM8(0, 8, false);
		} else if (wide == 2 && h->geom.kf == 2u && !ragged) { // KFREQ, direct blocks: the instantiation without hash, division and neighbour codes
			if (c15) DBGK_LAUNCH_UNIFORM(3, 15, false); else DBGK_LAUNCH_UNIFORM(3, 16, false);
		} else if (wide == 2) DBGK_LAUNCH_UNIFORM_W(2);
		else if (wide == 1) DBGK_LAUNCH_UNIFORM_W(1);
		else DBGK_LAUNCH_UNIFORM_W(0);
#undef DBGK_LAUNCH_UNIFORM8
#undef DBGK_LAUNCH_UNIFORM_W
#undef DBGK_LAUNCH_UNIFORM
	} else if (h->part) {
		const uint64_t n_tiles = (n_chunks + kL1Threads - 1) / kL1Threads;
		const int grid = (int)std::min<uint64_t>(n_tiles, (uint64_t)h->n_cu * l1_wgs_per_cu()); // 140 KiB of LDS: one workgroup per CU
		static const int dbg_mode = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_MODE")) : 0;
		const int wide_d = (h->geom.kf == 2u || h->geom.size >= (1ull << 32)) ? 2 : (h->geom.size >= (1ull << 31) ? 1 : 0); // how hash / size is computed
		const int force_lin = dbgk_hook("l1_linear") ? atoi(dbgk_hook("l1_linear")) : -1;
		if (!dbg_mode && (force_lin == 1 || (force_lin < 0 && h->geom.n1 > 320u))) { // many level-1 buckets: the linear form
#define DBGK_LAUNCH_FLAT_LIN(DEAD, WD)                                                                                                       \
	hipLaunchKernelGGL((k_extract_scatter_lin<DEAD, WD>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLdsLin<8>), h->stream, rb, h->geom, h->store, \
	                   h->d_ctr)
			if (has_long) {
				if (wide_d == 2) DBGK_LAUNCH_FLAT_LIN(true, 2); else if (wide_d == 1) DBGK_LAUNCH_FLAT_LIN(true, 1); else DBGK_LAUNCH_FLAT_LIN(true, 0);
			} else {
				if (wide_d == 2) DBGK_LAUNCH_FLAT_LIN(false, 2); else if (wide_d == 1) DBGK_LAUNCH_FLAT_LIN(false, 1); else DBGK_LAUNCH_FLAT_LIN(false, 0);
			}
#undef DBGK_LAUNCH_FLAT_LIN
		} else if (wide_d == 2 && has_long)
			hipLaunchKernelGGL((k_extract_scatter<true, 0, 2>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else if (wide_d == 2)
			hipLaunchKernelGGL((k_extract_scatter<false, 0, 2>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else if (wide_d && has_long)
			hipLaunchKernelGGL((k_extract_scatter<true, 0, 1>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else if (wide_d)
			hipLaunchKernelGGL((k_extract_scatter<false, 0, 1>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
#ifdef DBGK_EXPERIMENTS
		else if (dbg_mode == 1)
			hipLaunchKernelGGL((k_extract_scatter<false, 1>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else if (dbg_mode == 2)
			hipLaunchKernelGGL((k_extract_scatter<false, 2>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else if (dbg_mode == 3)
			hipLaunchKernelGGL((k_extract_scatter<false, 3>), dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
#endif
		else if (has_long)
			hipLaunchKernelGGL(k_extract_scatter<true>, dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
		else
			hipLaunchKernelGGL(k_extract_scatter<false>, dim3(grid), dim3(kL1Threads), sizeof(ScatterLds), h->stream, rb, h->geom, h->store, h->d_ctr);
	} else if (h->track) {
		if (has_long)
			hipLaunchKernelGGL((k_extract_insert<true, true>), dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->tref(), h->d_ctr,
			                   h->first_pos, h->pos_base);
		else
			hipLaunchKernelGGL((k_extract_insert<false, true>), dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->tref(), h->d_ctr,
			                   h->first_pos, h->pos_base);
		h->pos_base += n_bases;
	} else if (has_long) {
		hipLaunchKernelGGL((k_extract_insert<true, false>), dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->tref(), h->d_ctr,
		                   (unsigned long long *)nullptr, (uint64_t)0);
	} else {
		hipLaunchKernelGGL((k_extract_insert<false, false>), dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, rb, h->tref(), h->d_ctr,
		                   (unsigned long long *)nullptr, (uint64_t)0);
	}
	// bytes outside ACGTNacgtn were read as 'A'; if a kernel met one (Counters::other_seen) this batch's are counted now --
	// every workgroup of the launch leaves at once otherwise.  A packed batch has none: its packer counted them.
	if (!d_packed) hipLaunchKernelGGL(k_count_other_bytes, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, d_bases, n_bases, h->d_ctr);
	HIPCHK(hipGetLastError());
	return span_end(h, sp);
}

// pageable host buffer -> pinned staging buffer.  One memcpy thread moves ~12 GB/s, less than a third of
// what the H2D copy behind it can take, so large batches are cut over a few threads (DBGK_COPY_THREADS,
// default 8; the offset bookkeeping of the batch runs on the calling thread meanwhile).
static void staged_copy(char *dst, const char *src, size_t n, std::vector<std::thread> &workers)
{
	static const int want = getenv("DBGK_COPY_THREADS") ? atoi(getenv("DBGK_COPY_THREADS")) : 8;
	const size_t min_piece = 8u << 20;
	size_t pieces = want > 1 ? std::min<size_t>((size_t)want, n / min_piece) : 1;
	if (pieces <= 1) {
		memcpy(dst, src, n);
		return;
	}
	const size_t per = ((n + pieces - 1) / pieces + 4095) & ~(size_t)4095;
	for (size_t p = 1; p < pieces; p++) {
		const size_t lo = p * per, hi = std::min(n, lo + per);
		if (lo < hi) workers.emplace_back([=]() { memcpy(dst + lo, src + lo, hi - lo); });
	}
	memcpy(dst, src, std::min(n, per));
}

// Is the caller's buffer page-locked memory the GPU reads directly (hipHostMalloc / hipHostRegister -- a torch pinned tensor, a
// parser's own pinned arena)?  Then dbgk_push_reads copies host-to-device straight out of it: the staging copy, which is what
// bounds the pageable path (~30 GB/s against the link's 57), does not happen.
static bool device_readable_host(const char *p, size_t n)
{
	static const bool off = DBGK_EXPERIMENT_ENV("DBGK_NO_PINNED_SOURCE") && atoi(DBGK_EXPERIMENT_ENV("DBGK_NO_PINNED_SOURCE"));
	if (off || !p || !n) return false;
	void *dev[2] = {nullptr, nullptr};
	int i = 0;
	for (const char *q : {p, p + n - 1}) {
		hipPointerAttribute_t a;
		if (hipPointerGetAttributes(&a, q) != hipSuccess) {
			(void)hipGetLastError(); // plain malloc'ed memory: "invalid value", not an error of ours
			return false;
		}
		if (a.type != hipMemoryTypeHost) return false;
		dev[i++] = a.devicePointer;
	}
	// both ends page-locked is not enough: two registrations with pageable memory between them would pass.  One mapping means
	// one contiguous range of device addresses: the two ends must lie exactly n - 1 bytes apart there as well.
	if (dev[0] && dev[1] && (const char *)dev[1] - (const char *)dev[0] != (ptrdiff_t)(n - 1)) return false;
	return true;
}

static int ensure_slot(dbgk_handle *h, StageSlot &s)
{
	if (s.d_bases) return DBGK_OK;
	const uint64_t words = bitmap_words(h->cap_bases);
	if (hipHostMalloc(&s.h_bases, h->cap_bases, hipHostMallocDefault) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipHostMalloc(&s.h_offsets, (h->cap_reads + 1) * 8, hipHostMallocDefault) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&s.d_bases, h->cap_bases + 64) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&s.d_offsets, (h->cap_reads + 1) * 8) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&s.d_start, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&s.d_dead, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
	HIPCHK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
	return DBGK_OK;
}

// The host-to-device copies of a batch (sequences from `src`, offsets from the slot's pinned array) on the handle's COPY stream; the
// compute stream waits for them.  The copy of batch i+1 so overlaps the kernels of batch i (and the table reset in front of the
// first batch) instead of queueing behind them.  The slot's device buffers are free: the caller has waited for s.done.
static int h2d_batch(dbgk_handle *h, StageSlot &s, const char *src, uint64_t nb, uint64_t n_offsets, bool last_of_pinned_source)
{
	static const bool serial = DBGK_EXPERIMENT_ENV("DBGK_COPY_ON_COMPUTE_STREAM") && atoi(DBGK_EXPERIMENT_ENV("DBGK_COPY_ON_COMPUTE_STREAM")); // measurements
	if (!serial && !h->copy_stream) HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
	hipStream_t cs = serial ? h->stream : h->copy_stream;
	if (nb) HIPCHK(hipMemcpyAsync(s.d_bases, src, nb, hipMemcpyHostToDevice, cs));
	if (n_offsets) HIPCHK(hipMemcpyAsync(s.d_offsets, s.h_offsets, n_offsets * 8, hipMemcpyHostToDevice, cs)); // (0: reads of one length, no offsets travel)
	if (last_of_pinned_source) { // the caller's buffer is free again once the LAST copy out of it has run: waited for on return
		if (!h->source_read) HIPCHK(hipEventCreateWithFlags(&h->source_read, hipEventDisableTiming));
		HIPCHK(hipEventRecord(h->source_read, cs));
	}
	if (!serial) {
		HIPCHK(hipEventRecord(s.copied, cs));
		// level 2 of what the batches before this one stored goes onto the compute stream BEFORE that stream is told to wait for this
		// batch's copy: it runs while the batch is on the link
		int rc = early_l2(h);
		if (rc) return rc;
		HIPCHK(hipStreamWaitEvent(h->stream, s.copied, 0));
	}
	return DBGK_OK;
}

// Zero-copy hand-over of a batch: the caller writes the sequences and offsets straight into the handle's pinned staging
// buffers (dbgk_push_acquire) and commits them (dbgk_push_commit) -- what dbgk_push_reads does minus its copy of the batch.
extern "C" int dbgk_push_acquire(dbgk_handle *h, char **bases, uint64_t **offsets, uint64_t *cap_bases, uint64_t *cap_reads)
{
	if (!h || !bases || !offsets) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	StageSlot &s = h->slots[h->next_slot];
	rc = ensure_slot(h, s);
	if (rc) return rc;
	if (s.busy) {
		HIPCHK(hipEventSynchronize(s.done));
		s.busy = false;
	}
	s.acquired = true;
	*bases = s.h_bases;
	*offsets = s.h_offsets;
	if (cap_bases) *cap_bases = h->cap_bases;
	if (cap_reads) *cap_reads = h->cap_reads;
	return DBGK_OK;
}

static int push_commit_impl(dbgk_handle *h, uint64_t n_reads, bool packed)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (packed && h->seed) return DBGK_ERR_ARG;
	if (n_reads == 0) return DBGK_OK;
	if (n_reads > h->cap_reads) return DBGK_ERR_ARG;
	if (h->seed && h->total_reads + n_reads > 0xFFFFFFFFull) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	StageSlot &s = h->slots[h->next_slot];
	if (!s.h_offsets || s.busy || !s.acquired) return DBGK_ERR_STATE; // dbgk_push_acquire first, one commit per acquire
	s.acquired = false;
	const uint64_t K = (uint64_t)h->cfg.kmer_size, max_len = (uint64_t)h->cfg.max_read_len;
	if (s.h_offsets[0] != 0) return DBGK_ERR_ARG;
	uint64_t batch_windows = 0, len_max = 0;
	int has_long = 0;
	int64_t uniform_len = (int64_t)(s.h_offsets[1] - s.h_offsets[0]);
	for (uint64_t i = 1; i <= n_reads; i++) {
		if (s.h_offsets[i] < s.h_offsets[i - 1] || s.h_offsets[i] > h->cap_bases) return DBGK_ERR_ARG;
		const uint64_t len = s.h_offsets[i] - s.h_offsets[i - 1], rl = len > max_len ? max_len : len;
		if (rl >= K) batch_windows += rl - K + 1;
		if (len > max_len) has_long = 1;
		if ((int64_t)len != uniform_len) uniform_len = 0;
		len_max = std::max(len_max, len);
		if (h->seed && len >= (1ull << 30)) return DBGK_ERR_ARG;
	}
	const uint64_t nb = s.h_offsets[n_reads];
	const bool streaming = (h->part && !h->sharded) || (h->wpart && !h->wbuilt);
	if (streaming && h->pending_kmers > 0 && h->pending_kmers + batch_windows > h->store_capacity) { // the store is full: records -> table first
		rc = flush_records(h);
		if (rc) return rc;
	}
	rc = h2d_batch(h, s, s.h_bases, packed ? ((nb + 15) >> 4) * 4 : nb, n_reads + 1, false);
	if (rc) return rc;
	rc = launch_batch(h, packed ? nullptr : s.d_bases, s.d_offsets, n_reads, nb, s.d_start, s.d_dead, has_long, uniform_len, len_max,
	                  packed ? reinterpret_cast<const uint32_t *>(s.d_bases) : nullptr);
	if (rc) return rc;
	h->pending_kmers += batch_windows;
	HIPCHK(hipEventRecord(s.done, h->stream));
	s.busy = true;
	h->next_slot ^= 1;
	return DBGK_OK;
}

extern "C" int dbgk_push_commit(dbgk_handle *h, uint64_t n_reads) { return push_commit_impl(h, n_reads, false); }

extern "C" int dbgk_push_commit_packed(dbgk_handle *h, uint64_t n_reads, uint64_t other_bytes)
{
	const int rc = push_commit_impl(h, n_reads, true);
	if (rc == DBGK_OK) h->host_other_bytes += other_bytes;
	return rc;
}

extern "C" int dbgk_push_reads(dbgk_handle *h, const char *bases, const uint64_t *offsets, uint64_t n_reads)
{
	if (!h || !offsets || (n_reads && !bases && offsets[n_reads] != offsets[0])) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (h->seed && h->total_reads + n_reads > 0xFFFFFFFFull) return DBGK_ERR_ARG; // id is a 32-bit field
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t K = (uint64_t)h->cfg.kmer_size, max_len = (uint64_t)h->cfg.max_read_len;
	// PARTITION engine: the record store holds store_capacity occurrences; a batch that would not fit is
	// preceded by a flush (records -> table, dbgk_flush).  Batches are cut to the room that is left only when
	// the store is at least one staging batch large; a smaller store (expected_kmers a gross under-estimate)
	// takes whole batches and sends the excess through its overflow lists as before.
	const bool streaming = (h->part && !h->sharded) || (h->wpart && !h->wbuilt);
	const bool cut_to_room = streaming && h->store_capacity >= h->cap_bases && !h->wpart; // (a wide store is built once: no point in filling it to the brim)
	const bool pinned_source = n_reads && device_readable_host(bases + offsets[0], offsets[n_reads] - offsets[0]);
	bool source_in_flight = false;
	struct WaitSource { // EVERY exit path waits for the copies that still read the caller's buffer (the header: it may be reused on return)
		dbgk_handle *h; bool &on;
		~WaitSource() { if (on && h->source_read) (void)hipEventSynchronize(h->source_read); }
	} wait_source{h, source_in_flight};
	uint64_t r0 = 0;
	while (r0 < n_reads) {
		// largest [r0, r1) that fits the staging buffers (and the record store).  ONE read-only pass over the offsets validates them
		// and gathers everything the launch needs (windows, longest read, equal lengths); the rebased copy for the device is
		// written later, while the sequences are on their way
		uint64_t r1 = r0, batch_windows = 0, len_max = 0;
		const uint64_t base0 = offsets[r0];
		const uint64_t room = h->store_capacity > h->pending_kmers ? h->store_capacity - h->pending_kmers : 0;
		const uint64_t r_end = std::min<uint64_t>(n_reads, r0 + h->cap_reads);
		const uint64_t first_len = offsets[r0 + 1] >= base0 ? offsets[r0 + 1] - base0 : 0;
		bool uniform = true;
		for (uint64_t prev = base0; r1 < r_end; r1++) {
			const uint64_t next = offsets[r1 + 1];
			if (next < prev) return DBGK_ERR_ARG;
			if (next - base0 > h->cap_bases) break;
			const uint64_t len = next - prev, rl = len > max_len ? max_len : len, w = rl >= K ? rl - K + 1 : 0ull;
			if (cut_to_room && batch_windows + w > room && (r1 > r0 || h->pending_kmers > 0)) break;
			batch_windows += w;
			len_max = len > len_max ? len : len_max;
			uniform = uniform && len == first_len;
			prev = next;
		}
		if (streaming && !(h->wpart && h->wbuilt) && h->pending_kmers > 0 &&
		    (r1 == r0 || (!cut_to_room && h->pending_kmers + batch_windows > h->store_capacity))) {
			rc = flush_records(h);
			if (rc) return rc;
			if (r1 == r0) continue; // cut again with the whole store free
		}
		if (r1 == r0) return DBGK_ERR_ARG; // a single read larger than max_batch_bases
		if (h->seed && len_max >= (1ull << 30)) return DBGK_ERR_ARG; // pos is a 30-bit field
		StageSlot &s = h->slots[h->next_slot];
		rc = ensure_slot(h, s);
		if (rc) return rc;
		if (s.busy) {
			HIPCHK(hipEventSynchronize(s.done));
			s.busy = false;
		}
		s.acquired = false; // (the slot is overwritten: a batch acquired before this call and not committed is gone)
		const uint64_t nb = offsets[r1] - base0, nr = r1 - r0;
		std::vector<std::thread> copiers;
		struct Join { // every exit path below waits for the copy threads
			std::vector<std::thread> &w;
			~Join() { for (auto &t : w) if (t.joinable()) t.join(); }
		} join_copiers{copiers};
		if (nb && !pinned_source) staged_copy(s.h_bases, bases + base0, nb, copiers);
		const int has_long = len_max > max_len ? 1 : 0;
		const int64_t uniform_len = uniform ? (int64_t)first_len : 0;
		{
			const uint64_t *src = offsets + r0;
			uint64_t *dst = s.h_offsets;
			for (uint64_t i = 0; i <= nr; i++) dst[i] = src[i] - base0;
		}
		for (auto &t : copiers) t.join();
		rc = h2d_batch(h, s, pinned_source ? bases + base0 : s.h_bases, nb, nr + 1, pinned_source);
		if (rc) return rc;
		source_in_flight = source_in_flight || pinned_source;
		rc = launch_batch(h, s.d_bases, s.d_offsets, nr, nb, s.d_start, s.d_dead, has_long, uniform_len, len_max);
		if (rc) return rc;
		h->pending_kmers += batch_windows;
		HIPCHK(hipEventRecord(s.done, h->stream));
		s.busy = true;
		h->next_slot ^= 1;
		r0 = r1;
	}
	return DBGK_OK; // (wait_source: as with the staged path, `bases` may be reused on return)
}

extern "C" int dbgk_push_reads_device(dbgk_handle *h, const char *d_bases, const uint64_t *d_offsets,
                                      uint64_t n_reads, uint64_t n_bases)
{
	if (!h || !d_offsets || (n_bases && !d_bases)) return DBGK_ERR_ARG;
	if (((uintptr_t)d_bases & 15u) || ((uintptr_t)d_offsets & 7u)) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t words = bitmap_words(n_bases);
	if (words > h->dev_bits_words) {
		HIPCHK(hipStreamSynchronize(h->stream));
		if (h->dev_start) (void)hipFree(h->dev_start);
		if (h->dev_dead) (void)hipFree(h->dev_dead);
		h->dev_start = h->dev_dead = nullptr;
		h->dev_bits_words = 0;
		if (hipMalloc(&h->dev_start, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		if (hipMalloc(&h->dev_dead, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		h->dev_bits_words = words;
	}
	// the record store takes what it was sized for; the number of windows of a device batch is only known as
	// an upper bound here (one per base)
	if (((h->part && !h->sharded) || (h->wpart && !h->wbuilt)) && h->pending_kmers > 0 && h->pending_kmers + n_bases > h->store_capacity) {
		rc = flush_records(h);
		if (rc) return rc;
	}
	rc = launch_batch(h, d_bases, d_offsets, n_reads, n_bases, h->dev_start, h->dev_dead, -1);
	if (rc == DBGK_OK) h->pending_kmers += n_bases;
	return rc;
}

// dbgk_push_reads for a batch that is already 2 bits per base (include/dbgk.h).  Same cutting into staging batches; a batch
// must start on a word of its own on the device, so one that starts in the middle of a source word is shifted into place while
// it is copied into the pinned staging buffer (host threads; the copy is a quarter of the ASCII one).  A page-locked source is
// read by the copy engine directly whenever the batch starts on a word boundary -- later batches are cut where that holds.
extern "C" void dbgk_internal_shift_packed(const uint32_t *src, uint64_t first_base, uint64_t n_words, uint64_t src_words, uint32_t *dst);

// one pass over offsets[r0 .. r1]: monotone?  windows the reads hold, longest and shortest read.  Ten million reads are 80 MB of
// offsets -- a single thread needs longer for them than the packed sequences need for the PCIe link, so large ranges are cut over threads.
namespace {
struct OffsetScan {
	bool ok = true;
	uint64_t windows = 0, len_max = 0, len_min = ~0ull;
};
OffsetScan scan_offsets(const uint64_t *off, uint64_t r0, uint64_t r1, uint64_t K, uint64_t max_len)
{
	auto part = [=](uint64_t a, uint64_t b) {
		OffsetScan o;
		uint64_t prev = off[a];
		for (uint64_t i = a; i < b; i++) {
			const uint64_t next = off[i + 1];
			if (next < prev) { o.ok = false; break; }
			const uint64_t len = next - prev, rl = len > max_len ? max_len : len;
			o.windows += rl >= K ? rl - K + 1 : 0ull;
			o.len_max = len > o.len_max ? len : o.len_max;
			o.len_min = len < o.len_min ? len : o.len_min;
			prev = next;
		}
		return o;
	};
	const uint64_t n = r1 - r0;
	static const int want = getenv("DBGK_COPY_THREADS") ? atoi(getenv("DBGK_COPY_THREADS")) : 8;
	const uint64_t pieces = want > 1 ? std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)want, n >> 19)) : 1;
	if (pieces <= 1) return part(r0, r1);
	std::vector<OffsetScan> res(pieces);
	std::vector<std::thread> th;
	const uint64_t per = (n + pieces - 1) / pieces;
	for (uint64_t p = 0; p < pieces; p++) {
		const uint64_t a = r0 + p * per, b = std::min(r1, a + per);
		if (a >= b) continue;
		if (p + 1 < pieces) th.emplace_back([&res, part, p, a, b]() { res[p] = part(a, b); });
		else res[p] = part(a, b);
	}
	for (auto &t : th) t.join();
	OffsetScan o;
	for (const OffsetScan &x : res) {
		o.ok = o.ok && x.ok;
		o.windows += x.windows;
		o.len_max = std::max(o.len_max, x.len_max);
		o.len_min = std::min(o.len_min, x.len_min);
	}
	return o;
}
// dst[i] = src[i] - base for i in [0, n]: the offsets of a batch as the device sees them
void rebase_offsets(uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t base)
{
	auto part = [=](uint64_t a, uint64_t b) { for (uint64_t i = a; i < b; i++) dst[i] = src[i] - base; };
	static const int want = getenv("DBGK_COPY_THREADS") ? atoi(getenv("DBGK_COPY_THREADS")) : 8;
	const uint64_t pieces = want > 1 ? std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)want, (n + 1) >> 19)) : 1;
	if (pieces <= 1) { part(0, n + 1); return; }
	std::vector<std::thread> th;
	const uint64_t per = (n + 1 + pieces - 1) / pieces;
	for (uint64_t p = 1; p < pieces; p++) th.emplace_back(part, p * per, std::min(n + 1, (p + 1) * per));
	part(0, std::min(n + 1, per));
	for (auto &t : th) t.join();
}
} // namespace

extern "C" int dbgk_push_reads_packed(dbgk_handle *h, const uint32_t *packed, const uint64_t *offsets, uint64_t n_reads, uint64_t other_bytes)
{
	if (!h || !offsets || (n_reads && !packed && offsets[n_reads] != offsets[0])) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (h->seed) return DBGK_ERR_ARG; // windows of the seed index are cut at 'N'
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t K = (uint64_t)h->cfg.kmer_size, max_len = (uint64_t)h->cfg.max_read_len;
	const bool streaming = (h->part && !h->sharded) || (h->wpart && !h->wbuilt);
	const bool cut_to_room = streaming && h->store_capacity >= h->cap_bases && !h->wpart;
	const OffsetScan all = scan_offsets(offsets, 0, n_reads, K, max_len); // validates every offset before anything is queued
	if (!all.ok) return DBGK_ERR_ARG;
	const bool all_equal = n_reads && all.len_min == all.len_max;
	const uint64_t src_words = n_reads ? (offsets[n_reads] + 15) >> 4 : 0;
	const bool pinned_source = n_reads && offsets[n_reads] > offsets[0] &&
	                           device_readable_host(reinterpret_cast<const char *>(packed + (offsets[0] >> 4)), (src_words - (offsets[0] >> 4)) * 4);
	bool source_in_flight = false;
	struct WaitSource { // every exit path waits for the copies that still read the caller's buffer
		dbgk_handle *h; bool &on;
		~WaitSource() { if (on && h->source_read) (void)hipEventSynchronize(h->source_read); }
	} wait_source{h, source_in_flight};
	uint64_t r0 = 0;
	while (r0 < n_reads) {
		const uint64_t base0 = offsets[r0];
		const uint64_t room = h->store_capacity > h->pending_kmers ? h->store_capacity - h->pending_kmers : 0;
		const uint64_t r_end = std::min<uint64_t>(n_reads, r0 + h->cap_reads);
		// the longest [r0, r1) whose bases fit the staging buffers (the offsets are known to be monotone)
		uint64_t r1 = (uint64_t)(std::upper_bound(offsets + r0, offsets + r_end + 1, base0 + h->cap_bases) - offsets) - 1;
		if (r1 < n_reads && r1 > r0 + 64 && (offsets[r1] & 15u)) { // end the batch where the next one starts on a word boundary, if that is near
			for (uint64_t back = 1; back <= 64; back++)
				if ((offsets[r1 - back] & 15u) == 0) { r1 -= back; break; }
		}
		OffsetScan st;
		if (all_equal) { // (no second pass over the offsets)
			const uint64_t rl = all.len_max > max_len ? max_len : all.len_max;
			st.windows = (r1 - r0) * (rl >= K ? rl - K + 1 : 0ull);
			st.len_max = st.len_min = all.len_max;
		} else {
			st = scan_offsets(offsets, r0, r1, K, max_len);
		}
		if (cut_to_room && st.windows > room && (r1 > r0 + 1 || h->pending_kmers > 0)) { // the record store takes only part of it
			uint64_t w = 0, r = r0;
			st = OffsetScan();
			for (; r < r1; r++) {
				const uint64_t len = offsets[r + 1] - offsets[r], rl = len > max_len ? max_len : len, wr = rl >= K ? rl - K + 1 : 0ull;
				if (w + wr > room && (r > r0 || h->pending_kmers > 0)) break;
				w += wr;
				st.len_max = std::max(st.len_max, len);
				st.len_min = std::min(st.len_min, len);
			}
			st.windows = w;
			r1 = r;
		}
		const uint64_t batch_windows = st.windows, len_max = st.len_max;
		if (streaming && !(h->wpart && h->wbuilt) && h->pending_kmers > 0 &&
		    (r1 == r0 || (!cut_to_room && h->pending_kmers + batch_windows > h->store_capacity))) {
			rc = flush_records(h);
			if (rc) return rc;
			if (r1 == r0) continue;
		}
		if (r1 == r0) return DBGK_ERR_ARG; // a single read larger than max_batch_bases
		StageSlot &s = h->slots[h->next_slot];
		rc = ensure_slot(h, s);
		if (rc) return rc;
		if (s.busy) {
			HIPCHK(hipEventSynchronize(s.done));
			s.busy = false;
		}
		s.acquired = false;
		const uint64_t nb = offsets[r1] - base0, nr = r1 - r0, n_words = (nb + 15) >> 4;
		const bool direct = pinned_source && (base0 & 15u) == 0;
		std::vector<std::thread> copiers;
		struct Join {
			std::vector<std::thread> &w;
			~Join() { for (auto &t : w) if (t.joinable()) t.join(); }
		} join_copiers{copiers};
		if (n_words && !direct) {
			uint32_t *dst = reinterpret_cast<uint32_t *>(s.h_bases);
			static const int want = getenv("DBGK_COPY_THREADS") ? atoi(getenv("DBGK_COPY_THREADS")) : 8;
			const uint64_t min_piece = 1u << 20; // words
			const uint64_t pieces = want > 1 ? std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)want, n_words / min_piece)) : 1;
			const uint64_t per = (n_words + pieces - 1) / pieces;
			for (uint64_t pc = 1; pc < pieces; pc++) {
				const uint64_t lo = pc * per, hi = std::min(n_words, lo + per);
				if (lo < hi) copiers.emplace_back([=]() { dbgk_internal_shift_packed(packed, base0 + 16 * lo, hi - lo, src_words, dst + lo); });
			}
			dbgk_internal_shift_packed(packed, base0, std::min(n_words, per), src_words, dst);
		}
		const int has_long = len_max > max_len ? 1 : 0;
		const int64_t uniform_len = st.len_min == st.len_max ? (int64_t)st.len_max : 0;
		rebase_offsets(s.h_offsets, offsets + r0, nr, base0);
		for (auto &t : copiers) t.join();
		rc = h2d_batch(h, s, direct ? reinterpret_cast<const char *>(packed + (base0 >> 4)) : s.h_bases, n_words * 4, nr + 1, direct);
		if (rc) return rc;
		source_in_flight = source_in_flight || direct;
		rc = launch_batch(h, nullptr, s.d_offsets, nr, nb, s.d_start, s.d_dead, has_long, uniform_len, len_max, reinterpret_cast<const uint32_t *>(s.d_bases));
		if (rc) return rc;
		h->pending_kmers += batch_windows;
		HIPCHK(hipEventRecord(s.done, h->stream));
		s.busy = true;
		h->next_slot ^= 1;
		r0 = r1;
	}
	h->host_other_bytes += other_bytes;
	return DBGK_OK;
}

extern "C" int dbgk_push_reads_packed_device(dbgk_handle *h, const uint32_t *d_packed, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases)
{
	if (!h || !d_offsets || (n_bases && !d_packed)) return DBGK_ERR_ARG;
	if (((uintptr_t)d_packed & 15u) || ((uintptr_t)d_offsets & 7u)) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (h->seed) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t words = bitmap_words(n_bases);
	if (words > h->dev_bits_words) {
		HIPCHK(hipStreamSynchronize(h->stream));
		if (h->dev_start) (void)hipFree(h->dev_start);
		if (h->dev_dead) (void)hipFree(h->dev_dead);
		h->dev_start = h->dev_dead = nullptr;
		h->dev_bits_words = 0;
		if (hipMalloc(&h->dev_start, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		if (hipMalloc(&h->dev_dead, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		h->dev_bits_words = words;
	}
	if (((h->part && !h->sharded) || (h->wpart && !h->wbuilt)) && h->pending_kmers > 0 && h->pending_kmers + n_bases > h->store_capacity) {
		rc = flush_records(h);
		if (rc) return rc;
	}
	rc = launch_batch(h, nullptr, d_offsets, n_reads, n_bases, h->dev_start, h->dev_dead, -1, -1, 0, d_packed);
	if (rc == DBGK_OK) h->pending_kmers += n_bases;
	return rc;
}

// Reads of ONE length (what a sequencer writes before anything trims them), 2 bits per base, back to back: no offsets travel and no
// statistics pass runs in front of level 1 (launch_batch: no_offsets).  Batches are cut at reads where a word begins.
extern "C" int dbgk_push_reads_packed_uniform(dbgk_handle *h, const uint32_t *packed, uint64_t n_reads, uint32_t read_len, uint64_t other_bytes)
{
	if (!h || (n_reads && read_len && !packed)) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (h->seed) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (n_reads == 0) return DBGK_OK;
	if (read_len == 0) { // records without a sequence count as reads (DBGgraph.cpp:274)
		h->total_reads += n_reads;
		return DBGK_OK;
	}
	const uint64_t L = read_len, K = (uint64_t)h->cfg.kmer_size, max_len = (uint64_t)h->cfg.max_read_len;
	const uint64_t rl = L > max_len ? max_len : L, w_read = rl >= K ? rl - K + 1 : 0ull;
	uint64_t align_reads = 16; // a batch begins where a word begins: at a multiple of 16 / gcd(L, 16) reads
	for (uint64_t g = 16; g >= 1; g >>= 1)
		if (L % g == 0) { align_reads = 16 / g; break; }
	uint64_t per_batch = h->cap_bases / L;
	per_batch -= per_batch % align_reads;
	if (per_batch == 0) return DBGK_ERR_ARG; // reads larger than max_batch_bases
	// (large batches in whole level-1 tiles -- 1024 / Q reads, Q a power of two: no second launch for the reads behind the last whole tile)
	if (per_batch >= 64 * 1024 && align_reads <= 16) align_reads = 1024;
	per_batch -= per_batch % align_reads;
	const bool streaming = (h->part && !h->sharded) || (h->wpart && !h->wbuilt);
	const uint64_t src_words = (n_reads * L + 15) >> 4;
	const bool pinned_source = device_readable_host(reinterpret_cast<const char *>(packed), src_words * 4);
	bool source_in_flight = false;
	struct WaitSource {
		dbgk_handle *h; bool &on;
		~WaitSource() { if (on && h->source_read) (void)hipEventSynchronize(h->source_read); }
	} wait_source{h, source_in_flight};
	int ramp = h->pending_kmers == 0 ? 0 : 4; // index into the opening batch sizes of a fresh job (below); 4: full batches
	for (uint64_t r0 = 0; r0 < n_reads;) {
		uint64_t nr = std::min(per_batch, n_reads - r0);
		// the kernels of a batch cannot start before its copy has ended, and they take about 1.5 times as long as the copy: a job of
		// several batches opens with batches of 1/8, 1/4, 1/2 and 3/4 of the full size, so that the GPU waits for an eighth of a
		// batch's copy before it has work and hardly again (profiles/r05_h2d_region_timeline.txt)
		if (ramp < 4 && n_reads > per_batch) {
			static const uint64_t kEighths[4] = {1, 2, 4, 6};
			const uint64_t want = per_batch / 8 * kEighths[ramp++];
			if (want >= align_reads) nr = std::min(nr, want - want % align_reads);
		}
		if (streaming && h->pending_kmers > 0) {
			const uint64_t room = h->store_capacity > h->pending_kmers ? h->store_capacity - h->pending_kmers : 0;
			if (w_read && nr * w_read > room) { // what the record store still takes, in whole alignment groups; else flush first
				uint64_t fit = room / w_read;
				fit -= fit % align_reads;
				if (fit == 0 || !(h->store_capacity >= h->cap_bases && !h->wpart)) {
					rc = flush_records(h);
					if (rc) return rc;
					continue;
				}
				nr = std::min(nr, fit);
			}
		}
		StageSlot &s = h->slots[h->next_slot];
		rc = ensure_slot(h, s);
		if (rc) return rc;
		if (s.busy) {
			HIPCHK(hipEventSynchronize(s.done));
			s.busy = false;
		}
		s.acquired = false;
		const uint64_t base0 = r0 * L, nb = nr * L, n_words = (nb + 15) >> 4;
		const uint32_t *src = packed + (base0 >> 4); // (base0 is a multiple of 16)
		if (!pinned_source) {
			std::vector<std::thread> copiers;
			staged_copy(s.h_bases, reinterpret_cast<const char *>(src), n_words * 4, copiers);
			for (auto &t : copiers) t.join();
		}
		rc = h2d_batch(h, s, pinned_source ? reinterpret_cast<const char *>(src) : s.h_bases, n_words * 4, 0, pinned_source);
		if (rc) return rc;
		source_in_flight = source_in_flight || pinned_source;
		rc = launch_batch(h, nullptr, nullptr, nr, nb, s.d_start, s.d_dead, L > max_len ? 1 : 0, (int64_t)L, L, reinterpret_cast<const uint32_t *>(s.d_bases));
		if (rc) return rc;
		h->pending_kmers += nr * w_read;
		HIPCHK(hipEventRecord(s.done, h->stream));
		s.busy = true;
		h->next_slot ^= 1;
		r0 += nr;
	}
	h->host_other_bytes += other_bytes;
	return DBGK_OK;
}

extern "C" int dbgk_push_reads_packed_uniform_device(dbgk_handle *h, const uint32_t *d_packed, uint64_t n_reads, uint32_t read_len)
{
	if (!h || (n_reads && read_len && !d_packed)) return DBGK_ERR_ARG;
	if ((uintptr_t)d_packed & 15u) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (h->seed) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (n_reads == 0) return DBGK_OK;
	if (read_len == 0) {
		h->total_reads += n_reads;
		return DBGK_OK;
	}
	const uint64_t L = read_len, K = (uint64_t)h->cfg.kmer_size, max_len = (uint64_t)h->cfg.max_read_len, n_bases = n_reads * L;
	const uint64_t rl = L > max_len ? max_len : L, windows = (rl >= K ? rl - K + 1 : 0ull) * n_reads;
	const uint64_t words = bitmap_words(n_bases);
	if (words > h->dev_bits_words) {
		HIPCHK(hipStreamSynchronize(h->stream));
		if (h->dev_start) (void)hipFree(h->dev_start);
		if (h->dev_dead) (void)hipFree(h->dev_dead);
		h->dev_start = h->dev_dead = nullptr;
		h->dev_bits_words = 0;
		if (hipMalloc(&h->dev_start, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		if (hipMalloc(&h->dev_dead, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		h->dev_bits_words = words;
	}
	if (((h->part && !h->sharded) || (h->wpart && !h->wbuilt)) && h->pending_kmers > 0 && h->pending_kmers + windows > h->store_capacity) {
		rc = flush_records(h);
		if (rc) return rc;
	}
	rc = launch_batch(h, nullptr, nullptr, n_reads, n_bases, h->dev_start, h->dev_dead, L > max_len ? 1 : 0, (int64_t)L, L, d_packed);
	if (rc == DBGK_OK) h->pending_kmers += windows; // (exact here: the lengths are known)
	return rc;
}

// ASCII -> 2-bit on the device (the host twin is dbgk_pack_bases): d_packed gets (n_bases + 15) / 16 words; bytes outside
// ACGTNacgtn become 'A' and are added to the handle's stats.other_bytes
extern "C" int dbgk_pack_bases_device(dbgk_handle *h, const char *d_bases, uint64_t n_bases, uint32_t *d_packed)
{
	if (!h || (n_bases && (!d_bases || !d_packed))) return DBGK_ERR_ARG;
	if (((uintptr_t)d_bases & 15u) || ((uintptr_t)d_packed & 3u)) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (n_bases == 0) return DBGK_OK;
	const uint64_t n_chunks = (n_bases + 15) >> 4;
	hipLaunchKernelGGL(k_pack_bases, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, d_bases, n_bases, d_packed, h->d_ctr);
	hipLaunchKernelGGL(k_count_other_bytes, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, d_bases, n_bases, h->d_ctr);
	HIPCHK(hipGetLastError());
	return DBGK_OK;
}

extern "C" int dbgk_flush(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	return flush_records(h);
}

extern "C" int dbgk_store_room(dbgk_handle *h, uint64_t *pending_kmers, uint64_t *capacity_kmers)
{
	if (!h) return DBGK_ERR_ARG;
	const bool records = h->part || (h->wpart && !h->wbuilt);
	if (pending_kmers) *pending_kmers = records ? h->pending_kmers : 0;
	if (capacity_kmers) *capacity_kmers = records ? h->store_capacity : 0;
	return DBGK_OK;
}

static int read_counters(dbgk_handle *h)
{
	HIPCHK(hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	for (auto &s : h->slots) s.busy = false;
	return collect_spans(h);
}

static void fill_stats(const dbgk_handle *h, dbgk_stats *out)
{
	const Counters &c = *h->h_ctr;
	out->total_reads = h->total_reads;
	out->total_kmers = c.total_kmers;
	out->stored_kmers = c.stored_kmers;
	// + the key-0 node, always present (DBGgraph.cpp:418); of a sharded table only shard 0 reports it
	out->count = c.n_new + ((h->sharded && h->shard_rank != 0) ? 0 : 1);
	if (h->seed) out->count = c.n_new + (c.polyA_links ? 1 : 0); // key 0 is an ordinary key of the seed index
	out->count_conflict = c.n_conflict;
	out->table_slots = h->tslots;
	out->polyA_l_link = (uint32_t)(c.polyA_links & 0xFFFFFFFFu);
	out->polyA_r_link = (uint32_t)(c.polyA_links >> 32);
	out->other_bytes = c.other_bytes + h->host_other_bytes;
}

// PARTITION engine: records -> final buckets -> table regions, then the stragglers
template <int DBG>
static void launch_l2(dbgk_handle *h, int grid, uint32_t j0, uint32_t j1)
{
	if (h->three) { // mid pass, plan of the mid buckets filled so far, final pass (see the handle's comment)
		hipLaunchKernelGGL(k_scatter_l2<0>, dim3(grid), dim3(kL2Threads), sizeof(ScatterLdsL2), h->stream, h->g_mid, h->s_mid, h->tile_prefix, h->d_ctr, j0, j1);
		hipLaunchKernelGGL(k_plan_l2, dim3(1), dim3(kMaxBuckets), 0, h->stream, h->g_fin, h->s_fin, h->tile_prefix2);
		hipLaunchKernelGGL(k_scatter_l2<0>, dim3(grid), dim3(kL2Threads), sizeof(ScatterLdsL2), h->stream, h->g_fin, h->s_fin, h->tile_prefix2, h->d_ctr,
		                   j0 * h->fan_mid, j1 * h->fan_mid);
		return;
	}
	if (h->geom.n2 > 2048u) // tables of 2^33 slots and more
		hipLaunchKernelGGL((k_scatter_l2<0, 4096>), dim3(grid), dim3(kL2Threads), sizeof(ScatterLdsL2T<4096>), h->stream, h->geom, h->store, h->tile_prefix, h->d_ctr, j0, j1);
	else if (h->geom.n2 > (uint32_t)kMaxBuckets) // 2^32 .. 2^33 slots
		hipLaunchKernelGGL((k_scatter_l2<0, 2048>), dim3(grid), dim3(kL2Threads), sizeof(ScatterLdsL2T<2048>), h->stream, h->geom, h->store, h->tile_prefix, h->d_ctr, j0, j1);
	else if (h->geom.kf == 2u) // KFREQ, direct blocks: 32-bit level-1 records (n2 <= 1024 always)
		hipLaunchKernelGGL((k_scatter_l2<0, kMaxBuckets, true>), dim3(grid), dim3(kL2Threads), sizeof(ScatterLdsL2), h->stream, h->geom, h->store, h->tile_prefix, h->d_ctr, j0, j1);
	else
		hipLaunchKernelGGL(k_scatter_l2<DBG>, dim3(grid), dim3(kL2Threads), sizeof(ScatterLdsL2), h->stream, h->geom, h->store, h->tile_prefix, h->d_ctr, j0, j1);
}

static RedoList redo_list(dbgk_handle *h)
{
	return RedoList{h->region_cursor + kMaxBuildLaunches + 2, h->region_cursor + kMaxBuildLaunches + 1, h->geom.n_regions_own};
}

// FAST form of the insert (four records per thread in flight, plain adds on the link words, regions whose counters pass 255
// left to the exact pass) unless DBGK_BUILD_EXACT=1 asks for the saturating CAS loops everywhere
static bool build_fast()
{
	const bool exact = dbgk_hook("build_exact") && atoi(dbgk_hook("build_exact")) != 0;
	return !exact;
}

template <int DBG>
static void launch_build(dbgk_handle *h, hipStream_t stream, uint32_t first_region, uint32_t n_regions, unsigned int *cursor)
{
	static const int per_cu = DBGK_EXPERIMENT_ENV("DBGK_BUILD_PER_CU") ? std::max(1, atoi(DBGK_EXPERIMENT_ENV("DBGK_BUILD_PER_CU"))) : 2; // tuning knob
	const uint32_t grid = std::min<uint32_t>(n_regions, (uint32_t)h->n_cu * (uint32_t)per_cu); // persistent: two 66-KiB workgroups fit a CU
	const RedoList redo = redo_list(h);
	if (h->geom.kf == 2u) { // KFREQ, direct blocks
#define DBGK_KFB(INCR, FAST) \
	hipLaunchKernelGGL((k_kf_build_blocks<INCR, FAST>), dim3(grid), dim3(kBuildThreads), sizeof(KfBlockLds), stream, h->geom, h->store, h->counts, \
	                   h->d_ctr, first_region, n_regions, cursor, redo)
		if (build_fast()) { if (h->incr) DBGK_KFB(true, true); else DBGK_KFB(false, true); }
		else { if (h->incr) DBGK_KFB(true, false); else DBGK_KFB(false, false); }
#undef DBGK_KFB
		return;
	}
	Node *counts = reinterpret_cast<Node *>(h->counts);
#define DBGK_BUILD(D, KF, INCR, FAST, TABLE) \
	hipLaunchKernelGGL((k_build_regions<D, KF, INCR, FAST>), dim3(grid), dim3(kBuildThreads), sizeof(BuildLds), stream, h->geom, h->store, TABLE, h->d_ctr, \
	                   first_region, n_regions, cursor, redo)
	if (build_fast()) {
		if (h->kfreq && h->incr) DBGK_BUILD(0, true, true, true, counts);
		else if (h->kfreq) DBGK_BUILD(0, true, false, true, counts);
		else if (h->incr && DBG == 0) DBGK_BUILD(0, false, true, true, h->table);
		else DBGK_BUILD(DBG, false, false, true, h->table);
	} else {
		if (h->kfreq && h->incr) DBGK_BUILD(0, true, true, false, counts);
		else if (h->kfreq) DBGK_BUILD(0, true, false, false, counts);
		else if (h->incr && DBG == 0) DBGK_BUILD(0, false, true, false, h->table);
		else DBGK_BUILD(DBG, false, false, false, h->table);
	}
#undef DBGK_BUILD
}

// the exact pass over the regions the fast launches flagged (normally none: the kernel finds an empty list and returns)
static void launch_build_redo(dbgk_handle *h, hipStream_t stream)
{
	const uint32_t grid = (uint32_t)h->n_cu * 2u;
	const RedoList redo = redo_list(h);
	unsigned int *cursor = h->region_cursor + kMaxBuildLaunches;
	if (h->geom.kf == 2u) {
		if (h->incr)
			hipLaunchKernelGGL((k_kf_build_blocks<true, false, true>), dim3(grid), dim3(kBuildThreads), sizeof(KfBlockLds), stream, h->geom, h->store,
			                   h->counts, h->d_ctr, 0u, 0u, cursor, redo);
		else
			hipLaunchKernelGGL((k_kf_build_blocks<false, false, true>), dim3(grid), dim3(kBuildThreads), sizeof(KfBlockLds), stream, h->geom, h->store,
			                   h->counts, h->d_ctr, 0u, 0u, cursor, redo);
		return;
	}
	Node *counts = reinterpret_cast<Node *>(h->counts);
#define DBGK_REDO(KF, INCR, TABLE) \
	hipLaunchKernelGGL((k_build_regions<0, KF, INCR, false, true>), dim3(grid), dim3(kBuildThreads), sizeof(BuildLds), stream, h->geom, h->store, TABLE, \
	                   h->d_ctr, 0u, 0u, cursor, redo)
	if (h->kfreq && h->incr) DBGK_REDO(true, true, counts);
	else if (h->kfreq) DBGK_REDO(true, false, counts);
	else if (h->incr) DBGK_REDO(false, true, h->table);
	else DBGK_REDO(false, false, h->table);
#undef DBGK_REDO
}

// Level 2 is bound by the memory system (8 waves per CU and 72 KiB of LDS reach the same time as a
// full CU), the region build by instruction issue: they run CONCURRENTLY.  The own level-1 buckets
// are cut into chunks; level 2 of chunk c+1 runs on `stream` while the regions of chunk c are built
// on `stream2` (one 512-thread level-2 workgroup and one 1024-thread build workgroup fit a CU together).
// ---- pieces of the finalize of the PARTITION engine ---------------------------------------------
// part_plan: level-2 tile plan for all own buckets (needs every inbox fill count);
// part_build_range: level 2 + region build of the own buckets [j0, j1), asynchronous, level 2 on
// `stream`, the build behind it on `stream2`; part_finish: join, spill / overflow fix-ups.
// EARLY level 2.  Level 2 is an append into the final buckets, so it does not have to wait for the end of the input: every
// push first queues a level-2 round over the records the batches before it left in the level-1 buckets (tile plan from the fill
// counts minus what earlier rounds took, P.l2_done), then its own level 1.  A job whose batches come over the link (the reference
// overlaps reading and parsing the same way, DBGgraph.cpp:233-296) keeps the GPU busy with level 2 while the next batch travels;
// after the last batch only that batch's level 2 and the region build remain.  A job that is pushed in one piece (bench.py's
// resident step) is unchanged: its only round runs at dbgk_finalize.  Rounds are only worth their tiles' fixed costs when there
// is something to scatter: at least kEarlyL2Min occurrences since the last one.
constexpr uint64_t kEarlyL2Min = 8ull << 20;
static int early_l2(dbgk_handle *h)
{
	if (!h->part || !h->l2_done || h->part_planned || h->part_built) return DBGK_OK;
	const char *e_min = dbgk_hook("early_l2_min"); // (read per call: the tests ask for a round after every small batch)
	const uint64_t min_kmers = e_min ? strtoull(e_min, nullptr, 10) : kEarlyL2Min;
	if (h->pending_kmers < h->l2_seen_kmers + std::max<uint64_t>(min_kmers, 1)) return DBGK_OK;
	const PartGeom &G = h->geom;
	static const int l2_grid_env = DBGK_EXPERIMENT_ENV("DBGK_L2_GRID") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_L2_GRID")) : 0;
	const int l2_grid = l2_grid_env >= 8 ? (l2_grid_env & ~7) : h->n_cu;
	TimedSpan sp;
	int rc = span_begin(h, PH_PARTITION, sp);
	if (rc) return rc;
	hipLaunchKernelGGL(k_plan_l2, dim3(1), dim3(kMaxBuckets), 0, h->stream, h->geom, h->store, h->tile_prefix);
	launch_l2<0>(h, l2_grid, 0, G.nb_own);
	HIPCHK(hipGetLastError());
	h->l2_seen_kmers = h->pending_kmers;
	return span_end(h, sp);
}

static int part_plan(dbgk_handle *h)
{
	if (h->part_planned) return DBGK_OK;
	if (!h->stream2) {
		HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
		HIPCHK(hipEventCreateWithFlags(&h->join_ev, hipEventDisableTiming));
	}
	hipLaunchKernelGGL(k_plan_l2, dim3(1), dim3(kMaxBuckets), 0, h->stream, h->geom, h->store, h->tile_prefix);
	HIPCHK(hipGetLastError());
	h->cursors_used = 0; // (the work cursors were zeroed with the record stores' control arrays: clear_record_store)
	int rc = span_begin(h, PH_L2_BUILD_WALL, h->wall_span);
	if (rc) return rc;
	h->part_planned = true;
	h->next_bucket = 0;
	h->chunks_used = 0;
	return DBGK_OK;
}

static int part_build_range(dbgk_handle *h, uint32_t j0, uint32_t j1, bool two_streams)
{
	const PartGeom &G = h->geom;
	static const int l2_grid_env = DBGK_EXPERIMENT_ENV("DBGK_L2_GRID") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_L2_GRID")) : 0; // tuning knob: level-2 workgroups (multiple of 8)
	const int l2_grid = l2_grid_env >= 8 ? (l2_grid_env & ~7) : h->n_cu;
	static const int dbg_l2 = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_L2") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_L2")) : 0;       // timing experiments, results are wrong
	static const int dbg_build = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_BUILD") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_BUILD")) : 0;
	if (j0 >= j1) return DBGK_OK;
	hipStream_t bstream = two_streams ? h->stream2 : h->stream;
	TimedSpan sp;
	int rc = span_begin(h, PH_PARTITION, sp);
	if (rc) return rc;
	switch (dbg_l2) {
#ifdef DBGK_EXPERIMENTS
		case 1: launch_l2<1>(h, l2_grid, j0, j1); break;
		case 2: launch_l2<2>(h, l2_grid, j0, j1); break;
		case 3: launch_l2<3>(h, l2_grid, j0, j1); break;
#endif
		default: launch_l2<0>(h, l2_grid, j0, j1); break;
	}
	HIPCHK(hipGetLastError());
	rc = span_end(h, sp);
	if (rc) return rc;
	h->next_bucket = j1;
	if (dbg_l2) return DBGK_OK; // never build regions from the garbage a timing experiment leaves behind
	if (two_streams) {
		if (h->chunk_ev.size() <= h->chunks_used) {
			hipEvent_t e;
			HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			h->chunk_ev.push_back(e);
		}
		hipEvent_t ev = h->chunk_ev[h->chunks_used++];
		HIPCHK(hipEventRecord(ev, h->stream));
		HIPCHK(hipStreamWaitEvent(bstream, ev, 0));
	}
	const uint32_t r0 = j0 * G.n2, r1 = std::min(j1 * G.n2, G.n_regions_own);
	if (r1 <= r0) return DBGK_OK;
	rc = span_begin(h, PH_BUILD, sp, bstream);
	if (rc) return rc;
	if (h->cursors_used >= kMaxBuildLaunches) {
		g_last_error = "too many build ranges in one step";
		return DBGK_ERR_STATE;
	}
	unsigned int *cursor = h->region_cursor + h->cursors_used++;
	switch (dbg_build) {
#ifdef DBGK_EXPERIMENTS
		case 1: launch_build<1>(h, bstream, r0, r1 - r0, cursor); break;
		case 2: launch_build<2>(h, bstream, r0, r1 - r0, cursor); break;
		case 3: launch_build<3>(h, bstream, r0, r1 - r0, cursor); break;
#endif
		default: launch_build<0>(h, bstream, r0, r1 - r0, cursor); break;
	}
	HIPCHK(hipGetLastError());
	return span_end(h, sp, bstream);
}

// Level 2 is bound by the memory system (8 waves per CU and 72 KiB of LDS reach the same time as a
// full CU), the region build by instruction issue: they run CONCURRENTLY.  The own level-1 buckets
// are cut into chunks; level 2 of chunk c+1 runs on `stream` while the regions of chunk c are built
// on `stream2` (one 512-thread level-2 workgroup and one 1024-thread build workgroup fit a CU together).
static int build_from_records(dbgk_handle *h)
{
	const PartGeom &G = h->geom;
	static const int dbg_l2 = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_L2") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_L2")) : 0;
	static const int dbg_build = DBGK_EXPERIMENT_ENV("DBGK_DEBUG_BUILD") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_DEBUG_BUILD")) : 0;
	// 1 = level 2, then the build.  Round 5: with the leaner build and 16-byte record loads either kernel alone runs close to what the
	// memory system gives this traffic (4.1 + 3.7 ms for 2 x 19.2 GB), side by side they only share it: 1 / 2 / 3 / 4 / 6 / 12 chunk pairs ->
	// 12.86-12.92 / 12.86 / 13.00 / 13.03 / 13.07 / 13.06-13.24 ms per cfg2 step (profiles/r05_build_lean_walk_and_chunks_ab.txt; rounds 2-4: 12)
	static const int want_chunks = DBGK_EXPERIMENT_ENV("DBGK_OVERLAP_CHUNKS") ? atoi(DBGK_EXPERIMENT_ENV("DBGK_OVERLAP_CHUNKS")) : 1;
	uint32_t n_chunks = ((dbg_l2 | dbg_build) != 0 || want_chunks < 1) ? 1u : (uint32_t)want_chunks;
	int rc = part_plan(h);
	if (rc) return rc;
	// whatever the caller has not built by ranges yet (dbgk_shard_build_range): all of it, normally
	const uint32_t left = G.nb_own - std::min(h->next_bucket, G.nb_own);
	if (n_chunks > left) n_chunks = left ? left : 1u;
	const bool two_streams = n_chunks > 1 || h->chunks_used > 0;
	const uint32_t per = (left + n_chunks - 1) / n_chunks;
	for (uint32_t c = 0; c < n_chunks && h->next_bucket < G.nb_own; c++) {
		const uint32_t j0 = h->next_bucket, j1 = std::min(j0 + per, G.nb_own);
		rc = part_build_range(h, j0, j1, two_streams);
		if (rc) return rc;
	}
	if (h->chunks_used > 0) { // everything after this point is ordered behind the last build on `stream` again
		HIPCHK(hipEventRecord(h->join_ev, h->stream2));
		HIPCHK(hipStreamWaitEvent(h->stream, h->join_ev, 0));
	}
	rc = span_end(h, h->wall_span);
	if (rc) return rc;
	h->part_planned = false;
	if ((dbg_l2 | dbg_build) != 0) {
		g_last_error = "DBGK_DEBUG_L2 / DBGK_DEBUG_BUILD set: timing experiment, no valid table was built";
		return DBGK_ERR_STATE;
	}
	TimedSpan sp;
	h->zero_pending = false; // every slot has just been written
	rc = span_begin(h, PH_FIXUP, sp);
	if (rc) return rc;
	if (build_fast()) launch_build_redo(h, h->stream); // regions with a link counter beyond 255: rebuilt exactly (before their spill nodes are merged)
	if (h->kfreq) {
		Counters *track = h->kf_blocks ? h->d_ctr : nullptr; // direct blocks: the table summary is kept as the table is written
		hipLaunchKernelGGL(k_kf_apply, dim3(h->n_cu), dim3(kBlock), 0, h->stream, h->store.spill, &h->store.ovf_n[1], h->store.spill_cap, 0,
		                   reinterpret_cast<uint32_t *>(h->counts), track);
		hipLaunchKernelGGL(k_kf_apply, dim3(h->n_cu), dim3(kBlock), 0, h->stream, h->store.ovf, &h->store.ovf_n[0], h->store.ovf_cap, 1,
		                   reinterpret_cast<uint32_t *>(h->counts), track);
		hipLaunchKernelGGL(k_kf_apply_table, dim3(grid_for(h, h->store.hh_size)), dim3(kBlock), 0, h->stream, h->store.hh, h->store.hh_size,
		                   reinterpret_cast<uint32_t *>(h->counts), track);
		hipLaunchKernelGGL(k_kf_key0, dim3(1), dim3(64), 0, h->stream, h->d_ctr, h->counts, h->kf_blocks ? 1 : 0);
	} else if (!h->sharded) {
		hipLaunchKernelGGL(k_merge_spill, dim3(h->n_cu), dim3(kBlock), 0, h->stream, h->store.spill, &h->store.ovf_n[1], h->store.spill_cap,
		                   h->tref(), h->d_ctr);
		hipLaunchKernelGGL(k_insert_triples, dim3(h->n_cu), dim3(kBlock), 0, h->stream, h->store.ovf, &h->store.ovf_n[0], h->store.ovf_cap,
		                   h->tref(), h->d_ctr);
		// the aggregated surplus of heavy hitters (empty slots are all-zero records and add nothing)
		hipLaunchKernelGGL(k_merge_nodes, dim3(grid_for(h, h->store.hh_size)), dim3(kBlock), 0, h->stream, h->store.hh, h->store.hh_size,
		                   h->tref(), h->d_ctr, (const unsigned long long *)h->store.ovf_n, (unsigned long long)h->store.ovf_cap); // (in use only once the overflow list is full)
	} else {
		// spill nodes of this shard's regions stay in the shard unless they run off its end (-> outgoing);
		// overflow triples may belong to any shard: the caller exchanges them (dbgk_shard_overflow)
		hipLaunchKernelGGL(k_merge_sharded, dim3(h->n_cu), dim3(kBlock), 0, h->stream, h->store.spill, &h->store.ovf_n[1], (uint64_t)0,
		                   h->store.spill_cap, 0, 0, G, h->store, h->table, h->d_ctr);
	}
	HIPCHK(hipGetLastError());
	rc = span_end(h, sp);
	h->part_built = true;
	return rc;
}

static int kfreq_summary(dbgk_handle *h, uint64_t first, uint64_t n, unsigned long long res[2]);

extern "C" int dbgk_finalize(dbgk_handle *h, dbgk_stats *out)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (h->wpart && !h->finalized) {
		if (h->wmulti) { // shards / passes: the current pass is completed here, all passes must have run
			if (h->wpass_open) {
				rc = wide_end_pass(h);
				if (rc) return rc;
			}
			if (h->wpasses_done != h->wgeom.n_passes) {
				g_last_error = "dbgk_finalize: this WIDE handle reads its input in several passes (dbgk_wide_pass_info) and not all of them have run";
				return DBGK_ERR_STATE;
			}
			if (!h->wbuilt) {
				rc = wide_finish_records(h);
				if (rc) return rc;
			}
		} else if (!h->wbuilt && h->pending_kmers > 0) {
			rc = wide_build_from_records(h);
			if (rc) return rc;
		}
		rc = wide_ensure_zero(h); // nothing was pushed at all
		if (rc) return rc;
	}
	if (h->part && !h->part_built && !h->finalized) {
		if (h->sharded && !h->exchanged) {
			g_last_error = "sharded handle: exchange the level-1 buckets (dbgk_shard_buffers) and call dbgk_shard_mark_exchanged first";
			return DBGK_ERR_STATE;
		}
		if (!(h->incr && h->pending_kmers == 0 && !h->sharded)) { // (a flushed handle with nothing new: the table is complete)
			rc = build_from_records(h);
			if (rc) return rc;
			h->pending_kmers = 0;
		}
	}
	const bool kf_tracked = h->kfreq && h->part && h->kf_blocks; // direct blocks: the summary is kept while the table is written
	if (h->kfreq && !kf_tracked) {
		unsigned long long res[2];
		rc = kfreq_summary(h, 0, h->n_counts, res);
		if (rc) return rc;
		h->kf_distinct = res[0];
		h->kf_sum = res[1];
	}
	rc = read_counters(h);
	if (rc) return rc;
	if (kf_tracked) {
		h->kf_distinct = h->h_ctr->kf_nonzero;
		h->kf_sum = h->h_ctr->kf_sum;
	}
	h->finalized = true;
	if (out) fill_stats(h, out);
	if (h->kfreq) {
		if (out) {
			out->count = h->kf_distinct;
			out->count_conflict = 0;
			out->table_slots = h->n_counts;
		}
		if (h->h_ctr->error & 2u) return DBGK_ERR_CAPACITY; // through PARTITION: far more distinct k-mers than expected_kmers / 2
		return DBGK_OK;
	}
	if (h->h_ctr->error & 1u) return DBGK_ERR_TABLE_FULL;
	if (h->h_ctr->error & 2u) return DBGK_ERR_CAPACITY; // PARTITION overflow stores exhausted (expected_kmers too small)
	if (h->h_ctr->n_new + 1 > h->tslots) return DBGK_ERR_TABLE_FULL; // no free slot left for the key-0 node
	return DBGK_OK;
}

extern "C" int dbgk_refresh_stats(dbgk_handle *h, dbgk_stats *out)
{
	if (!h || !out) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	rc = read_counters(h);
	if (rc) return rc;
	fill_stats(h, out);
	if (h->kfreq && h->finalized) { // distinct canonical k-mers, as dbgk_finalize reports them
		out->count = h->kf_distinct;
		out->count_conflict = 0;
		out->table_slots = h->n_counts;
	}
	return (h->h_ctr->error & 1u) ? DBGK_ERR_TABLE_FULL : DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// results
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_build_flags_ctr(const Node *__restrict__ nodes, uint64_t size,
                                                            const Counters *__restrict__ ctr, uint8_t *__restrict__ flags)
{
	const uint64_t polyA_slot = ctr->polyA_slot;
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

// A large device-to-host copy into ORDINARY (pageable, malloc()ed) host memory -- the host KmerSet must be free()-able by the
// consumer.  hipMemcpy into pageable memory stages through the runtime's own bounce buffer on one thread (~20 GB/s); here the
// device fills pinned slices at the link's rate and several host threads move them on (their first touch also spreads the
// page faults of the fresh allocation).  Everything queued on the handle's stream before the call is complete on return.
static int d2h_pipelined(dbgk_handle *h, void *dst, const void *d_src, size_t bytes)
{
	constexpr size_t kSlice = 32ull << 20;
	constexpr int kBuffers = 8;
	static const int n_threads = getenv("DBGK_EXPORT_THREADS") ? std::max(1, atoi(getenv("DBGK_EXPORT_THREADS"))) : 6;
	if (bytes < 8 * kSlice || DBGK_EXPERIMENT_ENV("DBGK_EXPORT_PLAIN")) {
		HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		return DBGK_OK;
	}
	while (h->d2h_stage.size() < (size_t)kBuffers) {
		void *p = nullptr;
		hipEvent_t e = nullptr;
		if (hipHostMalloc(&p, kSlice, hipHostMallocDefault) != hipSuccess) return DBGK_ERR_NOMEM;
		h->d2h_stage.push_back(p);
		HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		h->d2h_ev.push_back(e);
	}
	const size_t n_slices = (bytes + kSlice - 1) / kSlice;
	std::vector<std::atomic<int>> issued(n_slices), done(n_slices);
	for (size_t i = 0; i < n_slices; i++) { issued[i].store(0); done[i].store(0); }
	std::atomic<int> failed{0};
	std::vector<std::thread> workers;
	for (int w = 0; w < n_threads; w++)
		workers.emplace_back([&, w]() {
			if (hipSetDevice(h->device) != hipSuccess) { failed.store(1); return; }
			for (size_t i = (size_t)w; i < n_slices; i += (size_t)n_threads) {
				while (!issued[i].load(std::memory_order_acquire)) {
					if (failed.load()) return;
					std::this_thread::yield();
				}
				const int b = (int)(i % kBuffers);
				if (hipEventSynchronize(h->d2h_ev[b]) != hipSuccess) { failed.store(1); return; }
				const size_t off = i * kSlice, len = std::min(kSlice, bytes - off);
				memcpy(static_cast<char *>(dst) + off, h->d2h_stage[b], len);
				done[i].store(1, std::memory_order_release);
			}
		});
	int rc = DBGK_OK;
	for (size_t i = 0; i < n_slices && rc == DBGK_OK; i++) {
		if (i >= (size_t)kBuffers)
			while (!done[i - kBuffers].load(std::memory_order_acquire)) { // its buffer is free again
				if (failed.load()) { rc = DBGK_ERR_HIP; break; }
				std::this_thread::yield();
			}
		if (rc) break;
		const int b = (int)(i % kBuffers);
		const size_t off = i * kSlice, len = std::min(kSlice, bytes - off);
		if (hipMemcpyAsync(h->d2h_stage[b], static_cast<const char *>(d_src) + off, len, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
		    hipEventRecord(h->d2h_ev[b], h->stream) != hipSuccess) {
			rc = DBGK_ERR_HIP;
			break;
		}
		issued[i].store(1, std::memory_order_release);
	}
	if (rc) failed.store(1);
	for (auto &t : workers) t.join();
	if (failed.load() && rc == DBGK_OK) rc = DBGK_ERR_HIP;
	if (rc) return hip_fail(hipGetLastError(), "d2h_pipelined", __LINE__);
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

// ---- the occupied nodes only ----------------------------------------------------------------------------------------------------
// A host table at the reference's load (-i: 0.3 - 0.6 of the slots hold a node) is mostly zeros: the copy above moves all of it over
// the link.  Here the device packs the occupied nodes, in slot order, into one stream (k_compact_nodes), only that stream and the
// occupancy bits cross the link, and the host threads that used to memcpy() the slices now lay the nodes out at their slots from the
// bits (zeros in between).  Same bytes in `array` and `nul_flag` as the plain copy (tests/test_gpu_parity.py compares the two).
constexpr uint32_t kCompactSpan = 4096; // slots per wavefront

__global__ __launch_bounds__(256) void k_flag_block_counts(const uint32_t *__restrict__ flags32, uint64_t n_dwords, uint64_t n_spans,
                                                           uint32_t *__restrict__ counts)
{
	const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
	const uint32_t lane = threadIdx.x & 63u;
	if (wave >= n_spans) return;
	uint32_t c = 0;
#pragma unroll
	for (uint32_t j = 0; j < kCompactSpan / 32 / 64; j++) {
		const uint64_t d = wave * (kCompactSpan / 32) + j * 64 + lane;
		if (d < n_dwords) c += (uint32_t)__popc(flags32[d]);
	}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
	if (lane == 0) counts[wave] = c;
}

__global__ __launch_bounds__(256) void k_compact_nodes(const Node *__restrict__ nodes, uint64_t size, const Counters *__restrict__ ctr,
                                                       const uint64_t *__restrict__ span_first, uint64_t n_spans, Node *__restrict__ out)
{
	const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
	const uint32_t lane = threadIdx.x & 63u;
	if (wave >= n_spans) return;
	const uint64_t polyA_slot = ctr->polyA_slot;
	uint64_t at = span_first[wave];
	const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(nodes);
	ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(out);
	for (uint32_t it = 0; it < kCompactSpan / 64; it++) {
		const uint64_t i = wave * kCompactSpan + it * 64 + lane;
		ulonglong2 nd = make_ulonglong2(0ull, 0ull);
		bool occ = false;
		if (i < size) {
			nd = src[i];
			occ = nd.x != 0ull || i == polyA_slot; // what k_build_flags_ctr calls occupied
		}
		const uint64_t m = __ballot(occ);
		if (occ) dst[at + (uint64_t)__popcll(m & ((1ull << lane) - 1ull))] = nd;
		at += (uint64_t)__popcll(m);
	}
}

// array[0, host_size) and nul_flag[0, host_size / 8 + 1) from the image T whose occupancy bits are d_flags (padded to whole dwords,
// the padding zero).  DBGK_ERR_STATE: "use the plain copy" (no memory for the stream, or the counts disagree).
// Page-locked memory: the staging buffers of the handle's batches where they exist -- idle once a table is finalized, and
// page-locking fresh memory for one copy costs about what the copy costs -- else the handle's own export buffers (d2h_stage).
static int d2h_compact(dbgk_handle *h, dbgk_node *array, uint8_t *nul_flag, const TableRef &T, const uint8_t *d_flags, uint64_t n_occ)
{
	constexpr size_t kSlice = 8ull << 20, kOwnSlice = 32ull << 20; // (d2h_stage holds pieces of 32 MiB: four slices each)
	constexpr size_t kMinBuffers = 16, kOwnBuffers = 8, kMaxBuffers = 48;
	static const int n_threads = getenv("DBGK_EXPORT_THREADS") ? std::max(1, atoi(getenv("DBGK_EXPORT_THREADS"))) : 12;
	const uint64_t size = T.size, n_spans = (size + kCompactSpan - 1) / kCompactSpan, n_flag_bytes = size / 8 + 1;
	static const bool lap_wanted = getenv("DBGK_TIMINGS") != nullptr;
	auto clock_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double laps[5] = {0, 0, 0, 0, 0}, lap_t = clock_s();
	auto lap = [&](int i) { const double t = clock_s(); laps[i] += t - lap_t; lap_t = t; };
	std::atomic<uint64_t> wait_link_us{0}, wait_host_us{0};
	Node *d_stream = nullptr;
	uint32_t *d_counts = nullptr;
	uint64_t *d_first = nullptr;
	auto cleanup = [&]() {
		for (void *p : {(void *)d_stream, (void *)d_counts, (void *)d_first})
			if (p) (void)hipFree(p);
	};
	if (hipMalloc(&d_stream, (n_occ + 1) * sizeof(Node)) != hipSuccess || hipMalloc(&d_counts, n_spans * 4) != hipSuccess ||
	    hipMalloc(&d_first, n_spans * 8) != hipSuccess) {
		(void)hipGetLastError();
		cleanup();
		return DBGK_ERR_STATE;
	}
	// page-locked pieces: [the occupancy bits] + the slices the stream passes through
	std::vector<char *> bufs;
	uint8_t *bits_pinned = nullptr;
	const size_t flag_room = (size_t)((n_flag_bytes + kSlice - 1) / kSlice) * kSlice;
	for (StageSlot &sl : h->slots) {
		if (!sl.h_bases || sl.acquired) continue;
		if (sl.busy && hipEventQuery(sl.done) != hipSuccess) continue; // (a batch still on its way: not after dbgk_finalize)
		size_t off = 0;
		if (!bits_pinned && h->cap_bases >= flag_room + kSlice) {
			bits_pinned = reinterpret_cast<uint8_t *>(sl.h_bases);
			off = flag_room;
		}
		for (; off + kSlice <= h->cap_bases && bufs.size() < kMaxBuffers; off += kSlice) bufs.push_back(sl.h_bases + off);
	}
	if (bufs.size() < kMinBuffers) {
		while (h->d2h_stage.size() < kOwnBuffers) {
			void *p = nullptr;
			if (hipHostMalloc(&p, kOwnSlice, hipHostMallocDefault) != hipSuccess) { cleanup(); return DBGK_ERR_NOMEM; }
			h->d2h_stage.push_back(p);
		}
		for (void *p : h->d2h_stage)
			for (size_t off = 0; off + kSlice <= kOwnSlice && bufs.size() < kMaxBuffers; off += kSlice) bufs.push_back(static_cast<char *>(p) + off);
	}
	while (h->d2h_ev.size() < bufs.size()) {
		hipEvent_t ev = nullptr;
		if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { cleanup(); return DBGK_ERR_HIP; }
		h->d2h_ev.push_back(ev);
	}
	const size_t n_bufs = bufs.size();
	lap(0);
	const unsigned int span_grid = (unsigned int)((n_spans + 3) / 4);
	hipLaunchKernelGGL(k_flag_block_counts, dim3(span_grid), dim3(256), 0, h->stream, reinterpret_cast<const uint32_t *>(d_flags),
	                   (n_flag_bytes + 3) / 4, n_spans, d_counts);
	std::vector<uint32_t> counts(n_spans);
	std::vector<uint64_t> first(n_spans + 1);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(counts.data(), d_counts, n_spans * 4, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) { cleanup(); return hip_fail(e, "d2h_compact(counts)", __LINE__); }
	first[0] = 0;
	for (uint64_t b = 0; b < n_spans; b++) first[b + 1] = first[b] + counts[b];
	if (first[n_spans] != n_occ) { // (never: the counters and the bits describe the same table)
		cleanup();
		return DBGK_ERR_STATE;
	}
	lap(1);
	e = hipMemcpyAsync(d_first, first.data(), n_spans * 8, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(bits_pinned ? bits_pinned : nul_flag, d_flags, n_flag_bytes, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_compact_nodes, dim3(span_grid), dim3(256), 0, h->stream, T.nodes, size, h->d_ctr, d_first, n_spans, d_stream);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream); // the bits are on the host: the threads below read them
	if (e != hipSuccess) { cleanup(); return hip_fail(e, "d2h_compact(bits)", __LINE__); }
	lap(2);
	// slices of the stream small enough that every thread gets several (the zeros between the nodes are written by the thread
	// whose nodes come next, so the slot ranges follow the ranks)
	const uint64_t per = std::min<uint64_t>(kSlice / sizeof(Node), std::max<uint64_t>(1u << 16, n_occ / (uint64_t)(4 * n_threads) + 1));
	const size_t n_slices = (size_t)((n_occ + per - 1) / per);
	const uint8_t *bits_src = bits_pinned ? bits_pinned : nul_flag;
	auto occupied = [&](uint64_t slot) { return (bits_src[slot >> 3] >> (7u - (uint32_t)(slot & 7u))) & 1u; };
	auto slot_of_rank = [&](uint64_t r) -> uint64_t { // the slot of the r-th occupied node (r < n_occ)
		const uint64_t b = (uint64_t)(std::upper_bound(first.begin(), first.end(), r) - first.begin()) - 1;
		uint64_t slot = b * kCompactSpan, left = r - first[b];
		for (;; slot++)
			if (occupied(slot)) {
				if (left == 0) return slot;
				left--;
			}
	};
	if (DBGK_EXPERIMENT_ENV("DBGK_EXPORT_PROBE")) { // (measurements) the link alone: the same copies with nobody reading the buffers
		const double t0 = clock_s();
		for (size_t i = 0; i < n_slices; i++) {
			const uint64_t r0 = (uint64_t)i * per, r1 = std::min(n_occ, r0 + per);
			(void)hipMemcpyAsync(bufs[i % n_bufs], d_stream + r0, (size_t)(r1 - r0) * sizeof(Node), hipMemcpyDeviceToHost, h->stream);
		}
		(void)hipStreamSynchronize(h->stream);
		const double t1 = clock_s();
		fprintf(stderr, "dbgk export probe: %zu copies of %.1f MB back to back %.4f s (%.1f GB/s)\n", n_slices, (double)per * 16e-6, t1 - t0,
		        (double)n_occ * 16e-9 / (t1 - t0));
		lap_t = clock_s();
	}
	std::vector<std::atomic<int>> issued(n_slices), done(n_slices);
	for (size_t i = 0; i < n_slices; i++) { issued[i].store(0); done[i].store(0); }
	std::atomic<int> failed{0};
	dbgk_node *dst = array;
	std::vector<std::thread> workers;
	for (int w = 0; w < n_threads; w++)
		workers.emplace_back([&, w]() {
			if (hipSetDevice(h->device) != hipSuccess) { failed.store(1); return; }
			if (bits_pinned) { // this thread's share of the bits -> the caller's nul_flag
				const uint64_t chunk = (n_flag_bytes + (uint64_t)n_threads - 1) / (uint64_t)n_threads;
				const uint64_t a = std::min(n_flag_bytes, chunk * (uint64_t)w), b = std::min(n_flag_bytes, a + chunk);
				memcpy(nul_flag + a, bits_pinned + a, (size_t)(b - a));
			}
			for (size_t i = (size_t)w; i < n_slices; i += (size_t)n_threads) {
				const uint64_t r0 = (uint64_t)i * per, r1 = std::min(n_occ, r0 + per);
				uint64_t slot = i == 0 ? 0 : slot_of_rank(r0);
				const uint64_t slot_end = i + 1 == n_slices ? size : slot_of_rank(r1);
				while (!issued[i].load(std::memory_order_acquire)) {
					if (failed.load()) return;
					std::this_thread::yield();
				}
				const size_t b = i % n_bufs;
				const double t_w = lap_wanted ? clock_s() : 0;
				if (hipEventSynchronize(h->d2h_ev[b]) != hipSuccess) { failed.store(1); return; }
				if (lap_wanted) wait_link_us += (uint64_t)((clock_s() - t_w) * 1e6);
				const dbgk_node *const src0 = reinterpret_cast<const dbgk_node *>(bufs[b]);
				const dbgk_node *src = src0;
				const dbgk_node zero{0, 0, 0};
				for (; slot < slot_end && (slot & 63u); slot++) dst[slot] = occupied(slot) ? *src++ : zero;
				// 64 slots at a time: zeros and the nodes the bits name (first slot = top bit) are put together in a buffer of one KiB
				// and leave with non-temporal stores -- the table is written once and not read here: no line is fetched for ownership
				const bool stream_out = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
				for (; slot + 64 <= slot_end; slot += 64) {
					uint64_t bits;
					memcpy(&bits, bits_src + (slot >> 3), 8);
					bits = __builtin_bswap64(bits);
					alignas(64) dbgk_node group[64];
					memset(static_cast<void *>(group), 0, sizeof group);
					while (bits) {
						const int j = __builtin_clzll(bits);
						group[j] = *src++;
						bits &= ~(0x8000000000000000ull >> j);
					}
					if (stream_out) {
#pragma unroll
						for (int q = 0; q < 64; q++)
							_mm_stream_si128(reinterpret_cast<__m128i *>(dst + slot + q), _mm_load_si128(reinterpret_cast<const __m128i *>(group + q)));
					} else {
						memcpy(static_cast<void *>(dst + slot), group, sizeof group);
					}
				}
				_mm_sfence();
				for (; slot < slot_end; slot++) dst[slot] = occupied(slot) ? *src++ : zero;
				if ((uint64_t)(src - src0) != r1 - r0) failed.store(2); // (never)
				done[i].store(1, std::memory_order_release);
			}
		});
	int rc = DBGK_OK;
	for (size_t i = 0; i < n_slices && rc == DBGK_OK; i++) {
		const double t_w = lap_wanted ? clock_s() : 0;
		if (i >= n_bufs)
			while (!done[i - n_bufs].load(std::memory_order_acquire)) { // its buffer is free again
				if (failed.load()) { rc = DBGK_ERR_HIP; break; }
				std::this_thread::yield();
			}
		if (lap_wanted) wait_host_us += (uint64_t)((clock_s() - t_w) * 1e6);
		if (rc) break;
		const size_t b = i % n_bufs;
		const uint64_t r0 = (uint64_t)i * per, r1 = std::min(n_occ, r0 + per);
		if (hipMemcpyAsync(bufs[b], d_stream + r0, (size_t)(r1 - r0) * sizeof(Node), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
		    hipEventRecord(h->d2h_ev[b], h->stream) != hipSuccess) {
			rc = DBGK_ERR_HIP;
			break;
		}
		issued[i].store(1, std::memory_order_release);
	}
	if (rc) failed.store(1);
	for (auto &t : workers) t.join();
	if (failed.load() && rc == DBGK_OK) rc = DBGK_ERR_HIP;
	if (rc == DBGK_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = DBGK_ERR_HIP;
	lap(3);
	cleanup();
	lap(4);
	if (rc) return hip_fail(hipGetLastError(), "d2h_compact", __LINE__);
	if (lap_wanted)
		fprintf(stderr, "dbgk export, occupied nodes only (s): buffers %.4f, counts %.4f, stream on the device + bits to the host %.4f, stream to the host and into "
		        "the slots %.4f (%zu slices of %.1f MB through %zu buffers%s; the %d threads waited %.4f for the link in all, the issuing thread %.4f for a free "
		        "buffer), release %.4f\n",
		        laps[0], laps[1], laps[2], laps[3], n_slices, (double)per * 16e-6, n_bufs, bits_pinned ? " of the batch staging" : "", n_threads,
		        (double)wait_link_us.load() * 1e-6, (double)wait_host_us.load() * 1e-6, laps[4]);
	return DBGK_OK;
}

// what the link pass of an export returns (all optional)
struct LinkOutputs {
	int32_t cutoff = 0;
	uint16_t *klink = nullptr;
	uint8_t *del_flag = nullptr;
	uint64_t *tips = nullptr, *branches = nullptr;
	uint64_t tip_cap = 0, branch_cap = 0;
	uint64_t *n_tips = nullptr, *n_branches = nullptr;
	dbgk_link_stats *stats = nullptr;
};

static int export_host_table_impl(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag, const LinkOutputs *LO);

extern "C" int dbgk_export_host_table(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag)
{
	return export_host_table_impl(h, host_size, array, nul_flag, nullptr);
}

extern "C" int dbgk_export_host_table_links(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag, int32_t kmer_freq_cutoff,
                                            uint16_t *klink, uint8_t *del_flag, uint64_t *tip_nodes, uint64_t tip_capacity, uint64_t *n_tips,
                                            uint64_t *branch_nodes, uint64_t branch_capacity, uint64_t *n_branches, dbgk_link_stats *stats)
{
	if (!klink || !del_flag || !n_tips || !n_branches) return DBGK_ERR_ARG;
	if (h && h->sharded) return DBGK_ERR_STATE; // slot numbers are those of ONE table: export the shards, assemble, then scan (or use one handle)
	LinkOutputs LO;
	LO.cutoff = kmer_freq_cutoff;
	LO.klink = klink;
	LO.del_flag = del_flag;
	LO.tips = tip_nodes;
	LO.branches = branch_nodes;
	LO.tip_cap = tip_nodes ? tip_capacity : 0;
	LO.branch_cap = branch_nodes ? branch_capacity : 0;
	LO.n_tips = n_tips;
	LO.n_branches = n_branches;
	LO.stats = stats;
	return export_host_table_impl(h, host_size, array, nul_flag, &LO);
}

// the link pass on the host-layout image T (key-0 node placed, ctr->polyA_slot set)
static int run_link_pass(dbgk_handle *h, const TableRef &T, const LinkOutputs &LO)
{
	const uint64_t n_blocks = (T.size + kLinkChunk - 1) / kLinkChunk;
	uint16_t *d_klink = nullptr;
	uint8_t *d_del = nullptr;
	unsigned long long *d_stats = nullptr, *d_base = nullptr, *d_tips = nullptr, *d_branches = nullptr;
	uint32_t *d_counts = nullptr;
	auto cleanup = [&]() {
		for (void *p : {(void *)d_klink, (void *)d_del, (void *)d_stats, (void *)d_base, (void *)d_tips, (void *)d_branches, (void *)d_counts})
			if (p) (void)hipFree(p);
	};
	if (hipMalloc(&d_klink, T.size * 2) != hipSuccess || hipMalloc(&d_del, T.size / 8 + 1) != hipSuccess || hipMalloc(&d_stats, 261 * 8) != hipSuccess ||
	    hipMalloc(&d_counts, n_blocks * 8) != hipSuccess || hipMalloc(&d_base, n_blocks * 16) != hipSuccess) {
		cleanup();
		return DBGK_ERR_NOMEM;
	}
	hipError_t e = hipMemsetAsync(d_stats, 0, 261 * 8, h->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_del, 0, T.size / 8 + 1, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_kmer_links<0>, dim3((unsigned)n_blocks), dim3(kBlock), 0, h->stream, T.nodes, T.size, h->d_ctr, (int)LO.cutoff, d_klink, d_del, d_stats,
		                   d_counts, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (unsigned long long *)nullptr);
		e = hipGetLastError();
	}
	std::vector<uint32_t> counts(n_blocks * 2);
	std::vector<unsigned long long> base(n_blocks * 2);
	unsigned long long res[261];
	if (e == hipSuccess) e = hipMemcpyAsync(counts.data(), d_counts, n_blocks * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(res, d_stats, sizeof res, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(LO.klink, d_klink, T.size * 2, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(LO.del_flag, d_del, T.size / 8 + 1, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) {
		cleanup();
		return hip_fail(e, "export_host_table_links", __LINE__);
	}
	unsigned long long nt = 0, nb = 0;
	for (uint64_t b = 0; b < n_blocks; b++) {
		base[2 * b] = nt;
		base[2 * b + 1] = nb;
		nt += counts[2 * b];
		nb += counts[2 * b + 1];
	}
	*LO.n_tips = nt;
	*LO.n_branches = nb;
	if (LO.stats) {
		for (int i = 0; i < 256; i++) LO.stats->depth_stat[i] = (int64_t)res[i];
		LO.stats->total_nodes = (int64_t)res[256];
		LO.stats->deleted_lowfreq = (int64_t)res[257];
		LO.stats->linear_nodes = (int64_t)res[258];
		LO.stats->tip_nodes = (int64_t)res[259];
		LO.stats->branch_nodes = (int64_t)res[260];
	}
	int rc = DBGK_OK;
	if ((LO.tips || LO.branches) && (nt || nb)) {
		if ((LO.tips && nt > LO.tip_cap) || (LO.branches && nb > LO.branch_cap)) {
			rc = DBGK_ERR_CAPACITY; // *n_tips / *n_branches say what is needed
		} else if (hipMalloc(&d_tips, (nt ? nt : 1) * 8) != hipSuccess || hipMalloc(&d_branches, (nb ? nb : 1) * 8) != hipSuccess) {
			rc = DBGK_ERR_NOMEM;
		} else {
			e = hipMemcpyAsync(d_base, base.data(), n_blocks * 16, hipMemcpyHostToDevice, h->stream);
			if (e == hipSuccess) {
				hipLaunchKernelGGL(k_kmer_links<1>, dim3((unsigned)n_blocks), dim3(kBlock), 0, h->stream, T.nodes, T.size, h->d_ctr, (int)LO.cutoff, d_klink, d_del,
				                   d_stats, d_counts, d_base, d_tips, d_branches);
				e = hipGetLastError();
			}
			if (e == hipSuccess && LO.tips && nt) e = hipMemcpyAsync(LO.tips, d_tips, nt * 8, hipMemcpyDeviceToHost, h->stream);
			if (e == hipSuccess && LO.branches && nb) e = hipMemcpyAsync(LO.branches, d_branches, nb * 8, hipMemcpyDeviceToHost, h->stream);
			if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
			if (e != hipSuccess) rc = hip_fail(e, "export_host_table_links(lists)", __LINE__);
		}
	}
	cleanup();
	return rc;
}

static int export_host_table_impl(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag, const LinkOutputs *LO)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !array || !nul_flag || host_size < 3) return DBGK_ERR_ARG;
	if (!h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	if (h->sharded) { // the shard's slice of the global table: slots [slot_lo, slot_hi), key-0 node not placed
		if (host_size != h->tslots) return DBGK_ERR_ARG;
		uint8_t *d_fl = nullptr;
		if (hipMalloc(&d_fl, host_size / 8 + 1) != hipSuccess) return DBGK_ERR_NOMEM;
		hipLaunchKernelGGL(k_build_flags_ctr, dim3(grid_for(h, host_size / 8 + 1)), dim3(kBlock), 0, h->stream, h->table, h->tslots,
		                   h->d_ctr, d_fl);
		hipError_t es = hipGetLastError();
		if (es == hipSuccess) es = hipMemcpyAsync(array, h->table, host_size * sizeof(Node), hipMemcpyDeviceToHost, h->stream);
		if (es == hipSuccess) es = hipMemcpyAsync(nul_flag, d_fl, host_size / 8 + 1, hipMemcpyDeviceToHost, h->stream);
		if (es == hipSuccess) es = hipStreamSynchronize(h->stream);
		(void)hipFree(d_fl);
		if (es != hipSuccess) return hip_fail(es, "export_host_table(shard)", __LINE__);
		return DBGK_OK;
	}
	if (h->h_ctr->n_new + 1 > host_size) return DBGK_ERR_TABLE_FULL;

	TableRef T = h->tref();
	Node *tmp = nullptr;
	uint8_t *d_flags = nullptr;
	auto cleanup = [&]() {
		if (tmp) (void)hipFree(tmp);
		if (d_flags) (void)hipFree(d_flags);
	};
	if (host_size != h->size) {
		if (hipMalloc(&tmp, host_size * sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
		T = TableRef{tmp, host_size, make_mod_magic(host_size)};
		if (hipMemsetAsync(tmp, 0, host_size * sizeof(Node), h->stream) != hipSuccess) { cleanup(); return DBGK_ERR_HIP; }
		hipLaunchKernelGGL(k_rehash, dim3(grid_for(h, h->size)), dim3(kBlock), 0, h->stream, h->table, h->size, T, h->d_ctr);
	}
	const uint64_t flag_bytes = host_size / 8 + 1, flag_alloc = (flag_bytes + 11) & ~7ull; // whole dwords for k_flag_block_counts
	if (hipMalloc(&d_flags, flag_alloc) != hipSuccess) { cleanup(); return DBGK_ERR_NOMEM; }
	if (hipMemsetAsync(d_flags, 0, flag_alloc, h->stream) != hipSuccess) { cleanup(); return DBGK_ERR_HIP; }
	hipLaunchKernelGGL(k_place_polyA, dim3(1), dim3(64), 0, h->stream, T, h->d_ctr);
	hipLaunchKernelGGL(k_build_flags_ctr, dim3(grid_for(h, host_size / 8 + 1)), dim3(kBlock), 0, h->stream, T.nodes, T.size,
	                   h->d_ctr, d_flags);
	hipError_t e = hipGetLastError();
	int copy_rc = DBGK_OK;
	static const bool lap_wanted = getenv("DBGK_TIMINGS") != nullptr;
	auto clock_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	if (lap_wanted && e == hipSuccess) e = hipStreamSynchronize(h->stream);
	const double t_a = clock_s();
	// large tables: the occupied nodes only (DBGK_EXPORT_FULL=1: every slot over the link, as before round 4)
	const bool full_copy = dbgk_hook("export_full") != nullptr; // (read at every call: the tests compare the two)
	bool bits_copied = false;
	if (e == hipSuccess) {
		copy_rc = DBGK_ERR_STATE;
		if (!full_copy && host_size * sizeof(Node) >= (dbgk_hook("export_compact_min") ? strtoull(dbgk_hook("export_compact_min"), nullptr, 10) : (256ull << 20))) copy_rc = d2h_compact(h, array, nul_flag, T, d_flags, h->h_ctr->n_new + 1);
		bits_copied = copy_rc == DBGK_OK;
		if (copy_rc == DBGK_ERR_STATE) copy_rc = d2h_pipelined(h, array, T.nodes, host_size * sizeof(Node));
	}
	const double t_b = clock_s();
	if (e == hipSuccess && copy_rc == DBGK_OK && !bits_copied)
		e = hipMemcpyAsync(nul_flag, d_flags, host_size / 8 + 1, hipMemcpyDeviceToHost, h->stream);
	int link_rc = copy_rc;
	if (e == hipSuccess && LO && copy_rc == DBGK_OK) link_rc = run_link_pass(h, T, *LO); // on the very image that is being copied out
	if (lap_wanted) fprintf(stderr, "dbgk export (s): node copy %.4f (%.1f GB/s)\n", t_b - t_a, (double)host_size * sizeof(Node) / (t_b - t_a) * 1e-9);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_unplace_polyA, dim3(1), dim3(64), 0, h->stream, T, h->d_ctr);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	cleanup();
	if (e != hipSuccess) return hip_fail(e, "export_host_table", __LINE__);
	if (h->h_ctr->error & 1u) return DBGK_ERR_TABLE_FULL;
	return link_rc;
}

extern "C" int dbgk_export_sorted(dbgk_handle *h, dbgk_node *out, uint64_t capacity, uint64_t *n_out)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !out || !n_out) return DBGK_ERR_ARG;
	if (!h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t n = h->h_ctr->n_new; // non-zero keys
	const uint64_t z = (h->sharded && h->geom.rank != 0) ? 0 : 1; // the key-0 node is reported by shard 0 only
	*n_out = n + z;
	if (capacity < n + z) return DBGK_ERR_CAPACITY;
	if (z) { // key 0 sorts first
		out[0].kmer = 0;
		out[0].l_link = (uint32_t)(h->h_ctr->polyA_links & 0xFFFFFFFFu);
		out[0].r_link = (uint32_t)(h->h_ctr->polyA_links >> 32);
	}
	if (n == 0) return DBGK_OK;

	uint64_t *d_keys = nullptr, *d_links = nullptr;
	unsigned long long *d_cursor = nullptr;
	auto cleanup = [&]() {
		if (d_keys) (void)hipFree(d_keys);
		if (d_links) (void)hipFree(d_links);
		if (d_cursor) (void)hipFree(d_cursor);
	};
	if (hipMalloc(&d_keys, n * 8) != hipSuccess || hipMalloc(&d_links, n * 8) != hipSuccess ||
	    hipMalloc(&d_cursor, 8) != hipSuccess) {
		cleanup();
		return DBGK_ERR_NOMEM;
	}
	hipError_t e = hipMemsetAsync(d_cursor, 0, 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_compact, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, d_keys, d_links,
		                   d_cursor, n);
		e = hipGetLastError();
	}
	unsigned long long found = 0;
	if (e == hipSuccess) e = hipMemcpyAsync(&found, d_cursor, 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) { cleanup(); return hip_fail(e, "export_sorted/compact", __LINE__); }
	if (found != n) {
		cleanup();
		g_last_error = "export_sorted: occupied slots != counted keys";
		return DBGK_ERR_STATE;
	}
	rc = dbgk_internal_sort_pairs(d_keys, d_links, n, h->stream);
	if (rc != DBGK_OK) { cleanup(); return rc; }
	std::vector<uint64_t> hk(n), hl(n);
	e = hipMemcpyAsync(hk.data(), d_keys, n * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(hl.data(), d_links, n * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	cleanup();
	if (e != hipSuccess) return hip_fail(e, "export_sorted/copy", __LINE__);
	for (uint64_t i = 0; i < n; i++) {
		out[i + z].kmer = hk[i];
		out[i + z].l_link = (uint32_t)(hl[i] & 0xFFFFFFFFu);
		out[i + z].r_link = (uint32_t)(hl[i] >> 32);
	}
	return DBGK_OK;
}

extern "C" int dbgk_export_first_seen_order(dbgk_handle *h, dbgk_node *out, uint64_t *first_pos, uint64_t capacity, uint64_t *n_out)
{
	if (!h || !out || !first_pos || !n_out) return DBGK_ERR_ARG;
	if (!h->finalized || !h->track) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t n = h->h_ctr->n_new;
	*n_out = n;
	if (capacity < n) return DBGK_ERR_CAPACITY;
	if (n == 0) return DBGK_OK;
	uint64_t *d_pos = nullptr, *d_slot = nullptr;
	unsigned long long *d_cursor = nullptr;
	Node *d_nodes = nullptr;
	auto cleanup = [&]() {
		for (void *p : {(void *)d_pos, (void *)d_slot, (void *)d_cursor, (void *)d_nodes})
			if (p) (void)hipFree(p);
	};
	if (hipMalloc(&d_pos, n * 8) != hipSuccess || hipMalloc(&d_slot, n * 8) != hipSuccess || hipMalloc(&d_cursor, 8) != hipSuccess ||
	    hipMalloc(&d_nodes, n * sizeof(Node)) != hipSuccess) {
		cleanup();
		return DBGK_ERR_NOMEM;
	}
	hipError_t e = hipMemsetAsync(d_cursor, 0, 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_compact_order, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->first_pos, h->tslots, d_pos,
		                   d_slot, d_cursor, n);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) { cleanup(); return hip_fail(e, "export_first_seen_order/compact", __LINE__); }
	rc = dbgk_internal_sort_pairs(d_pos, d_slot, n, h->stream); // positions are unique per key: a total order
	if (rc != DBGK_OK) { cleanup(); return rc; }
	hipLaunchKernelGGL(k_gather_nodes, dim3(grid_for(h, n)), dim3(kBlock), 0, h->stream, h->table, d_slot, n, d_nodes);
	e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(out, d_nodes, n * sizeof(Node), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(first_pos, d_pos, n * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	cleanup();
	if (e != hipSuccess) return hip_fail(e, "export_first_seen_order", __LINE__);
	return DBGK_OK;
}

extern "C" int dbgk_digest(dbgk_handle *h, uint64_t *digest)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !digest) return DBGK_ERR_ARG;
	if (!h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long *d_out = nullptr;
	if (hipMalloc(&d_out, 16) != hipSuccess) return DBGK_ERR_NOMEM;
	unsigned long long res[2] = {0, 0};
	hipError_t e = hipMemsetAsync(d_out, 0, 16, h->stream);
	if (e == hipSuccess) {
		if (h->wide)
			hipLaunchKernelGGL(k_wide_digest, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, h->wside, d_out);
		else
			hipLaunchKernelGGL(k_digest, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, d_out);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(res, d_out, 16, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_out);
	if (e != hipSuccess) return hip_fail(e, "digest", __LINE__);
	*digest = res[0] + ((h->sharded && h->shard_rank != 0) ? 0ull : node_digest(0ull, h->h_ctr->polyA_links));
	return DBGK_OK;
}

extern "C" int dbgk_link_stats_device(dbgk_handle *h, int32_t cutoff, dbgk_link_stats *out)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !out) return DBGK_ERR_ARG;
	if (!h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long *d_out = nullptr;
	const size_t bytes = 261 * sizeof(unsigned long long);
	if (hipMalloc(&d_out, bytes) != hipSuccess) return DBGK_ERR_NOMEM;
	unsigned long long res[261];
	hipError_t e = hipMemsetAsync(d_out, 0, bytes, h->stream);
	if (e == hipSuccess) {
		if (h->wide)
			hipLaunchKernelGGL(k_wide_link_stats, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, h->wside, (int)cutoff,
			                   (uint64_t)h->h_ctr->polyA_links, (h->sharded && h->shard_rank != 0) ? 0 : 1, d_out);
		else
			hipLaunchKernelGGL(k_link_stats, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, (int)cutoff,
			                   (uint64_t)h->h_ctr->polyA_links, (h->sharded && h->shard_rank != 0) ? 0 : 1, d_out);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(res, d_out, bytes, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_out);
	if (e != hipSuccess) return hip_fail(e, "link_stats", __LINE__);
	for (int i = 0; i < 256; i++) out->depth_stat[i] = (int64_t)res[i];
	out->total_nodes = (int64_t)res[256];
	out->deleted_lowfreq = (int64_t)res[257];
	out->linear_nodes = (int64_t)res[258];
	out->tip_nodes = (int64_t)res[259];
	out->branch_nodes = (int64_t)res[260];
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// SEEDIDX exports
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_seed_export_sorted(dbgk_handle *h, dbgk_node *out, uint64_t capacity, uint64_t *n_out)
{
	if (!h || !out || !n_out) return DBGK_ERR_ARG;
	if (!h->seed || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t n = h->h_ctr->n_new;
	const uint64_t z = h->h_ctr->polyA_links ? 1 : 0;
	*n_out = n + z;
	if (capacity < n + z) return DBGK_ERR_CAPACITY;
	if (z) {
		const uint64_t w = seed_payload_out(h->h_ctr->polyA_links);
		out[0].kmer = 0;
		out[0].l_link = (uint32_t)w;
		out[0].r_link = (uint32_t)(w >> 32);
	}
	if (n == 0) return DBGK_OK;
	uint64_t *d_keys = nullptr, *d_links = nullptr;
	unsigned long long *d_cursor = nullptr;
	auto cleanup = [&]() {
		for (void *p : {(void *)d_keys, (void *)d_links, (void *)d_cursor})
			if (p) (void)hipFree(p);
	};
	if (hipMalloc(&d_keys, n * 8) != hipSuccess || hipMalloc(&d_links, n * 8) != hipSuccess || hipMalloc(&d_cursor, 8) != hipSuccess) {
		cleanup();
		return DBGK_ERR_NOMEM;
	}
	hipError_t e = hipMemsetAsync(d_cursor, 0, 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_compact, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, d_keys, d_links, d_cursor, n);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) { cleanup(); return hip_fail(e, "seed_export_sorted/compact", __LINE__); }
	rc = dbgk_internal_sort_pairs(d_keys, d_links, n, h->stream);
	if (rc != DBGK_OK) { cleanup(); return rc; }
	std::vector<uint64_t> hk(n), hl(n);
	e = hipMemcpyAsync(hk.data(), d_keys, n * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(hl.data(), d_links, n * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	cleanup();
	if (e != hipSuccess) return hip_fail(e, "seed_export_sorted/copy", __LINE__);
	for (uint64_t i = 0; i < n; i++) {
		const uint64_t w = seed_payload_out(hl[i]);
		out[i + z].kmer = hk[i];
		out[i + z].l_link = (uint32_t)w;
		out[i + z].r_link = (uint32_t)(w >> 32);
	}
	return DBGK_OK;
}

extern "C" int dbgk_seed_export_host_table(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag)
{
	if (!h || !array || !nul_flag || host_size < 3) return DBGK_ERR_ARG;
	if (!h->seed || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	if (h->h_ctr->n_new + 1 > host_size) return DBGK_ERR_TABLE_FULL;
	Node *tmp = nullptr;
	uint8_t *d_flags = nullptr;
	if (hipMalloc(&tmp, host_size * sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&d_flags, host_size / 8 + 1) != hipSuccess) {
		(void)hipFree(tmp);
		return DBGK_ERR_NOMEM;
	}
	TableRef T{tmp, host_size, make_mod_magic(host_size)};
	hipError_t e = hipMemsetAsync(tmp, 0, host_size * sizeof(Node), h->stream);
	if (e == hipSuccess) {
		// always through a copy: the payload words are converted in place to the reference's bit-field
		hipLaunchKernelGGL(k_rehash, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, T, h->d_ctr,
		                   (const unsigned long long *)nullptr, (unsigned long long *)nullptr);
		hipLaunchKernelGGL(k_seed_convert, dim3(grid_for(h, host_size)), dim3(kBlock), 0, h->stream, tmp, host_size);
		hipLaunchKernelGGL(k_seed_place_key0, dim3(1), dim3(64), 0, h->stream, T, h->d_ctr);
		hipLaunchKernelGGL(k_build_flags_ctr, dim3(grid_for(h, host_size / 8 + 1)), dim3(kBlock), 0, h->stream, tmp, host_size, h->d_ctr, d_flags);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(array, tmp, host_size * sizeof(Node), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(nul_flag, d_flags, host_size / 8 + 1, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(tmp);
	(void)hipFree(d_flags);
	if (e != hipSuccess) return hip_fail(e, "seed_export_host_table", __LINE__);
	return (h->h_ctr->error & 1u) ? DBGK_ERR_TABLE_FULL : DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// WIDE exports (128-bit keys, include/dbgk_wide.h)
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_wide_export_sorted(dbgk_handle *h, dbgk_node32 *out, uint64_t capacity, uint64_t *n_out)
{
	if (!h || !out || !n_out) return DBGK_ERR_ARG;
	if (!h->wide || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t n = h->h_ctr->n_new; // non-zero keys (main + side table)
	const uint64_t z = (h->sharded && h->shard_rank != 0) ? 0 : 1; // of a sharded table only shard 0 reports the key-0 node
	*n_out = n + z;
	if (capacity < n + z) return DBGK_ERR_CAPACITY;
	if (z) out[0] = dbgk_node32{0, 0, (uint32_t)(h->h_ctr->polyA_links & 0xFFFFFFFFu), (uint32_t)(h->h_ctr->polyA_links >> 32), 0}; // key 0 sorts first
	dbgk_node32 *dst = out + z; // the non-zero keys follow the key-0 node (if this shard reports one)
	if (n == 0) return DBGK_OK;
	dbgk_node32 *d_out = nullptr;
	unsigned long long *d_cursor = nullptr;
	if (hipMalloc(&d_out, n * sizeof(dbgk_node32)) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&d_cursor, 8) != hipSuccess) {
		(void)hipFree(d_out);
		return DBGK_ERR_NOMEM;
	}
	unsigned long long found = 0;
	hipError_t e = hipMemsetAsync(d_cursor, 0, 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_wide_compact, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, h->wside, d_out, d_cursor, n);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(&found, d_cursor, 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dst, d_out, n * sizeof(dbgk_node32), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_out);
	(void)hipFree(d_cursor);
	if (e != hipSuccess) return hip_fail(e, "wide_export_sorted", __LINE__);
	if (found != n) {
		g_last_error = "wide_export_sorted: occupied slots != counted keys";
		return DBGK_ERR_STATE;
	}
	std::sort(dst, dst + n, [](const dbgk_node32 &a, const dbgk_node32 &b) {
		return a.kmer_hi < b.kmer_hi || (a.kmer_hi == b.kmer_hi && a.kmer_lo < b.kmer_lo);
	});
	return DBGK_OK;
}

// host-layout table of host_size == table_slots 32-byte nodes + nul_flag: every key reachable by linear probing from
// hash128(key) % size without crossing a clear flag.  The few nodes that live outside the main table on the
// device (keys whose low word is 0, the key-0 node) are put on their probe chains here, on the host.
extern "C" int dbgk_wide_export_host_table(dbgk_handle *h, uint64_t host_size, dbgk_node32 *array, uint8_t *nul_flag)
{
	if (!h || !array || !nul_flag) return DBGK_ERR_ARG;
	if (!h->wide || !h->finalized) return DBGK_ERR_STATE;
	if (host_size != h->tslots) {
		g_last_error = "dbgk_wide_export_host_table: host_size must be the handle's table_slots (a shard: the slots of its range)";
		return DBGK_ERR_ARG;
	}
	if (!h->sharded && h->h_ctr->n_new + 1 > host_size) return DBGK_ERR_TABLE_FULL;
	int rc = use_device(h);
	if (rc) return rc;
	dbgk_node32 *d_img = nullptr;
	uint8_t *d_flags = nullptr;
	std::vector<WNode> side(kWideSideSlots);
	if (hipMalloc(&d_img, host_size * sizeof(dbgk_node32)) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&d_flags, host_size / 8 + 1) != hipSuccess) {
		(void)hipFree(d_img);
		return DBGK_ERR_NOMEM;
	}
	hipLaunchKernelGGL(k_wide_image, dim3(grid_for(h, host_size / 8 + 1)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, d_img, d_flags);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(array, d_img, host_size * sizeof(dbgk_node32), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(nul_flag, d_flags, host_size / 8 + 1, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(side.data(), h->wside, kWideSideSlots * sizeof(WNode), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_img);
	(void)hipFree(d_flags);
	if (e != hipSuccess) return hip_fail(e, "wide_export_host_table", __LINE__);
	if (h->sharded) return DBGK_OK; // a shard's slice as it is: side-table nodes and the key-0 node are placed over the WHOLE table by the caller
	auto place = [&](dbgk_node32 nd) { // add_node_to_kmerset's rule (kmerSet.cpp:253-273): first slot without a flag on the key's chain
		uint64_t hc = dbgk_wide::hash128(dbgk_wide::Key128{nd.kmer_hi, nd.kmer_lo}) % host_size;
		while (nul_flag[hc >> 3] & (uint8_t)(128u >> (hc & 7u))) hc = (hc + 1 == host_size) ? 0 : hc + 1;
		array[hc] = nd;
		nul_flag[hc >> 3] |= (uint8_t)(128u >> (hc & 7u));
	};
	for (const WNode &s : side)
		if (s.hi1) place(dbgk_node32{s.hi1 - 1ull, 0ull, (uint32_t)s.links, (uint32_t)(s.links >> 32), 0});
	place(dbgk_node32{0, 0, (uint32_t)(h->h_ctr->polyA_links & 0xFFFFFFFFu), (uint32_t)(h->h_ctr->polyA_links >> 32), 0}); // DBGgraph.cpp:418
	return DBGK_OK;
}

// several GPUs with 128-bit keys: nodes grouped by owner, merged by the owner (dbgk_partition_* / dbgk_merge_nodes for
// 32-byte nodes).  counts[p] includes, for p == 0, this handle's key-0 node, which is written first.
extern "C" int dbgk_wide_partition_export(dbgk_handle *h, uint32_t n_parts, dbgk_node32 *d_nodes, uint64_t capacity, uint64_t *counts)
{
	if (!h || !counts || n_parts < 1 || n_parts > (uint32_t)kMaxParts) return DBGK_ERR_ARG;
	if (!h->wide || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long *d_counts = nullptr;
	if (hipMalloc(&d_counts, n_parts * 8) != hipSuccess) return DBGK_ERR_NOMEM;
	std::vector<unsigned long long> hc(n_parts, 0);
	hipError_t e = hipMemsetAsync(d_counts, 0, n_parts * 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_wide_partition, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, h->wside, n_parts, d_counts,
		                   (unsigned long long *)nullptr, (dbgk_node32 *)nullptr, (uint64_t)0);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(hc.data(), d_counts, n_parts * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) {
		(void)hipFree(d_counts);
		return hip_fail(e, "wide_partition_export/count", __LINE__);
	}
	hc[0] += 1; // the key-0 node travels with part 0
	uint64_t total = 0;
	std::vector<unsigned long long> cursors(n_parts);
	for (uint32_t p = 0; p < n_parts; p++) {
		counts[p] = hc[p];
		cursors[p] = total + (p == 0 ? 1 : 0);
		total += hc[p];
	}
	if (!d_nodes) { // counts only
		(void)hipFree(d_counts);
		return DBGK_OK;
	}
	if (total > capacity) {
		(void)hipFree(d_counts);
		return DBGK_ERR_CAPACITY;
	}
	const dbgk_node32 zero = {0, 0, (uint32_t)(h->h_ctr->polyA_links & 0xFFFFFFFFu), (uint32_t)(h->h_ctr->polyA_links >> 32), 0};
	e = hipMemcpyAsync(d_counts, cursors.data(), n_parts * 8, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_nodes, &zero, sizeof zero, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_wide_partition, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, h->wside, n_parts,
		                   (unsigned long long *)nullptr, d_counts, d_nodes, capacity);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_counts);
	if (e != hipSuccess) return hip_fail(e, "wide_partition_export", __LINE__);
	return DBGK_OK;
}

extern "C" int dbgk_wide_merge_nodes(dbgk_handle *h, const dbgk_node32 *d_nodes, uint64_t n)
{
	if (!h || (n && !d_nodes)) return DBGK_ERR_ARG;
	if (!h->wide) return DBGK_ERR_STATE;
	if (h->wpart && !h->finalized) { // pending records become the table first: the region build rewrites every slot
		int frc = use_device(h);
		if (frc) return frc;
		frc = h->pending_kmers && !h->wbuilt ? wide_build_from_records(h) : DBGK_OK;
		if (frc) return frc;
		frc = wide_ensure_zero(h);
		if (frc) return frc;
		h->wbuilt = true; // whatever comes later joins the table through the atomic kernels
	}
	int rc = use_device(h);
	if (rc) return rc;
	if (n == 0) return DBGK_OK;
	hipLaunchKernelGGL(k_wide_merge_nodes, dim3(grid_for(h, n)), dim3(kBlock), 0, h->stream, d_nodes, n, h->wref(), h->d_ctr);
	HIPCHK(hipGetLastError());
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// KFREQ exports
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_kfreq_export_counts(dbgk_handle *h, uint64_t first_kmer, uint64_t n, uint8_t *host_out)
{
	if (!h || !host_out) return DBGK_ERR_ARG;
	if (!h->kfreq || !h->finalized) return DBGK_ERR_STATE;
	const uint64_t total = 1ull << (2 * h->cfg.kmer_size);
	if (first_kmer > total || n > total - first_kmer) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (n == 0) return DBGK_OK;
	HIPCHK(hipMemcpyAsync(host_out, h->counts + first_kmer, n, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

extern "C" int dbgk_kfreq_export_bits(dbgk_handle *h, uint32_t cutoff, uint64_t first_byte, uint64_t n_bytes, uint8_t *host_out)
{
	if (!h || !host_out) return DBGK_ERR_ARG;
	if (!h->kfreq || !h->finalized) return DBGK_ERR_STATE;
	const uint64_t total_bytes = h->n_counts >> 3;
	if (first_byte > total_bytes || n_bytes > total_bytes - first_byte) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (n_bytes == 0) return DBGK_OK;
	uint8_t *d_bits = nullptr;
	if (hipMalloc(&d_bits, n_bytes) != hipSuccess) return DBGK_ERR_NOMEM;
	hipLaunchKernelGGL(k_counts_to_bits, dim3(grid_for(h, n_bytes)), dim3(kBlock), 0, h->stream, h->counts, first_byte, n_bytes, cutoff, d_bits);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(host_out, d_bits, n_bytes, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_bits);
	if (e != hipSuccess) return hip_fail(e, "kfreq_export_bits", __LINE__);
	return DBGK_OK;
}

// counts[first_kmer, first_kmer + n) += another partial table's slice held in device memory of this GPU
// (saturating).  The distinct-k-mer count of the handle is recomputed.
static int kfreq_summary(dbgk_handle *h, uint64_t first, uint64_t n, unsigned long long res[2])
{
	unsigned long long *d_sum = nullptr;
	res[0] = res[1] = 0;
	if (hipMalloc(&d_sum, 16) != hipSuccess) return DBGK_ERR_NOMEM;
	hipError_t e = hipMemsetAsync(d_sum, 0, 16, h->stream);
	if (e == hipSuccess && n) {
		hipLaunchKernelGGL(k_counts_summary, dim3(grid_for(h, n >> 3)), dim3(kBlock), 0, h->stream, h->counts + first, n, d_sum);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(res, d_sum, 16, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_sum);
	if (e != hipSuccess) return hip_fail(e, "kfreq summary", __LINE__);
	return DBGK_OK;
}

extern "C" int dbgk_kfreq_merge_counts(dbgk_handle *h, const uint8_t *d_counts, uint64_t first_kmer, uint64_t n)
{
	if (!h || !d_counts) return DBGK_ERR_ARG;
	if (!h->kfreq || !h->finalized) return DBGK_ERR_STATE;
	if (first_kmer > h->n_counts || n > h->n_counts - first_kmer) return DBGK_ERR_ARG;
	if ((first_kmer & 15u) || (n & 15u) || ((uintptr_t)d_counts & 15u)) {
		g_last_error = "dbgk_kfreq_merge_counts: first_kmer, n and the source address must be multiples of 16";
		return DBGK_ERR_ARG;
	}
	int rc = use_device(h);
	if (rc) return rc;
	if (n) {
		hipLaunchKernelGGL(k_counts_merge, dim3(grid_for(h, n >> 4)), dim3(kBlock), 0, h->stream, h->counts + first_kmer, d_counts, n);
		HIPCHK(hipGetLastError());
	}
	unsigned long long res[2];
	rc = kfreq_summary(h, 0, h->n_counts, res);
	if (rc) return rc;
	h->kf_distinct = res[0];
	h->kf_sum = res[1];
	return DBGK_OK;
}

extern "C" int dbgk_kfreq_device_counts(dbgk_handle *h, uint8_t **d_counts, uint64_t *n)
{
	if (!h || !d_counts || !n) return DBGK_ERR_ARG;
	if (!h->kfreq || !h->finalized) return DBGK_ERR_STATE;
	*d_counts = h->counts;
	*n = h->n_counts;
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// phase A alone
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_extract_kmers(dbgk_handle *h, const char *bases, const uint64_t *offsets, uint64_t n_reads,
                                  uint64_t *kmer, uint8_t *left, uint8_t *right, uint8_t *valid)
{
	if (!h || !offsets || !kmer || !left || !right || !valid) return DBGK_ERR_ARG;
	if (h->wide) return DBGK_ERR_STATE; // 64-bit keys only
	if (offsets[0] != 0) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t nb = offsets[n_reads];
	if (nb == 0) return DBGK_OK;
	const uint64_t words = bitmap_words(nb);
	char *d_bases = nullptr;
	uint64_t *d_off = nullptr, *d_kmer = nullptr;
	uint32_t *d_start = nullptr, *d_dead = nullptr;
	uint8_t *d_l = nullptr, *d_r = nullptr, *d_v = nullptr;
	Counters *d_ctr = nullptr;
	auto cleanup = [&]() {
		for (void *p : {(void *)d_bases, (void *)d_off, (void *)d_kmer, (void *)d_start, (void *)d_dead, (void *)d_l, (void *)d_r,
		                (void *)d_v, (void *)d_ctr})
			if (p) (void)hipFree(p);
	};
	hipError_t e = hipMalloc(&d_bases, nb + 64);
	if (e == hipSuccess) e = hipMalloc(&d_off, (n_reads + 1) * 8);
	if (e == hipSuccess) e = hipMalloc(&d_kmer, nb * 8);
	if (e == hipSuccess) e = hipMalloc(&d_start, words * 4);
	if (e == hipSuccess) e = hipMalloc(&d_dead, words * 4);
	if (e == hipSuccess) e = hipMalloc(&d_l, nb);
	if (e == hipSuccess) e = hipMalloc(&d_r, nb);
	if (e == hipSuccess) e = hipMalloc(&d_v, nb);
	if (e == hipSuccess) e = hipMalloc(&d_ctr, sizeof(Counters));
	if (e != hipSuccess) { cleanup(); return DBGK_ERR_NOMEM; }
	e = hipMemcpyAsync(d_bases, bases, nb, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_off, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_start, 0, words * 4, h->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_dead, 0, words * 4, h->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_ctr, 0, sizeof(Counters), h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_mark, dim3(grid_for(h, n_reads)), dim3(kBlock), 0, h->stream, d_off, n_reads, nb, h->cfg.kmer_size,
		                   h->cfg.max_read_len, d_start, d_dead, d_ctr);
		ReadBatch rb{d_bases, nb, d_start, d_dead, h->cfg.kmer_size, nullptr, &d_ctr->other_seen};
		hipLaunchKernelGGL(k_extract_store<true>, dim3(grid_for(h, (nb + 15) >> 4)), dim3(kBlock), 0, h->stream, rb, d_kmer, d_l,
		                   d_r, d_v);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(kmer, d_kmer, nb * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(left, d_l, nb, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(right, d_r, nb, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(valid, d_v, nb, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	cleanup();
	if (e != hipSuccess) return hip_fail(e, "extract_kmers", __LINE__);
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU building blocks
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_partition_counts(dbgk_handle *h, uint32_t n_parts, uint64_t *counts)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !counts || n_parts < 1 || n_parts > (uint32_t)kMaxParts) return DBGK_ERR_ARG;
	if (!h->finalized || h->sharded) return DBGK_ERR_STATE; // a sharded table is already owned by slot range
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long *d_counts = nullptr;
	if (hipMalloc(&d_counts, n_parts * 8) != hipSuccess) return DBGK_ERR_NOMEM;
	hipError_t e = hipMemsetAsync(d_counts, 0, n_parts * 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_partition_count, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, n_parts,
		                   d_counts);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(counts, d_counts, n_parts * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_counts);
	if (e != hipSuccess) return hip_fail(e, "partition_counts", __LINE__);
	counts[0] += 1; // the key-0 node travels with part 0
	return DBGK_OK;
}

__global__ void k_write_polyA_node(Node *out, uint64_t index, const Counters *ctr)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		out[index].kmer = 0ull;
		out[index].links = ctr->polyA_links;
	}
}

extern "C" int dbgk_partition_export(dbgk_handle *h, uint32_t n_parts, dbgk_node *d_nodes, uint64_t capacity)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !d_nodes || n_parts < 1 || n_parts > (uint32_t)kMaxParts) return DBGK_ERR_ARG;
	if (!h->finalized || h->sharded) return DBGK_ERR_STATE;
	std::vector<uint64_t> counts(n_parts);
	int rc = dbgk_partition_counts(h, n_parts, counts.data());
	if (rc) return rc;
	uint64_t total = 0;
	std::vector<unsigned long long> cursors(n_parts);
	for (uint32_t p = 0; p < n_parts; p++) {
		cursors[p] = total + (p == 0 ? 1 : 0); // slot 0 of part 0 is the key-0 node
		total += counts[p];
	}
	if (total > capacity) return DBGK_ERR_CAPACITY;
	unsigned long long *d_cursors = nullptr;
	if (hipMalloc(&d_cursors, n_parts * 8) != hipSuccess) return DBGK_ERR_NOMEM;
	hipError_t e = hipMemcpyAsync(d_cursors, cursors.data(), n_parts * 8, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_write_polyA_node, dim3(1), dim3(64), 0, h->stream, reinterpret_cast<Node *>(d_nodes), (uint64_t)0,
		                   h->d_ctr);
		hipLaunchKernelGGL(k_partition_scatter, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, n_parts,
		                   d_cursors, reinterpret_cast<Node *>(d_nodes), capacity);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_cursors);
	if (e != hipSuccess) return hip_fail(e, "partition_export", __LINE__);
	return DBGK_OK;
}

extern "C" int dbgk_merge_nodes(dbgk_handle *h, const dbgk_node *d_nodes, uint64_t n)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || (n && !d_nodes)) return DBGK_ERR_ARG;
	if ((uintptr_t)d_nodes & 15u) return DBGK_ERR_ARG;
	if (h->sharded) return DBGK_ERR_STATE; // use dbgk_shard_merge
	int rc = use_device(h);
	if (rc) return rc;
	if (n == 0) return DBGK_OK;
	if (h->zero_pending) { // PARTITION handle used as a merge target before any region build: the table must be
		rc = zero_table_now(h); // empty for the direct path; records pushed so far (or later) are built on top of it
		if (rc) return rc;
		h->incr = true;
	}
	TimedSpan sp;
	rc = span_begin(h, PH_FIXUP, sp);
	if (rc) return rc;
	hipLaunchKernelGGL(k_merge_nodes, dim3(grid_for(h, n)), dim3(kBlock), 0, h->stream, reinterpret_cast<const Node *>(d_nodes), n,
	                   h->tref(), h->d_ctr);
	HIPCHK(hipGetLastError());
	return span_end(h, sp);
}

extern "C" int dbgk_copy_nodes_peer(dbgk_handle *dst, dbgk_node *d_dst, dbgk_handle *src, const dbgk_node *d_src, uint64_t n)
{
	if (!dst || !src || (n && (!d_dst || !d_src))) return DBGK_ERR_ARG;
	if (n == 0) return DBGK_OK;
	int rc = dbgk_sync(src);
	if (rc) return rc;
	rc = use_device(dst);
	if (rc) return rc;
	HIPCHK(hipMemcpyPeer(d_dst, dst->device, d_src, src->device, n * sizeof(dbgk_node)));
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// sharded tables (one handle per GPU, each owning a contiguous slot range of one global table)
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_plan_partition(uint64_t table_slots, uint64_t expected_kmers, uint32_t shard_count, uint32_t shard_index, dbgk_plan_info *out)
{
	if (!out || table_slots == 0) return DBGK_ERR_ARG;
	dbgk_handle *h = new (std::nothrow) dbgk_handle();   // never touches a device: plan_partition is host arithmetic
	if (!h) return DBGK_ERR_NOMEM;
	memset(&h->cfg, 0, sizeof h->cfg);
	h->cfg.kmer_size = 31;
	h->cfg.engine = DBGK_ENGINE_PARTITION;
	h->cfg.table_slots = table_slots;
	h->cfg.expected_kmers = expected_kmers;
	h->cfg.shard_count = shard_count;
	h->cfg.shard_index = shard_index;
	h->size = table_slots;
	h->magic = make_mod_magic(table_slots);
	int rc = plan_partition(h);
	if (rc == DBGK_OK && !h->part) rc = DBGK_ERR_ARG;
	if (rc == DBGK_OK) {
		const PartGeom &G = h->geom;
		memset(out, 0, sizeof *out);
		out->table_slots = G.size;
		out->r = G.r;
		out->level1_buckets = G.n1;
		out->final_per_level1 = G.n2;
		out->three_level = h->three ? 1u : 0u;
		out->buckets_per_rank = G.B;
		out->own_buckets = G.nb_own;
		out->first_bucket = G.b_lo;
		out->slot_lo = G.slot_lo;
		out->slot_hi = G.slot_hi;
		out->records_per_level1_bucket = G.cap1;
		out->records_per_final_bucket = G.cap2;
		const uint64_t n_entries = (uint64_t)G.n_ranks * G.B * G.n_sub;
		out->table_bytes = (G.slot_hi - G.slot_lo) * sizeof(Node);
		out->level1_store_bytes = n_entries * G.cap1 * 8;
		out->inbox_bytes = h->sharded ? n_entries * G.cap1 * 8 : 0;
		out->final_store_bytes = (uint64_t)G.nb_own * G.n2 * G.cap2 * 8 + (h->three ? (uint64_t)G.nb_own * h->fan_mid * h->g_mid.cap2 * 8 : 0);
	}
	delete h;
	return rc;
}

extern "C" int dbgk_shard_buffers(dbgk_handle *h, dbgk_shard_info *out)
{
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !out) return DBGK_ERR_ARG;
	if (h->wide) { // 16-byte records; the buffers are those of the CURRENT pass (dbgk_wide_pass_info)
		if (!h->wpart || !h->wmulti) return DBGK_ERR_STATE;
		const WPartGeom &G = h->wgeom;
		memset(out, 0, sizeof(*out));
		out->n_ranks = G.n_ranks;
		out->rank = G.rank;
		out->slot_lo = G.slot_lo;
		out->slot_hi = G.slot_hi;
		out->table_slots_global = h->size;
		out->buckets_per_rank = G.Bp;
		out->own_buckets = wide_pass_buckets(h);
		out->bucket_bytes = G.cap1 * 16;
		out->cnt_bucket_bytes = 4;
		out->chunk_bytes = (uint64_t)G.Bp * G.cap1 * 16;
		out->cnt_chunk_bytes = (uint64_t)G.Bp * 4;
		out->d_send = h->wstore.l1;
		out->d_send_cnt = h->wstore.cnt1;
		out->d_recv = h->sharded ? (void *)h->winbox : (void *)h->wstore.l1;
		out->d_recv_cnt = h->sharded ? (void *)h->winbox_cnt : (void *)h->wstore.cnt1;
		return DBGK_OK;
	}
	if (!h->part) return DBGK_ERR_STATE;
	const PartGeom &G = h->geom;
	memset(out, 0, sizeof(*out));
	out->n_ranks = G.n_ranks;
	out->rank = G.rank;
	out->slot_lo = G.slot_lo;
	out->slot_hi = G.slot_hi;
	out->table_slots_global = h->size;
	out->buckets_per_rank = G.B;
	out->own_buckets = G.nb_own;
	out->bucket_bytes = (uint64_t)G.n_sub * G.cap1 * 8;
	out->cnt_bucket_bytes = (uint64_t)G.n_sub * 4;
	out->chunk_bytes = (uint64_t)G.B * G.n_sub * G.cap1 * 8;
	out->cnt_chunk_bytes = (uint64_t)G.B * G.n_sub * 4;
	out->d_send = h->store.l1;
	out->d_send_cnt = h->store.cnt1;
	out->d_recv = h->sharded ? (void *)h->inbox : (void *)h->store.l1;
	out->d_recv_cnt = h->sharded ? (void *)h->inbox_cnt : (void *)h->store.cnt1;
	return DBGK_OK;
}

extern "C" int dbgk_shard_mark_exchanged(dbgk_handle *h)
{
	if (!h || !(h->sharded || (h->wide && h->wmulti))) return DBGK_ERR_STATE;
	h->exchanged = true;
	return DBGK_OK;
}

extern "C" int dbgk_shard_plan(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->wide) {
		if (!h->wmulti || !h->wpass_open || h->finalized) return DBGK_ERR_STATE;
		int wrc = use_device(h);
		if (wrc) return wrc;
		return wide_plan_pass(h);
	}
	if (!h->part || h->part_built || h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	return part_plan(h);
}

extern "C" int dbgk_shard_build_range(dbgk_handle *h, uint32_t j0, uint32_t j1)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->wide) { // own-bucket indices of the current pass, in order
		if (!h->wmulti || !h->wpass_open || !h->wplanned || h->finalized) return DBGK_ERR_STATE;
		if (j0 != h->wnext || j1 < j0 || j1 > wide_pass_buckets(h)) return DBGK_ERR_ARG;
		int wrc = use_device(h);
		if (wrc) return wrc;
		return j1 > j0 ? wide_build_range(h, j0, j1) : DBGK_OK;
	}
	if (!h->part || !h->part_planned || h->part_built || h->finalized) return DBGK_ERR_STATE;
	if (j0 != h->next_bucket || j1 < j0 || j1 > h->geom.nb_own) return DBGK_ERR_ARG; // ranges are consumed in order, each bucket once
	int rc = use_device(h);
	if (rc) return rc;
	return part_build_range(h, j0, j1, true);
}

static int shard_list(dbgk_handle *h, void *list, unsigned long long *d_n, uint64_t cap, dbgk_node **d_nodes, uint64_t *n)
{
	if (!h || !d_nodes || !n) return DBGK_ERR_ARG;
	if (!(h->part || (h->wide && h->wmulti)) || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long v = 0;
	HIPCHK(hipMemcpyAsync(&v, d_n, 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	*d_nodes = reinterpret_cast<dbgk_node *>(list);
	*n = v < cap ? v : cap;
	if (!h->wide && list == (void *)h->store.ovf) return DBGK_OK; // the surplus beyond the list was aggregated in the side table (dbgk_shard_heavy)
	return v > cap ? DBGK_ERR_CAPACITY : DBGK_OK;
}

// (WIDE handles: the lists hold 32-byte dbgk_node32 entries -- nodes {hi, lo, l_link, r_link} / observations {hi, lo, lb, rb})
extern "C" int dbgk_shard_outgoing(dbgk_handle *h, dbgk_node **d_nodes, uint64_t *n)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->wide) return shard_list(h, h->wstore.outgoing, h->wstore.outgoing_n, h->wstore.outgoing_cap, d_nodes, n);
	return shard_list(h, h->store.outgoing, h->store.outgoing_n, h->store.outgoing_cap, d_nodes, n);
}

extern "C" int dbgk_shard_overflow(dbgk_handle *h, dbgk_node **d_triples, uint64_t *n)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->wide) return shard_list(h, h->wstore.ovf, &h->wstore.ovf_n[0], h->wstore.ovf_cap, d_triples, n);
	return shard_list(h, h->store.ovf, &h->store.ovf_n[0], h->store.ovf_cap, d_triples, n);
}

extern "C" int dbgk_shard_heavy(dbgk_handle *h, dbgk_node **d_table, uint64_t *n_slots)
{
	if (!h || !d_table || !n_slots) return DBGK_ERR_ARG;
	if (h->wide) { // no side table of aggregated surplus in the wide path: a full overflow list is DBGK_ERR_CAPACITY at finalize
		*d_table = nullptr;
		*n_slots = 0;
		return (h->wmulti && h->finalized) ? DBGK_OK : DBGK_ERR_STATE;
	}
	if (!h->part || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long v = 0;
	HIPCHK(hipMemcpyAsync(&v, &h->store.ovf_n[0], 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	*d_table = reinterpret_cast<dbgk_node *>(h->store.hh);
	*n_slots = (h->store.hh && v > h->store.ovf_cap) ? h->store.hh_size : 0; // unused unless the overflow list ran full
	return DBGK_OK;
}

extern "C" int dbgk_shard_merge(dbgk_handle *h, const dbgk_node *d_nodes, uint64_t n, int is_triple, int from_previous_shard)
{
	if (!h || (n && !d_nodes)) return DBGK_ERR_ARG;
	if (!(h->sharded || (h->wide && h->wmulti)) || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	if (n == 0) return DBGK_OK;
	TimedSpan sp;
	rc = span_begin(h, PH_FIXUP, sp);
	if (rc) return rc;
	if (h->wide) {
		hipLaunchKernelGGL(k_wide_merge_sharded, dim3(grid_for(h, n)), dim3(kBlock), 0, h->stream, reinterpret_cast<const dbgk_node32 *>(d_nodes),
		                   (const unsigned long long *)nullptr, n, n, is_triple ? 1 : 0, from_previous_shard ? 1 : 0, h->wgeom, h->wstore, h->wnodes, h->d_ctr);
		HIPCHK(hipGetLastError());
		return span_end(h, sp);
	}
	hipLaunchKernelGGL(k_merge_sharded, dim3(grid_for(h, n)), dim3(kBlock), 0, h->stream, reinterpret_cast<const Node *>(d_nodes),
	                   (const unsigned long long *)nullptr, n, n, is_triple ? 1 : 0, from_previous_shard ? 1 : 0, h->geom, h->store, h->table,
	                   h->d_ctr);
	HIPCHK(hipGetLastError());
	return span_end(h, sp);
}

extern "C" int dbgk_wide_pass_info(dbgk_handle *h, uint32_t *n_passes, uint32_t *passes_done)
{
	if (!h) return DBGK_ERR_ARG;
	if (!h->wide) return DBGK_ERR_STATE;
	if (n_passes) *n_passes = h->wpart ? h->wgeom.n_passes : 1u;
	if (passes_done) *passes_done = h->wpart ? h->wpasses_done : 0u;
	return DBGK_OK;
}

extern "C" int dbgk_wide_begin_pass(dbgk_handle *h, uint32_t pass)
{
	if (!h) return DBGK_ERR_ARG;
	if (!h->wide || !h->wpart || h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	if (pass == 0 && h->wpass_open && h->wpasses_done == 0 && h->pending_kmers == 0) return DBGK_OK; // pass 0 is open after create / reset
	return wide_begin_pass(h, pass);
}

extern "C" int dbgk_wide_end_pass(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	if (!h->wide || !h->wpart || !h->wmulti || h->finalized || !h->wpass_open) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	return wide_end_pass(h);
}

extern "C" int dbgk_shard_side_export(dbgk_handle *h, dbgk_node32 **d_nodes, uint64_t *n)
{
	if (!h || !d_nodes || !n) return DBGK_ERR_ARG;
	if (!h->wide || !h->wmulti || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipMemsetAsync(h->w_side_n, 0, 8, h->stream));
	hipLaunchKernelGGL(k_wide_side_export, dim3(16), dim3(kBlock), 0, h->stream, h->wside, h->d_ctr, h->w_side_out, h->w_side_n, (uint64_t)kWideSideSlots + 1);
	HIPCHK(hipGetLastError());
	unsigned long long v = 0;
	HIPCHK(hipMemcpyAsync(&v, h->w_side_n, 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	*d_nodes = h->w_side_out;
	*n = v + 1; // + the key-0 node in front
	return DBGK_OK;
}

extern "C" int dbgk_shard_side_clear(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	if (!h->wide || !h->wmulti || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	// the claims in the side table were counted as new keys of this handle: take them back with the table
	HIPCHK(hipMemsetAsync(h->w_side_n, 0, 8, h->stream));
	hipLaunchKernelGGL(k_wide_side_export, dim3(16), dim3(kBlock), 0, h->stream, h->wside, h->d_ctr, h->w_side_out, h->w_side_n, (uint64_t)kWideSideSlots + 1);
	HIPCHK(hipGetLastError());
	unsigned long long v = 0;
	HIPCHK(hipMemcpyAsync(&v, h->w_side_n, 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	Counters c;
	HIPCHK(hipMemcpyAsync(&c, h->d_ctr, sizeof c, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	c.n_new -= v;
	c.polyA_links = 0;
	HIPCHK(hipMemcpyAsync(h->d_ctr, &c, sizeof c, hipMemcpyHostToDevice, h->stream));
	HIPCHK(hipMemsetAsync(h->wside, 0, kWideSideSlots * sizeof(WNode), h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

extern "C" int dbgk_add_polyA(dbgk_handle *h, uint32_t l_link, uint32_t r_link)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->wide || h->kfreq || h->seed) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	Node nd;
	nd.kmer = 0;
	nd.links = (uint64_t)l_link | ((uint64_t)r_link << 32);
	Node *d = nullptr;
	if (hipMalloc(&d, sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
	hipError_t e = hipMemcpyAsync(d, &nd, sizeof(Node), hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) {
		if (h->sharded)
			hipLaunchKernelGGL(k_merge_sharded, dim3(1), dim3(kBlock), 0, h->stream, d, (const unsigned long long *)nullptr, (uint64_t)1, (uint64_t)1, 0,
			                   0, h->geom, h->store, h->table, h->d_ctr);
		else
			hipLaunchKernelGGL(k_merge_nodes, dim3(1), dim3(kBlock), 0, h->stream, d, (uint64_t)1, h->tref(), h->d_ctr);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return hip_fail(e, "add_polyA", __LINE__);
	return DBGK_OK;
}

extern "C" int dbgk_memcpy_d2d(dbgk_handle *h, void *d_dst, const void *d_src, size_t bytes)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// utilities
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_synth_reads_device(dbgk_handle *h, const dbgk_synth_params *p, uint64_t first_read, uint64_t n_reads,
                                       char *d_bases, uint64_t *d_offsets)
{
	if (!h || !p || !d_bases || !d_offsets) return DBGK_ERR_ARG;
	if (p->read_len == 0 || p->read_len > 1024 || p->genome_len < p->read_len) return DBGK_ERR_ARG;
	if ((uintptr_t)d_bases & 15u) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t chunks = (n_reads * (uint64_t)p->read_len + 15) >> 4;
	hipLaunchKernelGGL(k_synth_reads, dim3(grid_for(h, std::max<uint64_t>(chunks, n_reads + 1))), dim3(kBlock), 0, h->stream, *p,
	                   first_read, n_reads, d_bases, d_offsets);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

extern "C" int dbgk_device_malloc(dbgk_handle *h, size_t bytes, void **d_ptr)
{
	if (!h || !d_ptr) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (hipMalloc(d_ptr, bytes ? bytes : 16) != hipSuccess) {
		*d_ptr = nullptr;
		return DBGK_ERR_NOMEM;
	}
	return DBGK_OK;
}

extern "C" int dbgk_device_free(dbgk_handle *h, void *d_ptr)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(h->stream));
	HIPCHK(hipFree(d_ptr));
	return DBGK_OK;
}

extern "C" int dbgk_memcpy_d2h(dbgk_handle *h, void *dst, const void *d_src, size_t bytes)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

extern "C" int dbgk_memcpy_h2d(dbgk_handle *h, void *d_dst, const void *src, size_t bytes)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

extern "C" int dbgk_get_timings(dbgk_handle *h, dbgk_timings *out)
{
	if (!h || !out) return DBGK_ERR_ARG;
	memset(out, 0, sizeof(*out));
	out->mark_ms = h->phase_ms[PH_MARK];
	out->insert_ms = h->phase_ms[PH_INSERT];
	out->partition_ms = h->phase_ms[PH_PARTITION];
	out->build_ms = h->phase_ms[PH_BUILD];
	out->fixup_ms = h->phase_ms[PH_FIXUP];
	out->finalize_ms = h->phase_ms[PH_FINALIZE];
	out->insert_launches = h->insert_launches;
	out->l2_build_wall_ms = h->phase_ms[PH_L2_BUILD_WALL];
	out->partition_launches = h->partition_launches;
	out->uniform_launches = h->uniform_launches;
	out->prefix_launches = h->prefix_launches;
	return DBGK_OK;
}

extern "C" int dbgk_reset_timings(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	for (auto &v : h->phase_ms) v = 0.f;
	h->insert_launches = 0;
	h->partition_launches = 0;
	h->uniform_launches = 0;
	h->prefix_launches = 0;
	return DBGK_OK;
}

// The "measured HBM bandwidth" of the roofline.  A runtime DtoD memcpy reads 4.7-5.4 TB/s on this pool depending on the
// box; the guide's figure for a 16-byte-per-lane copy kernel is 6.3.  So the probe runs its OWN streaming kernels as well -- 16 bytes
// per lane, four loads in flight, 1024-thread persistent workgroups, default and non-temporal policy, one and two workgroups per CU
// (profiles/ubench/hbm_stream.hip is the sweep these shapes come from) -- and reports the BEST rate seen, copy bytes = read + written.
namespace {
typedef uint32_t probe_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(1024) void k_probe_copy(const probe_u32x4 *__restrict__ src, probe_u32x4 *__restrict__ dst, size_t n_vec)
{
	constexpr int U = 4;
	const size_t tile = (size_t)1024 * U, n_tiles = n_vec / tile;
	for (size_t g = blockIdx.x; g < n_tiles; g += gridDim.x) {
		probe_u32x4 v[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			const probe_u32x4 *p = src + g * tile + (size_t)u * 1024 + threadIdx.x;
			v[u] = NT ? __builtin_nontemporal_load(p) : *p;
		}
#pragma unroll
		for (int u = 0; u < U; u++) {
			probe_u32x4 *q = dst + g * tile + (size_t)u * 1024 + threadIdx.x;
			if (NT) __builtin_nontemporal_store(v[u], q); else *q = v[u];
		}
	}
}
} // namespace

extern "C" int dbgk_measure_copy_bandwidth(dbgk_handle *h, size_t bytes, int iters, double *gbps)
{
	if (!h || !gbps || bytes < 65536 || iters < 1) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	bytes &= ~(size_t)65535; // whole tiles of the probe kernels
	void *a = nullptr, *b = nullptr;
	if (hipMalloc(&a, bytes) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&b, bytes) != hipSuccess) {
		(void)hipFree(a);
		return DBGK_ERR_NOMEM;
	}
	hipEvent_t e0, e1;
	hipError_t e = hipEventCreate(&e0);
	if (e == hipSuccess) e = hipEventCreate(&e1);
	if (e == hipSuccess) e = hipMemsetAsync(a, 1, bytes, h->stream);
	double best = 0.0;
	for (int variant = 0; variant < 5 && e == hipSuccess; variant++) {
		auto run = [&]() -> hipError_t {
			const size_t n_vec = bytes / 16;
			switch (variant) {
			case 0: return hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, h->stream);
			case 1: hipLaunchKernelGGL(k_probe_copy<true>, dim3(h->n_cu), dim3(1024), 0, h->stream, (const probe_u32x4 *)a, (probe_u32x4 *)b, n_vec); break;
			case 2: hipLaunchKernelGGL(k_probe_copy<true>, dim3(2 * h->n_cu), dim3(1024), 0, h->stream, (const probe_u32x4 *)a, (probe_u32x4 *)b, n_vec); break;
			case 3: hipLaunchKernelGGL(k_probe_copy<false>, dim3(h->n_cu), dim3(1024), 0, h->stream, (const probe_u32x4 *)a, (probe_u32x4 *)b, n_vec); break;
			default: hipLaunchKernelGGL(k_probe_copy<false>, dim3(2 * h->n_cu), dim3(1024), 0, h->stream, (const probe_u32x4 *)a, (probe_u32x4 *)b, n_vec); break;
			}
			return hipGetLastError();
		};
		e = run(); // warm-up
		if (e == hipSuccess) e = hipEventRecord(e0, h->stream);
		for (int i = 0; i < iters && e == hipSuccess; i++) e = run();
		if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
		float ms = 0.f;
		if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
		if (e == hipSuccess && ms > 0.f) best = std::max(best, (2.0 * (double)bytes * iters) / (ms * 1e-3) / 1e9); // bytes read + bytes written
	}
	(void)hipFree(a);
	(void)hipFree(b);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	if (e != hipSuccess) return hip_fail(e, "measure_copy_bandwidth", __LINE__);
	*gbps = best;
	return DBGK_OK;
}

// Random 64-byte gather (the practical ceiling of the engines that touch one random node per k-mer
// occurrence, DIRECT / WIDE-atomic / SEEDIDX).  Four lanes fetch one 64-byte sector each (16 bytes per lane) at a
// pseudo-random sector of a buffer far larger than the caches; every lane group runs its own xorshift stream.
__global__ __launch_bounds__(kBlock) void k_random_gather64(const uint4 *__restrict__ buf, uint64_t n_sectors, uint32_t per_group,
                                                            unsigned long long *__restrict__ sink)
{
	const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	const uint64_t group = tid >> 2;
	const uint32_t part = (uint32_t)tid & 3u;
	uint64_t x = (group + 1u) * 0x9E3779B97F4A7C15ull;
	uint32_t acc = 0;
	for (uint32_t i = 0; i < per_group; i++) {
		x ^= x << 13;
		x ^= x >> 7;
		x ^= x << 17;
		const uint64_t sector = (uint64_t)(((unsigned __int128)x * n_sectors) >> 64);
		typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
		const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(buf) + sector * 4u + part);
		acc ^= v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345677u) atomicAdd(sink, 1ull); // keeps the loads alive
}

extern "C" int dbgk_measure_gather_bandwidth(dbgk_handle *h, size_t bytes, uint64_t n_accesses, double *gbps, double *gaccesses_per_s)
{
	if (!h || !gbps || bytes < (1u << 20) || n_accesses < 1024) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	void *a = nullptr;
	unsigned long long *sink = nullptr;
	if (hipMalloc(&a, bytes) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&sink, 8) != hipSuccess) {
		(void)hipFree(a);
		return DBGK_ERR_NOMEM;
	}
	const uint32_t per_group = 64;
	const uint64_t groups = (n_accesses + per_group - 1) / per_group;
	const uint64_t blocks = (groups * 4 + kBlock - 1) / kBlock;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	hipError_t e = hipEventCreate(&e0);
	if (e == hipSuccess) e = hipEventCreate(&e1);
	if (e == hipSuccess) e = hipMemsetAsync(a, 1, bytes, h->stream);
	if (e == hipSuccess) e = hipMemsetAsync(sink, 0, 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_random_gather64, dim3((unsigned)std::min<uint64_t>(blocks, 1u << 20)), dim3(kBlock), 0, h->stream, (const uint4 *)a, (uint64_t)(bytes >> 6), 4u,
		                   sink); // warm-up (page tables)
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipEventRecord(e0, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_random_gather64, dim3((unsigned)blocks), dim3(kBlock), 0, h->stream, (const uint4 *)a, (uint64_t)(bytes >> 6), per_group, sink);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	float ms = 0.f;
	if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
	(void)hipFree(a);
	(void)hipFree(sink);
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	if (e != hipSuccess) return hip_fail(e, "measure_gather_bandwidth", __LINE__);
	const double done = (double)(blocks * (kBlock / 4)) * per_group;
	*gbps = done * 64.0 / (ms * 1e-3) / 1e9;
	if (gaccesses_per_s) *gaccesses_per_s = done / (ms * 1e-3) / 1e9;
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// several GPUs in one process
// ---------------------------------------------------------------------------------------------
#include "dbgk_comm.h"
