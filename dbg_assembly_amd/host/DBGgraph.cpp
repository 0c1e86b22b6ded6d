// DBGgraph.cpp -- build_debruijn_graph() on an MI355X.
//
// Host responsibilities only: parse the (optionally gzip'ed) one-line FASTA/FASTQ files into
// batches of sequences, stream them to the GPU through the C ABI of include/dbgk.h, and
// materialise the finished table as the host KmerSet the reference's contig stage expects.  All
// k-mer work (canonicalisation, hashing, counting) happens in the HIP kernels.  The sequences travel
// 2 BITS PER BASE: the reader threads pack them (dbgk_pack_reads -- the reference's own alphabet[],
// seqKmer.cpp:9-19, is two bits wide) straight into the pinned staging buffers, a quarter of the bytes
// over PCIe; DBGK_HOST_ASCII=1 hands over the raw bytes instead (the GPU then packs them).
//
// Reference behaviour followed here (paths relative to /root/reference/DBG_contig/):
//   record detection            DBGgraph.cpp:244-272   (first character of a line; next line = sequence)
//   totals / stderr protocol    DBGgraph.cpp:380-428
//   table sizing                kmerSet.cpp:98-127 (initial), DBGgraph.cpp:337-351 + kmerSet.cpp:132-148 (doubling)
//   key-0 node appended last    DBGgraph.cpp:418
#include "DBGgraph.h"

#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <thread>

#include "dbgk.h"
#include "dbgk_env.h"
#include "reads_io.h"

// ---- the reference's globals (DBGgraph.cpp:10-34), same names and defaults -------------------
int KmerSize = 31;
int maxReadLen = 250;
int KmerNumInRead = 0;
int Input_file_format = 1;
string Output_prefix = "output";
int threadNum = 10;
KmerSet *kset = NULL;
KmerSet128 *kset_wide = NULL;
double initHashSize = 1.0;
uint64_t maxDoubleHashTimes = 10;
uint64_t doubleHashTimes = 0;
float hashLoadFactor = 0.7;
int BufferNum = 10000;
string *RawReads = NULL;
uint64_t *StoreKmer = NULL;
uint8_t *StoreLeftBase = NULL;
uint8_t *StoreRightBase = NULL;
uint8_t *Signal = NULL;
uint64_t Kmer_total_num = 0;
uint64_t Total_reads_num = 0;
uint64_t KmerHeadMaskVal = 0;
uint64_t KmerRCOrVal[4];
KmerNode *PolyA = NULL;
clock_t time_start;
clock_t time_end;

int DbgkLastStatus = 0;
uint16_t *DbgkKmerLinks = NULL;
vector<uint64_t> DbgkTipNodes;
vector<uint64_t> DbgkBranchNodes;
extern int KmerFreqCutoff __attribute__((weak)); // -D of the contig stage (defined by main.cpp / contig.cpp when linked in)

namespace {

const uint64_t kPartitionMinSlots = 67108879;   // smallest table the PARTITION engine takes (2^26 slots), a prime above it
const uint64_t kPartitionMaxSlots = (1ull << 34) - (1ull << 24); // the engine's geometry: < 2^34 slots

// bytes of the pending batch: grows without zero-filling (std::vector::resize would clear every batch's 256 MB first).  In
// EXTERNAL mode it is the handle's pinned staging buffer itself (dbgk_push_acquire): the batch is parsed straight into the memory
// the H2D copy reads from; if a batch outgrows that buffer it moves to the heap and is handed over with dbgk_push_reads.
struct ByteBuffer {
	char *p = nullptr;
	size_t n = 0, cap = 0;
	bool external = false;
	~ByteBuffer() { if (!external) free(p); }
	char *data() { return p; }
	size_t size() const { return n; }
	void clear() { n = 0; }
	void set_external(char *mem, size_t bytes)
	{
		if (!external) free(p);
		p = mem;
		cap = bytes;
		n = 0;
		external = true;
	}
	void detach() // give the external memory back (committed)
	{
		if (external) { p = nullptr; cap = 0; n = 0; external = false; }
	}
	void reserve(size_t want)
	{
		if (want <= cap) return;
		size_t c = cap ? cap : (1u << 16);
		while (c < want) c += c / 2 + (1u << 16);
		char *q = external ? static_cast<char *>(malloc(c)) : static_cast<char *>(realloc(p, c));
		if (!q) abort();
		if (external) { // outgrown the staging buffer: continue on the heap
			memcpy(q, p, n);
			external = false;
		}
		p = q;
		cap = c;
	}
	void append(const char *src, size_t len)
	{
		reserve(n + len);
		memcpy(p + n, src, len);
		n += len;
	}
	void grow_uninitialized(size_t len)
	{
		reserve(n + len);
		n += len;
	}
};

struct Session {
	dbgk_handle *h = nullptr;              // one GPU
	dbgk_comm *comm = nullptr;             // DBGK_GPUS=N / DBGK_GPU_LIST: N sharded handles of one table (always PARTITION)
	ByteBuffer bases;                 // sequences of the pending batch, back to back
	std::vector<uint64_t> offsets;    // offsets.size() == reads in batch + 1
	// plain files are mapped and parsed with several threads (reads_io.h): their reads are first only NOTED (pointer into the
	// mapping, length; offsets grows) and copied into `bases` by the same threads when the batch is handed over
	std::vector<dbgk_read_ref> noted;
	uint64_t noted_bytes = 0;
	int parse_threads = 4;
	// 2-bit hand-over (default): `bases` holds packed WORDS, its size() counts BASES (a quarter of the bytes are in use;
	// capacities are kept in bases = bytes of the ASCII form, so every limit below means the same in both modes)
	bool packed = true;
	uint64_t batch_other = 0;         // bytes outside ACGTNacgtn met while packing the pending batch (read as 'A', counted)
	std::thread creator;              // dbgk_create runs beside the reading of the first file window
	int create_rc = DBGK_OK;
	std::string create_err;
	bool zero_copy = false;           // one GPU: batches are assembled in the handle's pinned staging buffers (dbgk_push_acquire / _commit)
	uint64_t *staged_offsets = nullptr;
	uint64_t staged_cap_reads = 0;
	uint64_t batch_limit = 128ull << 20;
	uint64_t device_slots = 0;
	uint64_t next_progress = 0;
	int status = DBGK_OK;
	bool partition = false;                // PARTITION engine (streaming: records are flushed into the table as needed)
	bool wide = false;                     // -k 33..63: WIDE engine (128-bit keys); the table has the size -i asks for, no enlarge schedule
	// The reference's table, tracked exactly (DBGgraph.cpp:329-351): after every FULL block of BufferNum
	// reads `count > max` decides about a doubling, and at the -e cap about abandoning the file.
	uint64_t ref_size = 0, ref_max = 0;    // size / max of the kset the reference would hold now
	uint64_t count_known = 0;              // distinct non-zero keys on the device at the last exact reading
	uint64_t bound_since = 0;              // k-mer windows handed over (pushed or pending here) since then: new keys <= this
	uint64_t reads_in_block = 0;           // reads since the last full block of the current file
	bool stop_file = false;                // the reference would have left this file's block loop (-e cap)
	// host wall clock per phase (DBGK_TIMINGS): device calls made while parsing are timed on their own
	double t_create = 0, t_parse = 0, t_push = 0, t_count = 0, t_finalize = 0, t_export = 0, t_pack = 0;
	// the host table the consumer gets, allocated at the size -i asks for when the run starts and touched page by page by a few
	// threads while the reads are parsed (the kernel hands out -- and zeroes -- 16 bytes per slot: 9.6 GB for cfg2; left to the
	// export, that is most of its time).  Used if the reference's schedule ends at that size, dropped otherwise.
	KmerNode *early_array = nullptr;
	uint64_t early_size = 0;
	std::vector<std::thread> early_threads;
	std::atomic<bool> early_stop{false};
	double t_early = 0;
	// DBGK_LAYOUT=ref: reproduce the reference's -t 1 slot layout (first-seen order replay)
	bool ref_layout = false;
	uint64_t pos = 0;                      // bases handed to the device so far (+ pending batch)
	std::vector<uint64_t> full_block_ends; // position right after every FULL block of BufferNum reads
};

Session *g_session = nullptr;

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Stopwatch { // adds its lifetime to a counter
	double &acc, t0;
	explicit Stopwatch(double &a) : acc(a), t0(now_s()) {}
	~Stopwatch() { acc += now_s() - t0; }
};

void fail(Session &S, int rc, const char *what);

// the handle is created on a thread of its own while the first window of the first file is read: wait for it here
void ensure_created(Session &S)
{
	if (!S.creator.joinable()) return;
	S.creator.join();
	if (S.create_rc != DBGK_OK) {
		if (S.status == DBGK_OK) {
			S.status = S.create_rc;
			cerr << "\nAlert message: dbgk_create failed: " << dbgk_strerror(S.create_rc) << " [" << S.create_err << "]; the input is ignored" << endl;
		}
		S.zero_copy = false;
	}
}

void start_early_table(Session &S, uint64_t size)
{
	static const bool off = DBGK_EXPERIMENT_ENV("DBGK_NO_EARLY_TABLE") != nullptr;
	const size_t min_bytes = dbgk_hook("early_table_min") ? (size_t)strtoull(dbgk_hook("early_table_min"), NULL, 10) : ((size_t)64 << 20); // (tests: 0)
	if (off || size * sizeof(KmerNode) < min_bytes) return;
	S.early_array = static_cast<KmerNode *>(kmerset_alloc(size * sizeof(KmerNode), false));
	if (!S.early_array) return;
	S.early_size = size;
	const int T = dbgk_hook("early_threads") ? std::max(1, atoi(dbgk_hook("early_threads"))) : 4;
	const size_t bytes = size * sizeof(KmerNode), per = ((bytes / (size_t)T) + 4095) & ~(size_t)4095;
	const double t0 = now_s();
	for (int t = 0; t < T; t++)
		S.early_threads.emplace_back([&S, t, per, bytes, t0, T]() {
			volatile char *p = reinterpret_cast<volatile char *>(S.early_array);
			const size_t a = std::min(bytes, per * (size_t)t), b = std::min(bytes, a + per);
			for (size_t i = a; i < b && !S.early_stop.load(std::memory_order_relaxed); i += 4096) p[i] = 0; // (fresh pages: zero anyway)
			if (t == T - 1) S.t_early = now_s() - t0;
		});
}

// -> the array if it has `size` slots (all pages present), else NULL (released)
KmerNode *take_early_table(Session &S, uint64_t size)
{
	if (S.early_size != size) S.early_stop.store(true);
	for (auto &t : S.early_threads) t.join();
	S.early_threads.clear();
	KmerNode *a = S.early_array;
	S.early_array = nullptr;
	if (a && S.early_size != size) {
		free(a);
		a = nullptr;
	}
	S.early_size = 0;
	return a;
}

void fail(Session &S, int rc, const char *what)
{
	if (S.status == DBGK_OK) {
		S.status = rc;
		cerr << "\nAlert message: " << what << " failed: " << dbgk_strerror(rc);
		if (rc == DBGK_ERR_HIP || rc == DBGK_ERR_ARG || rc == DBGK_ERR_STATE) cerr << " [" << dbgk_last_error() << "]";
		cerr << "; the remaining input is ignored" << endl;
	}
}

inline uint64_t windows_of(size_t len)
{
	const uint64_t rl = std::min<uint64_t>(len, (uint64_t)maxReadLen);
	return rl >= (uint64_t)KmerSize ? rl - (uint64_t)KmerSize + 1 : 0;
}

inline float clamped_load_factor()
{
	return hashLoadFactor <= 0 ? 0.25f : (hashLoadFactor >= 1 ? 0.75f : hashLoadFactor); // kmerSet.cpp:107-108
}

// distinct non-zero keys on the device, exactly: everything handed over so far is in the table afterwards
// (the PARTITION engine flushes its record store for it)
void exact_count(Session &S)
{
	ensure_created(S);
	if (S.status != DBGK_OK) return;
	Stopwatch sw(S.t_count);
	int rc = S.comm ? dbgk_comm_flush(S.comm) : (S.partition ? dbgk_flush(S.h) : DBGK_OK);
	if (rc != DBGK_OK) return fail(S, rc, "dbgk_flush");
	dbgk_stats st;
	rc = S.comm ? dbgk_comm_refresh_stats(S.comm, &st) : dbgk_refresh_stats(S.h, &st);
	if (rc != DBGK_OK) return fail(S, rc, "dbgk_refresh_stats");
	S.count_known = st.count - 1;   // st.count includes the key-0 node, which the reference adds last (DBGgraph.cpp:418)
	S.bound_since = 0;
	for (size_t i = 1; i < S.offsets.size(); i++) S.bound_since += windows_of(S.offsets[i] - S.offsets[i - 1]); // not pushed yet
}

// make sure the device table can absorb everything handed over so far, growing it if necessary (the
// device-side counterpart of the reference's enlarge step; the size of the HOST table follows the
// reference's own schedule, see end_of_full_block).  No device round trip while the bound says it fits.
void reserve_device_slots(Session &S)
{
	if (S.wide) return; // a WIDE table is sized once (-i); a table too small ends in DBGK_ERR_TABLE_FULL at finalize
	if ((double)S.count_known + (double)S.bound_since <= 0.80 * (double)S.device_slots) return;
	exact_count(S);
	if (S.status != DBGK_OK) return;
	const double need = (double)S.count_known + (double)S.bound_since;
	if (need <= 0.80 * (double)S.device_slots) return;
	const uint64_t target = find_next_prime((uint64_t)(need / 0.55) + 16);
	int rc = S.comm ? dbgk_comm_resize(S.comm, std::min(target, kPartitionMaxSlots)) : dbgk_resize_table(S.h, target);
	if (rc != DBGK_OK) return fail(S, rc, "dbgk_resize_table");
	S.device_slots = target;
	cerr << "Enlarge device hash array size to be: " << target << endl;
}

// zero-copy batches: the staging buffers of the handle's next slot become the batch buffer
void acquire_staging(Session &S)
{
	ensure_created(S);
	if (!S.zero_copy || S.bases.external || S.bases.size() || S.status != DBGK_OK) return;
	char *mem = nullptr;
	uint64_t cap_bases = 0;
	Stopwatch sw(S.t_push);
	const int rc = dbgk_push_acquire(S.h, &mem, &S.staged_offsets, &cap_bases, &S.staged_cap_reads);
	if (rc != DBGK_OK) {
		S.zero_copy = false; // hand batches over by copy from here on
		return;
	}
	S.bases.set_external(mem, cap_bases);
}

// the reads noted since the last hand-over: their bytes into `bases`, several threads
void materialize_noted(Session &S)
{
	if (S.noted.empty()) return;
	acquire_staging(S);
	Stopwatch sw_pack(S.t_pack);
	const size_t first = S.offsets.size() - 1 - S.noted.size();
	S.bases.grow_uninitialized(S.noted_bytes);
	const int T = std::max(1, std::min(S.parse_threads, (int)(S.noted.size() / 4096 + 1)));
	const size_t per = (S.noted.size() + (size_t)T - 1) / (size_t)T;
	std::vector<uint64_t> other((size_t)T, 0);
	uint32_t *words = reinterpret_cast<uint32_t *>(S.bases.data());
	if (S.packed) {
		// every thread packs its reads as ONE 2-bit stream from the base position its share starts at; the words where two
		// shares meet (and the last one, if the batch ends inside it) are OR-ed into from both sides: zero beforehand.  The word
		// the appended range STARTS inside belongs to what is already there (its unused low bits are zero) and stays.
		const uint64_t p_start = S.offsets[first], p_end = S.offsets[first + S.noted.size()];
		for (int t = 0; t <= T; t++) {
			const uint64_t p = t == T ? p_end : S.offsets[first + std::min(S.noted.size(), per * (size_t)t)];
			if (t == T && (p & 15u) == 0) continue;                              // nothing of this batch reaches that word
			if ((p >> 4) == (p_start >> 4) && (p_start & 15u)) continue;        // holds earlier bases
			if (t > 0 && t < T && (p & 15u) == 0 && p == p_end) continue;       // (an empty share at the end)
			words[p >> 4] = 0u;
		}
	}
	auto copy = [&](int t) {
		const size_t a = std::min(S.noted.size(), per * (size_t)t), b = std::min(S.noted.size(), a + per);
		if (S.packed) {
			if (b > a) dbgk_pack_reads(&S.noted[a], b - a, words, S.offsets[first + a], &other[(size_t)t]);
			return;
		}
		for (size_t i = a; i < b; i++) memcpy(S.bases.data() + S.offsets[first + i], S.noted[i].seq, S.noted[i].len);
	};
	std::vector<std::thread> th;
	for (int t = 1; t < T; t++) th.emplace_back(copy, t);
	copy(0);
	for (auto &x : th) x.join();
	for (uint64_t o : other) S.batch_other += o;
	S.noted.clear();
	S.noted_bytes = 0;
}

void flush_batch(Session &S)
{
	materialize_noted(S);
	const uint64_t n_reads = S.offsets.size() - 1;
	if (n_reads == 0 || S.status != DBGK_OK) {
		S.bases.clear();
		S.offsets.assign(1, 0);
		return;
	}
	if (S.bases.external && n_reads > S.staged_cap_reads) S.bases.reserve(S.bases.cap + 1); // (more reads than offsets fit: hand over by copy)
	ensure_created(S);
	reserve_device_slots(S);
	if (S.status == DBGK_OK) {
		Stopwatch sw(S.t_push);
		int rc;
		const uint32_t *words = reinterpret_cast<const uint32_t *>(S.bases.data());
		if (S.bases.external && n_reads <= S.staged_cap_reads) { // the batch sits in the staging buffer already
			memcpy(S.staged_offsets, S.offsets.data(), (n_reads + 1) * sizeof(uint64_t));
			rc = S.packed ? dbgk_push_commit_packed(S.h, n_reads, S.batch_other) : dbgk_push_commit(S.h, n_reads);
		} else if (S.packed) {
			rc = S.comm ? dbgk_comm_push_reads_packed(S.comm, words, S.offsets.data(), n_reads, S.batch_other)
			            : dbgk_push_reads_packed(S.h, words, S.offsets.data(), n_reads, S.batch_other);
		} else {
			rc = S.comm ? dbgk_comm_push_reads(S.comm, S.bases.data(), S.offsets.data(), n_reads)
			            : dbgk_push_reads(S.h, S.bases.data(), S.offsets.data(), n_reads);
		}
		if (rc != DBGK_OK) fail(S, rc, "dbgk_push_reads");
	}
	S.batch_other = 0;
	S.bases.detach(); // (a staging buffer goes back to the handle; the next batch takes the other one)
	Total_reads_num += n_reads;
	if (Total_reads_num >= S.next_progress) {
		cerr << "Load reads block " << Total_reads_num << endl;
		S.next_progress = Total_reads_num + (uint64_t)std::max(BufferNum, 1) * 100;
	}
	S.bases.clear();
	S.offsets.assign(1, 0);
}

// The reference's check after every FULL block (DBGgraph.cpp:337-351): count > max -> one
// enlarge_kmerset_parallel(kset, 1, ...) while doublings are left, else the rest of the file is abandoned.
// count is only read from the device when the bound since the last exact reading could exceed max.
void end_of_full_block(Session &S)
{
	S.reads_in_block = 0;
	if (S.wide) return; // the reference's doubling schedule is defined for its 16-byte table only
	if (S.ref_layout) S.full_block_ends.push_back(S.pos);
	if (S.count_known + S.bound_since <= S.ref_max) return;
	flush_batch(S);
	exact_count(S);
	if (S.status != DBGK_OK || S.count_known <= S.ref_max) return;
	if (doubleHashTimes < maxDoubleHashTimes) {
		const float lf = clamped_load_factor();
		do {
			S.ref_size = find_next_prime(S.ref_size * 2);   // kmerSet.cpp:136
		} while ((float)S.ref_size * lf < (float)(S.count_known + 1));
		S.ref_max = (uint64_t)((float)S.ref_size * lf);     // kmerSet.cpp:145
		doubleHashTimes++;
		cerr << "Enlarge hash array size to be: " << S.ref_size << endl;
		cerr << "The expanded memory used now:  " << (double)S.ref_size / 1000000000 * 16 << " G" << endl;
	} else {
		cerr << "\nAlert message: Memory reach the maximum allowed, program have loaded " << Total_reads_num
		     << " reads, the left others are ignored\n" << endl;
		S.stop_file = true;
	}
}

inline void add_read(Session &S, const char *seq, size_t len)
{
	acquire_staging(S);
	if (S.bases.external && S.bases.size() + len > S.bases.cap && S.bases.size()) flush_batch(S), acquire_staging(S);
	if (S.packed) { // the read's codes go behind what is there; the words it newly reaches are cleared first (its ends are OR-ed in)
		const size_t at = S.bases.size();
		S.bases.grow_uninitialized(len);
		uint32_t *words = reinterpret_cast<uint32_t *>(S.bases.data());
		const size_t w0 = (at + 15) >> 4, w1 = (at + len + 15) >> 4;
		if (w1 > w0) memset(words + w0, 0, (w1 - w0) * sizeof(uint32_t));
		dbgk_pack_bases(seq, len, words, at, &S.batch_other);
	} else {
		S.bases.append(seq, len);
	}
	S.offsets.push_back(S.bases.size());
	S.bound_since += windows_of(len);
	S.pos += len;
	if (++S.reads_in_block == (uint64_t)std::max(BufferNum, 1)) end_of_full_block(S);
	else if (S.bases.size() >= S.batch_limit) flush_batch(S);
}

// the same bookkeeping for a read of a mapped file: its bytes are copied when the batch is handed over
inline void note_read(Session &S, const char *seq, size_t len)
{
	if (S.zero_copy && S.offsets.back() + len > S.batch_limit + (1u << 16) && S.offsets.size() > 1) flush_batch(S); // keep the batch inside the staging buffer
	S.noted.push_back(dbgk_read_ref{seq, (uint32_t)len});
	S.noted_bytes += len;
	S.offsets.push_back(S.offsets.back() + len);
	S.bound_since += windows_of(len);
	S.pos += len;
	if (++S.reads_in_block == (uint64_t)std::max(BufferNum, 1)) end_of_full_block(S);
	else if (S.offsets.back() >= S.batch_limit) flush_batch(S);
}

// note_read for a run of records (what one reader thread found in its slice of a file window): the same decisions at the same
// reads -- a flush when the batch would outgrow the staging buffer, the check after every FULL block of BufferNum reads
// (DBGgraph.cpp:337-351), a flush when the batch limit is reached -- but between two decisions the bookkeeping is one tight loop
static_assert(sizeof(ChunkedReadsFile::ReadRef) == sizeof(dbgk_read_ref) && offsetof(ChunkedReadsFile::ReadRef, len) == offsetof(dbgk_read_ref, len),
              "the reader's record reference is dbgk_read_ref");
void note_reads(Session &S, const ChunkedReadsFile::ReadRef *recs, size_t n)
{
	const uint64_t B = (uint64_t)std::max(BufferNum, 1), K = (uint64_t)KmerSize, R = (uint64_t)maxReadLen;
	const uint64_t limit = S.batch_limit;
	size_t i = 0;
	while (i < n && !S.stop_file) {
		const size_t base_idx = S.offsets.size();
		const size_t run = (size_t)std::min<uint64_t>(n - i, B - S.reads_in_block);
		const uint64_t start = S.offsets.back();
		uint64_t cur = start, bound = 0;
		bool flush_before = false, flush_after = false;
		S.offsets.resize(base_idx + run);
		uint64_t *off = S.offsets.data() + base_idx;
		size_t j = 0;
		for (; j < run; j++) {
			const uint64_t len = recs[i + j].len;
			if (S.zero_copy && cur + len > limit + (1u << 16) && base_idx + j > 1) { flush_before = true; break; } // keep the batch inside the staging buffer
			cur += len;
			off[j] = cur;
			const uint64_t rl = len < R ? len : R;
			bound += rl >= K ? rl - K + 1 : 0;
			if (cur >= limit) { j++; flush_after = true; break; }
		}
		S.offsets.resize(base_idx + j);
		const dbgk_read_ref *src = reinterpret_cast<const dbgk_read_ref *>(recs + i);
		S.noted.insert(S.noted.end(), src, src + j);
		S.noted_bytes += cur - start;
		S.bound_since += bound;
		S.pos += cur - start;
		S.reads_in_block += j;
		i += j;
		if (S.reads_in_block == B) end_of_full_block(S);
		else if (flush_after) flush_batch(S);
		if (flush_before) flush_batch(S); // (the read that did not fit opens the next batch)
	}
}

}  // namespace

void *thread_parseBlock(void *)
{
	cerr << "thread_parseBlock: this stage runs on the GPU in this build (k_extract_insert / k_extract_scatter)" << endl;
	abort();
}

void *thread_updatekmers(void *)
{
	cerr << "thread_updatekmers: this stage runs on the GPU in this build (k_extract_insert / k_build_regions)" << endl;
	abort();
}

// One reads file -> batches on the GPU (record rules: reads_io.h).
void parse_one_reads_file(string &reads_file)
{
	if (!g_session) return;
	Session &S = *g_session;
	S.stop_file = false;
	S.reads_in_block = 0;
	ChunkedReadsFile chunked;
	const bool sequential = dbgk_hook("parse_sequential") != nullptr;
	if (const char *pt = getenv("DBGK_PARSE_THREADS")) S.parse_threads = std::max(1, atoi(pt));
	if (!sequential && chunked.open(reads_file)) { // a plain file: windows read, lines found and bytes copied by several threads
		static const bool per_record = DBGK_EXPERIMENT_ENV("DBGK_PARSE_PER_RECORD") != nullptr; // (measurements: the record rules and the bookkeeping read by read on the calling thread)
		const bool ok = per_record ? chunked.for_each_read(Input_file_format, S.parse_threads, [&](const char *seq, size_t len) {
			if (chunked.too_long) { S.stop_file = true; return; } // (a 4 GiB line is no read)
			note_read(S, seq, len);
		}, [&]() { materialize_noted(S); }, &S.stop_file)
		                           : chunked.for_each_read_bulk(Input_file_format, S.parse_threads, [&](const ChunkedReadsFile::ReadRef *recs, size_t n) {
			if (chunked.too_long) { S.stop_file = true; return; } // (a 4 GiB line is no read)
			note_reads(S, recs, n);
		}, [&]() { materialize_noted(S); }, &S.stop_file);
		if (!ok) cerr << "fail to read reads file " << reads_file << endl;
		if (chunked.too_long) cerr << "Alert: a sequence line of 4 GiB or more in " << reads_file << ": the rest of this file is not read" << endl;
		if (getenv("DBGK_TIMINGS"))
			cerr << "Reader (s, calling thread): first window " << chunked.spent[0] << " records -> batches " << chunked.spent[1] << " copy/pack + hand-over "
			     << chunked.spent[2] << " waiting for the next window " << chunked.spent[3] << " (" << S.parse_threads << " threads); of all that, packing / copying the reads into the batches " << S.t_pack << endl;
		flush_batch(S);
	} else {
		if (!for_each_read_in_file(reads_file, Input_file_format, [&](const char *seq, size_t len) { add_read(S, seq, len); }, &S.stop_file)) {
			cerr << "fail to open reads file " << reads_file << endl;
			return;
		}
		flush_batch(S);
	}
	S.reads_in_block = 0; // a file's last, short block is never followed by an enlarge check (DBGgraph.cpp:329-331)
	if (!S.stop_file) cerr << "this block has reach the end of file " << endl;
}

// DBGK_LAYOUT=ref: rebuild the host table exactly as the reference's single-threaded path lays it
// out.  At -t 1 the reference inserts k-mers in (file, read, position) order (DBGgraph.cpp:139-205),
// a new key goes to the first slot with kmer == 0 on its probe chain, and after every FULL block of
// BufferNum reads the table is enlarged in place when count > max (:337-343, kmerSet.cpp:132-189).
// The slot layout therefore depends only on the ORDER in which distinct keys first appear and on
// where the block boundaries fall between them -- both known here: the device returns the nodes
// sorted by first-seen position, the parser recorded the position after every full block (and left a
// file where the reference would have, so no read beyond the -e cap is in the input).
static KmerSet *replay_reference_layout(Session &S, const std::vector<dbgk_node> &nodes, const std::vector<uint64_t> &first_pos,
                                        uint64_t initial_size, uint32_t polyA_l, uint32_t polyA_r)
{
	KmerSet *ks = init_kmerset_parallel(initial_size, hashLoadFactor, std::max(threadNum, 1));
	if (!ks) return NULL;
	size_t i = 0;
	uint64_t doublings = 0;
	auto insert_until = [&](uint64_t limit) {
		for (; i < nodes.size() && first_pos[i] < limit; i++) {
			uint64_t hc = hash_code(nodes[i].kmer) % ks->size;
			while (ks->array[hc].kmer != 0) {
				ks->count_conflict++;
				hc = (hc + 1 == ks->size) ? 0 : hc + 1;
			}
			ks->array[hc].kmer = nodes[i].kmer;
			ks->array[hc].l_link = nodes[i].l_link;
			ks->array[hc].r_link = nodes[i].r_link;
			set_entity_fill(ks->nul_flag, hc);
			ks->count++;
		}
	};
	for (uint64_t end : S.full_block_ends) {
		insert_until(end);
		if (ks->count > ks->max && doublings < maxDoubleHashTimes) {
			enlarge_kmerset_parallel(ks, 1, std::max(threadNum, 1));
			doublings++;
		}
	}
	insert_until(~0ull);
	KmerNode zero = {0, polyA_l, polyA_r};
	add_node_to_kmerset(ks, &zero);  // DBGgraph.cpp:418
	return ks;
}

static void release_session()
{
	if (!g_session) return;
	if (g_session->creator.joinable()) g_session->creator.join();
	(void)take_early_table(*g_session, 0); // (an early host table nobody took)
	if (g_session->h) dbgk_destroy(g_session->h);
	if (g_session->comm) dbgk_comm_destroy(g_session->comm);
	delete g_session;
	g_session = nullptr;
}

void dbgk_host_release_session() { release_session(); }

// plain (not gzip'ed) input: the sum of the file sizes bounds the number of k-mer windows; 0 = unknown
static uint64_t input_size_bound(const vector<string> &files)
{
	uint64_t total = 0;
	for (const string &f : files) {
		FILE *fp = fopen(f.c_str(), "rb");
		if (!fp) continue;
		unsigned char magic[2] = {0, 0};
		const size_t got = fread(magic, 1, 2, fp);
		const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
		fseek(fp, 0, SEEK_END);
		const long sz = ftell(fp);
		fclose(fp);
		if (gz || sz < 0) return 0;
		total += (uint64_t)sz;
	}
	return total;
}

// -k 33..63 (this build only: the reference stops at 31, DBG_contig/main.cpp:100; PARITY UNPINNED).  Same host duties -- parse,
// stream, materialise -- through the WIDE engine: 128-bit keys, 32-byte nodes, the table of the size -i asks for (the reference's
// doubling schedule is defined for its own table only).  With a known input size (plain files) the occurrences travel as 16-byte
// records; a table with more level-1 buckets than one pass can fan out to is built in several passes over the input files
// (dbgk_wide_begin_pass); DBGK_GPUS=N: N slot-range shards of the table in this process (dbgk_comm_*, one pass).
static void build_debruijn_graph_wide(vector<string> &reads_files, Session *S, uint64_t initial_size)
{
	S->wide = true;
	dbgk_config cfg;
	memset(&cfg, 0, sizeof cfg);
	cfg.kmer_size = KmerSize;
	cfg.max_read_len = maxReadLen;
	cfg.device_id = getenv("DBGK_DEVICE") ? atoi(getenv("DBGK_DEVICE")) : 0;
	cfg.engine = DBGK_ENGINE_WIDE;
	cfg.table_slots = initial_size;
	cfg.max_batch_bases = S->batch_limit + (1u << 16);
	const uint64_t bound = input_size_bound(reads_files);
	const bool records = bound > 0 && initial_size >= kPartitionMinSlots && initial_size <= kPartitionMaxSlots && !dbgk_hook("wide_direct");
	cfg.expected_kmers = records ? bound : 0; // unknown input size (compressed files): fused extract + atomic insert
	if (records) cfg.n_passes = 1; // this caller follows the pass protocol: as many passes over the input files as the geometry needs
	if (records && getenv("DBGK_WIDE_PASSES")) cfg.n_passes = (uint64_t)std::max(1, atoi(getenv("DBGK_WIDE_PASSES"))); // more passes than the geometry needs (small devices, tests)
	S->device_slots = initial_size;
	std::vector<int32_t> devices;
	if (const char *lst = getenv("DBGK_GPU_LIST")) {
		for (const char *p = lst; *p;) {
			devices.push_back((int32_t)strtol(p, const_cast<char **>(&p), 10));
			while (*p == ',' || *p == ' ') p++;
		}
	} else if (const char *ng = getenv("DBGK_GPUS")) {
		for (int i = 0; i < atoi(ng); i++) devices.push_back(i);
	}
	int rc;
	uint32_t n_passes = 1;
	if (devices.size() > 1 && records) {
		cfg.expected_kmers = std::max<uint64_t>(bound / devices.size() + (bound >> 4), 1024); // per handle (batches are dealt round robin)
		rc = dbgk_comm_create(&cfg, devices.data(), (uint32_t)devices.size(), &S->comm);
		if (rc != DBGK_OK) fail(*S, rc, "dbgk_comm_create");
		else cerr << "k-mer table of " << S->device_slots << " entries (32 bytes each) over " << devices.size() << " GPU shards" << endl;
	} else {
		rc = dbgk_create(&cfg, &S->h);
		if (rc != DBGK_OK) fail(*S, rc, "dbgk_create");
		else if (dbgk_wide_pass_info(S->h, &n_passes, NULL) != DBGK_OK) n_passes = 1;
		S->zero_copy = rc == DBGK_OK && !dbgk_hook("no_zero_copy");
	}
	S->t_create = double(clock() - time_start) / CLOCKS_PER_SEC;
	cerr << "Hash initialization array size:  " << initHashSize << " G" << endl;
	cerr << "The initialization memory used:  " << initHashSize * 32 << " G" << endl;
	time_end = clock();
	cerr << "Finished! Run time: " << double(time_end - time_start) / CLOCKS_PER_SEC << endl;

	cerr << "\nparse input reads files: " << endl;
	for (uint32_t pass = 0; pass < n_passes && S->status == DBGK_OK; pass++) {
		if (n_passes > 1) {
			cerr << "\nPass " << pass + 1 << " of " << n_passes << " over the input (the table is completed part by part)" << endl;
			rc = dbgk_wide_begin_pass(S->h, pass);
			if (rc != DBGK_OK) fail(*S, rc, "dbgk_wide_begin_pass");
			Total_reads_num = 0; // every pass reads all files again
			S->next_progress = 0;
		}
		for (size_t i = 0; i < reads_files.size(); i++) {
			cerr << "\nStart to parse reads file: " << reads_files[i] << endl;
			if (S->status == DBGK_OK) {
				const double t0 = now_s(), dev0 = S->t_push + S->t_count;
				parse_one_reads_file(reads_files[i]);
				S->t_parse += (now_s() - t0) - (S->t_push + S->t_count - dev0);
			}
			dbgk_stats st;
			if (S->status == DBGK_OK && pass == 0 && (S->comm ? dbgk_comm_refresh_stats(S->comm, &st) : dbgk_refresh_stats(S->h, &st)) == DBGK_OK)
				Kmer_total_num = st.total_kmers;
			cerr << "\nTotal number of reads loaded into memory: " << Total_reads_num << endl;
			cerr << "Total number of kmers loaded into memory: " << Kmer_total_num << endl;
			time_end = clock();
			cerr << "Finished! Run time: " << double(time_end - time_start) / CLOCKS_PER_SEC << endl;
		}
		if (n_passes > 1 && S->status == DBGK_OK) {
			rc = dbgk_wide_end_pass(S->h);
			if (rc != DBGK_OK) fail(*S, rc, "dbgk_wide_end_pass");
		}
	}
	dbgk_stats st;
	memset(&st, 0, sizeof st);
	if (S->status == DBGK_OK) {
		Stopwatch sw(S->t_finalize);
		rc = S->comm ? dbgk_comm_finalize(S->comm, &st) : dbgk_finalize(S->h, &st);
		if (rc != DBGK_OK) fail(*S, rc, "dbgk_finalize");
	}
	if (S->status == DBGK_OK && st.other_bytes)
		cerr << "\nAlert message: " << st.other_bytes << " sequence bytes are none of ACGTNacgtn; they were read as A (like N)" << endl;
	const double t_export0 = now_s();
	KmerSet128 *result = NULL;
	if (S->status == DBGK_OK) {
		Kmer_total_num = st.total_kmers;
		KmerNode32 *array = static_cast<KmerNode32 *>(kmerset_alloc(initial_size * sizeof(KmerNode32), false));
		uint8_t *nul = static_cast<uint8_t *>(kmerset_alloc(initial_size / 8 + 1, false)), *del = static_cast<uint8_t *>(kmerset_alloc(initial_size / 8 + 1, true));
		if (!array || !nul || !del) {
			free(array), free(nul), free(del);
			fail(*S, DBGK_ERR_NOMEM, "host table allocation");
		} else {
			rc = S->comm ? dbgk_comm_wide_export_host_table(S->comm, initial_size, reinterpret_cast<dbgk_node32 *>(array), nul)
			             : dbgk_wide_export_host_table(S->h, initial_size, reinterpret_cast<dbgk_node32 *>(array), nul);
			if (rc != DBGK_OK) {
				free(array), free(nul), free(del);
				fail(*S, rc, "dbgk_wide_export_host_table");
			} else {
				result = adopt_kmerset128(initial_size, hashLoadFactor, st.count, st.count_conflict, array, nul, del);
			}
		}
	}
	if (kset_wide) free_hash128(kset_wide);
	kset_wide = result;
	// the 64-bit container stays valid and empty (only the key-0 node): code that looks at `kset` keeps working
	if (kset) free_hash(kset);
	kset = init_kmerset_parallel(3, hashLoadFactor, std::max(threadNum, 1));
	KmerNode zero = {0, 0, 0};
	add_node_to_kmerset(kset, &zero);
	DbgkLastStatus = S->status;
	S->t_export = now_s() - t_export0;
	if (getenv("DBGK_TIMINGS"))
		cerr << "Host phases (s): create " << S->t_create << " read+parse " << S->t_parse << " push " << S->t_push << " count/flush " << S->t_count
		     << " finalize " << S->t_finalize << " host table " << S->t_export << endl;
	cerr << "\nKmerset hash parameters (128-bit keys, 32-byte nodes):" << endl;
	if (kset_wide)
		cerr << "array_size: " << kset_wide->size << "\nload_factor: " << kset_wide->load_factor << "\nmax_cutoff: " << kset_wide->max
		     << "\ncount: " << kset_wide->count << "\ncount_conflict: " << kset_wide->count_conflict << endl;
}

void build_debruijn_graph(vector<string> &reads_files)
{
	time_start = clock();
	release_session();
	DbgkLastStatus = DBGK_OK;

	KmerHeadMaskVal = pow_integer(2, KmerSize * 2) - 1;              // DBGgraph.cpp:371-376
	for (int b = 0; b < 4; b++) KmerRCOrVal[b] = (uint64_t)(3 - b) << (2 * KmerSize - 2);
	KmerNumInRead = maxReadLen - KmerSize + 1;
	Kmer_total_num = Total_reads_num = 0;
	doubleHashTimes = 0;

	cerr << "Start to initialize the kmerset hash" << endl;
	const uint64_t wanted = (uint64_t)(initHashSize * 1000000000);
	const uint64_t initial_size = wanted < 3 ? 3 : find_next_prime(wanted);

	Session *S = new Session();
	g_session = S;
	S->offsets.assign(1, 0);
	S->ref_size = initial_size;
	S->ref_max = (uint64_t)((float)initial_size * clamped_load_factor());   // kmerSet.cpp:114
	if (const char *bb = getenv("DBGK_BATCH_BYTES")) S->batch_limit = std::max<uint64_t>(1024, strtoull(bb, NULL, 10)); // tests: many small batches
	S->bases.reserve(S->batch_limit + (1u << 16));
	S->packed = !getenv("DBGK_HOST_ASCII");
	S->parse_threads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
	if (KmerSize > 32) return build_debruijn_graph_wide(reads_files, S, initial_size);
	if (kset_wide) {
		free_hash128(kset_wide);
		kset_wide = NULL;
	}

	S->ref_layout = getenv("DBGK_LAYOUT") && string(getenv("DBGK_LAYOUT")) == "ref";
	dbgk_config cfg;
	memset(&cfg, 0, sizeof cfg);
	cfg.kmer_size = KmerSize;
	cfg.max_read_len = maxReadLen;
	cfg.device_id = getenv("DBGK_DEVICE") ? atoi(getenv("DBGK_DEVICE")) : 0;
	// Engine: PARTITION (records radix-partitioned by slot range, table regions built in LDS; input of unknown
	// size streams through its record store) unless DBGK_ENGINE=1 asks for DIRECT (global atomics) or the
	// reference's slot layout is wanted (needs DIRECT's first-seen tracking).  The device table is independent
	// of the host table the consumer gets: at least 2^26 slots for PARTITION, grown as needed.
	const int want_engine = getenv("DBGK_ENGINE") ? atoi(getenv("DBGK_ENGINE")) : DBGK_ENGINE_AUTO;
	S->partition = !S->ref_layout && want_engine != DBGK_ENGINE_DIRECT && initial_size <= kPartitionMaxSlots;
	S->device_slots = S->partition ? std::max(initial_size, kPartitionMinSlots) : initial_size;
	cfg.table_slots = S->device_slots;
	cfg.engine = S->partition ? DBGK_ENGINE_PARTITION : DBGK_ENGINE_DIRECT;
	if (S->partition) {
		// record store: what the input can hold at most (plain files: one window per byte), at most 2^31
		// occurrences (~38 GB of stores); more input is flushed into the table in rounds
		uint64_t store = getenv("DBGK_STORE_KMERS") ? strtoull(getenv("DBGK_STORE_KMERS"), NULL, 10) : 0;
		if (!store) {
			const uint64_t bound = input_size_bound(reads_files);
			store = bound ? std::min<uint64_t>(bound, 1ull << 31) : (1ull << 31);
		}
		cfg.expected_kmers = std::max<uint64_t>(store, 1024);
	}
	if (S->ref_layout) cfg.flags |= DBGK_FLAG_TRACK_FIRST_SEEN;
	cfg.max_batch_bases = S->batch_limit + (1u << 16);
	// several GPUs: DBGK_GPUS=N (devices 0..N-1) or DBGK_GPU_LIST=a,b,c (ordinals, repeats allowed)
	std::vector<int32_t> devices;
	if (const char *lst = getenv("DBGK_GPU_LIST")) {
		for (const char *p = lst; *p;) {
			devices.push_back((int32_t)strtol(p, const_cast<char **>(&p), 10));
			while (*p == ',' || *p == ' ') p++;
		}
	} else if (const char *ng = getenv("DBGK_GPUS")) {
		for (int i = 0; i < atoi(ng); i++) devices.push_back(i);
	}
	int rc;
	if (devices.size() > 1 && S->partition) {
		// ONE table over all GPUs, sized once: every k-mer window could be a new node (plain files: their size
		// bounds the windows; compressed input: 2^32 entries unless -i asks for more)
		const uint64_t bound = input_size_bound(reads_files);
		const uint64_t want_slots = bound ? (uint64_t)((double)bound / 0.75) : (1ull << 32);
		S->device_slots = find_next_prime(std::min(std::max(S->device_slots, want_slots), kPartitionMaxSlots));
		cfg.table_slots = S->device_slots;
		cfg.expected_kmers = std::max<uint64_t>(cfg.expected_kmers / devices.size(), 1024); // per handle
		rc = dbgk_comm_create(&cfg, devices.data(), (uint32_t)devices.size(), &S->comm);
		if (rc != DBGK_OK) fail(*S, rc, "dbgk_comm_create");
		else cerr << "k-mer table of " << S->device_slots << " entries over " << devices.size() << " GPU shards" << endl;
	} else {
		// the handle (device table, record stores, streams: ~0.1 s) is made on a thread of its own while the first window of the
		// first file is being read; whoever needs it first waits for it (ensure_created)
		rc = DBGK_OK;
		S->zero_copy = !dbgk_hook("no_zero_copy"); // batches are parsed straight into the pinned staging buffers
		const double t0c = now_s();
		// (only here: page-locking the staging buffers happens on the creating thread, beside the first file read.  A communicator
		// creates its handles on the calling thread and a fresh set at every resize: there the buffers are made when first used)
		cfg.flags |= DBGK_FLAG_PREALLOC_STAGING;
		S->creator = std::thread([S, cfg, t0c]() {
			S->create_rc = dbgk_create(&cfg, &S->h);
			if (S->create_rc != DBGK_OK) S->create_err = dbgk_last_error();
			S->t_create = now_s() - t0c;
		});
		if (DBGK_EXPERIMENT_ENV("DBGK_CREATE_SYNC")) ensure_created(*S);
	}

	if (!S->ref_layout) start_early_table(*S, initial_size);
	if (S->comm) S->t_create = double(clock() - time_start) / CLOCKS_PER_SEC;
	cerr << "Hash initialization array size:  " << initHashSize << " G" << endl;
	cerr << "The initialization memory used:  " << initHashSize * 16 << " G" << endl;
	time_end = clock();
	cerr << "Finished! Run time: " << double(time_end - time_start) / CLOCKS_PER_SEC << endl;

	cerr << "\nparse input reads files: " << endl;
	for (size_t i = 0; i < reads_files.size(); i++) {
		cerr << "\nStart to parse reads file: " << reads_files[i] << endl;
		if (S->status == DBGK_OK) {
			const double t0 = now_s(), dev0 = S->t_push + S->t_count;
			parse_one_reads_file(reads_files[i]);
			S->t_parse += (now_s() - t0) - (S->t_push + S->t_count - dev0);
		}
		dbgk_stats st;
		ensure_created(*S);
		if (S->status == DBGK_OK && (S->comm ? dbgk_comm_refresh_stats(S->comm, &st) : dbgk_refresh_stats(S->h, &st)) == DBGK_OK)
			Kmer_total_num = st.total_kmers;
		cerr << "\nTotal number of reads loaded into memory: " << Total_reads_num << endl;
		cerr << "Total number of kmers loaded into memory: " << Kmer_total_num << endl;
		time_end = clock();
		cerr << "Finished! Run time: " << double(time_end - time_start) / CLOCKS_PER_SEC << endl;
	}

	// hand the graph over as a host KmerSet
	dbgk_stats st;
	memset(&st, 0, sizeof st);
	ensure_created(*S);
	if (S->status == DBGK_OK) {
		Stopwatch sw(S->t_finalize);
		rc = S->comm ? dbgk_comm_finalize(S->comm, &st) : dbgk_finalize(S->h, &st);
		if (rc != DBGK_OK) fail(*S, rc, "dbgk_finalize");
	}
	if (S->status == DBGK_OK && st.other_bytes)
		cerr << "\nAlert message: " << st.other_bytes << " sequence bytes are none of ACGTNacgtn; they were read as A (like N)" << endl;
	const double t_export0 = now_s();
	KmerSet *result = NULL;
	if (S->status == DBGK_OK && S->ref_layout) {
		Kmer_total_num = st.total_kmers;
		std::vector<dbgk_node> nodes(st.count ? st.count : 1);
		std::vector<uint64_t> first_pos(st.count ? st.count : 1);
		uint64_t n = 0;
		rc = dbgk_export_first_seen_order(S->h, nodes.data(), first_pos.data(), nodes.size(), &n);
		if (rc != DBGK_OK) {
			fail(*S, rc, "dbgk_export_first_seen_order");
		} else {
			nodes.resize(n);
			first_pos.resize(n);
			result = replay_reference_layout(*S, nodes, first_pos, wanted, st.polyA_l_link, st.polyA_r_link);
			if (!result) fail(*S, DBGK_ERR_NOMEM, "host table allocation");
		}
	} else if (S->status == DBGK_OK) {
		Kmer_total_num = st.total_kmers;
		// the host table has the size the reference's own schedule arrived at (tracked block by block above)
		const uint64_t use_size = S->ref_size;
		KmerNode *array = NULL;
		uint8_t *nul = NULL, *del = NULL;
		if (st.count > use_size) {
			// the reference would now spin forever looking for a free slot (DBGgraph.cpp:170-205)
			cerr << "\nAlert message: " << st.count << " kmer nodes do not fit the hash array of " << use_size
			     << " entries the -i/-e settings allow" << endl;
			fail(*S, DBGK_ERR_TABLE_FULL, "host table");
		} else {
			array = take_early_table(*S, use_size);
			if (!array) array = static_cast<KmerNode *>(kmerset_alloc(use_size * sizeof(KmerNode), false));
			nul = static_cast<uint8_t *>(kmerset_alloc(use_size / 8 + 1, false));
			del = static_cast<uint8_t *>(kmerset_alloc(use_size / 8 + 1, true));
			if (!array || !nul || !del) {
				free(array), free(nul), free(del);
				fail(*S, DBGK_ERR_NOMEM, "host table allocation");
			} else {
				const bool links = getenv("DBGK_LINKS") && atoi(getenv("DBGK_LINKS")) != 0;
				if (links) { // the consumer's first pass on the device, for exactly this table
					free(DbgkKmerLinks);
					DbgkKmerLinks = static_cast<uint16_t *>(malloc(use_size * sizeof(uint16_t)));
					DbgkTipNodes.assign(st.count ? st.count : 1, 0);
					DbgkBranchNodes.assign(st.count ? st.count : 1, 0);
					uint64_t nt = 0, nb = 0;
					const int cutoff = &KmerFreqCutoff ? KmerFreqCutoff : 2;
					if (!DbgkKmerLinks) rc = DBGK_ERR_NOMEM;
					else if (S->comm) // several GPU shards: assembled in one table of use_size slots on the first GPU, then the same pass
						rc = dbgk_comm_export_host_table_links(S->comm, use_size, reinterpret_cast<dbgk_node *>(array), nul, cutoff, DbgkKmerLinks, del,
						                                       DbgkTipNodes.data(), DbgkTipNodes.size(), &nt, DbgkBranchNodes.data(), DbgkBranchNodes.size(), &nb, NULL);
					else
						rc = dbgk_export_host_table_links(S->h, use_size, reinterpret_cast<dbgk_node *>(array), nul, cutoff, DbgkKmerLinks, del,
						                                  DbgkTipNodes.data(), DbgkTipNodes.size(), &nt, DbgkBranchNodes.data(), DbgkBranchNodes.size(), &nb, NULL);
					DbgkTipNodes.resize(rc == DBGK_OK ? nt : 0);
					DbgkBranchNodes.resize(rc == DBGK_OK ? nb : 0);
					if (rc == DBGK_OK) cerr << "First pass of the contig stage done on the GPU: " << nt << " tip nodes, " << nb << " branching nodes" << endl;
				} else {
					rc = S->comm ? dbgk_comm_export_host_table(S->comm, use_size, reinterpret_cast<dbgk_node *>(array), nul)
					             : dbgk_export_host_table(S->h, use_size, reinterpret_cast<dbgk_node *>(array), nul);
				}
				if (rc != DBGK_OK) {
					free(array), free(nul), free(del);
					fail(*S, rc, "dbgk_export_host_table");
				} else {
					result = adopt_kmerset(use_size, hashLoadFactor, st.count, st.count_conflict, array, nul, del);
				}
			}
		}
	}
	if (!result) {  // keep the consumer alive: an empty but valid set holding only the key-0 node
		result = init_kmerset_parallel(initial_size, hashLoadFactor, std::max(threadNum, 1));
		KmerNode zero = {0, 0, 0};
		add_node_to_kmerset(result, &zero);
	}
	if (kset) free_hash(kset);
	kset = result;
	DbgkLastStatus = S->status;
	S->t_export = now_s() - t_export0;
	if (getenv("DBGK_TIMINGS"))
		cerr << "Host phases (s): create " << S->t_create << " read+parse " << S->t_parse << " push " << S->t_push << " count/flush " << S->t_count
		     << " finalize " << S->t_finalize << " host table " << S->t_export << " (its pages touched beside the parse in " << S->t_early << ")" << endl;
	if (getenv("DBGK_TIMINGS") && S->h) { // device time per phase, summed over the run (HIP events on the library's streams)
		dbgk_timings tm;
		if (dbgk_get_timings(S->h, &tm) == DBGK_OK)
			cerr << "GPU phases (ms): engine " << (S->partition ? "partition" : "direct") << " mark " << tm.mark_ms << " level1/insert " << tm.insert_ms
			     << " level2 " << tm.partition_ms << " build " << tm.build_ms << " level2+build wall " << tm.l2_build_wall_ms << " fixup "
			     << tm.fixup_ms << " launches " << tm.insert_launches << endl;
	}

	print_kmerset_parameter(kset);
}

int write_kmer_freq_file(const string &path, int kmer_freq_cutoff)
{
	// `<prefix>.contig.kmer.freq` of the consumer's first pass (contig.cpp:186-203): header, then
	// rows 1..255 of DepthStat (row 0 is not written).  Computed on the device table.
	if (!g_session || (!g_session->h && !g_session->comm) || g_session->status != DBGK_OK) return DBGK_ERR_STATE;
	dbgk_link_stats ls;
	int rc = g_session->comm ? dbgk_comm_link_stats(g_session->comm, kmer_freq_cutoff, &ls)
	                         : dbgk_link_stats_device(g_session->h, kmer_freq_cutoff, &ls);
	if (rc != DBGK_OK) return rc;
	ofstream out(path.c_str());
	if (!out) {
		cerr << "fail to open file " << path << endl;
		return DBGK_ERR_ARG;
	}
	cerr << "\nTotal kmer nodes number:    " << ls.total_nodes << endl;
	cerr << "Deleted lowfreq kmer nodes: " << ls.deleted_lowfreq << "\t" << (double)ls.deleted_lowfreq / ls.total_nodes << endl;
	cerr << "Used linear kmer nodes:     " << ls.linear_nodes << "\t" << (double)ls.linear_nodes / ls.total_nodes << endl;
	cerr << "Used tip kmer nodes:        " << ls.tip_nodes << "\t" << (double)ls.tip_nodes / ls.total_nodes << endl;
	cerr << "Used branching kmer nodes:  " << ls.branch_nodes << "\t" << (double)ls.branch_nodes / ls.total_nodes << endl;
	out << "Kmer_depth\tAppear_times\n";
	for (int i = 1; i <= 255; i++) out << i << "\t" << ls.depth_stat[i] << endl;
	return DBGK_OK;
}

int write_links_dump(const string &path)
{
	// what DBGK_LINKS=1 left behind: `K slot record(hex) deleted` for every occupied slot, then `T slot` / `B slot` lines
	if (!kset || !DbgkKmerLinks) return DBGK_ERR_STATE;
	FILE *fp = fopen(path.c_str(), "w");
	if (!fp) return DBGK_ERR_ARG;
	fprintf(fp, "#size %llu tips %llu branches %llu\n", (unsigned long long)kset->size, (unsigned long long)DbgkTipNodes.size(),
	        (unsigned long long)DbgkBranchNodes.size());
	for (uint64_t i = 0; i < kset->size; i++)
		if (!is_entity_null(kset->nul_flag, i)) fprintf(fp, "K\t%llu\t%04x\t%d\n", (unsigned long long)i, DbgkKmerLinks[i], is_entity_delete(kset->del_flag, i));
	for (uint64_t v : DbgkTipNodes) fprintf(fp, "T\t%llu\n", (unsigned long long)v);
	for (uint64_t v : DbgkBranchNodes) fprintf(fp, "B\t%llu\n", (unsigned long long)v);
	fclose(fp);
	return DBGK_OK;
}

int write_table_image(const string &path)
{
	// raw image of the host KmerSet: size, count, the node array, the nul_flag bytes (layout tests)
	if (!kset) return DBGK_ERR_STATE;
	FILE *fp = fopen(path.c_str(), "wb");
	if (!fp) return DBGK_ERR_ARG;
	const uint64_t hdr[2] = {kset->size, kset->count};
	fwrite(hdr, 8, 2, fp);
	fwrite(kset->array, sizeof(KmerNode), kset->size, fp);
	fwrite(kset->nul_flag, 1, kset->size / 8 + 1, fp);
	fclose(fp);
	return DBGK_OK;
}

int write_sorted_dump(const string &path)
{
	if (KmerSize > 32) { // 128-bit keys: kmer_hi, kmer_lo, l_link, r_link of every node, sorted by (hi, lo)
		if (!kset_wide) return DBGK_ERR_STATE;
		std::vector<KmerNode32> wn;
		wn.reserve(kset_wide->count);
		for (uint64_t i = 0; i < kset_wide->size; i++)
			if (!is_entity_null(kset_wide->nul_flag, i)) wn.push_back(kset_wide->array[i]);
		std::sort(wn.begin(), wn.end(), [](const KmerNode32 &a, const KmerNode32 &b) { return a.kmer_hi < b.kmer_hi || (a.kmer_hi == b.kmer_hi && a.kmer_lo < b.kmer_lo); });
		FILE *wf = fopen(path.c_str(), "w");
		if (!wf) return DBGK_ERR_ARG;
		fprintf(wf, "#reads %llu kmers %llu count %llu\n", (unsigned long long)Total_reads_num, (unsigned long long)Kmer_total_num,
		        (unsigned long long)kset_wide->count);
		for (const KmerNode32 &n : wn)
			fprintf(wf, "%llu\t%llu\t%08x\t%08x\n", (unsigned long long)n.kmer_hi, (unsigned long long)n.kmer_lo, n.l_link, n.r_link);
		fclose(wf);
		return DBGK_OK;
	}
	if (!kset) return DBGK_ERR_STATE;
	std::vector<KmerNode> nodes;
	nodes.reserve(kset->count);
	for (uint64_t i = 0; i < kset->size; i++)
		if (!is_entity_null(kset->nul_flag, i)) nodes.push_back(kset->array[i]);
	std::sort(nodes.begin(), nodes.end(), [](const KmerNode &a, const KmerNode &b) { return a.kmer < b.kmer; });
	FILE *fp = fopen(path.c_str(), "w");
	if (!fp) return DBGK_ERR_ARG;
	fprintf(fp, "#reads %llu kmers %llu count %llu\n", (unsigned long long)Total_reads_num,
	        (unsigned long long)Kmer_total_num, (unsigned long long)kset->count);
	for (const KmerNode &n : nodes) fprintf(fp, "%llu\t%08x\t%08x\n", (unsigned long long)n.kmer, n.l_link, n.r_link);
	fclose(fp);
	return DBGK_OK;
}
