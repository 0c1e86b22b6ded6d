// DBGgraph.cpp -- build_debruijn_graph() on an MI355X.
//
// Host responsibilities only: parse the (optionally gzip'ed) one-line FASTA/FASTQ files into
// batches of raw sequence bytes, stream them to the GPU through the C ABI of include/dbgk.h, and
// materialise the finished table as the host KmerSet the reference's contig stage expects.  All
// k-mer work (2-bit packing, canonicalisation, hashing, counting) happens in the HIP kernels.
//
// Reference behaviour followed here (paths relative to /root/reference/DBG_contig/):
//   record detection            DBGgraph.cpp:244-272   (first character of a line; next line = sequence)
//   totals / stderr protocol    DBGgraph.cpp:380-428
//   table sizing                kmerSet.cpp:98-127 (initial), DBGgraph.cpp:337-351 + kmerSet.cpp:132-148 (doubling)
//   key-0 node appended last    DBGgraph.cpp:418
#include "DBGgraph.h"

#include <zlib.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "dbgk.h"
#include "reads_io.h"

// ---- the reference's globals (DBGgraph.cpp:10-34), same names and defaults -------------------
int KmerSize = 31;
int maxReadLen = 250;
int KmerNumInRead = 0;
int Input_file_format = 1;
string Output_prefix = "output";
int threadNum = 10;
KmerSet *kset = NULL;
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

namespace {

struct Session {
	dbgk_handle *h = nullptr;
	std::vector<char> bases;          // sequences of the pending batch, back to back
	std::vector<uint64_t> offsets;    // offsets.size() == reads in batch + 1
	uint64_t batch_limit = 128ull << 20;
	uint64_t device_slots = 0;
	uint64_t next_progress = 0;
	int status = DBGK_OK;
	// DBGK_LAYOUT=ref: reproduce the reference's -t 1 slot layout (first-seen order replay)
	bool ref_layout = false;
	uint64_t pos = 0;                      // bases handed to the device so far (+ pending batch)
	uint64_t reads_in_block = 0;           // reads since the last full block of the current file
	std::vector<uint64_t> full_block_ends; // position right after every FULL block of BufferNum reads
};

Session *g_session = nullptr;

void fail(Session &S, int rc, const char *what)
{
	if (S.status == DBGK_OK) {
		S.status = rc;
		cerr << "\nAlert message: " << what << " failed: " << dbgk_strerror(rc);
		if (rc == DBGK_ERR_HIP) cerr << " [" << dbgk_last_error() << "]";
		cerr << "; the remaining input is ignored" << endl;
	}
}

// make sure the device table can absorb `incoming` more distinct keys, growing it if necessary
// (the device-side counterpart of the reference's enlarge step; the size the HOST table finally
// gets is decided separately in final_host_size()).
void reserve_device_slots(Session &S, uint64_t incoming)
{
	dbgk_stats st;
	int rc = dbgk_refresh_stats(S.h, &st);
	if (rc != DBGK_OK) return fail(S, rc, "dbgk_refresh_stats");
	const double need = (double)st.count + (double)incoming;
	if (need <= 0.80 * (double)S.device_slots) return;
	const uint64_t target = find_next_prime((uint64_t)(need / 0.55) + 16);
	rc = dbgk_resize_table(S.h, target);
	if (rc != DBGK_OK) return fail(S, rc, "dbgk_resize_table");
	S.device_slots = target;
	cerr << "Enlarge device hash array size to be: " << target << endl;
}

void flush_batch(Session &S)
{
	const uint64_t n_reads = S.offsets.size() - 1;
	if (n_reads == 0 || S.status != DBGK_OK) {
		S.bases.clear();
		S.offsets.assign(1, 0);
		return;
	}
	reserve_device_slots(S, S.bases.size());  // every base starts at most one new k-mer
	if (S.status == DBGK_OK) {
		int rc = dbgk_push_reads(S.h, S.bases.data(), S.offsets.data(), n_reads);
		if (rc != DBGK_OK) fail(S, rc, "dbgk_push_reads");
	}
	Total_reads_num += n_reads;
	if (Total_reads_num >= S.next_progress) {
		cerr << "Load reads block " << Total_reads_num << endl;
		S.next_progress = Total_reads_num + (uint64_t)std::max(BufferNum, 1) * 100;
	}
	S.bases.clear();
	S.offsets.assign(1, 0);
}

inline void add_read(Session &S, const char *seq, size_t len)
{
	S.bases.insert(S.bases.end(), seq, seq + len);
	S.offsets.push_back(S.bases.size());
	if (S.ref_layout) {
		S.pos += len;
		if (++S.reads_in_block == (uint64_t)std::max(BufferNum, 1)) { // the reference checks count > max here (DBGgraph.cpp:337)
			S.full_block_ends.push_back(S.pos);
			S.reads_in_block = 0;
		}
	}
	if (S.bases.size() >= S.batch_limit) flush_batch(S);
}

// the host table size the reference would end with for `keys` distinct non-zero keys: start from
// the initial "prime" and double (find_next_prime(2*size)) while count > max, at most
// maxDoubleHashTimes times (DBGgraph.cpp:337-351, kmerSet.cpp:140-145).  The reference tests this
// after every full block of BufferNum reads; evaluating it once on the final count gives the
// same chain element except when only the last, short block pushes count over max.
uint64_t final_host_size(uint64_t initial, uint64_t keys, float lf, uint64_t &doublings, bool &capped)
{
	if (lf <= 0) lf = 0.25f; else if (lf >= 1) lf = 0.75f;
	uint64_t size = initial;
	doublings = 0;
	capped = false;
	while (keys > (uint64_t)((float)size * lf)) {
		if (doublings >= maxDoubleHashTimes) { capped = true; break; }
		size = find_next_prime(size * 2);
		doublings++;
	}
	return size;
}

}  // namespace

void *thread_parseBlock(void *)
{
	cerr << "thread_parseBlock: this stage runs on the GPU in this build (k_extract_insert)" << endl;
	abort();
}

void *thread_updatekmers(void *)
{
	cerr << "thread_updatekmers: this stage runs on the GPU in this build (k_extract_insert)" << endl;
	abort();
}

// One reads file -> batches on the GPU (record rules: reads_io.h).
void parse_one_reads_file(string &reads_file)
{
	if (!g_session) return;
	Session &S = *g_session;
	if (!for_each_read_in_file(reads_file, Input_file_format, [&](const char *seq, size_t len) { add_read(S, seq, len); })) {
		cerr << "fail to open reads file " << reads_file << endl;
		return;
	}
	flush_batch(S);
	S.reads_in_block = 0; // a file's last, short block is never followed by an enlarge check (DBGgraph.cpp:329-331)
	cerr << "this block has reach the end of file " << endl;
}

// DBGK_LAYOUT=ref: rebuild the host table exactly as the reference's single-threaded path lays it
// out.  At -t 1 the reference inserts k-mers in (file, read, position) order (DBGgraph.cpp:139-205),
// a new key goes to the first slot with kmer == 0 on its probe chain, and after every FULL block of
// BufferNum reads the table is enlarged in place when count > max (:337-343, kmerSet.cpp:132-189).
// The slot layout therefore depends only on the ORDER in which distinct keys first appear and on
// where the block boundaries fall between them -- both known here: the device returns the nodes
// sorted by first-seen position, the parser recorded the position after every full block.
static KmerSet *replay_reference_layout(Session &S, const std::vector<dbgk_node> &nodes, const std::vector<uint64_t> &first_pos,
                                        uint64_t initial_size, uint32_t polyA_l, uint32_t polyA_r)
{
	KmerSet *ks = init_kmerset_parallel(initial_size, hashLoadFactor, std::max(threadNum, 1));
	if (!ks) return NULL;
	size_t i = 0;
	bool alerted = false;
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
		if (ks->count > ks->max) {
			if (doubleHashTimes >= maxDoubleHashTimes && !alerted) {
				// the reference drops the rest of the file here (DBGgraph.cpp:346-350); every read is kept
				cerr << "\nAlert message: Memory reach the maximum allowed by -e " << maxDoubleHashTimes << "; all reads were kept" << endl;
				alerted = true;
			}
			enlarge_kmerset_parallel(ks, 1, std::max(threadNum, 1));
			doubleHashTimes++;
			cerr << "Enlarge hash array size to be: " << ks->size << endl;
			cerr << "The expanded memory used now:  " << (double)ks->size / 1000000000 * 16 << " G" << endl;
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
	if (g_session->h) dbgk_destroy(g_session->h);
	delete g_session;
	g_session = nullptr;
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
	S->device_slots = initial_size;
	if (const char *mb = getenv("DBGK_BATCH_MB")) S->batch_limit = std::max<uint64_t>(1, strtoull(mb, NULL, 10)) << 20;
	S->bases.reserve(S->batch_limit + (1u << 16));

	S->ref_layout = getenv("DBGK_LAYOUT") && string(getenv("DBGK_LAYOUT")) == "ref";
	dbgk_config cfg;
	memset(&cfg, 0, sizeof cfg);
	cfg.kmer_size = KmerSize;
	cfg.max_read_len = maxReadLen;
	cfg.table_slots = initial_size;
	cfg.device_id = getenv("DBGK_DEVICE") ? atoi(getenv("DBGK_DEVICE")) : 0;
	cfg.engine = getenv("DBGK_ENGINE") ? atoi(getenv("DBGK_ENGINE")) : DBGK_ENGINE_AUTO;
	if (S->ref_layout) {
		cfg.engine = DBGK_ENGINE_DIRECT;
		cfg.flags |= DBGK_FLAG_TRACK_FIRST_SEEN;
	}
	cfg.max_batch_bases = S->batch_limit + (1u << 16);
	int rc = dbgk_create(&cfg, &S->h);
	if (rc != DBGK_OK) fail(*S, rc, "dbgk_create");

	cerr << "Hash initialization array size:  " << initHashSize << " G" << endl;
	cerr << "The initialization memory used:  " << initHashSize * 16 << " G" << endl;
	time_end = clock();
	cerr << "Finished! Run time: " << double(time_end - time_start) / CLOCKS_PER_SEC << endl;

	cerr << "\nparse input reads files: " << endl;
	for (size_t i = 0; i < reads_files.size(); i++) {
		cerr << "\nStart to parse reads file: " << reads_files[i] << endl;
		if (S->status == DBGK_OK) parse_one_reads_file(reads_files[i]);
		dbgk_stats st;
		if (S->status == DBGK_OK && dbgk_refresh_stats(S->h, &st) == DBGK_OK) Kmer_total_num = st.total_kmers;
		cerr << "\nTotal number of reads loaded into memory: " << Total_reads_num << endl;
		cerr << "Total number of kmers loaded into memory: " << Kmer_total_num << endl;
		time_end = clock();
		cerr << "Finished! Run time: " << double(time_end - time_start) / CLOCKS_PER_SEC << endl;
	}

	// hand the graph over as a host KmerSet
	dbgk_stats st;
	memset(&st, 0, sizeof st);
	if (S->status == DBGK_OK) {
		rc = dbgk_finalize(S->h, &st);
		if (rc != DBGK_OK) fail(*S, rc, "dbgk_finalize");
	}
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
		bool capped = false;
		const uint64_t host_size = final_host_size(initial_size, st.count - 1, hashLoadFactor, doubleHashTimes, capped);
		if (doubleHashTimes) {
			cerr << "Enlarge hash array size to be: " << host_size << endl;
			cerr << "The expanded memory used now:  " << (double)host_size / 1000000000 * 16 << " G" << endl;
		}
		uint64_t use_size = host_size;
		if (capped) {
			// The reference stops reading here and drops the rest of the file (DBGgraph.cpp:346-350).
			// All reads are already in the graph on the device, so keep them and size the host table
			// to fit instead of silently losing data.
			cerr << "\nAlert message: Memory reach the maximum allowed by -e " << maxDoubleHashTimes
			     << "; all " << Total_reads_num << " reads were kept, the host table is sized to hold them" << endl;
			while (st.count > (uint64_t)((float)use_size * 0.95f)) use_size = find_next_prime(use_size * 2);
		}
		KmerNode *array = static_cast<KmerNode *>(malloc(use_size * sizeof(KmerNode)));
		uint8_t *nul = static_cast<uint8_t *>(malloc(use_size / 8 + 1));
		uint8_t *del = static_cast<uint8_t *>(calloc(use_size / 8 + 1, 1));
		if (!array || !nul || !del) {
			free(array), free(nul), free(del);
			fail(*S, DBGK_ERR_NOMEM, "host table allocation");
		} else {
			rc = dbgk_export_host_table(S->h, use_size, reinterpret_cast<dbgk_node *>(array), nul);
			if (rc != DBGK_OK) {
				free(array), free(nul), free(del);
				fail(*S, rc, "dbgk_export_host_table");
			} else {
				result = adopt_kmerset(use_size, hashLoadFactor, st.count, st.count_conflict, array, nul, del);
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

	print_kmerset_parameter(kset);
}

int write_kmer_freq_file(const string &path, int kmer_freq_cutoff)
{
	// `<prefix>.contig.kmer.freq` of the consumer's first pass (contig.cpp:186-203): header, then
	// rows 1..255 of DepthStat (row 0 is not written).  Computed on the device table.
	if (!g_session || !g_session->h || g_session->status != DBGK_OK) return DBGK_ERR_STATE;
	dbgk_link_stats ls;
	int rc = dbgk_link_stats_device(g_session->h, kmer_freq_cutoff, &ls);
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
