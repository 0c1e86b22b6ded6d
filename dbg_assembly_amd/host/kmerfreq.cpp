// kmerfreq -- writer of the k-mer frequency table that DBG_assembly's correct_error module loads.
//
// The reference calls an external program for this (`../../kmerfreq/kmerfreq -k 17 -m 1 -q 10
// clean_reads.lib`, test/01.clean_correct/work.sh:18) which is not part of its repository; only the
// CONSUMER side is defined there, and that is what this tool writes for:
//   1-bit format  (-b 1, default)  read by correct_error/main_parallel_senior.cpp:334-408
//       4^k bits, bit 128 >> (v % 8) of byte v / 8 (bitAll, correct_error/seqKmer.cpp:34) set when
//       the canonical k-mer v occurs more than <cutoff> times; cut into blocks of 8 Mi k-mers
//       (1 MiB), each zlib compress()ed and appended to <out>.cz; <out>.cz.len holds one decimal
//       compressed size per line (2048 lines at k = 17, like
//       test/01.clean_correct/clean_reads.lib.kmer.freq.cz.len).  The loader itself mirrors a set
//       bit v to rc(v) when v <= rc(v), so only the canonical (smaller) k-mer is marked.
//   8-bit format  (-b 8)           read by correct_error/main.cpp:161-220
//       4^k saturating byte counts indexed by k-mer value, blocks of 8 MiB, same .cz / .cz.len
//       scheme; that loader applies its own -l cutoff and sets v and rc(v).
// Counting runs on the GPU (KFREQ engine of include/dbgk.h): every k-mer window of every read, N
// counted as A (correct_error/ReadMe.txt), canonical = min(forward, reverse complement).
//
// usage: kmerfreq [-k 17] [-f 1|2] [-b 1|8] [-m cutoff] [-t threads] [-o prefix] [-e store size | -a] <reads.lib>
//        output: <prefix>.kmer.freq.cz, <prefix>.kmer.freq.cz.len   (prefix defaults to <reads.lib>)
#include <unistd.h>
#include <zlib.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <mutex>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "dbgk.h"
#include "reads_io.h"

using namespace std;

static const uint64_t kBlockKmers = 8ull * 1024 * 1024;  // SrcBlockSize, correct_error/main_parallel_senior.cpp:71

static void die(const char *what, int rc)
{
	cerr << what << " failed: " << dbgk_strerror(rc);
	if (rc == DBGK_ERR_HIP) cerr << " [" << dbgk_last_error() << "]";
	cerr << endl;
	exit(1);
}

int main(int argc, char **argv)
{
	int k = 17, fmt = 1, bits = 1, threads = 8, max_read_len = 1000000;
	long cutoff = 1;
	string prefix;
	int c;
	unsigned long long expected = 0;
	bool atomics = false;
	while ((c = getopt(argc, argv, "k:f:b:m:t:o:q:r:e:ah")) != -1) {
		switch (c) {
			case 'k': k = atoi(optarg); break;
			case 'f': fmt = atoi(optarg); break;
			case 'b': bits = atoi(optarg); break;
			case 'm': cutoff = atol(optarg); break;
			case 't': threads = atoi(optarg); break;
			case 'o': prefix = optarg; break;
			case 'r': max_read_len = atoi(optarg); break;
			case 'e': expected = strtoull(optarg, NULL, 10); break;
			case 'a': atomics = true; break;
			case 'q': break;  // quality cutoff of the original tool: accepted, sequences carry no qualities here
			default:
				cout << "\nkmerfreq [-k 17] [-f 1:fq|2:fa] [-b 1|8 bit table] [-m cutoff, 1-bit: mark k-mers seen more than this, default 1]"
				     << " [-t threads] [-o prefix] [-e k-mer occurrences the partitioned counting engine holds before it merges"
				     << " them into the table (default: the input size, at most 2^30)] [-a count with atomics on the table instead]"
				     << " <reads.lib>\n" << endl;
				return 0;
		}
	}
	if (optind >= argc || k < 1 || k > 18 || (bits != 1 && bits != 8)) { cerr << "bad arguments (see -h)" << endl; return 2; }
	const string lib = argv[optind];
	if (prefix.empty()) prefix = lib;
	if (threads < 1) threads = 1;

	dbgk_config cfg;
	memset(&cfg, 0, sizeof cfg);
	cfg.kmer_size = k;
	cfg.max_read_len = max_read_len < k ? k : max_read_len;
	cfg.engine = DBGK_ENGINE_KFREQ;
	cfg.device_id = getenv("DBGK_DEVICE") ? atoi(getenv("DBGK_DEVICE")) : 0;
	const uint64_t batch_bytes = getenv("DBGK_BATCH_BYTES") ? max<uint64_t>(1, strtoull(getenv("DBGK_BATCH_BYTES"), NULL, 10)) : (128ull << 20);
	cfg.max_batch_bases = batch_bytes + 65536;
	// partitioned counting (occurrences radix-partitioned by hash, aggregated per key in LDS; input of any size
	// streams through the record store in rounds) unless -a asks for atomics on the byte table
	if (!atomics && expected == 0) {
		uint64_t bound = 0;
		bool known = true;
		ifstream probe(lib.c_str());
		for (string path; getline(probe, path);) {
			if (path.empty()) continue;
			FILE *fp = fopen(path.c_str(), "rb");
			if (!fp) continue;
			unsigned char magic[2] = {0, 0};
			const bool gz = fread(magic, 1, 2, fp) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
			fseek(fp, 0, SEEK_END);
			const long sz = ftell(fp);
			fclose(fp);
			if (gz || sz < 0) known = false; else bound += (uint64_t)sz;
		}
		expected = known && bound ? min<uint64_t>(bound, 1ull << 30) : (1ull << 30);
		if (expected < 1024) expected = 1024;
	}
	cfg.expected_kmers = atomics ? 0 : expected;
	// several GPUs (DBGK_GPUS=N: devices 0..N-1, or DBGK_GPU_LIST=a,b,c): every GPU counts the batches dealt to it
	// into a table of its own, the tables are combined by range at the end (dbgk_comm_* in include/dbgk.h)
	vector<int32_t> devices;
	if (const char *lst = getenv("DBGK_GPU_LIST")) {
		for (const char *p = lst; *p;) {
			devices.push_back((int32_t)strtol(p, const_cast<char **>(&p), 10));
			while (*p == ',' || *p == ' ') p++;
		}
	} else if (const char *ng = getenv("DBGK_GPUS")) {
		for (int i = 0; i < atoi(ng); i++) devices.push_back(i);
	}
	dbgk_handle *h = nullptr;
	dbgk_comm *comm = nullptr;
	int rc;
	if (devices.size() > 1) {
		if (cfg.expected_kmers) cfg.expected_kmers = max<uint64_t>(cfg.expected_kmers / devices.size(), 1024);
		rc = dbgk_comm_create(&cfg, devices.data(), (uint32_t)devices.size(), &comm);
		if (rc) die("dbgk_comm_create", rc);
		cerr << "counting on " << devices.size() << " GPUs" << endl;
	} else {
		if (devices.size() == 1) cfg.device_id = devices[0];
		rc = dbgk_create(&cfg, &h);
		if (rc) die("dbgk_create", rc);
	}

	vector<char> bases;
	vector<uint64_t> offsets(1, 0);
	auto flush = [&]() {
		if (offsets.size() > 1) {
			rc = comm ? dbgk_comm_push_reads(comm, bases.data(), offsets.data(), offsets.size() - 1)
			          : dbgk_push_reads(h, bases.data(), offsets.data(), offsets.size() - 1);
			if (rc) die("dbgk_push_reads", rc);
		}
		bases.clear();
		offsets.assign(1, 0);
	};
	ifstream list(lib.c_str());
	if (!list) { cerr << "fail to open " << lib << endl; return 1; }
	for (string path; getline(list, path);) {
		if (path.empty()) continue;
		cerr << "counting k-mers of " << path << endl;
		const bool ok = for_each_read_in_file(path, fmt, [&](const char *seq, size_t len) {
			bases.insert(bases.end(), seq, seq + len);
			offsets.push_back(bases.size());
			if (bases.size() >= batch_bytes) flush();
		});
		if (!ok) cerr << "fail to open reads file " << path << endl;
	}
	flush();
	dbgk_stats st;
	rc = comm ? dbgk_comm_finalize(comm, &st) : dbgk_finalize(h, &st);
	if (rc) die("dbgk_finalize", rc);
	cerr << "reads " << st.total_reads << "  k-mers " << st.stored_kmers << "  distinct canonical k-mers " << st.count << endl;
	if (st.other_bytes) cerr << "Alert message: " << st.other_bytes << " sequence bytes are none of ACGTNacgtn; they were read as A (like N)" << endl;

	// blocks of 8 Mi k-mers; compressed independently by a small thread pool, written in order
	const uint64_t total = 1ull << (2 * k);
	const uint64_t n_blocks = (total + kBlockKmers - 1) / kBlockKmers;
	const uint64_t block_bytes_full = bits == 1 ? kBlockKmers / 8 : kBlockKmers;
	vector<vector<unsigned char>> packed(n_blocks);
	atomic<uint64_t> next(0);
	atomic<int> failed(0);
	vector<thread> pool;
	mutex *export_lock = new mutex();  // one handle = one host thread at a time
	for (int t = 0; t < threads; t++) {
		pool.emplace_back([&]() {
			vector<unsigned char> raw(block_bytes_full);
			for (uint64_t b = next++; b < n_blocks; b = next++) {
				const uint64_t first = b * kBlockKmers;
				const uint64_t n = min<uint64_t>(kBlockKmers, total - first);
				const uint64_t nbytes = bits == 1 ? n / 8 : n;
				int erc;
				{
					lock_guard<mutex> g(*export_lock);
					if (comm)
						erc = bits == 1 ? dbgk_comm_kfreq_export_bits(comm, (uint32_t)cutoff, first / 8, nbytes, raw.data())
						                : dbgk_comm_kfreq_export_counts(comm, first, n, raw.data());
					else
						erc = bits == 1 ? dbgk_kfreq_export_bits(h, (uint32_t)cutoff, first / 8, nbytes, raw.data())
						                : dbgk_kfreq_export_counts(h, first, n, raw.data());
				}
				if (erc) { failed = erc; return; }
				uLongf clen = compressBound(nbytes);
				packed[b].resize(clen);
				if (compress(packed[b].data(), &clen, raw.data(), nbytes) != Z_OK) { failed = -100; return; }
				packed[b].resize(clen);
			}
		});
	}
	for (auto &t : pool) t.join();
	if (failed) die("table export", failed);
	const string cz = prefix + ".kmer.freq.cz";
	FILE *fz = fopen(cz.c_str(), "wb");
	ofstream flen((cz + ".len").c_str());
	if (!fz || !flen) { cerr << "fail to open output " << cz << endl; return 1; }
	for (uint64_t b = 0; b < n_blocks; b++) {
		fwrite(packed[b].data(), 1, packed[b].size(), fz);
		flen << packed[b].size() << "\n";
	}
	fclose(fz);
	cerr << "wrote " << cz << " (" << n_blocks << " blocks, " << bits << "-bit format)" << endl;
	if (comm) dbgk_comm_destroy(comm);
	else dbgk_destroy(h);
	return 0;
}
