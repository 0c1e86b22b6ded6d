// TEST INFRASTRUCTURE ONLY -- driver around the *real* reference hot path.
//
// This file is ours; it is compiled together with the reference's own sources
// where they lie under /root/reference/DBG_contig (seqKmer.cpp kmerSet.cpp
// DBGgraph.cpp gzstream.cpp) by oracle/Makefile into oracle/_ref/ref_dbg.
// Nothing from the reference is copied into this repository.
//
// It calls the reference's public API exactly as DBG_contig/main.cpp:162-204
// does (set the extern globals of DBGgraph.h:25-36, reading_file_list(),
// build_debruijn_graph()) and then prints the canonical dump used as the
// parity artefact: every non-null slot's (kmer, l_link, r_link), sorted by kmer.
//
// Sub-commands
//   ref_dbg build [-k -r -f -t -i -l -e -b as in main.cpp:166] [-d dump.txt] [-T table.img] [-S] [-q] <reads.lib>
//                 -T writes the raw table image (size, count, node array, nul_flag): the slot LAYOUT
//                 -S adds "digest" (order-independent sum over the non-null slots, the definition of
//                    dbgk_digest in include/dbgk.h) and "depth_stat" (histogram of the 8 link counters of
//                    every node, get_next_kmer_depth kmerSet.cpp:341-344) to the JSON line: the full-size
//                    record tests/golden/cfg2_full.json is confirmed against it (make_cfg2_full.py --ref)
//                 -H adds "sorted_sha256": SHA-256 over the canonical dump as PACKED records -- every non-null slot's
//                    KmerNode (kmerSet.h:70-75: u64 kmer, u32 l_link, u32 r_link, 16 bytes little-endian) sorted by kmer,
//                    nothing else -- i.e. over exactly the bytes dbgk_export_sorted returns: "bit-exact kmerSet" made
//                    literal at full size (tests/golden/cfg2_full.json, make_cfg2_full.py --ref)
//   ref_dbg kat                        known-answer values of the codec / hash helpers
//   ref_dbg prime <n> [<n> ...]        find_next_prime(n)
#include "DBGgraph.h"

#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

static inline uint64_t mix64(uint64_t x)   // splitmix64 finaliser (ours: the digest of include/dbgk.h, not reference code)
{
	uint64_t z = x + 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

// SHA-256 (FIPS 180-4), ours: the driver hashes gigabytes of records without a library dependency
struct Sha256 {
	uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
	unsigned char buf[64];
	uint64_t len = 0;
	static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
	void block(const unsigned char *p)
	{
		static const uint32_t K[64] = {
			0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
			0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
			0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
			0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
			0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
			0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
		uint32_t w[64];
		for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
		for (int i = 16; i < 64; i++) {
			const uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
			w[i] = w[i - 16] + s0 + w[i - 7] + s1;
		}
		uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
		for (int i = 0; i < 64; i++) {
			const uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
			const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
			hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
		}
		h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
	}
	void update(const void *data, size_t n)
	{
		const unsigned char *p = static_cast<const unsigned char *>(data);
		size_t fill = (size_t)(len & 63);
		len += n;
		if (fill) {
			const size_t take = std::min(n, (size_t)64 - fill);
			memcpy(buf + fill, p, take);
			p += take; n -= take; fill += take;
			if (fill < 64) return;
			block(buf);
		}
		for (; n >= 64; p += 64, n -= 64) block(p);
		if (n) memcpy(buf, p, n);
	}
	std::string hex()
	{
		const uint64_t bits = len * 8;
		const unsigned char one = 0x80, zero = 0;
		update(&one, 1);
		while ((len & 63) != 56) update(&zero, 1);
		unsigned char be[8];
		for (int i = 0; i < 8; i++) be[i] = (unsigned char)(bits >> (56 - 8 * i));
		update(be, 8);
		char out[65];
		for (int i = 0; i < 8; i++) snprintf(out + 8 * i, 9, "%08x", h[i]);
		return std::string(out, 64);
	}
};

static int cmd_build(int argc, char **argv)
{
	std::string dump_path, image_path;
	int quiet = 0, summary = 0, sorted_hash = 0;
	int c;
	optind = 1;
	while ((c = getopt(argc, argv, "k:r:f:t:i:l:e:b:d:T:SHq")) != -1) {
		switch (c) {
			case 'k': KmerSize = atoi(optarg); break;
			case 'r': maxReadLen = atoi(optarg); break;
			case 'f': Input_file_format = atoi(optarg); break;
			case 't': threadNum = atoi(optarg); break;
			case 'i': initHashSize = atof(optarg); break;
			case 'l': hashLoadFactor = atof(optarg); break;
			case 'e': maxDoubleHashTimes = atoi(optarg); break;
			case 'b': BufferNum = atoi(optarg); break;
			case 'd': dump_path = optarg; break;
			case 'T': image_path = optarg; break;
			case 'S': summary = 1; break;
			case 'H': sorted_hash = 1; break;
			case 'q': quiet = 1; break;
			default: return 2;
		}
	}
	if (optind >= argc) { fprintf(stderr, "ref_dbg build: missing <reads.lib>\n"); return 2; }
	std::string lib = argv[optind];

	if (quiet) { if (!freopen("/dev/null", "w", stderr)) return 3; }

	std::vector<std::string> files;
	reading_file_list(lib, files);

	auto t0 = std::chrono::steady_clock::now();
	build_debruijn_graph(files);
	auto t1 = std::chrono::steady_clock::now();
	double wall = std::chrono::duration<double>(t1 - t0).count();

	printf("{\"reads\": %llu, \"kmers\": %llu, \"count\": %llu, \"size\": %llu, \"max\": %llu, "
	       "\"conflict\": %llu, \"threads\": %d, \"wall_s\": %.6f",
	       (unsigned long long)Total_reads_num, (unsigned long long)Kmer_total_num,
	       (unsigned long long)kset->count, (unsigned long long)kset->size,
	       (unsigned long long)kset->max, (unsigned long long)kset->count_conflict,
	       threadNum, wall);
	if (summary) {
		uint64_t digest = 0, nodes = 0;
		std::vector<long long> depth(256, 0);
		for (uint64_t i = 0; i < kset->size; i++) {
			if (is_entity_null(kset->nul_flag, i)) continue;
			const KmerNode &e = kset->array[i];
			digest += mix64(e.kmer ^ mix64(((uint64_t)e.l_link << 32) | e.r_link));
			for (uint8_t b = 0; b < 4; b++) {
				depth[get_next_kmer_depth(e.l_link, b)]++;
				depth[get_next_kmer_depth(e.r_link, b)]++;
			}
			nodes++;
		}
		printf(", \"nonnull_slots\": %llu, \"digest\": %llu, \"depth_stat\": [", (unsigned long long)nodes, (unsigned long long)digest);
		for (int d = 0; d < 256; d++) printf("%s%lld", d ? ", " : "", depth[d]);
		printf("]");
	}
	if (sorted_hash) {
		static_assert(sizeof(KmerNode) == 16, "the packed record is the reference's own node");
		std::vector<KmerNode> nodes;
		nodes.reserve(kset->count);
		for (uint64_t i = 0; i < kset->size; i++)
			if (!is_entity_null(kset->nul_flag, i)) nodes.push_back(kset->array[i]);
		std::sort(nodes.begin(), nodes.end(), [](const KmerNode &a, const KmerNode &b) { return a.kmer < b.kmer; });
		Sha256 sha;
		sha.update(nodes.data(), nodes.size() * sizeof(KmerNode));
		printf(", \"sorted_records\": %llu, \"sorted_sha256\": \"%s\"", (unsigned long long)nodes.size(), sha.hex().c_str());
	}
	printf("}\n");

	if (!image_path.empty()) {
		FILE *fp = fopen(image_path.c_str(), "wb");
		if (!fp) { perror("image"); return 4; }
		const uint64_t hdr[2] = {kset->size, kset->count};
		fwrite(hdr, 8, 2, fp);
		fwrite(kset->array, sizeof(KmerNode), kset->size, fp);
		fwrite(kset->nul_flag, 1, kset->size / 8 + 1, fp);
		fclose(fp);
	}
	if (!dump_path.empty()) {
		std::vector<KmerNode> nodes;
		nodes.reserve(kset->count);
		for (uint64_t i = 0; i < kset->size; i++) {
			if (!is_entity_null(kset->nul_flag, i)) nodes.push_back(kset->array[i]);
		}
		std::sort(nodes.begin(), nodes.end(),
		          [](const KmerNode &a, const KmerNode &b) { return a.kmer < b.kmer; });
		FILE *fp = fopen(dump_path.c_str(), "w");
		if (!fp) { perror("dump"); return 4; }
		fprintf(fp, "#reads %llu kmers %llu count %llu\n",
		        (unsigned long long)Total_reads_num, (unsigned long long)Kmer_total_num,
		        (unsigned long long)kset->count);
		for (size_t i = 0; i < nodes.size(); i++) {
			fprintf(fp, "%llu\t%08x\t%08x\n", (unsigned long long)nodes[i].kmer,
			        nodes[i].l_link, nodes[i].r_link);
		}
		fclose(fp);
	}
	return 0;
}

static int cmd_kat()
{
	const char *seqs[] = {
		"ACGTACGTACGTACGTACGTACGTACGTACG", "ANNTG", "acgtn", "TTTTTTTTTTTTTTTTT",
		"GATTACAGATTACAGATTACAGATTACAGATT", "CCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCC", "A", "T",
	};
	for (size_t i = 0; i < sizeof(seqs) / sizeof(seqs[0]); i++) {
		std::string s = seqs[i];
		uint64_t b = seq2bit(s);
		uint64_t rc = get_rev_com_kbit(b, (uint8_t)s.size());
		printf("seq2bit\t%s\t%llu\trc\t%llu\tback\t%s\n", s.c_str(), (unsigned long long)b,
		       (unsigned long long)rc, bit2seq(b, (int)s.size()).c_str());
	}
	const uint64_t hs[] = {0ULL, 1ULL, 2ULL, 0x0123456789ABCDEFULL, 488296166657017542ULL,
	                       (1ULL << 62) - 1, 0xFFFFFFFFFFFFFFFFULL, 0x8000000000000000ULL};
	for (size_t i = 0; i < sizeof(hs) / sizeof(hs[0]); i++) {
		printf("hash_code\t%llu\t%llu\n", (unsigned long long)hs[i], (unsigned long long)hash_code(hs[i]));
	}
	for (int b = 0; b < 4; b++) {
		printf("get_next_kmer_depth\t0x01020304\t%d\t%u\n", b, (unsigned)get_next_kmer_depth(0x01020304u, (uint8_t)b));
	}
	for (int e = 0; e <= 64; e += 8) {
		printf("pow_integer\t2\t%d\t%llu\n", e, (unsigned long long)pow_integer(2, e));
	}
	const uint64_t ps[] = {9, 15, 25, 49, 121, 169, 1000003, 1000005};
	for (size_t i = 0; i < sizeof(ps) / sizeof(ps[0]); i++) {
		printf("is_prime\t%llu\t%d\n", (unsigned long long)ps[i], is_prime(ps[i]));
	}
	return 0;
}

static int cmd_prime(int argc, char **argv)
{
	for (int i = 1; i < argc; i++) {
		uint64_t n = strtoull(argv[i], NULL, 10);
		printf("find_next_prime\t%llu\t%llu\n", (unsigned long long)n, (unsigned long long)find_next_prime(n));
	}
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: ref_dbg build|kat|prime ...\n"); return 2; }
	std::string cmd = argv[1];
	if (cmd == "build") return cmd_build(argc - 1, argv + 1);
	if (cmd == "kat") return cmd_kat();
	if (cmd == "prime") return cmd_prime(argc - 1, argv + 1);
	fprintf(stderr, "unknown sub-command %s\n", cmd.c_str());
	return 2;
}
