// Test driver for the host layer's reference-compatible API (no GPU needed): prints one line per
// check; tests/test_host_api.py compares the lines with the oracle / golden KATs.
#include <cstdio>
#include "DBGgraph.h"

int main()
{
	const char *seqs[] = {"ACGTACGTACGTACGTACGTACGTACGTACG", "ANNTG", "acgtn", "TTTTTTTTTTTTTTTTT",
	                      "GATTACAGATTACAGATTACAGATTACAGATT", "CCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCC", "A", "T"};
	for (const char *q : seqs) {
		string s = q;
		uint64_t b = seq2bit(s);
		printf("seq2bit\t%s\t%llu\trc\t%llu\tback\t%s\n", q, (unsigned long long)b,
		       (unsigned long long)get_rev_com_kbit(b, (uint8_t)s.size()), bit2seq(b, (int)s.size()).c_str());
	}
	const uint64_t hs[] = {0ULL, 1ULL, 2ULL, 0x0123456789ABCDEFULL, 488296166657017542ULL, (1ULL << 62) - 1,
	                       0xFFFFFFFFFFFFFFFFULL, 0x8000000000000000ULL};
	for (uint64_t h : hs) printf("hash_code\t%llu\t%llu\n", (unsigned long long)h, (unsigned long long)hash_code(h));
	for (int b = 0; b < 4; b++) printf("get_next_kmer_depth\t0x01020304\t%d\t%u\n", b, (unsigned)get_next_kmer_depth(0x01020304u, (uint8_t)b));
	for (int e = 0; e <= 64; e += 8) printf("pow_integer\t2\t%d\t%llu\n", e, (unsigned long long)pow_integer(2, e));
	const uint64_t ps[] = {9, 15, 25, 49, 121, 169, 1000003, 1000005};
	for (uint64_t p : ps) printf("is_prime\t%llu\t%d\n", (unsigned long long)p, is_prime(p));
	const uint64_t ns[] = {3, 4, 1000, 10000, 20000, 100000, 200006, 200014, 1000000, 2000006, 10000000, 20000038,
	                       100000000, 1000000000, 2000000014, 24, 48, 120, 168, 288, 360};
	for (uint64_t n : ns) printf("find_next_prime\t%llu\t%llu\n", (unsigned long long)n, (unsigned long long)find_next_prime(n));

	// set maintenance: insert via add_node_to_kmerset, enlarge twice, look everything up, delete some
	KmerSet *s = init_kmerset_parallel(1000, 0.7f, 3);
	printf("init\t%llu\t%llu\t%u\n", (unsigned long long)s->size, (unsigned long long)s->max, s->e_size);
	uint64_t x = 88172645463325252ULL;
	std::vector<uint64_t> keys;
	for (int i = 0; i < 650; i++) {
		x ^= x << 13; x ^= x >> 7; x ^= x << 17;
		KmerNode n = {x | 1, (uint32_t)i, (uint32_t)(i * 7)};
		add_node_to_kmerset(s, &n);
		keys.push_back(n.kmer);
	}
	enlarge_kmerset_parallel(s, 1, 2);
	printf("enlarge1\t%llu\t%llu\t%llu\n", (unsigned long long)s->size, (unsigned long long)s->max, (unsigned long long)s->count);
	enlarge_kmerset_parallel(s, 3000, 2);
	printf("enlarge2\t%llu\t%llu\t%llu\n", (unsigned long long)s->size, (unsigned long long)s->max, (unsigned long long)s->count);
	int found = 0, links_ok = 0;
	for (size_t i = 0; i < keys.size(); i++) {
		uint64_t idx = exist_kmerset(s, keys[i]);
		if (idx != s->size) {
			found++;
			if (s->array[idx].l_link == (uint32_t)i && s->array[idx].r_link == (uint32_t)(i * 7)) links_ok++;
		}
	}
	printf("lookup\t%d\t%d\t%d\n", found, links_ok, exist_kmerset(s, 12345678ULL * 2) == s->size);
	for (int i = 0; i < 100; i++) delete_kmerset(s, keys[i]);
	printf("delete\t%llu\t%d\t%d\n", (unsigned long long)s->count, exist_kmerset(s, keys[5]) == s->size, exist_kmerset(s, keys[500]) != s->size);
	// slot layout after the enlarges, as (slot,kmer) checksum: must equal the oracle's enlarge
	unsigned long long chk = 0;
	for (uint64_t i = 0; i < s->size; i++)
		if (!is_entity_null(s->nul_flag, i)) chk = chk * 1000003ULL + (i ^ s->array[i].kmer);
	printf("layout\t%llu\n", chk);
	string rc;
	string in = "AACGTN";
	reverse_complement(in, rc);
	printf("revcomp\t%s\t%c\t%d\n", rc.c_str(), complement_base('g'), check_seq(in));
	free_hash(s);
	return 0;
}
