// TEST INFRASTRUCTURE ONLY -- driver around the REAL reference seed index of the link_scaffold
// module: chop_contig_to_kmerset (link_scaffold/map_func.cpp:119-173) over
// add_kmerset (link_scaffold/kmerSet.cpp:168-210).  Ours; compiled by oracle/Makefile with the
// reference's kmerSet.cpp map_func.cpp seqKmer.cpp gzstream.cpp where they lie under
// /root/reference/link_scaffold.
//
// usage: ref_seed <contigs.fa> <k> <hash_size (0 = 3 * total length, as map_pair.cpp:122)> <load_factor> <dump.txt>
// dump: "#contigs N size S count C" then one line per stored node sorted by kmer:
//       kmer <TAB> id <TAB> pos <TAB> freq <TAB> direct
#include "map_func.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv)
{
	if (argc < 6) { fprintf(stderr, "usage: ref_seed <contigs.fa> <k> <hash_size> <load_factor> <dump.txt>\n"); return 2; }
	string contig_file = argv[1];
	KmerSize = atoi(argv[2]);
	uint64_t hash_size = strtoull(argv[3], NULL, 10);
	double load_factor = atof(argv[4]);
	vector<string> contig_ids, contig_seqs;
	read_contig_file(contig_file, contig_ids, contig_seqs);
	uint64_t total_len = 0;
	for (size_t i = 0; i < contig_seqs.size(); i++) total_len += contig_seqs[i].size();
	if (hash_size == 0) hash_size = total_len * 3;
	KmerSet *kset = init_kmerset(hash_size, load_factor);
	chop_contig_to_kmerset(kset, contig_seqs);
	vector<KmerNode> nodes;
	for (uint64_t i = 0; i < kset->size; i++)
		if (!is_entity_null(kset->nul_flag, i)) nodes.push_back(kset->array[i]);
	std::sort(nodes.begin(), nodes.end(), [](const KmerNode &a, const KmerNode &b) { return a.kmer < b.kmer; });
	FILE *fp = fopen(argv[5], "w");
	if (!fp) { perror(argv[5]); return 3; }
	fprintf(fp, "#contigs %llu size %llu count %llu\n", (unsigned long long)contig_seqs.size(), (unsigned long long)kset->size,
	        (unsigned long long)kset->count);
	for (size_t i = 0; i < nodes.size(); i++)
		fprintf(fp, "%llu\t%u\t%u\t%u\t%u\n", (unsigned long long)nodes[i].kmer, (unsigned)nodes[i].id, (unsigned)nodes[i].pos,
		        (unsigned)nodes[i].freq, (unsigned)nodes[i].direct);
	fclose(fp);
	return 0;
}
