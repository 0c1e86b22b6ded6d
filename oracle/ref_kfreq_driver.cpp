// TEST INFRASTRUCTURE ONLY -- driver around the REAL reference loaders of the correct_error
// module's k-mer frequency table (the consumer side of BASELINE cfg4 / SURVEY section 8(f)-2).
//
// Ours; compiled by oracle/Makefile together with the reference's own sources where they lie under
// /root/reference/correct_error.  Two binaries, because the two loaders live in two different
// programs of the reference with clashing globals:
//   ref_kfreq1  (REF_KFREQ_1BIT)  main_parallel_senior.cpp:334-408
//                                 make_kmerFreq_1bit_table_from_1BitGz_pthread(): 1-bit .cz -> bit table,
//                                 then mirrors every set bit i to rc(i) when i <= rc(i) (:310-329)
//   ref_kfreq8  (REF_KFREQ_8BIT)  main.cpp:161-220
//                                 make_kmerFreq_1bit_table_from_8BitGz(): 8-bit .cz -> bit table with the
//                                 low-frequency cutoff, sets idx and rc(idx)
// The reference's main() in those files is renamed on the compiler command line (-Dmain=...).
//
// usage: ref_kfreq1 <file.cz> <k> <threads> <out.bits>
//        ref_kfreq8 <file.cz> <k> <low_freq_cutoff> <out.bits>
// writes the 4^k/8-byte bit table and prints one JSON line of the loader's own statistics.
#include <inttypes.h>
#include <cstdio>
#include <cstdlib>
#include <string>
using namespace std;

extern int KmerSize;
extern uint8_t *KmerFreq;

#ifdef REF_KFREQ_1BIT
extern int threadNum;
extern uint64_t Kmer_theory_total, Kmer_hifreq_num;
uint8_t *make_kmerFreq_1bit_table_from_1BitGz_pthread(string &kmer_freq_file, int Ksize, uint64_t &total);
#else
uint8_t *make_kmerFreq_1bit_table_from_8BitGz(string &kmer_freq_file, int Ksize, uint64_t &total, uint64_t &num_total_kmers,
                                              uint64_t &num_effect_kmers, int low_freq_cutoff, double &low_freq_ratio);
#endif

#undef main
int main(int argc, char **argv)
{
	if (argc < 5) { fprintf(stderr, "usage: %s <file.cz> <k> <threads|cutoff> <out.bits>\n", argv[0]); return 2; }
	string path = argv[1];
	KmerSize = atoi(argv[2]);
	uint64_t total = 0;
#ifdef REF_KFREQ_1BIT
	threadNum = atoi(argv[3]);
	Kmer_theory_total = 0;
	make_kmerFreq_1bit_table_from_1BitGz_pthread(path, KmerSize, Kmer_theory_total); // fills the global KmerFreq
	total = Kmer_theory_total;
	printf("{\"total\": %llu, \"hifreq\": %llu}\n", (unsigned long long)total, (unsigned long long)Kmer_hifreq_num);
#else
	uint64_t n_total = 0, n_effect = 0;
	double low_ratio = 0;
	KmerFreq = make_kmerFreq_1bit_table_from_8BitGz(path, KmerSize, total, n_total, n_effect, atoi(argv[3]), low_ratio);
	printf("{\"total\": %llu, \"kmers\": %llu, \"effect\": %llu, \"low_ratio\": %.9f}\n", (unsigned long long)total,
	       (unsigned long long)n_total, (unsigned long long)n_effect, low_ratio);
#endif
	FILE *fp = fopen(argv[4], "wb");
	if (!fp) { perror(argv[4]); return 3; }
	fwrite(KmerFreq, 1, total / 8, fp);
	fclose(fp);
	return 0;
}
