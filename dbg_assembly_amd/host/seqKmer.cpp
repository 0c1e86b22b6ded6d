// seqKmer.cpp -- host-side k-mer codec (see seqKmer.h).  Behaviour follows
// /root/reference/DBG_contig/seqKmer.cpp (line numbers in the header); the code is new.
#include "seqKmer.h"

namespace {
struct AlphabetInit {
	AlphabetInit()
	{
		for (int i = 0; i < 128; i++) alphabet[i] = 4;
		const char *zero = "AaNn";
		for (const char *p = zero; *p; p++) alphabet[(int)*p] = 0;
		alphabet['C'] = alphabet['c'] = 1;
		alphabet['G'] = alphabet['g'] = 2;
		alphabet['T'] = alphabet['t'] = 3;
	}
};
}  // namespace

char alphabet[128];
static AlphabetInit alphabet_init_once;
char bases[5] = {'A', 'C', 'G', 'T', 'N'};
char c_bases[5] = {'T', 'G', 'C', 'A', 'N'};

static inline int base_code(char c) { return alphabet[(unsigned char)c & 127]; }

uint64_t seq2bit(string &kseq)
{
	uint64_t packed = 0;
	for (char c : kseq) packed = (packed << 2) | (uint64_t)base_code(c);
	return packed;
}

string bit2seq(uint64_t kbit, int kmerSize)
{
	string out((size_t)kmerSize, 'A');
	for (int i = kmerSize - 1; i >= 0; i--, kbit >>= 2) out[(size_t)i] = bases[kbit & 3u];
	return out;
}

int check_seq(string &seq)
{
	for (char c : seq)
		if (base_code(c) == 4) return 0;
	return 1;
}

void reverse_complement(string &in_str, string &out_str)
{
	for (auto it = in_str.rbegin(); it != in_str.rend(); ++it) out_str.push_back(c_bases[base_code(*it)]);
}

void complement_sequence(string &str)
{
	for (char &c : str) c = c_bases[base_code(c)];
}

uint64_t get_rev_com_kbit(uint64_t kbit, uint8_t ksize)
{
	// complement every base (bitwise not), then reverse the order of the 32 two-bit groups with a
	// log-step butterfly; the k-mer ends up in the low 2*ksize bits.
	uint64_t x = ~kbit;
	x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
	x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
	x = __builtin_bswap64(x);
	return x >> (64 - 2 * (int)ksize);
}

void reading_file_list(string &file_list, vector<string> &files)
{
	ifstream in(file_list.c_str());
	if (!in) cerr << "fail to open input file" << file_list << endl;
	for (string line; getline(in, line);)
		if (!line.empty()) files.push_back(line);  // used verbatim, no trimming (seqKmer.cpp:109-112)
}

void display_num_in_bits(uint64_t num, int len)
{
	cout << num << "\t" << len << "\t";
	for (int i = len - 1; i >= 0; i--) cerr << ((num >> i) & 1u);
	cerr << endl;
}

uint64_t pow_integer(int base, int exponent)
{
	uint64_t r = 1;
	while (exponent-- > 0) r *= (uint64_t)(int64_t)base;
	return r;
}
