// seqKmer.h -- k-mer codec helpers of the host layer.
//
// Source-compatible with the reference's DBG_contig/seqKmer.h:21-61 (same names, argument meaning
// and results, so DBG_contig/main.cpp and contig.cpp compile against it unchanged); implementation
// is new.  The per-read work these helpers do in the reference's hot loop (DBGgraph.cpp:64-98) runs
// on the GPU here; the host versions remain for the consumer (contig.cpp uses bit2seq,
// get_rev_com_kbit, ... on single k-mers).
#ifndef DBGK_HOST_SEQKMER_H_
#define DBGK_HOST_SEQKMER_H_

#include <inttypes.h>
#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <string>
#include <vector>

using namespace std;  // the reference's headers export std into every includer (seqKmer.h:16); consumers rely on it

extern char alphabet[128];   // ASCII -> 0..3 (A/a/N/n=0, C=1, G=2, T=3), 4 for anything else   (seqKmer.cpp:9-19)
extern char bases[5];        // "ACGTN"                                                        (seqKmer.cpp:22-24)
extern char c_bases[5];      // complement of bases[]: "TGCAN"                                 (seqKmer.cpp:27-29)

uint64_t seq2bit(string &kseq);                            // 2 bits per base, first base most significant
string bit2seq(uint64_t kbit, int kmerSize);
int check_seq(string &seq);                                // 1 when every character is one of ACGTNacgtn... (alphabet != 4)
inline char complement_base(char base) { return c_bases[(int)alphabet[(unsigned char)base & 127]]; }
void reverse_complement(string &in_str, string &out_str);  // APPENDS to out_str
void complement_sequence(string &str);
void reading_file_list(string &file_list, vector<string> &files);
uint64_t get_rev_com_kbit(uint64_t kbit, uint8_t ksize);
void display_num_in_bits(uint64_t num, int len);
uint64_t pow_integer(int base, int exponent);              // wrapping u64 product (2^64 -> 0)

#endif
