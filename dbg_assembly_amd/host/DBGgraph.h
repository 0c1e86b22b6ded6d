// DBGgraph.h -- construction of the k-mer de Bruijn graph (the hot path), host side.
//
// Source-compatible with the reference's DBG_contig/DBGgraph.h:25-66: same extern globals (set by
// the command line before the call, read by the contig stage after it) and the same entry point
// build_debruijn_graph().  The work behind it -- read -> 2-bit k-mers -> canonical form -> hash
// insert with saturating neighbour counters -- runs on an MI355X through include/dbgk.h; the
// result is handed over as an ordinary host KmerSet in `kset`.
#ifndef DBGK_HOST_DBGGRAPH_H_
#define DBGK_HOST_DBGGRAPH_H_

#include <inttypes.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <ctime>
#include <iostream>
#include <zlib.h>   // the reference's DBGgraph.h:18 pulls in gzstream.h -> <zlib.h> (-> <unistd.h>: main.cpp's getopt relies on it)
#include "kmerSet.h"
#include "seqKmer.h"

using namespace std;

// parameters (defaults in DBGgraph.cpp; the reference's are at DBG_contig/DBGgraph.cpp:10-21)
extern int KmerSize;                 // -k
extern int maxReadLen;               // -r  reads are trimmed to this length
extern int KmerNumInRead;            // maxReadLen - KmerSize + 1
extern int Input_file_format;        // -f  1 = one-line FASTQ(.gz), 2 = one-line FASTA(.gz)
extern string Output_prefix;         // -o
extern int threadNum;                // -t  host threads (table zeroing, consumer); the GPU does the hashing
extern KmerSet *kset;                // THE result
extern KmerSet128 *kset_wide;        // ... for -k 33..63 (this build only; kset then holds just the key-0 node)
extern double initHashSize;          // -i  initial table size in units of 1e9 slots (16 GB each)
extern uint64_t maxDoubleHashTimes;  // -e
extern uint64_t doubleHashTimes;     // doublings the reference would have performed for this input
extern float hashLoadFactor;         // -l
extern int BufferNum;                // -b  reads per block in the reference; here only the progress-line granularity
extern string *RawReads;             // staging buffers of the reference's pthread pipeline: not used
extern uint64_t *StoreKmer;          //   by this build (kept so that code naming them still links),
extern uint8_t *StoreLeftBase;       //   always NULL
extern uint8_t *StoreRightBase;
extern uint8_t *Signal;
extern uint64_t Kmer_total_num;      // sum over reads with len >= K of (len - K + 1), untrimmed length
extern uint64_t Total_reads_num;     // records seen, including too-short ones
extern uint64_t KmerHeadMaskVal;     // 2^(2K) - 1
extern uint64_t KmerRCOrVal[4];      // (3 - b) << (2K - 2)
extern KmerNode *PolyA;              // the key-0 (poly-A / poly-T) node while building; NULL afterwards

extern clock_t time_start;
extern clock_t time_end;

// The reference declares its two pthread routines and the per-file driver here (DBGgraph.h:53-62).
// parse_one_reads_file streams one file to the GPU; the two thread routines have no counterpart
// (their work is the HIP kernels) and abort if called.
void *thread_parseBlock(void *threadId_p);
void *thread_updatekmers(void *threadId_p);
void parse_one_reads_file(string &reads_file);

void build_debruijn_graph(vector<string> &reads_files);

// extras of this build (not in the reference) -----------------------------------------------------
// status of the last build: 0 ok, otherwise a DBGK_ERR_* code (the reference has no error channel:
// problems are printed and execution continues; kset is then an empty, valid set)
extern int DbgkLastStatus;
// counts as computed on the device by the consumer's first pass (contig.cpp:119-181); filled when
// DBGK_LINK_STATS is set in the environment or write_kmer_freq_file() is called before teardown
int write_kmer_freq_file(const string &path, int kmer_freq_cutoff);
// canonical dump (every node sorted by k-mer) of the current kset, the parity artefact
int write_sorted_dump(const string &path);
// raw image (size, count, node array, nul_flag) of the current kset
int write_table_image(const string &path);
// DBGK_LINKS=1 in the environment: build_debruijn_graph() also runs the consumer's whole first pass, calculate_kmer_links
// (DBG_contig/contig.cpp:107-181), on the device for the table it hands over (dbgk_export_host_table_links): kset->del_flag
// holds the low-frequency nodes' delete bits, and below are the per-slot 2-byte KmerLink records (contig.h:31-42; kset->size
// of them, malloc()ed, NULL when not computed) and the tip / branching slots in ascending slot order -- what the serial scan
// of contig.cpp:119-181 would produce with -D KmerFreqCutoff, ready to be adopted by the contig stage
int write_links_dump(const string &path);   // text dump of the three artefacts below (tests)
extern uint16_t *DbgkKmerLinks;
extern vector<uint64_t> DbgkTipNodes;
extern vector<uint64_t> DbgkBranchNodes;
// build_debruijn_graph() keeps its GPU handle (table, record stores, pinned staging buffers) alive for write_kmer_freq_file /
// DBGK_LINKS users that follow; a program that embeds libdbgasm_host.so and goes on living gives it back with this
// (debruijn_contig itself leaves through _exit: the process ends faster than the teardown runs)
void dbgk_host_release_session();

#endif
