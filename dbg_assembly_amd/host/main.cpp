// debruijn_contig -- command line of the graph-construction stage on MI355X.
//
// Same options, defaults and positional argument as the reference's DBG_contig/main.cpp:97-124,
// 166-193 (getopt string "k:r:f:o:t:i:l:e:b:D:T:I:P:W:C:G:B:U:L:E:M:h"); the help wording is this build's own.  The graph stage
// (build_debruijn_graph) runs on the GPU.  The contig stage (tip/bubble removal, contig read-out:
// DBG_contig/contig.cpp) is the reference's unchanged host code and is NOT part of this repository;
// when this program is linked together with it build_contig_sequence() is
// called exactly as in the reference, otherwise the program stops after the graph stage and
// writes the artefacts that stage defines: <prefix>.contig.kmer.freq (first pass of
// calculate_kmer_links) and, if DBGK_DUMP is set, the canonical node dump.
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "DBGgraph.h"
#include "dbgk_env.h"

// parameters of the contig stage: parsed for command-line compatibility (main.cpp:177-189) and
// handed to the contig stage when it is linked in (it defines the same globals; these are weak).
int KmerFreqCutoff __attribute__((weak)) = 2;
int is_remove_tip __attribute__((weak)) = 1;
int Tip_len_cutoff __attribute__((weak)) = 100;
double Tip_depth_cutoff __attribute__((weak)) = 3.0;
int is_remove_lowedge __attribute__((weak)) = 1;
int LowCovEdge_len_cutoff __attribute__((weak)) = 100;
double LowCovEdge_depth_cutoff __attribute__((weak)) = 3.0;
int is_remove_bubble __attribute__((weak)) = 1;
int Bubble_len_cutoff __attribute__((weak)) = 100;
double Bubble_len_diff_rate_cutoff __attribute__((weak)) = 0.1;
double Bubble_base_diff_rate_cutoff __attribute__((weak)) = 0.1;
int Contig_len_cutoff __attribute__((weak)) = 125;

void build_contig_sequence() __attribute__((weak));  // DBG_contig/contig.h:67, present only when linked with contig.cpp

static void print_options(ostream &os, bool with_k_max)
{
	// option letters, argument kinds, defaults and order follow the reference (DBG_contig/main.cpp:97-124); the wording is ours
	os << "   -k <int>    k-mer length" << (with_k_max ? " (the contig stage: at most 31; graph stage alone: up to 63, 128-bit keys)" : "") << " [" << KmerSize << "]" << endl
	   << "   -r <int>    longest read length used; longer reads are cut to this [" << maxReadLen << "]" << endl
	   << "   -f <int>    input format: 1 = FASTQ, 2 = FASTA, one sequence per line, plain or .gz [" << Input_file_format << "]" << endl
	   << "   -o <str>    prefix of the output files [" << Output_prefix << "]" << endl
	   << "   -t <int>    host threads (table zeroing, contig stage; the k-mer work runs on the GPU) [" << threadNum << "]" << endl
	   << "   -i <float>  initial size of the k-mer hash table in units of 1e9 entries, 16 bytes each [" << initHashSize << "]" << endl
	   << "   -l <float>  load factor at which the hash table is enlarged [" << hashLoadFactor << "]" << endl
	   << "   -e <int>    how many times the hash table may double before further input is dropped [" << maxDoubleHashTimes << "]" << endl
	   << "   -b <int>    reads per block; the table is checked for enlarging after every full block [" << BufferNum << "]" << endl
	   << "   -D <int>    drop k-mer links seen at most this many times [" << KmerFreqCutoff << "]" << endl
	   << "   -T <int>    remove tips: 1 = yes, 0 = no [" << is_remove_tip << "]" << endl
	   << "   -I <int>    longest tip that is removed [" << Tip_len_cutoff << "]" << endl
	   << "   -P <float>  highest depth of a tip that is removed [" << Tip_depth_cutoff << "]" << endl
	   << "   -W <int>    remove low-coverage edges between two branching nodes: 1 = yes, 0 = no [" << is_remove_lowedge << "]" << endl
	   << "   -C <int>    longest low-coverage edge that is removed [" << LowCovEdge_len_cutoff << "]" << endl
	   << "   -G <float>  highest depth of a low-coverage edge that is removed [" << LowCovEdge_depth_cutoff << "]" << endl
	   << "   -B <int>    merge bubbles (drop the lower-coverage branch of a pair): 1 = yes, 0 = no [" << is_remove_bubble << "]" << endl
	   << "   -U <int>    longest bubble branch considered [" << Bubble_len_cutoff << "]" << endl
	   << "   -L <float>  largest relative length difference of the two branches of a bubble [" << Bubble_len_diff_rate_cutoff << "]" << endl
	   << "   -E <float>  largest relative base difference of the two branches of a bubble [" << Bubble_base_diff_rate_cutoff << "]" << endl
	   << "   -M <int>    shortest contig that is written [" << Contig_len_cutoff << "]" << endl;
}

static void usage()
{
	cout << "\ndebruijn_contig   <reads_file.lib>\n"
	     << "   \nBuilds the k-mer de Bruijn graph of the reads listed in <reads_file.lib> (one path per line) on an\n"
	     << "   AMD MI355X GPU -- the graph stage of the DBG_assembly contig builder -- and passes it to the contig stage\n" << endl
	     << "   Version: 1.0 (gfx950)\n" << endl;
	print_options(cout, true);
	cout << "   -h          this help" << endl << endl
	     << "   environment: DBGK_DEVICE=<gpu ordinal>  DBGK_BATCH_BYTES=<host batch size>  DBGK_DUMP=<file: sorted node dump>" << endl
	     << "                DBGK_ENGINE=1|2   1 = global-atomic insert, 2 = partitioned records + LDS-built table regions (default)" << endl
	     << "                DBGK_STORE_KMERS=<n>  k-mer occurrences the partitioned engine holds before merging them into the table" << endl
	     << "                DBGK_LAYOUT=ref   lay the hash table out slot for slot like `debruijn_contig -t 1` of the reference" << endl
	     << "                DBGK_LINKS=1      also run the contig stage's first pass (link records, delete flags, tip / branch lists) on the GPU" << endl
	     << "                DBGK_GPUS=<n> | DBGK_GPU_LIST=a,b,..   one table over several GPUs;  DBGK_WIDE_PASSES=<n>  (-k > 32) passes over the input" << endl
	     << "\nExample: \ndebruijn_contig  -k 31 -r 250  -t 10  -i 0.1  -M 125 -o Ecoli reads_files.lib   2> reads_files.debruijn_contig.log \n" << endl;
	exit(0);
}

static double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Leave without tearing down: the results are on disk, and returning from main() would spend a good part of a second unmapping
// the 16 bytes x table-size host array page by page and unloading the HIP runtime with all its device allocations.
// DBGK_SLOW_EXIT=1 returns normally (leak checkers).
static void leave(int code)
{
	cout.flush();
	cerr.flush();
	fflush(NULL);
	if (DBGK_EXPERIMENT_ENV("DBGK_SLOW_EXIT")) exit(code);
	_exit(code);
}

int main(int argc, char *argv[])
{
	const double t_main = wall_now();
	int c;
	while ((c = getopt(argc, argv, "k:r:f:o:t:i:l:e:b:D:T:I:P:W:C:G:B:U:L:E:M:h")) != -1) {
		switch (c) {
			case 'k': KmerSize = atoi(optarg); break;
			case 'r': maxReadLen = atoi(optarg); break;
			case 'f': Input_file_format = atoi(optarg); break;
			case 'o': Output_prefix = optarg; break;
			case 't': threadNum = atoi(optarg); break;
			case 'i': initHashSize = atof(optarg); break;
			case 'l': hashLoadFactor = atof(optarg); break;
			case 'e': maxDoubleHashTimes = atoi(optarg); break;
			case 'b': BufferNum = atoi(optarg); break;
			case 'D': KmerFreqCutoff = atoi(optarg); break;
			case 'T': is_remove_tip = atoi(optarg); break;
			case 'I': Tip_len_cutoff = atoi(optarg); break;
			case 'P': Tip_depth_cutoff = atof(optarg); break;
			case 'W': is_remove_lowedge = atoi(optarg); break;
			case 'C': LowCovEdge_len_cutoff = atoi(optarg); break;
			case 'G': LowCovEdge_depth_cutoff = atof(optarg); break;
			case 'B': is_remove_bubble = atoi(optarg); break;
			case 'U': Bubble_len_cutoff = atoi(optarg); break;
			case 'L': Bubble_len_diff_rate_cutoff = atof(optarg); break;
			case 'E': Bubble_base_diff_rate_cutoff = atof(optarg); break;
			case 'M': Contig_len_cutoff = atof(optarg); break;
			default: usage();
		}
	}
	if (argc < 2 || optind >= argc) usage();

	cerr << "\nProgram parameters setting:" << endl;
	print_options(cerr, false);
	cerr << endl;

	string reads_lib_file = argv[optind++];
	vector<string> reads_files;
	reading_file_list(reads_lib_file, reads_files);

	const double t_build0 = wall_now();
	build_debruijn_graph(reads_files);
	const double t_build1 = wall_now();
	cerr << "\nLoad reads, chop kmer, build kmer graph finished !" << endl;
	if (DbgkLastStatus != 0) {
		cerr << "graph construction failed with status " << DbgkLastStatus << endl;
		leave(1);
	}
	if (const char *dump = getenv("DBGK_DUMP")) write_sorted_dump(dump);
	if (const char *img = getenv("DBGK_DUMP_TABLE")) write_table_image(img);
	if (const char *lk = getenv("DBGK_DUMP_LINKS")) write_links_dump(lk);

	if (KmerSize > 32) { // the reference's consumer is written for 64-bit k-mers (uint64_t kmer, DBG_contig/kmerSet.h:71)
		cerr << "\nStart to calulate kmer links information!" << endl;
		write_kmer_freq_file(Output_prefix + ".contig.kmer.freq", KmerFreqCutoff);
		cerr << "\nGraph stage finished: k = " << KmerSize << " graph in kset_wide (32-byte nodes); the contig stage handles k <= 31 only" << endl;
	} else if (build_contig_sequence) {
		build_contig_sequence();
		cerr << "\nRemove tips, merge bubbles, output contig sequence finished !" << endl;
		cerr << "\nAssembly completely finished!" << endl;
	} else {
		cerr << "\nStart to calulate kmer links information!" << endl;
		write_kmer_freq_file(Output_prefix + ".contig.kmer.freq", KmerFreqCutoff);
		cerr << "\nGraph stage finished (contig stage not linked in, see INTEGRATION.md)" << endl;
	}
	if (getenv("DBGK_TIMINGS"))
		cerr << "Wall phases (s): start-up " << t_build0 - t_main << " build_debruijn_graph " << t_build1 - t_build0 << " after " << wall_now() - t_build1 << endl;
	if (getenv("DBGK_TIMINGS")) { // how much of the process sits in transparent huge pages
		if (FILE *fp = fopen("/proc/self/smaps_rollup", "r")) {
			char line[256];
			while (fgets(line, sizeof line, fp))
				if (!strncmp(line, "Rss:", 4) || !strncmp(line, "AnonHugePages:", 14)) cerr << line;
			fclose(fp);
		}
	}
	if (DBGK_EXPERIMENT_ENV("DBGK_SLOW_EXIT") && getenv("DBGK_TIMINGS")) { // what the teardown that leave() skips would cost
		const double t0 = wall_now();
		free_hash(kset);
		kset = NULL;
		const double t1 = wall_now();
		vector<string> none;
		dbgk_host_release_session();
		cerr << "Teardown (s): free host KmerSet " << t1 - t0 << " destroy GPU handle " << wall_now() - t1 << endl;
	}
	if (DBGK_EXPERIMENT_ENV("DBGK_EXIT_DESTROY")) { // experiment: give the device memory back before leaving
		dbgk_host_release_session();
	}
	leave(0);
	return 0;
}
