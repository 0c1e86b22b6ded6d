// debruijn_contig -- command line of the graph-construction stage on MI355X.
//
// Same options, defaults and positional argument as the reference's DBG_contig/main.cpp:97-124,
// 166-193 (getopt string "k:r:f:o:t:i:l:e:b:D:T:I:P:W:C:G:B:U:L:E:M:h").  The graph stage
// (build_debruijn_graph) runs on the GPU.  The contig stage (tip/bubble removal, contig read-out:
// DBG_contig/contig.cpp) is the reference's unchanged host code and is NOT part of this repository;
// when this program is linked together with it build_contig_sequence() is
// called exactly as in the reference, otherwise the program stops after the graph stage and
// writes the artefacts that stage defines: <prefix>.contig.kmer.freq (first pass of
// calculate_kmer_links) and, if DBGK_DUMP is set, the canonical node dump.
#include <unistd.h>
#include <cstdlib>

#include "DBGgraph.h"

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
	os << "   -k <int>   set kmer size" << (with_k_max ? ", max 31" : "") << ", default=" << KmerSize << endl
	   << "   -r <int>   set maximum allowed read length, trimmed if longer, default=" << maxReadLen << endl
	   << "   -f <int>   set the input file format: 1: fq|gz(one-line), 2: fa|gz(one-line), default=" << Input_file_format << endl
	   << "   -o <str>   set the output file prefix, default = " << Output_prefix << endl
	   << "   -t <int>   thread number to run in parallel, default=" << threadNum << endl
	   << "   -i <float>  set initialization size (uint:G) of the kmer-hash, memory consumption ( * 16 G bytes ), default=" << initHashSize << endl
	   << "   -l <float>  set loading factor of the kmer hash, default=" << hashLoadFactor << endl
	   << "   -e <int>  max doubling times of hash size allowed to enlarge memory consumption, default=" << maxDoubleHashTimes << endl
	   << "   -b <int>  buffer size: number of reads loading into the buffer memory, default=" << BufferNum << endl
	   << "   -D <int>   delete kmer-links with frequency no larger than, default=" << KmerFreqCutoff << endl
	   << "   -T <int>   whether cut off tip-branch, 1:yes; 0:no; default=" << is_remove_tip << endl
	   << "   -I <int>   set the max allowed tip-branch length, default=" << Tip_len_cutoff << endl
	   << "   -P <float>  set the max allowed tip-branch depth, default=" << Tip_depth_cutoff << endl
	   << "   -W <int>   wheter cut off low-coverage branch between two branching nodes, 1:yes; 0:no; default=" << is_remove_lowedge << endl
	   << "   -C <int>    set the max allowed length for low-coverage branch, default=" << LowCovEdge_len_cutoff << endl
	   << "   -G <float>  set the max allowed depth for low-coverage branch, default=" << LowCovEdge_depth_cutoff << endl
	   << "   -B <int>   whether cut off the low-coverage branch for pairs of bubble branches, 1:yes; 0:no; default=" << is_remove_bubble << endl
	   << "   -U <int>   set the max allowed bubble-branch length, default=" << Bubble_len_cutoff << endl
	   << "   -L <float>   set the max allowed length difference rate between the two bubble-branchess, default=" << Bubble_len_diff_rate_cutoff << endl
	   << "   -E <float>  set the max allowed base difference rate between the two bubble-branches, default=" << Bubble_base_diff_rate_cutoff << endl
	   << "   -M <int>    set the minimum length for contig to output, default=" << Contig_len_cutoff << endl;
}

static void usage()
{
	cout << "\ndebruijn_contig   <reads_file.lib>\n"
	     << "   \nFunction: build the k-mer de Bruijn graph of the reads on an AMD MI355X GPU (graph stage of the\n"
	     << "   DBG_assembly contig builder) and hand it to the contig stage\n" << endl
	     << "   Verion: 1.0 (gfx950)\n" << endl;
	print_options(cout, true);
	cout << "   -h         get the help information" << endl << endl
	     << "   environment: DBGK_DEVICE=<gpu ordinal>  DBGK_BATCH_MB=<host batch size>  DBGK_DUMP=<file: sorted node dump>" << endl
	     << "                DBGK_LAYOUT=ref  lay the hash table out slot for slot like `debruijn_contig -t 1` of the reference" << endl
	     << "\nExample: \ndebruijn_contig  -k 31 -r 250  -t 10  -i 0.1  -M 125 -o Ecoli reads_files.lib   2> reads_files.debruijn_contig.log \n" << endl;
	exit(0);
}

int main(int argc, char *argv[])
{
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

	build_debruijn_graph(reads_files);
	cerr << "\nLoad reads, chop kmer, build kmer graph finished !" << endl;
	if (DbgkLastStatus != 0) {
		cerr << "graph construction failed with status " << DbgkLastStatus << endl;
		return 1;
	}
	if (const char *dump = getenv("DBGK_DUMP")) write_sorted_dump(dump);
	if (const char *img = getenv("DBGK_DUMP_TABLE")) write_table_image(img);

	if (build_contig_sequence) {
		build_contig_sequence();
		cerr << "\nRemove tips, merge bubbles, output contig sequence finished !" << endl;
		cerr << "\nAssembly completely finished!" << endl;
	} else {
		cerr << "\nStart to calulate kmer links information!" << endl;
		write_kmer_freq_file(Output_prefix + ".contig.kmer.freq", KmerFreqCutoff);
		cerr << "\nGraph stage finished (contig stage not linked in, see INTEGRATION.md)" << endl;
	}
	return 0;
}
