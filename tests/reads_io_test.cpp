// Test driver (no GPU): the mapped, multi-threaded reader of plain files (ChunkedReadsFile) must deliver exactly the records of
// the sequential reader (for_each_read_in_file), which follows the reference's rules (DBG_contig/DBGgraph.cpp:244-272), for any
// input and any number of threads.  usage: reads_io_test <file> <format> <threads>; prints one line per record: "<len>\t<seq>".
#include <cstdio>
#include <cstdlib>
#include "reads_io.h"

int main(int argc, char **argv)
{
	if (argc < 4) return 2;
	const std::string path = argv[1];
	const int format = atoi(argv[2]), threads = atoi(argv[3]);
	auto print = [](const char *seq, size_t len) { printf("%zu\t%.*s\n", len, (int)len, seq); };
	if (threads == 0) return for_each_read_in_file(path, format, print) ? 0 : 1;
	ChunkedReadsFile m;
	if (!m.open(path)) return 3; // gzip'ed or unreadable
	std::vector<std::pair<const char *, size_t>> shown; // records may only be used until the end of their window
	bool ok = m.for_each_read(format, threads, [&](const char *seq, size_t len) { shown.push_back({seq, len}); },
	                          [&]() { for (auto &r : shown) print(r.first, r.second); shown.clear(); });
	return ok ? 0 : 4;
}
