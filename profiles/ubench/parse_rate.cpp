// How fast can the host turn a one-line FASTA file into batches?  (measurement aid)  parse_rate <file> -- for 1, 2, 4, 8, 16 threads:
// the windowed reader alone (records only counted), and with the bytes of every window copied out by the same number of threads.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "reads_io.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
	if (argc < 2) return 2;
	const std::string path = argv[1];
	{
		uint64_t n = 0, total = 0;
		const double t0 = now();
		for_each_read_in_file(path, 2, [&](const char *, size_t len) { n++; total += len; });
		printf("sequential reader: %.3f s (%llu reads, %llu bases)\n", now() - t0, (unsigned long long)n, (unsigned long long)total);
	}
	std::vector<char> out(400u << 20);
	for (int threads : {1, 2, 4, 8, 16}) {
		for (int with_copy = 0; with_copy < 2; with_copy++) {
			ChunkedReadsFile f;
			if (!f.open(path)) return 3;
			std::vector<std::pair<const char *, uint32_t>> noted;
			std::vector<uint64_t> offs(1, 0);
			uint64_t n = 0;
			const double t0 = now();
			f.for_each_read(2, threads, [&](const char *s, size_t len) { noted.push_back({s, (uint32_t)len}); offs.push_back(offs.back() + len); n++; },
			                [&]() {
				                if (with_copy) {
					                std::vector<std::thread> th;
					                const size_t per = (noted.size() + threads - 1) / threads;
					                for (int t = 0; t < threads; t++)
						                th.emplace_back([&, t]() {
							                const size_t a = std::min(noted.size(), per * t), b = std::min(noted.size(), a + per);
							                for (size_t i = a; i < b; i++) memcpy(out.data() + (offs[i] % (300u << 20)), noted[i].first, noted[i].second);
						                });
					                for (auto &x : th) x.join();
				                }
				                noted.clear();
				                offs.assign(1, 0);
			                });
			printf("windowed reader, %2d threads%s: %.3f s (%llu reads)\n", threads, with_copy ? " + copy" : "       ", now() - t0, (unsigned long long)n);
		}
	}
	return 0;
}
