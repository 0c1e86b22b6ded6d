// Fixed costs of a HIP process on the box (measurement aid): runtime initialisation, a 30 GB device allocation, a 9.6 GB host
// allocation touched page by page (plain malloc against 2 MiB-aligned + MADV_HUGEPAGE), and what the exit of such a process costs
// (measured by the caller around the whole run).   hip_startup <mode>   0 = init only, 1 = + device memory, 2 = + host table 4 KiB
// pages, 3 = + host table huge pages
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
	const int mode = argc > 1 ? atoi(argv[1]) : 0;
	const double t0 = now();
	int n = 0;
	hipGetDeviceCount(&n);
	hipSetDevice(0);
	void *p = nullptr;
	hipMalloc(&p, 1 << 20);
	const double t1 = now();
	void *big = nullptr;
	if (mode >= 1) {
		hipMalloc(&big, 30ull << 30);
		hipMemset(big, 0, 30ull << 30);
		hipDeviceSynchronize();
	}
	const double t2 = now();
	const size_t bytes = 9600000016ull;
	char *host = nullptr;
	if (mode == 2) host = (char *)malloc(bytes);
	if (mode == 3) {
		if (posix_memalign((void **)&host, 2u << 20, bytes)) host = nullptr;
		if (host) madvise(host, bytes, MADV_HUGEPAGE);
	}
	if (host) { // first touch by 8 threads
		std::vector<std::thread> th;
		for (int t = 0; t < 8; t++) th.emplace_back([=]() { memset(host + bytes / 8 * t, 1, bytes / 8); });
		for (auto &x : th) x.join();
	}
	const double t3 = now();
	printf("mode %d: hip init %.3f s, device 30 GB %.3f s, host 9.6 GB touched %.3f s\n", mode, t1 - t0, t2 - t1, t3 - t2);
	fflush(stdout);
	_exit(0);
}
