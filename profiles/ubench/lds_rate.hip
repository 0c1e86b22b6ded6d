// Rate of LDS operations per CU on gfx950 (measurement aid, not part of the library): what a 64-lane LDS instruction costs
// when its addresses are random slots of a 4096-entry table (the region image of k_build_regions), by kind of operation:
// 64-bit read, 64-bit compare-swap with return, 64-bit add with / without return, 32-bit add with / without return, 64-bit write.
// Every thread issues OPS operations, four independent ones in flight; 1024 threads, 2 workgroups per CU, all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kSlots = 4096, kThreads = 1024, kOps = 4096;

__device__ __forceinline__ uint32_t rnd(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 16; }

template <int KIND, int STRIDE>   // STRIDE 0: random slots; 1: lane i -> slot base + i (conflict-free)
__global__ __launch_bounds__(kThreads, 8) void k_rate(unsigned long long *out)
{
	__shared__ unsigned long long tab[kSlots];
	for (int i = threadIdx.x; i < kSlots; i += kThreads) tab[i] = 0;
	__syncthreads();
	uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
	unsigned long long acc = 0;
	uint32_t *tab32 = reinterpret_cast<uint32_t *>(tab);
	for (int it = 0; it < kOps / 4; it++) {
		uint32_t a[4];
#pragma unroll
		for (int u = 0; u < 4; u++) a[u] = STRIDE ? ((rnd(s) & ~63u) + (threadIdx.x & 63u)) & (kSlots - 1) : rnd(s) & (kSlots - 1);
		unsigned long long r[4] = {0, 0, 0, 0};
#pragma unroll
		for (int u = 0; u < 4; u++) {
			if (KIND == 0) r[u] = tab[a[u]];
			if (KIND == 1) r[u] = atomicCAS(&tab[a[u]], 0ull, (unsigned long long)a[u] + 1ull);
			if (KIND == 2) r[u] = atomicAdd(&tab[a[u]], 1ull);
			if (KIND == 3) atomicAdd(&tab[a[u]], 1ull);                       // result unused: ds_add_u64
			if (KIND == 4) r[u] = atomicAdd(&tab32[a[u]], 1u);
			if (KIND == 5) atomicAdd(&tab32[a[u]], 1u);                       // ds_add_u32
			if (KIND == 6) tab[a[u]] = s;
			if (KIND == 7) r[u] = tab32[a[u]];
		}
#pragma unroll
		for (int u = 0; u < 4; u++) acc += r[u];
	}
	__syncthreads();
	if (acc == 0x123456789ull) out[0] = acc + tab[threadIdx.x];
}

template <int KIND, int STRIDE>
static int run(const char *name, unsigned long long *out, int n_cu)
{
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	float best = 1e9f;
	for (int rep = 0; rep < 4; rep++) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_rate<KIND, STRIDE>), dim3(n_cu * 2), dim3(kThreads), 0, 0, out);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (rep && ms < best) best = ms;
	}
	const double lane_ops_per_cu = 2.0 * kThreads * kOps;
	printf("%-34s %-8s %8.3f ms  %6.2f lane-ops per ns and CU  (%.1f ns per wave instruction)\n", name, STRIDE ? "linear" : "random", best,
	       lane_ops_per_cu / (best * 1e6), best * 1e6 / (lane_ops_per_cu / 64.0));
	return 0;
}

int main()
{
	unsigned long long *out;
	CHECK(hipMalloc(&out, 4096 * 8));
	hipDeviceProp_t p;
	CHECK(hipGetDeviceProperties(&p, 0));
	const int n_cu = p.multiProcessorCount;
	printf("%s, %d CUs\n", p.name, n_cu);
#define BOTH(K, NAME) if (run<K, 0>(NAME, out, n_cu) || run<K, 1>(NAME, out, n_cu)) return 1
	BOTH(0, "ds_read_b64");
	BOTH(7, "ds_read_b32");
	BOTH(6, "ds_write_b64");
	BOTH(1, "ds_cmpst_rtn_b64");
	BOTH(2, "ds_add_rtn_u64");
	BOTH(3, "ds_add_u64 (no return)");
	BOTH(4, "ds_add_rtn_u32");
	BOTH(5, "ds_add_u32 (no return)");
	return 0;
}
