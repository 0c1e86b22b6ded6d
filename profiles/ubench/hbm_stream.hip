// What does a streaming kernel reach on this box?  (Measurement aid, not part of the library.)
// read-only / write-only / copy over 4 GiB buffers, for: bytes per lane and instruction (8, 16), cache policy (default,
// non-temporal), workgroup size (256, 512, 1024), loads in flight per lane (1, 4, 8), persistent grid (workgroups per CU).
// The library's record streams are read 8 bytes per lane with non-temporal loads by 512- and 1024-thread persistent
// workgroups; this probe says what that shape costs against the best one, and gives the roofline's "measured" denominator.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T, bool NT>
__device__ __forceinline__ T ld(const T *p)
{
	if (NT) return __builtin_nontemporal_load(p);
	return *p;
}
template <typename T, bool NT>
__device__ __forceinline__ void st(T *p, T v)
{
	if (NT) __builtin_nontemporal_store(v, p);
	else *p = v;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ uint32_t fold(T v);
template <> __device__ __forceinline__ uint32_t fold<u32x4>(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <> __device__ __forceinline__ uint32_t fold<u32x2>(u32x2 v) { return v.x ^ v.y; }

// tile = THREADS * U elements; workgroups stride over tiles; lane reads elements tile*T*U + u*THREADS + tid (coalesced)
template <typename T, bool NT, int THREADS, int U>
__global__ __launch_bounds__(THREADS) void k_read(const T *__restrict__ p, size_t n, uint32_t *out)
{
	uint32_t acc = 0;
	const size_t tile = (size_t)THREADS * U, n_tiles = n / tile;
	for (size_t g = blockIdx.x; g < n_tiles; g += gridDim.x) {
		T v[U];
#pragma unroll
		for (int u = 0; u < U; u++) v[u] = ld<T, NT>(p + g * tile + (size_t)u * THREADS + threadIdx.x);
#pragma unroll
		for (int u = 0; u < U; u++) acc ^= fold<T>(v[u]);
	}
	if (acc == 0x12345678u) out[0] = acc;
}
template <typename T, bool NT, int THREADS, int U>
__global__ __launch_bounds__(THREADS) void k_write(T *__restrict__ p, size_t n, uint32_t x)
{
	const size_t tile = (size_t)THREADS * U, n_tiles = n / tile;
	T v;
	for (int i = 0; i < (int)(sizeof(T) / 4); i++) v[i] = x + i;
	for (size_t g = blockIdx.x; g < n_tiles; g += gridDim.x) {
#pragma unroll
		for (int u = 0; u < U; u++) st<T, NT>(p + g * tile + (size_t)u * THREADS + threadIdx.x, v);
	}
}
template <typename T, bool NT, int THREADS, int U>
__global__ __launch_bounds__(THREADS) void k_copy(const T *__restrict__ p, T *__restrict__ q, size_t n)
{
	const size_t tile = (size_t)THREADS * U, n_tiles = n / tile;
	for (size_t g = blockIdx.x; g < n_tiles; g += gridDim.x) {
		T v[U];
#pragma unroll
		for (int u = 0; u < U; u++) v[u] = ld<T, NT>(p + g * tile + (size_t)u * THREADS + threadIdx.x);
#pragma unroll
		for (int u = 0; u < U; u++) st<T, NT>(q + g * tile + (size_t)u * THREADS + threadIdx.x, v[u]);
	}
}

static hipEvent_t e0, e1;
template <typename F>
static float timed(F f, int reps = 5)
{
	f(); // warm-up
	float best = 1e30f;
	for (int r = 0; r < reps; r++) {
		hipEventRecord(e0);
		f();
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	return best;
}

template <typename T, bool NT, int THREADS, int U>
static void run(const char *tname, void *a, void *b, size_t bytes, uint32_t *out, int per_cu, int cus)
{
	const size_t n = bytes / sizeof(T);
	const int grid = per_cu * cus;
	const float r = timed([&] { hipLaunchKernelGGL((k_read<T, NT, THREADS, U>), dim3(grid), dim3(THREADS), 0, 0, (const T *)a, n, out); });
	const float w = timed([&] { hipLaunchKernelGGL((k_write<T, NT, THREADS, U>), dim3(grid), dim3(THREADS), 0, 0, (T *)b, n, 7u); });
	const float c = timed([&] { hipLaunchKernelGGL((k_copy<T, NT, THREADS, U>), dim3(grid), dim3(THREADS), 0, 0, (const T *)a, (T *)b, n); });
	printf("%-6s %-3s threads %4d  in flight %2d  wg/CU %2d   read %7.1f GB/s   write %7.1f GB/s   copy %7.1f GB/s (read+write bytes)\n", tname, NT ? "nt" : "def",
	       THREADS, U, per_cu, bytes / (r * 1e-3) / 1e9, bytes / (w * 1e-3) / 1e9, 2.0 * bytes / (c * 1e-3) / 1e9);
	fflush(stdout);
}

int main(int argc, char **argv)
{
	const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 4096ull) << 20;
	void *a, *b;
	uint32_t *out;
	CHECK(hipMalloc(&a, bytes));
	CHECK(hipMalloc(&b, bytes));
	CHECK(hipMalloc(&out, 64));
	CHECK(hipMemset(a, 1, bytes));
	CHECK(hipMemset(b, 2, bytes));
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("%s, %d CUs, buffers of %zu MiB\n", prop.name, cus, bytes >> 20);
	{ // the runtime's own copy
		const float ms = timed([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
		printf("hipMemcpyAsync DtoD: %7.1f GB/s (read+write bytes)\n", 2.0 * bytes / (ms * 1e-3) / 1e9);
	}
	for (int per_cu : {1, 2, 4, 8}) {
		run<u32x4, false, 256, 4>("16 B", a, b, bytes, out, per_cu, cus);
		run<u32x4, true, 256, 4>("16 B", a, b, bytes, out, per_cu, cus);
	}
	run<u32x4, false, 256, 1>("16 B", a, b, bytes, out, 8, cus);
	run<u32x4, true, 256, 8>("16 B", a, b, bytes, out, 4, cus);
	run<u32x4, true, 512, 4>("16 B", a, b, bytes, out, 2, cus);
	run<u32x4, true, 512, 8>("16 B", a, b, bytes, out, 1, cus);
	run<u32x4, true, 1024, 4>("16 B", a, b, bytes, out, 1, cus);
	run<u32x4, true, 1024, 4>("16 B", a, b, bytes, out, 2, cus);
	run<u32x4, false, 1024, 4>("16 B", a, b, bytes, out, 2, cus);
	// the shapes of the library's record streams: 8 bytes per lane
	run<u32x2, true, 512, 16>("8 B", a, b, bytes, out, 1, cus);   // level 2: 512 threads, 16 records per thread, 1 workgroup per CU
	run<u32x2, true, 512, 16>("8 B", a, b, bytes, out, 2, cus);
	run<u32x2, true, 1024, 4>("8 B", a, b, bytes, out, 2, cus);   // build: 1024 threads, 4 records per thread, 2 workgroups per CU
	run<u32x2, true, 1024, 4>("8 B", a, b, bytes, out, 1, cus);
	run<u32x2, false, 1024, 4>("8 B", a, b, bytes, out, 2, cus);
	run<u32x2, true, 1024, 8>("8 B", a, b, bytes, out, 2, cus);
	run<u32x2, true, 256, 8>("8 B", a, b, bytes, out, 8, cus);
	CHECK(hipDeviceSynchronize());
	return 0;
}
