// Does data a kernel has just WRITTEN get served from the Infinity Cache (256 MiB, memory side) when another kernel
// reads it right afterwards?  (Measurement aid, not part of the library.)  For buffer sizes X: kernel W writes X bytes,
// kernel R reads them; "hot" = R right after W; "cold" = a 2 GiB stream runs in between (evicts everything).
// Also "interleaved": W(i+1) of another buffer runs between W(i) and R(i), the pattern of level 2 -> region build.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_write(uint4 *p, size_t n, uint32_t v)
{
	const size_t stride = (size_t)gridDim.x * 256;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = make_uint4(v, (uint32_t)i, v ^ 7u, 1u);
}
__global__ __launch_bounds__(256) void k_read(const uint4 *p, size_t n, uint32_t *out)
{
	const size_t stride = (size_t)gridDim.x * 256;
	uint32_t acc = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
		const uint4 v = p[i];
		acc ^= v.x + v.y + v.z + v.w;
	}
	if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
	const size_t big = 2ull << 30;
	uint4 *a, *b, *flush;
	uint32_t *out;
	CHECK(hipMalloc(&a, 1ull << 30));
	CHECK(hipMalloc(&b, 1ull << 30));
	CHECK(hipMalloc(&flush, big));
	CHECK(hipMalloc(&out, 64));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const int grid = 256 * 8;
	auto timed_read = [&](uint4 *p, size_t bytes, float &ms) {
		hipEventRecord(e0);
		hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, p, bytes / 16, out);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		hipEventElapsedTime(&ms, e0, e1);
	};
	for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 512, 1024}) {
		const size_t bytes = mb << 20;
		float hot = 0, cold = 0, inter = 0, ms;
		for (int rep = 0; rep < 5; rep++) {
			hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, bytes / 16, (uint32_t)rep);
			timed_read(a, bytes, ms);
			if (rep) hot += ms;
			hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, bytes / 16, (uint32_t)rep + 9u);
			hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, flush, big / 16, 3u);
			timed_read(a, bytes, ms);
			if (rep) cold += ms;
			hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, bytes / 16, (uint32_t)rep + 5u);
			hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, bytes / 16, (uint32_t)rep + 6u);
			timed_read(a, bytes, ms);
			if (rep) inter += ms;
		}
		printf("%5zu MiB  read after write: %7.1f GB/s   after a 2 GiB stream: %7.1f GB/s   after writing another buffer of the same size: %7.1f GB/s\n",
		       mb, bytes / (hot / 4 * 1e-3) / 1e9, bytes / (cold / 4 * 1e-3) / 1e9, bytes / (inter / 4 * 1e-3) / 1e9);
	}
	CHECK(hipDeviceSynchronize());
	return 0;
}
