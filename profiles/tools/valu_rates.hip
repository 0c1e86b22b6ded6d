// valu_rates.hip -- issue rate of the integer VALU instructions the level-1 kernels are made of, on the box it runs on:
// hipcc --offload-arch=gfx950 -O3 profiles/tools/valu_rates.hip -o gpurun_out/valu_rates && gpurun_out/valu_rates
// Every kernel runs ITER iterations of 8 independent chains of one instruction (inline asm, so that the compiler neither merges nor
// reorders them); 16 waves per CU, 4 per SIMD.  Reported: clocks per wave-instruction and SIMD at the clock rate measured with v_add_u32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define K32(NAME, ASM)                                                                                         \
	__global__ __launch_bounds__(1024) void NAME(uint32_t *out, uint32_t s)                                    \
	{                                                                                                          \
		uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = s | 3u;    \
		for (int i = 0; i < ITER; i++) {                                                                       \
			asm volatile(ASM " %0, %0, %1" : "+v"(a0) : "v"(b));                                               \
			asm volatile(ASM " %0, %0, %1" : "+v"(a1) : "v"(b));                                               \
			asm volatile(ASM " %0, %0, %1" : "+v"(a2) : "v"(b));                                               \
			asm volatile(ASM " %0, %0, %1" : "+v"(a3) : "v"(b));                                               \
			asm volatile(ASM " %0, %0, %1" : "+v"(a4) : "v"(b));                                               \
			asm volatile(ASM " %0, %0, %1" : "+v"(a5) : "v"(b));                                               \
			asm volatile(ASM " %0, %0, %1" : "+v"(a6) : "v"(b));                                               \
			asm volatile(ASM " %0, %0, %1" : "+v"(a7) : "v"(b));                                               \
		}                                                                                                      \
		out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                          \
	}
K32(k_add, "v_add_u32")
K32(k_mul_lo, "v_mul_lo_u32")
K32(k_mul_hi, "v_mul_hi_u32")
K32(k_mul_u24, "v_mul_u32_u24")
K32(k_xor, "v_xor_b32")
K32(k_lshl, "v_lshlrev_b32")
#define K64(NAME, STMT)                                                                                        \
	__global__ __launch_bounds__(1024) void NAME(uint32_t *out, uint32_t s)                                    \
	{                                                                                                          \
		uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                \
		uint32_t b = s | 3u, c = s | 5u;                                                                       \
		for (int i = 0; i < ITER; i++) {                                                                       \
			STMT(a0) STMT(a1) STMT(a2) STMT(a3) STMT(a4) STMT(a5) STMT(a6) STMT(a7)                            \
		}                                                                                                      \
		const uint64_t x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                               \
		out[blockIdx.x * 1024 + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32) ^ c;                          \
	}
#define S_MAD64(A) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(A) : "v"(b), "v"(c) : "vcc");
#define S_SHL64(A) asm volatile("v_lshlrev_b64 %0, 13, %0" : "+v"(A));
#define S_SHR64(A) asm volatile("v_lshrrev_b64 %0, 22, %0" : "+v"(A));
#define S_LSHLADD64(A) asm volatile("v_lshl_add_u64 %0, %0, 3, %0" : "+v"(A));
K64(k_mad64, S_MAD64)
K64(k_shl64, S_SHL64)
K64(k_shr64, S_SHR64)
K64(k_lshladd64, S_LSHLADD64)
int main()
{
	uint32_t *d;
	hipMalloc(&d, 1024 * 1024 * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	double base = 0;
#define RUN(K)                                                                                              \
	{                                                                                                       \
		hipLaunchKernelGGL(K, dim3(256), dim3(1024), 0, 0, d, 1u);                                           \
		hipDeviceSynchronize();                                                                              \
		hipEventRecord(e0);                                                                                  \
		hipLaunchKernelGGL(K, dim3(256), dim3(1024), 0, 0, d, 1u);                                           \
		hipEventRecord(e1);                                                                                  \
		hipEventSynchronize(e1);                                                                             \
		float ms;                                                                                            \
		hipEventElapsedTime(&ms, e0, e1);                                                                    \
		const double per = ms * 1e-3 / (4.0 * ITER * 8); /* seconds per wave-instruction and SIMD: 4 waves per SIMD */ \
		if (base == 0) base = per;                                                                           \
		printf("%-14s %8.3f ms  %6.2f clocks per wave-instruction (v_add_u32 = 4)\n", #K, ms, 4.0 * per / base); \
	}
	RUN(k_add) RUN(k_xor) RUN(k_lshl) RUN(k_mul_u24) RUN(k_mul_lo) RUN(k_mul_hi) RUN(k_mad64) RUN(k_shl64) RUN(k_shr64) RUN(k_lshladd64)
	printf("clock if v_add_u32 issues every 4 clocks: %.2f GHz\n", 4.0 / base * 1e-9);
	return 0;
}
