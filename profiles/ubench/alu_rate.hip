// Instruction issue-rate microbenchmark for gfx950 (measurement aid, not part of the library).
// Each kernel runs a long chain of ONE VALU instruction on 8 independent accumulators per lane;
// reported: wave-instructions per clock per SIMD (1/4 CU), from the wall time at the clock rate read
// back via wall_clock64 vs s_memtime is avoided: we report ns per wave-instruction per SIMD instead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 1 << 17;
constexpr int ACC = 8;

// operands: %0 = 64-bit accumulator (rw), %1 = 32-bit accumulator (rw), %2 %3 = 64-bit inputs, %4 %5 = 32-bit inputs
#define KERNEL(name, ASM)                                                                    \
__global__ __launch_bounds__(256) void name(uint32_t *out, uint32_t seed)                    \
{                                                                                            \
	uint64_t a64[ACC], b64 = ((uint64_t)seed << 32) | 1u, c64 = seed ^ 0x9E3779B97F4A7C15ull; \
	uint32_t a32[ACC], b32 = seed | 1u, c32 = seed ^ 0x9E3779B9u;                            \
	for (int i = 0; i < ACC; i++) { a64[i] = threadIdx.x + i * 77u + seed; a32[i] = (uint32_t)a64[i] * 3u; } \
	for (int it = 0; it < ITER; it++) {                                                      \
		_Pragma("unroll") for (int i = 0; i < ACC; i++)                                      \
			asm volatile(ASM : "+v"(a64[i]), "+v"(a32[i]) : "v"(b64), "v"(c64), "v"(b32), "v"(c32) : "vcc", "s10", "s11"); \
	}                                                                                        \
	uint64_t x = 0;                                                                          \
	for (int i = 0; i < ACC; i++) x ^= a64[i] + a32[i];                                      \
	if (x == 0x12345u) out[threadIdx.x] = (uint32_t)x;                                       \
}

KERNEL(k_add_u32, "v_add_u32 %1, %1, %4")
KERNEL(k_xor_b32, "v_xor_b32 %1, %1, %4")
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %1, %1, %4")
KERNEL(k_mul_hi_u32, "v_mul_hi_u32 %1, %1, %4")
KERNEL(k_mul_u24, "v_mul_u32_u24 %1, %1, %4")
KERNEL(k_mad_u24, "v_mad_u32_u24 %1, %1, %4, %5")
KERNEL(k_alignbit, "v_alignbit_b32 %1, %1, %4, 7")
KERNEL(k_bfe, "v_bfe_u32 %1, %1, 3, 9")
KERNEL(k_lshl_or, "v_lshl_or_b32 %1, %1, 3, %4")
KERNEL(k_add3, "v_add3_u32 %1, %1, %4, %5")
KERNEL(k_cndmask, "v_cndmask_b32 %1, %1, %4, vcc")
KERNEL(k_perm, "v_perm_b32 %1, %1, %4, %5")
KERNEL(k_lshl_b64, "v_lshlrev_b64 %0, 3, %0")
KERNEL(k_lshr_b64, "v_lshrrev_b64 %0, 3, %0")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 3, %2")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %4, %0")
KERNEL(k_cmp_lt_u64, "v_cmp_lt_u64 vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc")
KERNEL(k_cmp_lt_u32, "v_cmp_lt_u32 vcc, %1, %5\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc")
KERNEL(k_add_addc, "v_add_co_u32 %1, vcc, %1, %4\n\tv_addc_co_u32 %1, vcc, %1, %5, vcc")

KERNEL(k_and_b32, "v_and_b32 %1, %1, %4")
KERNEL(k_or_b32, "v_or_b32 %1, %1, %4")
KERNEL(k_lshl_b32, "v_lshlrev_b32 %1, 3, %1")
KERNEL(k_lshr_b32, "v_lshrrev_b32 %1, 3, %1")
KERNEL(k_sub_u32, "v_sub_u32 %1, %1, %4")
KERNEL(k_not_b32, "v_not_b32 %1, %1")
KERNEL(k_mov_b32, "v_mov_b32 %1, %4")
KERNEL(k_min_u32, "v_min_u32 %1, %1, %4")
KERNEL(k_xor_e64, "v_xor_b32_e64 %1, %1, %4")
KERNEL(k_add_e64, "v_add_u32_e64 %1, %1, %4")
KERNEL(k_add_co, "v_add_co_u32 %1, vcc, %1, %4")
KERNEL(k_cnd_e64, "v_cndmask_b32_e64 %1, %1, %4, s[10:11]")
KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %1, %4")
KERNEL(k_cmp_u64, "v_cmp_lt_u64 vcc, %0, %2")
KERNEL(k_or3, "v_or3_b32 %1, %1, %4, %5")
KERNEL(k_and_or, "v_and_or_b32 %1, %1, %4, %5")
KERNEL(k_xad, "v_xad_u32 %1, %1, %4, %5")
KERNEL(k_lshl_add_u32, "v_lshl_add_u32 %1, %1, 3, %4")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %1, %1, %4")
KERNEL(k_ashr_i64, "v_ashrrev_i64 %0, 3, %0")
KERNEL(k_mul_f32, "v_mul_f32 %1, %1, %4")
KERNEL(k_fma_f32, "v_fma_f32 %1, %1, %4, %5")
KERNEL(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %2, %3")
KERNEL(k_fma_f64, "v_fma_f64 %0, %0, %2, %3")
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %1, %1")
KERNEL(k_mul_f64, "v_mul_f64 %0, %0, %2")
KERNEL(k_cnd_e64_vcc, "v_cndmask_b32_e64 %1, %1, %4, vcc")
KERNEL(k_cmp_cnd_e32, "v_cmp_lt_u32 vcc, %1, %5\n\tv_cndmask_b32 %1, %1, %4, vcc")
KERNEL(k_cmp_cnd_e64, "v_cmp_lt_u32_e64 s[10:11], %1, %5\n\tv_cndmask_b32_e64 %1, %1, %4, s[10:11]")
KERNEL(k_cnd_e32_other, "v_cndmask_b32 %1, %4, %5, vcc")
KERNEL(k_cnd_e32_chain, "v_cndmask_b32 %1, %4, %1, vcc")
KERNEL(k_bitop3, "v_bitop3_b32 %1, %1, %4, %5 bitop3:0x6c")
KERNEL(k_and_const, "v_and_b32 %1, 0x3fffffff, %1")
KERNEL(k_and_inline, "v_and_b32 %1, 15, %1")
KERNEL(k_xor_m1, "v_xor_b32 %1, -1, %1")
KERNEL(k_lshr_inline, "v_lshrrev_b32 %1, 2, %1")
KERNEL(k_sub_co, "v_sub_co_u32 %1, vcc, %1, %4\n\tv_subb_co_u32 %1, vcc, %1, %5, vcc")
KERNEL(k_readlane, "v_readlane_b32 s10, %1, 3\n\tv_add_u32 %1, s10, %1")
KERNEL(k_mov_b64, "v_mov_b64 %0, %2")
KERNEL(k_pk_mov, "v_pk_mov_b32 %0, %2, %3")
KERNEL(k_cmp_3cnd, "v_cmp_lt_u32 vcc, %1, %5\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %1, %4, %1, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc")
KERNEL(k_smov_cnd, "s_mov_b64 vcc, s[10:11]\n\tv_cndmask_b32 %1, %1, %4, vcc")
KERNEL(k_cmp_xor_cnd, "v_cmp_lt_u32 vcc, %1, %5\n\tv_xor_b32 %1, %1, %4\n\tv_cndmask_b32 %1, %1, %4, vcc")
KERNEL(k_cmp_e64_3cnd, "v_cmp_lt_u32_e64 s[10:11], %1, %5\n\tv_cndmask_b32_e64 %1, %1, %4, s[10:11]\n\tv_cndmask_b32_e64 %1, %4, %1, s[10:11]\n\tv_cndmask_b32_e64 %1, %1, %5, s[10:11]")
KERNEL(k_and_swapped, "v_and_b32 %1, %4, %1")
KERNEL(k_and_e64, "v_and_b32_e64 %1, %1, %4")
KERNEL(k_or_swapped, "v_or_b32 %1, %4, %1")
KERNEL(k_lshl_by_reg, "v_lshlrev_b32 %1, %4, %1")

struct K { const char *name; void (*fn)(uint32_t *, uint32_t); int instr_per_step; };

int main()
{
	uint32_t *d;
	CHECK(hipMalloc(&d, 4096));
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	const double mhz = prop.clockRate / 1000.0;
	printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, mhz);
	std::vector<K> ks = {
		{"v_add_u32", k_add_u32, 1}, {"v_xor_b32", k_xor_b32, 1}, {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
		{"v_mul_u32_u24", k_mul_u24, 1}, {"v_mad_u32_u24", k_mad_u24, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_bfe_u32", k_bfe, 1},
		{"v_lshl_or_b32", k_lshl_or, 1}, {"v_add3_u32", k_add3, 1}, {"v_cndmask_b32", k_cndmask, 1}, {"v_perm_b32", k_perm, 1},
		{"v_lshlrev_b64", k_lshl_b64, 1}, {"v_lshrrev_b64", k_lshr_b64, 1}, {"v_lshl_add_u64", k_lshl_add_u64, 1},
		{"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_cmp_lt_u64+addc", k_cmp_lt_u64, 2}, {"v_cmp_lt_u32+addc", k_cmp_lt_u32, 2},
		{"add_addc", k_add_addc, 2}, {"and_b32", k_and_b32, 1}, {"or_b32", k_or_b32, 1}, {"lshl_b32", k_lshl_b32, 1}, {"lshr_b32", k_lshr_b32, 1}, {"sub_u32", k_sub_u32, 1}, {"not_b32", k_not_b32, 1}, {"mov_b32", k_mov_b32, 1}, {"min_u32", k_min_u32, 1}, {"xor_e64", k_xor_e64, 1}, {"add_e64", k_add_e64, 1}, {"add_co", k_add_co, 1}, {"cnd_e64", k_cnd_e64, 1}, {"cmp_u32", k_cmp_u32, 1}, {"cmp_u64", k_cmp_u64, 1}, {"or3", k_or3, 1}, {"and_or", k_and_or, 1}, {"xad", k_xad, 1}, {"lshl_add_u32", k_lshl_add_u32, 1}, {"pk_add_u16", k_pk_add_u16, 1}, {"ashr_i64", k_ashr_i64, 1}, {"mul_f32", k_mul_f32, 1}, {"fma_f32", k_fma_f32, 1}, {"pk_fma_f32", k_pk_fma_f32, 1}, {"fma_f64", k_fma_f64, 1}, {"cvt_f32_u32", k_cvt_f32_u32, 1}, {"mul_f64", k_mul_f64, 1},
		{"cnd_e64_vcc", k_cnd_e64_vcc, 1}, {"cmp+cnd_e32", k_cmp_cnd_e32, 2}, {"cmp_e64+cnd_e64", k_cmp_cnd_e64, 2}, {"cnd_e32 dst!=src", k_cnd_e32_other, 1},
		{"cnd_e32 src1=dst", k_cnd_e32_chain, 1}, {"bitop3", k_bitop3, 1}, {"and literal", k_and_const, 1}, {"and inline", k_and_inline, 1}, {"xor -1", k_xor_m1, 1},
		{"lshr inline", k_lshr_inline, 1}, {"sub_co+subb", k_sub_co, 2}, {"readlane+add", k_readlane, 2}, {"mov_b64", k_mov_b64, 1}, {"pk_mov_b32", k_pk_mov, 1},
		{"cmp+3 cnd_e32", k_cmp_3cnd, 4}, {"s_mov vcc+cnd_e32", k_smov_cnd, 1}, {"cmp+xor+cnd_e32", k_cmp_xor_cnd, 3}, {"cmp_e64+3 cnd_e64", k_cmp_e64_3cnd, 4},
		{"and swapped", k_and_swapped, 1}, {"and e64", k_and_e64, 1}, {"or swapped", k_or_swapped, 1}, {"lshl by reg", k_lshl_by_reg, 1},
	};
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	// 4 waves per SIMD resident: 256-thread blocks (4 waves -> one per SIMD), 4 blocks per CU
	const int blocks = cus * 4;
	for (auto &k : ks) {
		for (int rep = 0; rep < 4; rep++) {
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 12345u + rep);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			{
				// per SIMD: 4 waves, each ITER*ACC steps
				const double wave_instr = 4.0 * ITER * ACC * k.instr_per_step;
				const double clocks = ms * 1e-3 * mhz * 1e6;
				if (rep == 0) printf("%-22s", k.name);
				printf(" %6.2f", clocks / wave_instr);
				if (rep == 3) printf("   clocks per wave-instruction per SIMD (4 runs)\n");
			}
		}
	}
	return 0;
}
