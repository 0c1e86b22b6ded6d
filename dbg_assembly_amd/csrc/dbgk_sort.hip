// dbgk_sort.hip -- device radix sort of (key, links) pairs for the canonical dump
// (dbgk_export_sorted).  Not on the hot path; kept in its own translation unit because the rocPRIM
// templates dominate compile time.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>

extern "C" int dbgk_internal_sort_pairs(uint64_t *d_keys, uint64_t *d_vals, uint64_t n, hipStream_t stream)
{
	if (n == 0) return 0;
	uint64_t *keys_out = nullptr, *vals_out = nullptr;
	void *tmp = nullptr;
	size_t tmp_bytes = 0;
	int rc = 0;
	if (hipMalloc(&keys_out, n * 8) != hipSuccess || hipMalloc(&vals_out, n * 8) != hipSuccess) rc = -5;
	if (rc == 0 &&
	    rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, keys_out, d_vals, vals_out, (size_t)n, 0, 64, stream) != hipSuccess)
		rc = -2;
	if (rc == 0 && hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16) != hipSuccess) rc = -5;
	if (rc == 0 &&
	    rocprim::radix_sort_pairs(tmp, tmp_bytes, d_keys, keys_out, d_vals, vals_out, (size_t)n, 0, 64, stream) != hipSuccess)
		rc = -2;
	if (rc == 0 && hipMemcpyAsync(d_keys, keys_out, n * 8, hipMemcpyDeviceToDevice, stream) != hipSuccess) rc = -2;
	if (rc == 0 && hipMemcpyAsync(d_vals, vals_out, n * 8, hipMemcpyDeviceToDevice, stream) != hipSuccess) rc = -2;
	if (rc == 0 && hipStreamSynchronize(stream) != hipSuccess) rc = -2;
	if (keys_out) (void)hipFree(keys_out);
	if (vals_out) (void)hipFree(vals_out);
	if (tmp) (void)hipFree(tmp);
	return rc;
}
