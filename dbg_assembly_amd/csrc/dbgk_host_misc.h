// dbgk_host_misc.h -- part of libdbgk.so's host side (one translation unit: included by dbgk.hip, in this order).
// key-0 links, device memory helpers, timings, the copy / gather bandwidth probes
#pragma once

extern "C" int dbgk_add_polyA(dbgk_handle *h, uint32_t l_link, uint32_t r_link)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->wide || h->kfreq || h->seed) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	Node nd;
	nd.kmer = 0;
	nd.links = (uint64_t)l_link | ((uint64_t)r_link << 32);
	Node *d = nullptr;
	if (hipMalloc(&d, sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
	hipError_t e = hipMemcpyAsync(d, &nd, sizeof(Node), hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) {
		if (h->sharded)
			hipLaunchKernelGGL(k_merge_sharded, dim3(1), dim3(kBlock), 0, h->stream, d, (const unsigned long long *)nullptr, (uint64_t)1, (uint64_t)1, 0,
			                   0, h->geom, h->store, h->table, h->d_ctr);
		else
			hipLaunchKernelGGL(k_merge_nodes, dim3(1), dim3(kBlock), 0, h->stream, d, (uint64_t)1, h->tref(), h->d_ctr);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return hip_fail(e, "add_polyA", __LINE__);
	return DBGK_OK;
}

extern "C" int dbgk_memcpy_d2d(dbgk_handle *h, void *d_dst, const void *d_src, size_t bytes)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// utilities
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_synth_reads_device(dbgk_handle *h, const dbgk_synth_params *p, uint64_t first_read, uint64_t n_reads,
                                       char *d_bases, uint64_t *d_offsets)
{
	if (!h || !p || !d_bases || !d_offsets) return DBGK_ERR_ARG;
	if (p->read_len == 0 || p->read_len > 1024 || p->genome_len < p->read_len) return DBGK_ERR_ARG;
	if ((uintptr_t)d_bases & 15u) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t chunks = (n_reads * (uint64_t)p->read_len + 15) >> 4;
	hipLaunchKernelGGL(k_synth_reads, dim3(grid_for(h, std::max<uint64_t>(chunks, n_reads + 1))), dim3(kBlock), 0, h->stream, *p,
	                   first_read, n_reads, d_bases, d_offsets);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

extern "C" int dbgk_device_malloc(dbgk_handle *h, size_t bytes, void **d_ptr)
{
	if (!h || !d_ptr) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (hipMalloc(d_ptr, bytes ? bytes : 16) != hipSuccess) {
		*d_ptr = nullptr;
		return DBGK_ERR_NOMEM;
	}
	return DBGK_OK;
}

extern "C" int dbgk_device_free(dbgk_handle *h, void *d_ptr)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(h->stream));
	HIPCHK(hipFree(d_ptr));
	return DBGK_OK;
}

extern "C" int dbgk_memcpy_d2h(dbgk_handle *h, void *dst, const void *d_src, size_t bytes)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

extern "C" int dbgk_memcpy_h2d(dbgk_handle *h, void *d_dst, const void *src, size_t bytes)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

extern "C" int dbgk_get_timings(dbgk_handle *h, dbgk_timings *out)
{
	if (!h || !out) return DBGK_ERR_ARG;
	memset(out, 0, sizeof(*out));
	out->mark_ms = h->phase_ms[PH_MARK];
	out->insert_ms = h->phase_ms[PH_INSERT];
	out->partition_ms = h->phase_ms[PH_PARTITION];
	out->build_ms = h->phase_ms[PH_BUILD];
	out->fixup_ms = h->phase_ms[PH_FIXUP];
	out->finalize_ms = h->phase_ms[PH_FINALIZE];
	out->insert_launches = h->insert_launches;
	out->l2_build_wall_ms = h->phase_ms[PH_L2_BUILD_WALL];
	out->partition_launches = h->partition_launches;
	out->uniform_launches = h->uniform_launches;
	out->prefix_launches = h->prefix_launches;
	return DBGK_OK;
}

extern "C" int dbgk_reset_timings(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	for (auto &v : h->phase_ms) v = 0.f;
	h->insert_launches = 0;
	h->partition_launches = 0;
	h->uniform_launches = 0;
	h->prefix_launches = 0;
	return DBGK_OK;
}

// The "measured HBM bandwidth" of the roofline.  A runtime DtoD memcpy reads 4.7-5.4 TB/s on this pool depending on the
// box; the guide's figure for a 16-byte-per-lane copy kernel is 6.3.  So the probe runs its OWN streaming kernels as well -- 16 bytes
// per lane, four loads in flight, 1024-thread persistent workgroups, default and non-temporal policy, one and two workgroups per CU
// (profiles/ubench/hbm_stream.hip is the sweep these shapes come from) -- and reports the BEST rate seen, copy bytes = read + written.
namespace {
typedef uint32_t probe_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(1024) void k_probe_copy(const probe_u32x4 *__restrict__ src, probe_u32x4 *__restrict__ dst, size_t n_vec)
{
	constexpr int U = 4;
	const size_t tile = (size_t)1024 * U, n_tiles = n_vec / tile;
	for (size_t g = blockIdx.x; g < n_tiles; g += gridDim.x) {
		probe_u32x4 v[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			const probe_u32x4 *p = src + g * tile + (size_t)u * 1024 + threadIdx.x;
			v[u] = NT ? __builtin_nontemporal_load(p) : *p;
		}
#pragma unroll
		for (int u = 0; u < U; u++) {
			probe_u32x4 *q = dst + g * tile + (size_t)u * 1024 + threadIdx.x;
			if (NT) __builtin_nontemporal_store(v[u], q); else *q = v[u];
		}
	}
}
} // namespace

static int measure_copy_bandwidth(dbgk_handle *h, size_t bytes, int iters, double *gbps, double *runtime_memcpy);

extern "C" int dbgk_measure_copy_bandwidth(dbgk_handle *h, size_t bytes, int iters, double *gbps)
{
	return measure_copy_bandwidth(h, bytes, iters, gbps, nullptr);
}

extern "C" int dbgk_measure_copy_bandwidth2(dbgk_handle *h, size_t bytes, int iters, double *gbps, double *runtime_memcpy_gbps)
{
	return measure_copy_bandwidth(h, bytes, iters, gbps, runtime_memcpy_gbps);
}

static int measure_copy_bandwidth(dbgk_handle *h, size_t bytes, int iters, double *gbps, double *runtime_memcpy)
{
	if (!h || !gbps || bytes < 65536 || iters < 1) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	bytes &= ~(size_t)65535; // whole tiles of the probe kernels
	void *a = nullptr, *b = nullptr;
	if (hipMalloc(&a, bytes) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&b, bytes) != hipSuccess) {
		(void)hipFree(a);
		return DBGK_ERR_NOMEM;
	}
	hipEvent_t e0, e1;
	hipError_t e = hipEventCreate(&e0);
	if (e == hipSuccess) e = hipEventCreate(&e1);
	if (e == hipSuccess) e = hipMemsetAsync(a, 1, bytes, h->stream);
	double best = 0.0;
	for (int variant = 0; variant < 5 && e == hipSuccess; variant++) {
		auto run = [&]() -> hipError_t {
			const size_t n_vec = bytes / 16;
			switch (variant) {
			case 0: return hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, h->stream);
			case 1: hipLaunchKernelGGL(k_probe_copy<true>, dim3(h->n_cu), dim3(1024), 0, h->stream, (const probe_u32x4 *)a, (probe_u32x4 *)b, n_vec); break;
			case 2: hipLaunchKernelGGL(k_probe_copy<true>, dim3(2 * h->n_cu), dim3(1024), 0, h->stream, (const probe_u32x4 *)a, (probe_u32x4 *)b, n_vec); break;
			case 3: hipLaunchKernelGGL(k_probe_copy<false>, dim3(h->n_cu), dim3(1024), 0, h->stream, (const probe_u32x4 *)a, (probe_u32x4 *)b, n_vec); break;
			default: hipLaunchKernelGGL(k_probe_copy<false>, dim3(2 * h->n_cu), dim3(1024), 0, h->stream, (const probe_u32x4 *)a, (probe_u32x4 *)b, n_vec); break;
			}
			return hipGetLastError();
		};
		e = run(); // warm-up
		if (e == hipSuccess) e = hipEventRecord(e0, h->stream);
		for (int i = 0; i < iters && e == hipSuccess; i++) e = run();
		if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
		float ms = 0.f;
		if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
		const double rate = (e == hipSuccess && ms > 0.f) ? (2.0 * (double)bytes * iters) / (ms * 1e-3) / 1e9 : 0.0; // bytes read + bytes written
		best = std::max(best, rate);
		if (variant == 0 && runtime_memcpy) *runtime_memcpy = rate;
	}
	(void)hipFree(a);
	(void)hipFree(b);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	if (e != hipSuccess) return hip_fail(e, "measure_copy_bandwidth", __LINE__);
	*gbps = best;
	return DBGK_OK;
}

// Random 64-byte gather (the practical ceiling of the engines that touch one random node per k-mer
// occurrence, DIRECT / WIDE-atomic / SEEDIDX).  Four lanes fetch one 64-byte sector each (16 bytes per lane) at a
// pseudo-random sector of a buffer far larger than the caches; every lane group runs its own xorshift stream.
__global__ __launch_bounds__(kBlock) void k_random_gather64(const uint4 *__restrict__ buf, uint64_t n_sectors, uint32_t per_group,
                                                            unsigned long long *__restrict__ sink)
{
	const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	const uint64_t group = tid >> 2;
	const uint32_t part = (uint32_t)tid & 3u;
	uint64_t x = (group + 1u) * 0x9E3779B97F4A7C15ull;
	uint32_t acc = 0;
	for (uint32_t i = 0; i < per_group; i++) {
		x ^= x << 13;
		x ^= x >> 7;
		x ^= x << 17;
		const uint64_t sector = (uint64_t)(((unsigned __int128)x * n_sectors) >> 64);
		typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
		const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(buf) + sector * 4u + part);
		acc ^= v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345677u) atomicAdd(sink, 1ull); // keeps the loads alive
}

extern "C" int dbgk_measure_gather_bandwidth(dbgk_handle *h, size_t bytes, uint64_t n_accesses, double *gbps, double *gaccesses_per_s)
{
	if (!h || !gbps || bytes < (1u << 20) || n_accesses < 1024) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	void *a = nullptr;
	unsigned long long *sink = nullptr;
	if (hipMalloc(&a, bytes) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&sink, 8) != hipSuccess) {
		(void)hipFree(a);
		return DBGK_ERR_NOMEM;
	}
	const uint32_t per_group = 64;
	const uint64_t groups = (n_accesses + per_group - 1) / per_group;
	const uint64_t blocks = (groups * 4 + kBlock - 1) / kBlock;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	hipError_t e = hipEventCreate(&e0);
	if (e == hipSuccess) e = hipEventCreate(&e1);
	if (e == hipSuccess) e = hipMemsetAsync(a, 1, bytes, h->stream);
	if (e == hipSuccess) e = hipMemsetAsync(sink, 0, 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_random_gather64, dim3((unsigned)std::min<uint64_t>(blocks, 1u << 20)), dim3(kBlock), 0, h->stream, (const uint4 *)a, (uint64_t)(bytes >> 6), 4u,
		                   sink); // warm-up (page tables)
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipEventRecord(e0, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_random_gather64, dim3((unsigned)blocks), dim3(kBlock), 0, h->stream, (const uint4 *)a, (uint64_t)(bytes >> 6), per_group, sink);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	float ms = 0.f;
	if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
	(void)hipFree(a);
	(void)hipFree(sink);
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	if (e != hipSuccess) return hip_fail(e, "measure_gather_bandwidth", __LINE__);
	const double done = (double)(blocks * (kBlock / 4)) * per_group;
	*gbps = done * 64.0 / (ms * 1e-3) / 1e9;
	if (gaccesses_per_s) *gaccesses_per_s = done / (ms * 1e-3) / 1e9;
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// several GPUs in one process
// ---------------------------------------------------------------------------------------------
