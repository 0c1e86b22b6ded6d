// dbgk_host_export.h -- part of libdbgk.so's host side (one translation unit: included by dbgk.hip, in this order).
// dbgk_finalize, statistics, device-to-host copies of the table (plain, pipelined, occupied nodes only), sorted dump, digest, link pass
#pragma once

extern "C" int dbgk_finalize(dbgk_handle *h, dbgk_stats *out)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (h->wpart && !h->finalized) {
		if (h->wmulti) { // shards / passes: the current pass is completed here, all passes must have run
			if (h->wpass_open) {
				rc = wide_end_pass(h);
				if (rc) return rc;
			}
			if (h->wpasses_done != h->wgeom.n_passes) {
				g_last_error = "dbgk_finalize: this WIDE handle reads its input in several passes (dbgk_wide_pass_info) and not all of them have run";
				return DBGK_ERR_STATE;
			}
			if (!h->wbuilt) {
				rc = wide_finish_records(h);
				if (rc) return rc;
			}
		} else if (!h->wbuilt && h->pending_kmers > 0) {
			rc = wide_build_from_records(h);
			if (rc) return rc;
		}
		rc = wide_ensure_zero(h); // nothing was pushed at all
		if (rc) return rc;
	}
	if (h->part && !h->part_built && !h->finalized) {
		if (h->sharded && !h->exchanged) {
			g_last_error = "sharded handle: exchange the level-1 buckets (dbgk_shard_buffers) and call dbgk_shard_mark_exchanged first";
			return DBGK_ERR_STATE;
		}
		if (!(h->incr && h->pending_kmers == 0 && !h->sharded)) { // (a flushed handle with nothing new: the table is complete)
			rc = build_from_records(h);
			if (rc) return rc;
			h->pending_kmers = 0;
		}
	}
	const bool kf_tracked = h->kfreq && h->part && h->kf_blocks; // direct blocks: the summary is kept while the table is written
	if (h->kfreq && !kf_tracked) {
		unsigned long long res[2];
		rc = kfreq_summary(h, 0, h->n_counts, res);
		if (rc) return rc;
		h->kf_distinct = res[0];
		h->kf_sum = res[1];
	}
	rc = read_counters(h);
	if (rc) return rc;
	if (kf_tracked) {
		h->kf_distinct = h->h_ctr->kf_nonzero;
		h->kf_sum = h->h_ctr->kf_sum;
	}
	h->finalized = true;
	if (out) fill_stats(h, out);
	if (h->kfreq) {
		if (out) {
			out->count = h->kf_distinct;
			out->count_conflict = 0;
			out->table_slots = h->n_counts;
		}
		if (h->h_ctr->error & 2u) return DBGK_ERR_CAPACITY; // through PARTITION: far more distinct k-mers than expected_kmers / 2
		return DBGK_OK;
	}
	if (h->h_ctr->error & 1u) return DBGK_ERR_TABLE_FULL;
	if (h->h_ctr->error & 2u) return DBGK_ERR_CAPACITY; // PARTITION overflow stores exhausted (expected_kmers too small)
	if (h->h_ctr->n_new + 1 > h->tslots) return DBGK_ERR_TABLE_FULL; // no free slot left for the key-0 node
	return DBGK_OK;
}

extern "C" int dbgk_refresh_stats(dbgk_handle *h, dbgk_stats *out)
{
	if (!h || !out) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	rc = read_counters(h);
	if (rc) return rc;
	fill_stats(h, out);
	if (h->kfreq && h->finalized) { // distinct canonical k-mers, as dbgk_finalize reports them
		out->count = h->kf_distinct;
		out->count_conflict = 0;
		out->table_slots = h->n_counts;
	}
	return (h->h_ctr->error & 1u) ? DBGK_ERR_TABLE_FULL : DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// results
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_build_flags_ctr(const Node *__restrict__ nodes, uint64_t size,
                                                            const Counters *__restrict__ ctr, uint8_t *__restrict__ flags)
{
	const uint64_t polyA_slot = ctr->polyA_slot;
	const uint64_t n_bytes = size / 8 + 1;
	const uint64_t stride = (uint64_t)gridDim.x * kBlock;
	for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_bytes; b += stride) {
		uint32_t byte = 0;
#pragma unroll
		for (uint32_t j = 0; j < 8; j++) {
			const uint64_t i = b * 8 + j;
			if (i < size && (nodes[i].kmer != 0ull || i == polyA_slot)) byte |= 0x80u >> j;
		}
		flags[b] = (uint8_t)byte;
	}
}

// A large device-to-host copy into ORDINARY (pageable, malloc()ed) host memory -- the host KmerSet must be free()-able by the
// consumer.  hipMemcpy into pageable memory stages through the runtime's own bounce buffer on one thread (~20 GB/s); here the
// device fills pinned slices at the link's rate and several host threads move them on (their first touch also spreads the
// page faults of the fresh allocation).  Everything queued on the handle's stream before the call is complete on return.
static int d2h_pipelined(dbgk_handle *h, void *dst, const void *d_src, size_t bytes)
{
	constexpr size_t kSlice = 32ull << 20;
	constexpr int kBuffers = 8;
	static const int n_threads = getenv("DBGK_EXPORT_THREADS") ? std::max(1, atoi(getenv("DBGK_EXPORT_THREADS"))) : 6;
	if (bytes < 8 * kSlice || DBGK_EXPERIMENT_ENV("DBGK_EXPORT_PLAIN")) {
		HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		return DBGK_OK;
	}
	while (h->d2h_stage.size() < (size_t)kBuffers) {
		void *p = nullptr;
		hipEvent_t e = nullptr;
		if (hipHostMalloc(&p, kSlice, hipHostMallocDefault) != hipSuccess) return DBGK_ERR_NOMEM;
		h->d2h_stage.push_back(p);
		HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		h->d2h_ev.push_back(e);
	}
	const size_t n_slices = (bytes + kSlice - 1) / kSlice;
	std::vector<std::atomic<int>> issued(n_slices), done(n_slices);
	for (size_t i = 0; i < n_slices; i++) { issued[i].store(0); done[i].store(0); }
	std::atomic<int> failed{0};
	std::vector<std::thread> workers;
	for (int w = 0; w < n_threads; w++)
		workers.emplace_back([&, w]() {
			if (hipSetDevice(h->device) != hipSuccess) { failed.store(1); return; }
			for (size_t i = (size_t)w; i < n_slices; i += (size_t)n_threads) {
				while (!issued[i].load(std::memory_order_acquire)) {
					if (failed.load()) return;
					std::this_thread::yield();
				}
				const int b = (int)(i % kBuffers);
				if (hipEventSynchronize(h->d2h_ev[b]) != hipSuccess) { failed.store(1); return; }
				const size_t off = i * kSlice, len = std::min(kSlice, bytes - off);
				memcpy(static_cast<char *>(dst) + off, h->d2h_stage[b], len);
				done[i].store(1, std::memory_order_release);
			}
		});
	int rc = DBGK_OK;
	for (size_t i = 0; i < n_slices && rc == DBGK_OK; i++) {
		if (i >= (size_t)kBuffers)
			while (!done[i - kBuffers].load(std::memory_order_acquire)) { // its buffer is free again
				if (failed.load()) { rc = DBGK_ERR_HIP; break; }
				std::this_thread::yield();
			}
		if (rc) break;
		const int b = (int)(i % kBuffers);
		const size_t off = i * kSlice, len = std::min(kSlice, bytes - off);
		if (hipMemcpyAsync(h->d2h_stage[b], static_cast<const char *>(d_src) + off, len, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
		    hipEventRecord(h->d2h_ev[b], h->stream) != hipSuccess) {
			rc = DBGK_ERR_HIP;
			break;
		}
		issued[i].store(1, std::memory_order_release);
	}
	if (rc) failed.store(1);
	for (auto &t : workers) t.join();
	if (failed.load() && rc == DBGK_OK) rc = DBGK_ERR_HIP;
	if (rc) return hip_fail(hipGetLastError(), "d2h_pipelined", __LINE__);
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

// ---- the occupied nodes only ----------------------------------------------------------------------------------------------------
// A host table at the reference's load (-i: 0.3 - 0.6 of the slots hold a node) is mostly zeros: the copy above moves all of it over
// the link.  Here the device packs the occupied nodes, in slot order, into one stream (k_compact_nodes), only that stream and the
// occupancy bits cross the link, and the host threads that used to memcpy() the slices now lay the nodes out at their slots from the
// bits (zeros in between).  Same bytes in `array` and `nul_flag` as the plain copy (tests/test_gpu_parity.py compares the two).
constexpr uint32_t kCompactSpan = 4096; // slots per wavefront

__global__ __launch_bounds__(256) void k_flag_block_counts(const uint32_t *__restrict__ flags32, uint64_t n_dwords, uint64_t n_spans,
                                                           uint32_t *__restrict__ counts)
{
	const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
	const uint32_t lane = threadIdx.x & 63u;
	if (wave >= n_spans) return;
	uint32_t c = 0;
#pragma unroll
	for (uint32_t j = 0; j < kCompactSpan / 32 / 64; j++) {
		const uint64_t d = wave * (kCompactSpan / 32) + j * 64 + lane;
		if (d < n_dwords) c += (uint32_t)__popc(flags32[d]);
	}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
	if (lane == 0) counts[wave] = c;
}

__global__ __launch_bounds__(256) void k_compact_nodes(const Node *__restrict__ nodes, uint64_t size, const Counters *__restrict__ ctr,
                                                       const uint64_t *__restrict__ span_first, uint64_t n_spans, Node *__restrict__ out)
{
	const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
	const uint32_t lane = threadIdx.x & 63u;
	if (wave >= n_spans) return;
	const uint64_t polyA_slot = ctr->polyA_slot;
	uint64_t at = span_first[wave];
	const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(nodes);
	ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(out);
	for (uint32_t it = 0; it < kCompactSpan / 64; it++) {
		const uint64_t i = wave * kCompactSpan + it * 64 + lane;
		ulonglong2 nd = make_ulonglong2(0ull, 0ull);
		bool occ = false;
		if (i < size) {
			nd = src[i];
			occ = nd.x != 0ull || i == polyA_slot; // what k_build_flags_ctr calls occupied
		}
		const uint64_t m = __ballot(occ);
		if (occ) dst[at + (uint64_t)__popcll(m & ((1ull << lane) - 1ull))] = nd;
		at += (uint64_t)__popcll(m);
	}
}

// array[0, host_size) and nul_flag[0, host_size / 8 + 1) from the image T whose occupancy bits are d_flags (padded to whole dwords,
// the padding zero).  DBGK_ERR_STATE: "use the plain copy" (no memory for the stream, or the counts disagree).
// Page-locked memory: the staging buffers of the handle's batches where they exist -- idle once a table is finalized, and
// page-locking fresh memory for one copy costs about what the copy costs -- else the handle's own export buffers (d2h_stage).
static int d2h_compact(dbgk_handle *h, dbgk_node *array, uint8_t *nul_flag, const TableRef &T, const uint8_t *d_flags, uint64_t n_occ)
{
	constexpr size_t kSlice = 8ull << 20, kOwnSlice = 32ull << 20; // (d2h_stage holds pieces of 32 MiB: four slices each)
	constexpr size_t kMinBuffers = 16, kOwnBuffers = 8, kMaxBuffers = 48;
	static const int n_threads = getenv("DBGK_EXPORT_THREADS") ? std::max(1, atoi(getenv("DBGK_EXPORT_THREADS"))) : 12;
	const uint64_t size = T.size, n_spans = (size + kCompactSpan - 1) / kCompactSpan, n_flag_bytes = size / 8 + 1;
	static const bool lap_wanted = getenv("DBGK_TIMINGS") != nullptr;
	auto clock_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double laps[5] = {0, 0, 0, 0, 0}, lap_t = clock_s();
	auto lap = [&](int i) { const double t = clock_s(); laps[i] += t - lap_t; lap_t = t; };
	std::atomic<uint64_t> wait_link_us{0}, wait_host_us{0};
	Node *d_stream = nullptr;
	uint32_t *d_counts = nullptr;
	uint64_t *d_first = nullptr;
	auto cleanup = [&]() {
		for (void *p : {(void *)d_stream, (void *)d_counts, (void *)d_first})
			if (p) (void)hipFree(p);
	};
	if (hipMalloc(&d_stream, (n_occ + 1) * sizeof(Node)) != hipSuccess || hipMalloc(&d_counts, n_spans * 4) != hipSuccess ||
	    hipMalloc(&d_first, n_spans * 8) != hipSuccess) {
		(void)hipGetLastError();
		cleanup();
		return DBGK_ERR_STATE;
	}
	// page-locked pieces: [the occupancy bits] + the slices the stream passes through
	std::vector<char *> bufs;
	uint8_t *bits_pinned = nullptr;
	const size_t flag_room = (size_t)((n_flag_bytes + kSlice - 1) / kSlice) * kSlice;
	for (StageSlot &sl : h->slots) {
		if (!sl.h_bases || sl.acquired) continue;
		if (sl.busy && hipEventQuery(sl.done) != hipSuccess) continue; // (a batch still on its way: not after dbgk_finalize)
		size_t off = 0;
		if (!bits_pinned && h->cap_bases >= flag_room + kSlice) {
			bits_pinned = reinterpret_cast<uint8_t *>(sl.h_bases);
			off = flag_room;
		}
		for (; off + kSlice <= h->cap_bases && bufs.size() < kMaxBuffers; off += kSlice) bufs.push_back(sl.h_bases + off);
	}
	if (bufs.size() < kMinBuffers) {
		while (h->d2h_stage.size() < kOwnBuffers) {
			void *p = nullptr;
			if (hipHostMalloc(&p, kOwnSlice, hipHostMallocDefault) != hipSuccess) { cleanup(); return DBGK_ERR_NOMEM; }
			h->d2h_stage.push_back(p);
		}
		for (void *p : h->d2h_stage)
			for (size_t off = 0; off + kSlice <= kOwnSlice && bufs.size() < kMaxBuffers; off += kSlice) bufs.push_back(static_cast<char *>(p) + off);
	}
	while (h->d2h_ev.size() < bufs.size()) {
		hipEvent_t ev = nullptr;
		if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { cleanup(); return DBGK_ERR_HIP; }
		h->d2h_ev.push_back(ev);
	}
	const size_t n_bufs = bufs.size();
	lap(0);
	const unsigned int span_grid = (unsigned int)((n_spans + 3) / 4);
	hipLaunchKernelGGL(k_flag_block_counts, dim3(span_grid), dim3(256), 0, h->stream, reinterpret_cast<const uint32_t *>(d_flags),
	                   (n_flag_bytes + 3) / 4, n_spans, d_counts);
	std::vector<uint32_t> counts(n_spans);
	std::vector<uint64_t> first(n_spans + 1);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(counts.data(), d_counts, n_spans * 4, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) { cleanup(); return hip_fail(e, "d2h_compact(counts)", __LINE__); }
	first[0] = 0;
	for (uint64_t b = 0; b < n_spans; b++) first[b + 1] = first[b] + counts[b];
	if (first[n_spans] != n_occ) { // (never: the counters and the bits describe the same table)
		cleanup();
		return DBGK_ERR_STATE;
	}
	lap(1);
	e = hipMemcpyAsync(d_first, first.data(), n_spans * 8, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(bits_pinned ? bits_pinned : nul_flag, d_flags, n_flag_bytes, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_compact_nodes, dim3(span_grid), dim3(256), 0, h->stream, T.nodes, size, h->d_ctr, d_first, n_spans, d_stream);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream); // the bits are on the host: the threads below read them
	if (e != hipSuccess) { cleanup(); return hip_fail(e, "d2h_compact(bits)", __LINE__); }
	lap(2);
	// slices of the stream small enough that every thread gets several (the zeros between the nodes are written by the thread
	// whose nodes come next, so the slot ranges follow the ranks)
	const uint64_t per = std::min<uint64_t>(kSlice / sizeof(Node), std::max<uint64_t>(1u << 16, n_occ / (uint64_t)(4 * n_threads) + 1));
	const size_t n_slices = (size_t)((n_occ + per - 1) / per);
	const uint8_t *bits_src = bits_pinned ? bits_pinned : nul_flag;
	auto occupied = [&](uint64_t slot) { return (bits_src[slot >> 3] >> (7u - (uint32_t)(slot & 7u))) & 1u; };
	auto slot_of_rank = [&](uint64_t r) -> uint64_t { // the slot of the r-th occupied node (r < n_occ)
		const uint64_t b = (uint64_t)(std::upper_bound(first.begin(), first.end(), r) - first.begin()) - 1;
		uint64_t slot = b * kCompactSpan, left = r - first[b];
		for (;; slot++)
			if (occupied(slot)) {
				if (left == 0) return slot;
				left--;
			}
	};
	if (DBGK_EXPERIMENT_ENV("DBGK_EXPORT_PROBE")) { // (measurements) the link alone: the same copies with nobody reading the buffers
		const double t0 = clock_s();
		for (size_t i = 0; i < n_slices; i++) {
			const uint64_t r0 = (uint64_t)i * per, r1 = std::min(n_occ, r0 + per);
			(void)hipMemcpyAsync(bufs[i % n_bufs], d_stream + r0, (size_t)(r1 - r0) * sizeof(Node), hipMemcpyDeviceToHost, h->stream);
		}
		(void)hipStreamSynchronize(h->stream);
		const double t1 = clock_s();
		fprintf(stderr, "dbgk export probe: %zu copies of %.1f MB back to back %.4f s (%.1f GB/s)\n", n_slices, (double)per * 16e-6, t1 - t0,
		        (double)n_occ * 16e-9 / (t1 - t0));
		lap_t = clock_s();
	}
	std::vector<std::atomic<int>> issued(n_slices), done(n_slices);
	for (size_t i = 0; i < n_slices; i++) { issued[i].store(0); done[i].store(0); }
	std::atomic<int> failed{0};
	dbgk_node *dst = array;
	std::vector<std::thread> workers;
	for (int w = 0; w < n_threads; w++)
		workers.emplace_back([&, w]() {
			if (hipSetDevice(h->device) != hipSuccess) { failed.store(1); return; }
			if (bits_pinned) { // this thread's share of the bits -> the caller's nul_flag
				const uint64_t chunk = (n_flag_bytes + (uint64_t)n_threads - 1) / (uint64_t)n_threads;
				const uint64_t a = std::min(n_flag_bytes, chunk * (uint64_t)w), b = std::min(n_flag_bytes, a + chunk);
				memcpy(nul_flag + a, bits_pinned + a, (size_t)(b - a));
			}
			for (size_t i = (size_t)w; i < n_slices; i += (size_t)n_threads) {
				const uint64_t r0 = (uint64_t)i * per, r1 = std::min(n_occ, r0 + per);
				uint64_t slot = i == 0 ? 0 : slot_of_rank(r0);
				const uint64_t slot_end = i + 1 == n_slices ? size : slot_of_rank(r1);
				while (!issued[i].load(std::memory_order_acquire)) {
					if (failed.load()) return;
					std::this_thread::yield();
				}
				const size_t b = i % n_bufs;
				const double t_w = lap_wanted ? clock_s() : 0;
				if (hipEventSynchronize(h->d2h_ev[b]) != hipSuccess) { failed.store(1); return; }
				if (lap_wanted) wait_link_us += (uint64_t)((clock_s() - t_w) * 1e6);
				const dbgk_node *const src0 = reinterpret_cast<const dbgk_node *>(bufs[b]);
				const dbgk_node *src = src0;
				const dbgk_node zero{0, 0, 0};
				for (; slot < slot_end && (slot & 63u); slot++) dst[slot] = occupied(slot) ? *src++ : zero;
				// 64 slots at a time: zeros and the nodes the bits name (first slot = top bit) are put together in a buffer of one KiB
				// and leave with non-temporal stores -- the table is written once and not read here: no line is fetched for ownership
				const bool stream_out = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
				for (; slot + 64 <= slot_end; slot += 64) {
					uint64_t bits;
					memcpy(&bits, bits_src + (slot >> 3), 8);
					bits = __builtin_bswap64(bits);
					alignas(64) dbgk_node group[64];
					memset(static_cast<void *>(group), 0, sizeof group);
					while (bits) {
						const int j = __builtin_clzll(bits);
						group[j] = *src++;
						bits &= ~(0x8000000000000000ull >> j);
					}
					if (stream_out) {
#pragma unroll
						for (int q = 0; q < 64; q++)
							_mm_stream_si128(reinterpret_cast<__m128i *>(dst + slot + q), _mm_load_si128(reinterpret_cast<const __m128i *>(group + q)));
					} else {
						memcpy(static_cast<void *>(dst + slot), group, sizeof group);
					}
				}
				_mm_sfence();
				for (; slot < slot_end; slot++) dst[slot] = occupied(slot) ? *src++ : zero;
				if ((uint64_t)(src - src0) != r1 - r0) failed.store(2); // (never)
				done[i].store(1, std::memory_order_release);
			}
		});
	int rc = DBGK_OK;
	for (size_t i = 0; i < n_slices && rc == DBGK_OK; i++) {
		const double t_w = lap_wanted ? clock_s() : 0;
		if (i >= n_bufs)
			while (!done[i - n_bufs].load(std::memory_order_acquire)) { // its buffer is free again
				if (failed.load()) { rc = DBGK_ERR_HIP; break; }
				std::this_thread::yield();
			}
		if (lap_wanted) wait_host_us += (uint64_t)((clock_s() - t_w) * 1e6);
		if (rc) break;
		const size_t b = i % n_bufs;
		const uint64_t r0 = (uint64_t)i * per, r1 = std::min(n_occ, r0 + per);
		if (hipMemcpyAsync(bufs[b], d_stream + r0, (size_t)(r1 - r0) * sizeof(Node), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
		    hipEventRecord(h->d2h_ev[b], h->stream) != hipSuccess) {
			rc = DBGK_ERR_HIP;
			break;
		}
		issued[i].store(1, std::memory_order_release);
	}
	if (rc) failed.store(1);
	for (auto &t : workers) t.join();
	if (failed.load() && rc == DBGK_OK) rc = DBGK_ERR_HIP;
	if (rc == DBGK_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = DBGK_ERR_HIP;
	lap(3);
	cleanup();
	lap(4);
	if (rc) return hip_fail(hipGetLastError(), "d2h_compact", __LINE__);
	if (lap_wanted)
		fprintf(stderr, "dbgk export, occupied nodes only (s): buffers %.4f, counts %.4f, stream on the device + bits to the host %.4f, stream to the host and into "
		        "the slots %.4f (%zu slices of %.1f MB through %zu buffers%s; the %d threads waited %.4f for the link in all, the issuing thread %.4f for a free "
		        "buffer), release %.4f\n",
		        laps[0], laps[1], laps[2], laps[3], n_slices, (double)per * 16e-6, n_bufs, bits_pinned ? " of the batch staging" : "", n_threads,
		        (double)wait_link_us.load() * 1e-6, (double)wait_host_us.load() * 1e-6, laps[4]);
	return DBGK_OK;
}

// what the link pass of an export returns (all optional)
struct LinkOutputs {
	int32_t cutoff = 0;
	uint16_t *klink = nullptr;
	uint8_t *del_flag = nullptr;
	uint64_t *tips = nullptr, *branches = nullptr;
	uint64_t tip_cap = 0, branch_cap = 0;
	uint64_t *n_tips = nullptr, *n_branches = nullptr;
	dbgk_link_stats *stats = nullptr;
};

static int export_host_table_impl(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag, const LinkOutputs *LO);

extern "C" int dbgk_export_host_table(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag)
{
	return export_host_table_impl(h, host_size, array, nul_flag, nullptr);
}

extern "C" int dbgk_export_host_table_links(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag, int32_t kmer_freq_cutoff,
                                            uint16_t *klink, uint8_t *del_flag, uint64_t *tip_nodes, uint64_t tip_capacity, uint64_t *n_tips,
                                            uint64_t *branch_nodes, uint64_t branch_capacity, uint64_t *n_branches, dbgk_link_stats *stats)
{
	if (!klink || !del_flag || !n_tips || !n_branches) return DBGK_ERR_ARG;
	if (h && h->sharded) return DBGK_ERR_STATE; // slot numbers are those of ONE table: export the shards, assemble, then scan (or use one handle)
	LinkOutputs LO;
	LO.cutoff = kmer_freq_cutoff;
	LO.klink = klink;
	LO.del_flag = del_flag;
	LO.tips = tip_nodes;
	LO.branches = branch_nodes;
	LO.tip_cap = tip_nodes ? tip_capacity : 0;
	LO.branch_cap = branch_nodes ? branch_capacity : 0;
	LO.n_tips = n_tips;
	LO.n_branches = n_branches;
	LO.stats = stats;
	return export_host_table_impl(h, host_size, array, nul_flag, &LO);
}

// the link pass on the host-layout image T (key-0 node placed, ctr->polyA_slot set)
static int run_link_pass(dbgk_handle *h, const TableRef &T, const LinkOutputs &LO)
{
	const uint64_t n_blocks = (T.size + kLinkChunk - 1) / kLinkChunk;
	uint16_t *d_klink = nullptr;
	uint8_t *d_del = nullptr;
	unsigned long long *d_stats = nullptr, *d_base = nullptr, *d_tips = nullptr, *d_branches = nullptr;
	uint32_t *d_counts = nullptr;
	auto cleanup = [&]() {
		for (void *p : {(void *)d_klink, (void *)d_del, (void *)d_stats, (void *)d_base, (void *)d_tips, (void *)d_branches, (void *)d_counts})
			if (p) (void)hipFree(p);
	};
	if (hipMalloc(&d_klink, T.size * 2) != hipSuccess || hipMalloc(&d_del, T.size / 8 + 1) != hipSuccess || hipMalloc(&d_stats, 261 * 8) != hipSuccess ||
	    hipMalloc(&d_counts, n_blocks * 8) != hipSuccess || hipMalloc(&d_base, n_blocks * 16) != hipSuccess) {
		cleanup();
		return DBGK_ERR_NOMEM;
	}
	hipError_t e = hipMemsetAsync(d_stats, 0, 261 * 8, h->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_del, 0, T.size / 8 + 1, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_kmer_links<0>, dim3((unsigned)n_blocks), dim3(kBlock), 0, h->stream, T.nodes, T.size, h->d_ctr, (int)LO.cutoff, d_klink, d_del, d_stats,
		                   d_counts, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (unsigned long long *)nullptr);
		e = hipGetLastError();
	}
	std::vector<uint32_t> counts(n_blocks * 2);
	std::vector<unsigned long long> base(n_blocks * 2);
	unsigned long long res[261];
	if (e == hipSuccess) e = hipMemcpyAsync(counts.data(), d_counts, n_blocks * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(res, d_stats, sizeof res, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(LO.klink, d_klink, T.size * 2, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(LO.del_flag, d_del, T.size / 8 + 1, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) {
		cleanup();
		return hip_fail(e, "export_host_table_links", __LINE__);
	}
	unsigned long long nt = 0, nb = 0;
	for (uint64_t b = 0; b < n_blocks; b++) {
		base[2 * b] = nt;
		base[2 * b + 1] = nb;
		nt += counts[2 * b];
		nb += counts[2 * b + 1];
	}
	*LO.n_tips = nt;
	*LO.n_branches = nb;
	if (LO.stats) {
		for (int i = 0; i < 256; i++) LO.stats->depth_stat[i] = (int64_t)res[i];
		LO.stats->total_nodes = (int64_t)res[256];
		LO.stats->deleted_lowfreq = (int64_t)res[257];
		LO.stats->linear_nodes = (int64_t)res[258];
		LO.stats->tip_nodes = (int64_t)res[259];
		LO.stats->branch_nodes = (int64_t)res[260];
	}
	int rc = DBGK_OK;
	if ((LO.tips || LO.branches) && (nt || nb)) {
		if ((LO.tips && nt > LO.tip_cap) || (LO.branches && nb > LO.branch_cap)) {
			rc = DBGK_ERR_CAPACITY; // *n_tips / *n_branches say what is needed
		} else if (hipMalloc(&d_tips, (nt ? nt : 1) * 8) != hipSuccess || hipMalloc(&d_branches, (nb ? nb : 1) * 8) != hipSuccess) {
			rc = DBGK_ERR_NOMEM;
		} else {
			e = hipMemcpyAsync(d_base, base.data(), n_blocks * 16, hipMemcpyHostToDevice, h->stream);
			if (e == hipSuccess) {
				hipLaunchKernelGGL(k_kmer_links<1>, dim3((unsigned)n_blocks), dim3(kBlock), 0, h->stream, T.nodes, T.size, h->d_ctr, (int)LO.cutoff, d_klink, d_del,
				                   d_stats, d_counts, d_base, d_tips, d_branches);
				e = hipGetLastError();
			}
			if (e == hipSuccess && LO.tips && nt) e = hipMemcpyAsync(LO.tips, d_tips, nt * 8, hipMemcpyDeviceToHost, h->stream);
			if (e == hipSuccess && LO.branches && nb) e = hipMemcpyAsync(LO.branches, d_branches, nb * 8, hipMemcpyDeviceToHost, h->stream);
			if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
			if (e != hipSuccess) rc = hip_fail(e, "export_host_table_links(lists)", __LINE__);
		}
	}
	cleanup();
	return rc;
}

static int export_host_table_impl(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag, const LinkOutputs *LO)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !array || !nul_flag || host_size < 3) return DBGK_ERR_ARG;
	if (!h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	if (h->sharded) { // the shard's slice of the global table: slots [slot_lo, slot_hi), key-0 node not placed
		if (host_size != h->tslots) return DBGK_ERR_ARG;
		uint8_t *d_fl = nullptr;
		if (hipMalloc(&d_fl, host_size / 8 + 1) != hipSuccess) return DBGK_ERR_NOMEM;
		hipLaunchKernelGGL(k_build_flags_ctr, dim3(grid_for(h, host_size / 8 + 1)), dim3(kBlock), 0, h->stream, h->table, h->tslots,
		                   h->d_ctr, d_fl);
		hipError_t es = hipGetLastError();
		if (es == hipSuccess) es = hipMemcpyAsync(array, h->table, host_size * sizeof(Node), hipMemcpyDeviceToHost, h->stream);
		if (es == hipSuccess) es = hipMemcpyAsync(nul_flag, d_fl, host_size / 8 + 1, hipMemcpyDeviceToHost, h->stream);
		if (es == hipSuccess) es = hipStreamSynchronize(h->stream);
		(void)hipFree(d_fl);
		if (es != hipSuccess) return hip_fail(es, "export_host_table(shard)", __LINE__);
		return DBGK_OK;
	}
	if (h->h_ctr->n_new + 1 > host_size) return DBGK_ERR_TABLE_FULL;

	TableRef T = h->tref();
	Node *tmp = nullptr;
	uint8_t *d_flags = nullptr;
	auto cleanup = [&]() {
		if (tmp) (void)hipFree(tmp);
		if (d_flags) (void)hipFree(d_flags);
	};
	if (host_size != h->size) {
		if (hipMalloc(&tmp, host_size * sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
		T = TableRef{tmp, host_size, make_mod_magic(host_size)};
		if (hipMemsetAsync(tmp, 0, host_size * sizeof(Node), h->stream) != hipSuccess) { cleanup(); return DBGK_ERR_HIP; }
		hipLaunchKernelGGL(k_rehash, dim3(grid_for(h, h->size)), dim3(kBlock), 0, h->stream, h->table, h->size, T, h->d_ctr);
	}
	const uint64_t flag_bytes = host_size / 8 + 1, flag_alloc = (flag_bytes + 11) & ~7ull; // whole dwords for k_flag_block_counts
	if (hipMalloc(&d_flags, flag_alloc) != hipSuccess) { cleanup(); return DBGK_ERR_NOMEM; }
	if (hipMemsetAsync(d_flags, 0, flag_alloc, h->stream) != hipSuccess) { cleanup(); return DBGK_ERR_HIP; }
	hipLaunchKernelGGL(k_place_polyA, dim3(1), dim3(64), 0, h->stream, T, h->d_ctr);
	hipLaunchKernelGGL(k_build_flags_ctr, dim3(grid_for(h, host_size / 8 + 1)), dim3(kBlock), 0, h->stream, T.nodes, T.size,
	                   h->d_ctr, d_flags);
	hipError_t e = hipGetLastError();
	int copy_rc = DBGK_OK;
	static const bool lap_wanted = getenv("DBGK_TIMINGS") != nullptr;
	auto clock_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	if (lap_wanted && e == hipSuccess) e = hipStreamSynchronize(h->stream);
	const double t_a = clock_s();
	// large tables: the occupied nodes only (DBGK_EXPORT_FULL=1: every slot over the link, as before round 4)
	const bool full_copy = dbgk_hook("export_full") != nullptr; // (read at every call: the tests compare the two)
	bool bits_copied = false;
	if (e == hipSuccess) {
		copy_rc = DBGK_ERR_STATE;
		if (!full_copy && host_size * sizeof(Node) >= (dbgk_hook("export_compact_min") ? strtoull(dbgk_hook("export_compact_min"), nullptr, 10) : (256ull << 20))) copy_rc = d2h_compact(h, array, nul_flag, T, d_flags, h->h_ctr->n_new + 1);
		bits_copied = copy_rc == DBGK_OK;
		if (copy_rc == DBGK_ERR_STATE) copy_rc = d2h_pipelined(h, array, T.nodes, host_size * sizeof(Node));
	}
	const double t_b = clock_s();
	if (e == hipSuccess && copy_rc == DBGK_OK && !bits_copied)
		e = hipMemcpyAsync(nul_flag, d_flags, host_size / 8 + 1, hipMemcpyDeviceToHost, h->stream);
	int link_rc = copy_rc;
	if (e == hipSuccess && LO && copy_rc == DBGK_OK) link_rc = run_link_pass(h, T, *LO); // on the very image that is being copied out
	if (lap_wanted) fprintf(stderr, "dbgk export (s): node copy %.4f (%.1f GB/s)\n", t_b - t_a, (double)host_size * sizeof(Node) / (t_b - t_a) * 1e-9);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_unplace_polyA, dim3(1), dim3(64), 0, h->stream, T, h->d_ctr);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	cleanup();
	if (e != hipSuccess) return hip_fail(e, "export_host_table", __LINE__);
	if (h->h_ctr->error & 1u) return DBGK_ERR_TABLE_FULL;
	return link_rc;
}

extern "C" int dbgk_export_sorted(dbgk_handle *h, dbgk_node *out, uint64_t capacity, uint64_t *n_out)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !out || !n_out) return DBGK_ERR_ARG;
	if (!h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t n = h->h_ctr->n_new; // non-zero keys
	const uint64_t z = (h->sharded && h->geom.rank != 0) ? 0 : 1; // the key-0 node is reported by shard 0 only
	*n_out = n + z;
	if (capacity < n + z) return DBGK_ERR_CAPACITY;
	if (z) { // key 0 sorts first
		out[0].kmer = 0;
		out[0].l_link = (uint32_t)(h->h_ctr->polyA_links & 0xFFFFFFFFu);
		out[0].r_link = (uint32_t)(h->h_ctr->polyA_links >> 32);
	}
	if (n == 0) return DBGK_OK;

	uint64_t *d_keys = nullptr, *d_links = nullptr;
	unsigned long long *d_cursor = nullptr;
	auto cleanup = [&]() {
		if (d_keys) (void)hipFree(d_keys);
		if (d_links) (void)hipFree(d_links);
		if (d_cursor) (void)hipFree(d_cursor);
	};
	if (hipMalloc(&d_keys, n * 8) != hipSuccess || hipMalloc(&d_links, n * 8) != hipSuccess ||
	    hipMalloc(&d_cursor, 8) != hipSuccess) {
		cleanup();
		return DBGK_ERR_NOMEM;
	}
	hipError_t e = hipMemsetAsync(d_cursor, 0, 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_compact, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, d_keys, d_links,
		                   d_cursor, n);
		e = hipGetLastError();
	}
	unsigned long long found = 0;
	if (e == hipSuccess) e = hipMemcpyAsync(&found, d_cursor, 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) { cleanup(); return hip_fail(e, "export_sorted/compact", __LINE__); }
	if (found != n) {
		cleanup();
		g_last_error = "export_sorted: occupied slots != counted keys";
		return DBGK_ERR_STATE;
	}
	rc = dbgk_internal_sort_pairs(d_keys, d_links, n, h->stream);
	if (rc != DBGK_OK) { cleanup(); return rc; }
	std::vector<uint64_t> hk(n), hl(n);
	e = hipMemcpyAsync(hk.data(), d_keys, n * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(hl.data(), d_links, n * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	cleanup();
	if (e != hipSuccess) return hip_fail(e, "export_sorted/copy", __LINE__);
	for (uint64_t i = 0; i < n; i++) {
		out[i + z].kmer = hk[i];
		out[i + z].l_link = (uint32_t)(hl[i] & 0xFFFFFFFFu);
		out[i + z].r_link = (uint32_t)(hl[i] >> 32);
	}
	return DBGK_OK;
}

extern "C" int dbgk_export_first_seen_order(dbgk_handle *h, dbgk_node *out, uint64_t *first_pos, uint64_t capacity, uint64_t *n_out)
{
	if (!h || !out || !first_pos || !n_out) return DBGK_ERR_ARG;
	if (!h->finalized || !h->track) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t n = h->h_ctr->n_new;
	*n_out = n;
	if (capacity < n) return DBGK_ERR_CAPACITY;
	if (n == 0) return DBGK_OK;
	uint64_t *d_pos = nullptr, *d_slot = nullptr;
	unsigned long long *d_cursor = nullptr;
	Node *d_nodes = nullptr;
	auto cleanup = [&]() {
		for (void *p : {(void *)d_pos, (void *)d_slot, (void *)d_cursor, (void *)d_nodes})
			if (p) (void)hipFree(p);
	};
	if (hipMalloc(&d_pos, n * 8) != hipSuccess || hipMalloc(&d_slot, n * 8) != hipSuccess || hipMalloc(&d_cursor, 8) != hipSuccess ||
	    hipMalloc(&d_nodes, n * sizeof(Node)) != hipSuccess) {
		cleanup();
		return DBGK_ERR_NOMEM;
	}
	hipError_t e = hipMemsetAsync(d_cursor, 0, 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_compact_order, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->first_pos, h->tslots, d_pos,
		                   d_slot, d_cursor, n);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) { cleanup(); return hip_fail(e, "export_first_seen_order/compact", __LINE__); }
	rc = dbgk_internal_sort_pairs(d_pos, d_slot, n, h->stream); // positions are unique per key: a total order
	if (rc != DBGK_OK) { cleanup(); return rc; }
	hipLaunchKernelGGL(k_gather_nodes, dim3(grid_for(h, n)), dim3(kBlock), 0, h->stream, h->table, d_slot, n, d_nodes);
	e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(out, d_nodes, n * sizeof(Node), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(first_pos, d_pos, n * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	cleanup();
	if (e != hipSuccess) return hip_fail(e, "export_first_seen_order", __LINE__);
	return DBGK_OK;
}

extern "C" int dbgk_digest(dbgk_handle *h, uint64_t *digest)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !digest) return DBGK_ERR_ARG;
	if (!h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long *d_out = nullptr;
	if (hipMalloc(&d_out, 16) != hipSuccess) return DBGK_ERR_NOMEM;
	unsigned long long res[2] = {0, 0};
	hipError_t e = hipMemsetAsync(d_out, 0, 16, h->stream);
	if (e == hipSuccess) {
		if (h->wide)
			hipLaunchKernelGGL(k_wide_digest, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, h->wside, d_out);
		else
			hipLaunchKernelGGL(k_digest, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, d_out);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(res, d_out, 16, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_out);
	if (e != hipSuccess) return hip_fail(e, "digest", __LINE__);
	*digest = res[0] + ((h->sharded && h->shard_rank != 0) ? 0ull : node_digest(0ull, h->h_ctr->polyA_links));
	return DBGK_OK;
}

extern "C" int dbgk_link_stats_device(dbgk_handle *h, int32_t cutoff, dbgk_link_stats *out)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !out) return DBGK_ERR_ARG;
	if (!h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long *d_out = nullptr;
	const size_t bytes = 261 * sizeof(unsigned long long);
	if (hipMalloc(&d_out, bytes) != hipSuccess) return DBGK_ERR_NOMEM;
	unsigned long long res[261];
	hipError_t e = hipMemsetAsync(d_out, 0, bytes, h->stream);
	if (e == hipSuccess) {
		if (h->wide)
			hipLaunchKernelGGL(k_wide_link_stats, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, h->wside, (int)cutoff,
			                   (uint64_t)h->h_ctr->polyA_links, (h->sharded && h->shard_rank != 0) ? 0 : 1, d_out);
		else
			hipLaunchKernelGGL(k_link_stats, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, (int)cutoff,
			                   (uint64_t)h->h_ctr->polyA_links, (h->sharded && h->shard_rank != 0) ? 0 : 1, d_out);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(res, d_out, bytes, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_out);
	if (e != hipSuccess) return hip_fail(e, "link_stats", __LINE__);
	for (int i = 0; i < 256; i++) out->depth_stat[i] = (int64_t)res[i];
	out->total_nodes = (int64_t)res[256];
	out->deleted_lowfreq = (int64_t)res[257];
	out->linear_nodes = (int64_t)res[258];
	out->tip_nodes = (int64_t)res[259];
	out->branch_nodes = (int64_t)res[260];
	return DBGK_OK;
}
