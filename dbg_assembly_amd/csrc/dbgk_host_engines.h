// dbgk_host_engines.h -- part of libdbgk.so's host side (one translation unit: included by dbgk.hip, in this order).
// SEEDIDX / WIDE / KFREQ exports, node partition + merge (hash-owner flow), shard_* (slot-range flow), WIDE passes
#pragma once

// ---------------------------------------------------------------------------------------------
// SEEDIDX exports
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_seed_export_sorted(dbgk_handle *h, dbgk_node *out, uint64_t capacity, uint64_t *n_out)
{
	if (!h || !out || !n_out) return DBGK_ERR_ARG;
	if (!h->seed || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t n = h->h_ctr->n_new;
	const uint64_t z = h->h_ctr->polyA_links ? 1 : 0;
	*n_out = n + z;
	if (capacity < n + z) return DBGK_ERR_CAPACITY;
	if (z) {
		const uint64_t w = seed_payload_out(h->h_ctr->polyA_links);
		out[0].kmer = 0;
		out[0].l_link = (uint32_t)w;
		out[0].r_link = (uint32_t)(w >> 32);
	}
	if (n == 0) return DBGK_OK;
	uint64_t *d_keys = nullptr, *d_links = nullptr;
	unsigned long long *d_cursor = nullptr;
	auto cleanup = [&]() {
		for (void *p : {(void *)d_keys, (void *)d_links, (void *)d_cursor})
			if (p) (void)hipFree(p);
	};
	if (hipMalloc(&d_keys, n * 8) != hipSuccess || hipMalloc(&d_links, n * 8) != hipSuccess || hipMalloc(&d_cursor, 8) != hipSuccess) {
		cleanup();
		return DBGK_ERR_NOMEM;
	}
	hipError_t e = hipMemsetAsync(d_cursor, 0, 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_compact, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, d_keys, d_links, d_cursor, n);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) { cleanup(); return hip_fail(e, "seed_export_sorted/compact", __LINE__); }
	rc = dbgk_internal_sort_pairs(d_keys, d_links, n, h->stream);
	if (rc != DBGK_OK) { cleanup(); return rc; }
	std::vector<uint64_t> hk(n), hl(n);
	e = hipMemcpyAsync(hk.data(), d_keys, n * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(hl.data(), d_links, n * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	cleanup();
	if (e != hipSuccess) return hip_fail(e, "seed_export_sorted/copy", __LINE__);
	for (uint64_t i = 0; i < n; i++) {
		const uint64_t w = seed_payload_out(hl[i]);
		out[i + z].kmer = hk[i];
		out[i + z].l_link = (uint32_t)w;
		out[i + z].r_link = (uint32_t)(w >> 32);
	}
	return DBGK_OK;
}

extern "C" int dbgk_seed_export_host_table(dbgk_handle *h, uint64_t host_size, dbgk_node *array, uint8_t *nul_flag)
{
	if (!h || !array || !nul_flag || host_size < 3) return DBGK_ERR_ARG;
	if (!h->seed || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	if (h->h_ctr->n_new + 1 > host_size) return DBGK_ERR_TABLE_FULL;
	Node *tmp = nullptr;
	uint8_t *d_flags = nullptr;
	if (hipMalloc(&tmp, host_size * sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&d_flags, host_size / 8 + 1) != hipSuccess) {
		(void)hipFree(tmp);
		return DBGK_ERR_NOMEM;
	}
	TableRef T{tmp, host_size, make_mod_magic(host_size)};
	hipError_t e = hipMemsetAsync(tmp, 0, host_size * sizeof(Node), h->stream);
	if (e == hipSuccess) {
		// always through a copy: the payload words are converted in place to the reference's bit-field
		hipLaunchKernelGGL(k_rehash, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, T, h->d_ctr,
		                   (const unsigned long long *)nullptr, (unsigned long long *)nullptr);
		hipLaunchKernelGGL(k_seed_convert, dim3(grid_for(h, host_size)), dim3(kBlock), 0, h->stream, tmp, host_size);
		hipLaunchKernelGGL(k_seed_place_key0, dim3(1), dim3(64), 0, h->stream, T, h->d_ctr);
		hipLaunchKernelGGL(k_build_flags_ctr, dim3(grid_for(h, host_size / 8 + 1)), dim3(kBlock), 0, h->stream, tmp, host_size, h->d_ctr, d_flags);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(array, tmp, host_size * sizeof(Node), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(nul_flag, d_flags, host_size / 8 + 1, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(tmp);
	(void)hipFree(d_flags);
	if (e != hipSuccess) return hip_fail(e, "seed_export_host_table", __LINE__);
	return (h->h_ctr->error & 1u) ? DBGK_ERR_TABLE_FULL : DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// WIDE exports (128-bit keys, include/dbgk_wide.h)
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_wide_export_sorted(dbgk_handle *h, dbgk_node32 *out, uint64_t capacity, uint64_t *n_out)
{
	if (!h || !out || !n_out) return DBGK_ERR_ARG;
	if (!h->wide || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t n = h->h_ctr->n_new; // non-zero keys (main + side table)
	const uint64_t z = (h->sharded && h->shard_rank != 0) ? 0 : 1; // of a sharded table only shard 0 reports the key-0 node
	*n_out = n + z;
	if (capacity < n + z) return DBGK_ERR_CAPACITY;
	if (z) out[0] = dbgk_node32{0, 0, (uint32_t)(h->h_ctr->polyA_links & 0xFFFFFFFFu), (uint32_t)(h->h_ctr->polyA_links >> 32), 0}; // key 0 sorts first
	dbgk_node32 *dst = out + z; // the non-zero keys follow the key-0 node (if this shard reports one)
	if (n == 0) return DBGK_OK;
	dbgk_node32 *d_out = nullptr;
	unsigned long long *d_cursor = nullptr;
	if (hipMalloc(&d_out, n * sizeof(dbgk_node32)) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&d_cursor, 8) != hipSuccess) {
		(void)hipFree(d_out);
		return DBGK_ERR_NOMEM;
	}
	unsigned long long found = 0;
	hipError_t e = hipMemsetAsync(d_cursor, 0, 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_wide_compact, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, h->wside, d_out, d_cursor, n);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(&found, d_cursor, 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dst, d_out, n * sizeof(dbgk_node32), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_out);
	(void)hipFree(d_cursor);
	if (e != hipSuccess) return hip_fail(e, "wide_export_sorted", __LINE__);
	if (found != n) {
		g_last_error = "wide_export_sorted: occupied slots != counted keys";
		return DBGK_ERR_STATE;
	}
	std::sort(dst, dst + n, [](const dbgk_node32 &a, const dbgk_node32 &b) {
		return a.kmer_hi < b.kmer_hi || (a.kmer_hi == b.kmer_hi && a.kmer_lo < b.kmer_lo);
	});
	return DBGK_OK;
}

// host-layout table of host_size == table_slots 32-byte nodes + nul_flag: every key reachable by linear probing from
// hash128(key) % size without crossing a clear flag.  The few nodes that live outside the main table on the
// device (keys whose low word is 0, the key-0 node) are put on their probe chains here, on the host.
extern "C" int dbgk_wide_export_host_table(dbgk_handle *h, uint64_t host_size, dbgk_node32 *array, uint8_t *nul_flag)
{
	if (!h || !array || !nul_flag) return DBGK_ERR_ARG;
	if (!h->wide || !h->finalized) return DBGK_ERR_STATE;
	if (host_size != h->tslots) {
		g_last_error = "dbgk_wide_export_host_table: host_size must be the handle's table_slots (a shard: the slots of its range)";
		return DBGK_ERR_ARG;
	}
	if (!h->sharded && h->h_ctr->n_new + 1 > host_size) return DBGK_ERR_TABLE_FULL;
	int rc = use_device(h);
	if (rc) return rc;
	dbgk_node32 *d_img = nullptr;
	uint8_t *d_flags = nullptr;
	std::vector<WNode> side(kWideSideSlots);
	if (hipMalloc(&d_img, host_size * sizeof(dbgk_node32)) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&d_flags, host_size / 8 + 1) != hipSuccess) {
		(void)hipFree(d_img);
		return DBGK_ERR_NOMEM;
	}
	hipLaunchKernelGGL(k_wide_image, dim3(grid_for(h, host_size / 8 + 1)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, d_img, d_flags);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(array, d_img, host_size * sizeof(dbgk_node32), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(nul_flag, d_flags, host_size / 8 + 1, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(side.data(), h->wside, kWideSideSlots * sizeof(WNode), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_img);
	(void)hipFree(d_flags);
	if (e != hipSuccess) return hip_fail(e, "wide_export_host_table", __LINE__);
	if (h->sharded) return DBGK_OK; // a shard's slice as it is: side-table nodes and the key-0 node are placed over the WHOLE table by the caller
	auto place = [&](dbgk_node32 nd) { // add_node_to_kmerset's rule (kmerSet.cpp:253-273): first slot without a flag on the key's chain
		uint64_t hc = dbgk_wide::hash128(dbgk_wide::Key128{nd.kmer_hi, nd.kmer_lo}) % host_size;
		while (nul_flag[hc >> 3] & (uint8_t)(128u >> (hc & 7u))) hc = (hc + 1 == host_size) ? 0 : hc + 1;
		array[hc] = nd;
		nul_flag[hc >> 3] |= (uint8_t)(128u >> (hc & 7u));
	};
	for (const WNode &s : side)
		if (s.hi1) place(dbgk_node32{s.hi1 - 1ull, 0ull, (uint32_t)s.links, (uint32_t)(s.links >> 32), 0});
	place(dbgk_node32{0, 0, (uint32_t)(h->h_ctr->polyA_links & 0xFFFFFFFFu), (uint32_t)(h->h_ctr->polyA_links >> 32), 0}); // DBGgraph.cpp:418
	return DBGK_OK;
}

// several GPUs with 128-bit keys: nodes grouped by owner, merged by the owner (dbgk_partition_* / dbgk_merge_nodes for
// 32-byte nodes).  counts[p] includes, for p == 0, this handle's key-0 node, which is written first.
extern "C" int dbgk_wide_partition_export(dbgk_handle *h, uint32_t n_parts, dbgk_node32 *d_nodes, uint64_t capacity, uint64_t *counts)
{
	if (!h || !counts || n_parts < 1 || n_parts > (uint32_t)kMaxParts) return DBGK_ERR_ARG;
	if (!h->wide || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long *d_counts = nullptr;
	if (hipMalloc(&d_counts, n_parts * 8) != hipSuccess) return DBGK_ERR_NOMEM;
	std::vector<unsigned long long> hc(n_parts, 0);
	hipError_t e = hipMemsetAsync(d_counts, 0, n_parts * 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_wide_partition, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, h->wside, n_parts, d_counts,
		                   (unsigned long long *)nullptr, (dbgk_node32 *)nullptr, (uint64_t)0);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(hc.data(), d_counts, n_parts * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) {
		(void)hipFree(d_counts);
		return hip_fail(e, "wide_partition_export/count", __LINE__);
	}
	hc[0] += 1; // the key-0 node travels with part 0
	uint64_t total = 0;
	std::vector<unsigned long long> cursors(n_parts);
	for (uint32_t p = 0; p < n_parts; p++) {
		counts[p] = hc[p];
		cursors[p] = total + (p == 0 ? 1 : 0);
		total += hc[p];
	}
	if (!d_nodes) { // counts only
		(void)hipFree(d_counts);
		return DBGK_OK;
	}
	if (total > capacity) {
		(void)hipFree(d_counts);
		return DBGK_ERR_CAPACITY;
	}
	const dbgk_node32 zero = {0, 0, (uint32_t)(h->h_ctr->polyA_links & 0xFFFFFFFFu), (uint32_t)(h->h_ctr->polyA_links >> 32), 0};
	e = hipMemcpyAsync(d_counts, cursors.data(), n_parts * 8, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_nodes, &zero, sizeof zero, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_wide_partition, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->wnodes, h->tslots, h->wside, n_parts,
		                   (unsigned long long *)nullptr, d_counts, d_nodes, capacity);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_counts);
	if (e != hipSuccess) return hip_fail(e, "wide_partition_export", __LINE__);
	return DBGK_OK;
}

extern "C" int dbgk_wide_merge_nodes(dbgk_handle *h, const dbgk_node32 *d_nodes, uint64_t n)
{
	if (!h || (n && !d_nodes)) return DBGK_ERR_ARG;
	if (!h->wide) return DBGK_ERR_STATE;
	if (h->wpart && !h->finalized) { // pending records become the table first: the region build rewrites every slot
		int frc = use_device(h);
		if (frc) return frc;
		frc = h->pending_kmers && !h->wbuilt ? wide_build_from_records(h) : DBGK_OK;
		if (frc) return frc;
		frc = wide_ensure_zero(h);
		if (frc) return frc;
		h->wbuilt = true; // whatever comes later joins the table through the atomic kernels
	}
	int rc = use_device(h);
	if (rc) return rc;
	if (n == 0) return DBGK_OK;
	hipLaunchKernelGGL(k_wide_merge_nodes, dim3(grid_for(h, n)), dim3(kBlock), 0, h->stream, d_nodes, n, h->wref(), h->d_ctr);
	HIPCHK(hipGetLastError());
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// KFREQ exports
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_kfreq_export_counts(dbgk_handle *h, uint64_t first_kmer, uint64_t n, uint8_t *host_out)
{
	if (!h || !host_out) return DBGK_ERR_ARG;
	if (!h->kfreq || !h->finalized) return DBGK_ERR_STATE;
	const uint64_t total = 1ull << (2 * h->cfg.kmer_size);
	if (first_kmer > total || n > total - first_kmer) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (n == 0) return DBGK_OK;
	HIPCHK(hipMemcpyAsync(host_out, h->counts + first_kmer, n, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}

extern "C" int dbgk_kfreq_export_bits(dbgk_handle *h, uint32_t cutoff, uint64_t first_byte, uint64_t n_bytes, uint8_t *host_out)
{
	if (!h || !host_out) return DBGK_ERR_ARG;
	if (!h->kfreq || !h->finalized) return DBGK_ERR_STATE;
	const uint64_t total_bytes = h->n_counts >> 3;
	if (first_byte > total_bytes || n_bytes > total_bytes - first_byte) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (n_bytes == 0) return DBGK_OK;
	uint8_t *d_bits = nullptr;
	if (hipMalloc(&d_bits, n_bytes) != hipSuccess) return DBGK_ERR_NOMEM;
	hipLaunchKernelGGL(k_counts_to_bits, dim3(grid_for(h, n_bytes)), dim3(kBlock), 0, h->stream, h->counts, first_byte, n_bytes, cutoff, d_bits);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(host_out, d_bits, n_bytes, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_bits);
	if (e != hipSuccess) return hip_fail(e, "kfreq_export_bits", __LINE__);
	return DBGK_OK;
}

// counts[first_kmer, first_kmer + n) += another partial table's slice held in device memory of this GPU
// (saturating).  The distinct-k-mer count of the handle is recomputed.
static int kfreq_summary(dbgk_handle *h, uint64_t first, uint64_t n, unsigned long long res[2])
{
	unsigned long long *d_sum = nullptr;
	res[0] = res[1] = 0;
	if (hipMalloc(&d_sum, 16) != hipSuccess) return DBGK_ERR_NOMEM;
	hipError_t e = hipMemsetAsync(d_sum, 0, 16, h->stream);
	if (e == hipSuccess && n) {
		hipLaunchKernelGGL(k_counts_summary, dim3(grid_for(h, n >> 3)), dim3(kBlock), 0, h->stream, h->counts + first, n, d_sum);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(res, d_sum, 16, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_sum);
	if (e != hipSuccess) return hip_fail(e, "kfreq summary", __LINE__);
	return DBGK_OK;
}

extern "C" int dbgk_kfreq_merge_counts(dbgk_handle *h, const uint8_t *d_counts, uint64_t first_kmer, uint64_t n)
{
	if (!h || !d_counts) return DBGK_ERR_ARG;
	if (!h->kfreq || !h->finalized) return DBGK_ERR_STATE;
	if (first_kmer > h->n_counts || n > h->n_counts - first_kmer) return DBGK_ERR_ARG;
	if ((first_kmer & 15u) || (n & 15u) || ((uintptr_t)d_counts & 15u)) {
		g_last_error = "dbgk_kfreq_merge_counts: first_kmer, n and the source address must be multiples of 16";
		return DBGK_ERR_ARG;
	}
	int rc = use_device(h);
	if (rc) return rc;
	if (n) {
		hipLaunchKernelGGL(k_counts_merge, dim3(grid_for(h, n >> 4)), dim3(kBlock), 0, h->stream, h->counts + first_kmer, d_counts, n);
		HIPCHK(hipGetLastError());
	}
	unsigned long long res[2];
	rc = kfreq_summary(h, 0, h->n_counts, res);
	if (rc) return rc;
	h->kf_distinct = res[0];
	h->kf_sum = res[1];
	return DBGK_OK;
}

extern "C" int dbgk_kfreq_device_counts(dbgk_handle *h, uint8_t **d_counts, uint64_t *n)
{
	if (!h || !d_counts || !n) return DBGK_ERR_ARG;
	if (!h->kfreq || !h->finalized) return DBGK_ERR_STATE;
	*d_counts = h->counts;
	*n = h->n_counts;
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// phase A alone
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_extract_kmers(dbgk_handle *h, const char *bases, const uint64_t *offsets, uint64_t n_reads,
                                  uint64_t *kmer, uint8_t *left, uint8_t *right, uint8_t *valid)
{
	if (!h || !offsets || !kmer || !left || !right || !valid) return DBGK_ERR_ARG;
	if (h->wide) return DBGK_ERR_STATE; // 64-bit keys only
	if (offsets[0] != 0) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t nb = offsets[n_reads];
	if (nb == 0) return DBGK_OK;
	const uint64_t words = bitmap_words(nb);
	char *d_bases = nullptr;
	uint64_t *d_off = nullptr, *d_kmer = nullptr;
	uint32_t *d_start = nullptr, *d_dead = nullptr;
	uint8_t *d_l = nullptr, *d_r = nullptr, *d_v = nullptr;
	Counters *d_ctr = nullptr;
	auto cleanup = [&]() {
		for (void *p : {(void *)d_bases, (void *)d_off, (void *)d_kmer, (void *)d_start, (void *)d_dead, (void *)d_l, (void *)d_r,
		                (void *)d_v, (void *)d_ctr})
			if (p) (void)hipFree(p);
	};
	hipError_t e = hipMalloc(&d_bases, nb + 64);
	if (e == hipSuccess) e = hipMalloc(&d_off, (n_reads + 1) * 8);
	if (e == hipSuccess) e = hipMalloc(&d_kmer, nb * 8);
	if (e == hipSuccess) e = hipMalloc(&d_start, words * 4);
	if (e == hipSuccess) e = hipMalloc(&d_dead, words * 4);
	if (e == hipSuccess) e = hipMalloc(&d_l, nb);
	if (e == hipSuccess) e = hipMalloc(&d_r, nb);
	if (e == hipSuccess) e = hipMalloc(&d_v, nb);
	if (e == hipSuccess) e = hipMalloc(&d_ctr, sizeof(Counters));
	if (e != hipSuccess) { cleanup(); return DBGK_ERR_NOMEM; }
	e = hipMemcpyAsync(d_bases, bases, nb, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_off, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_start, 0, words * 4, h->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_dead, 0, words * 4, h->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_ctr, 0, sizeof(Counters), h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_mark, dim3(grid_for(h, n_reads)), dim3(kBlock), 0, h->stream, d_off, n_reads, nb, h->cfg.kmer_size,
		                   h->cfg.max_read_len, d_start, d_dead, d_ctr);
		ReadBatch rb{d_bases, nb, d_start, d_dead, h->cfg.kmer_size, nullptr, &d_ctr->other_seen};
		hipLaunchKernelGGL(k_extract_store<true>, dim3(grid_for(h, (nb + 15) >> 4)), dim3(kBlock), 0, h->stream, rb, d_kmer, d_l,
		                   d_r, d_v);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(kmer, d_kmer, nb * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(left, d_l, nb, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(right, d_r, nb, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(valid, d_v, nb, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	cleanup();
	if (e != hipSuccess) return hip_fail(e, "extract_kmers", __LINE__);
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU building blocks
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_partition_counts(dbgk_handle *h, uint32_t n_parts, uint64_t *counts)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !counts || n_parts < 1 || n_parts > (uint32_t)kMaxParts) return DBGK_ERR_ARG;
	if (!h->finalized || h->sharded) return DBGK_ERR_STATE; // a sharded table is already owned by slot range
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long *d_counts = nullptr;
	if (hipMalloc(&d_counts, n_parts * 8) != hipSuccess) return DBGK_ERR_NOMEM;
	hipError_t e = hipMemsetAsync(d_counts, 0, n_parts * 8, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_partition_count, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, n_parts,
		                   d_counts);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(counts, d_counts, n_parts * 8, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_counts);
	if (e != hipSuccess) return hip_fail(e, "partition_counts", __LINE__);
	counts[0] += 1; // the key-0 node travels with part 0
	return DBGK_OK;
}

__global__ void k_write_polyA_node(Node *out, uint64_t index, const Counters *ctr)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		out[index].kmer = 0ull;
		out[index].links = ctr->polyA_links;
	}
}

extern "C" int dbgk_partition_export(dbgk_handle *h, uint32_t n_parts, dbgk_node *d_nodes, uint64_t capacity)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !d_nodes || n_parts < 1 || n_parts > (uint32_t)kMaxParts) return DBGK_ERR_ARG;
	if (!h->finalized || h->sharded) return DBGK_ERR_STATE;
	std::vector<uint64_t> counts(n_parts);
	int rc = dbgk_partition_counts(h, n_parts, counts.data());
	if (rc) return rc;
	uint64_t total = 0;
	std::vector<unsigned long long> cursors(n_parts);
	for (uint32_t p = 0; p < n_parts; p++) {
		cursors[p] = total + (p == 0 ? 1 : 0); // slot 0 of part 0 is the key-0 node
		total += counts[p];
	}
	if (total > capacity) return DBGK_ERR_CAPACITY;
	unsigned long long *d_cursors = nullptr;
	if (hipMalloc(&d_cursors, n_parts * 8) != hipSuccess) return DBGK_ERR_NOMEM;
	hipError_t e = hipMemcpyAsync(d_cursors, cursors.data(), n_parts * 8, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_write_polyA_node, dim3(1), dim3(64), 0, h->stream, reinterpret_cast<Node *>(d_nodes), (uint64_t)0,
		                   h->d_ctr);
		hipLaunchKernelGGL(k_partition_scatter, dim3(grid_for(h, h->tslots)), dim3(kBlock), 0, h->stream, h->table, h->tslots, n_parts,
		                   d_cursors, reinterpret_cast<Node *>(d_nodes), capacity);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	(void)hipFree(d_cursors);
	if (e != hipSuccess) return hip_fail(e, "partition_export", __LINE__);
	return DBGK_OK;
}

extern "C" int dbgk_merge_nodes(dbgk_handle *h, const dbgk_node *d_nodes, uint64_t n)
{
	if (h && h->seed) return DBGK_ERR_STATE; // SEEDIDX handles: use dbgk_seed_export_*
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || (n && !d_nodes)) return DBGK_ERR_ARG;
	if ((uintptr_t)d_nodes & 15u) return DBGK_ERR_ARG;
	if (h->sharded) return DBGK_ERR_STATE; // use dbgk_shard_merge
	int rc = use_device(h);
	if (rc) return rc;
	if (n == 0) return DBGK_OK;
	if (h->zero_pending) { // PARTITION handle used as a merge target before any region build: the table must be
		rc = zero_table_now(h); // empty for the direct path; records pushed so far (or later) are built on top of it
		if (rc) return rc;
		h->incr = true;
	}
	TimedSpan sp;
	rc = span_begin(h, PH_FIXUP, sp);
	if (rc) return rc;
	hipLaunchKernelGGL(k_merge_nodes, dim3(grid_for(h, n)), dim3(kBlock), 0, h->stream, reinterpret_cast<const Node *>(d_nodes), n,
	                   h->tref(), h->d_ctr);
	HIPCHK(hipGetLastError());
	return span_end(h, sp);
}

extern "C" int dbgk_copy_nodes_peer(dbgk_handle *dst, dbgk_node *d_dst, dbgk_handle *src, const dbgk_node *d_src, uint64_t n)
{
	if (!dst || !src || (n && (!d_dst || !d_src))) return DBGK_ERR_ARG;
	if (n == 0) return DBGK_OK;
	int rc = dbgk_sync(src);
	if (rc) return rc;
	rc = use_device(dst);
	if (rc) return rc;
	HIPCHK(hipMemcpyPeer(d_dst, dst->device, d_src, src->device, n * sizeof(dbgk_node)));
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// sharded tables (one handle per GPU, each owning a contiguous slot range of one global table)
// ---------------------------------------------------------------------------------------------
extern "C" int dbgk_plan_partition(uint64_t table_slots, uint64_t expected_kmers, uint32_t shard_count, uint32_t shard_index, dbgk_plan_info *out)
{
	if (!out || table_slots == 0) return DBGK_ERR_ARG;
	dbgk_handle *h = new (std::nothrow) dbgk_handle();   // never touches a device: plan_partition is host arithmetic
	if (!h) return DBGK_ERR_NOMEM;
	memset(&h->cfg, 0, sizeof h->cfg);
	h->cfg.kmer_size = 31;
	h->cfg.engine = DBGK_ENGINE_PARTITION;
	h->cfg.table_slots = table_slots;
	h->cfg.expected_kmers = expected_kmers;
	h->cfg.shard_count = shard_count;
	h->cfg.shard_index = shard_index;
	h->size = table_slots;
	h->magic = make_mod_magic(table_slots);
	int rc = plan_partition(h);
	if (rc == DBGK_OK && !h->part) rc = DBGK_ERR_ARG;
	if (rc == DBGK_OK) {
		const PartGeom &G = h->geom;
		memset(out, 0, sizeof *out);
		out->table_slots = G.size;
		out->r = G.r;
		out->level1_buckets = G.n1;
		out->final_per_level1 = G.n2;
		out->three_level = h->three ? 1u : 0u;
		out->buckets_per_rank = G.B;
		out->own_buckets = G.nb_own;
		out->first_bucket = G.b_lo;
		out->slot_lo = G.slot_lo;
		out->slot_hi = G.slot_hi;
		out->records_per_level1_bucket = G.cap1;
		out->records_per_final_bucket = G.cap2;
		const uint64_t n_entries = (uint64_t)G.n_ranks * G.B * G.n_sub;
		out->table_bytes = (G.slot_hi - G.slot_lo) * sizeof(Node);
		out->level1_store_bytes = n_entries * G.cap1 * 8;
		out->inbox_bytes = h->sharded ? n_entries * G.cap1 * 8 : 0;
		out->final_store_bytes = (uint64_t)G.nb_own * G.n2 * G.cap2 * 8 + (h->three ? (uint64_t)G.nb_own * h->fan_mid * h->g_mid.cap2 * 8 : 0);
	}
	delete h;
	return rc;
}

extern "C" int dbgk_shard_buffers(dbgk_handle *h, dbgk_shard_info *out)
{
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || !out) return DBGK_ERR_ARG;
	if (h->wide) { // 16-byte records; the buffers are those of the CURRENT pass (dbgk_wide_pass_info)
		if (!h->wpart || !h->wmulti) return DBGK_ERR_STATE;
		const WPartGeom &G = h->wgeom;
		memset(out, 0, sizeof(*out));
		out->n_ranks = G.n_ranks;
		out->rank = G.rank;
		out->slot_lo = G.slot_lo;
		out->slot_hi = G.slot_hi;
		out->table_slots_global = h->size;
		out->buckets_per_rank = G.Bp;
		out->own_buckets = wide_pass_buckets(h);
		out->bucket_bytes = G.cap1 * 16;
		out->cnt_bucket_bytes = 4;
		out->chunk_bytes = (uint64_t)G.Bp * G.cap1 * 16;
		out->cnt_chunk_bytes = (uint64_t)G.Bp * 4;
		out->d_send = h->wstore.l1;
		out->d_send_cnt = h->wstore.cnt1;
		out->d_recv = h->sharded ? (void *)h->winbox : (void *)h->wstore.l1;
		out->d_recv_cnt = h->sharded ? (void *)h->winbox_cnt : (void *)h->wstore.cnt1;
		return DBGK_OK;
	}
	if (!h->part) return DBGK_ERR_STATE;
	const PartGeom &G = h->geom;
	memset(out, 0, sizeof(*out));
	out->n_ranks = G.n_ranks;
	out->rank = G.rank;
	out->slot_lo = G.slot_lo;
	out->slot_hi = G.slot_hi;
	out->table_slots_global = h->size;
	out->buckets_per_rank = G.B;
	out->own_buckets = G.nb_own;
	out->bucket_bytes = (uint64_t)G.n_sub * G.cap1 * 8;
	out->cnt_bucket_bytes = (uint64_t)G.n_sub * 4;
	out->chunk_bytes = (uint64_t)G.B * G.n_sub * G.cap1 * 8;
	out->cnt_chunk_bytes = (uint64_t)G.B * G.n_sub * 4;
	out->d_send = h->store.l1;
	out->d_send_cnt = h->store.cnt1;
	out->d_recv = h->sharded ? (void *)h->inbox : (void *)h->store.l1;
	out->d_recv_cnt = h->sharded ? (void *)h->inbox_cnt : (void *)h->store.cnt1;
	return DBGK_OK;
}

extern "C" int dbgk_shard_mark_exchanged(dbgk_handle *h)
{
	if (!h || !(h->sharded || (h->wide && h->wmulti))) return DBGK_ERR_STATE;
	h->exchanged = true;
	return DBGK_OK;
}

extern "C" int dbgk_shard_plan(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->wide) {
		if (!h->wmulti || !h->wpass_open || h->finalized) return DBGK_ERR_STATE;
		int wrc = use_device(h);
		if (wrc) return wrc;
		return wide_plan_pass(h);
	}
	if (!h->part || h->part_built || h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	return part_plan(h);
}

extern "C" int dbgk_shard_build_range(dbgk_handle *h, uint32_t j0, uint32_t j1)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->wide) { // own-bucket indices of the current pass, in order
		if (!h->wmulti || !h->wpass_open || !h->wplanned || h->finalized) return DBGK_ERR_STATE;
		if (j0 != h->wnext || j1 < j0 || j1 > wide_pass_buckets(h)) return DBGK_ERR_ARG;
		int wrc = use_device(h);
		if (wrc) return wrc;
		return j1 > j0 ? wide_build_range(h, j0, j1) : DBGK_OK;
	}
	if (!h->part || !h->part_planned || h->part_built || h->finalized) return DBGK_ERR_STATE;
	if (j0 != h->next_bucket || j1 < j0 || j1 > h->geom.nb_own) return DBGK_ERR_ARG; // ranges are consumed in order, each bucket once
	int rc = use_device(h);
	if (rc) return rc;
	return part_build_range(h, j0, j1, true);
}

static int shard_list(dbgk_handle *h, void *list, unsigned long long *d_n, uint64_t cap, dbgk_node **d_nodes, uint64_t *n)
{
	if (!h || !d_nodes || !n) return DBGK_ERR_ARG;
	if (!(h->part || (h->wide && h->wmulti)) || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long v = 0;
	HIPCHK(hipMemcpyAsync(&v, d_n, 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	*d_nodes = reinterpret_cast<dbgk_node *>(list);
	*n = v < cap ? v : cap;
	if (!h->wide && list == (void *)h->store.ovf) return DBGK_OK; // the surplus beyond the list was aggregated in the side table (dbgk_shard_heavy)
	return v > cap ? DBGK_ERR_CAPACITY : DBGK_OK;
}

// (WIDE handles: the lists hold 32-byte dbgk_node32 entries -- nodes {hi, lo, l_link, r_link} / observations {hi, lo, lb, rb})
extern "C" int dbgk_shard_outgoing(dbgk_handle *h, dbgk_node **d_nodes, uint64_t *n)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->wide) return shard_list(h, h->wstore.outgoing, h->wstore.outgoing_n, h->wstore.outgoing_cap, d_nodes, n);
	return shard_list(h, h->store.outgoing, h->store.outgoing_n, h->store.outgoing_cap, d_nodes, n);
}

extern "C" int dbgk_shard_overflow(dbgk_handle *h, dbgk_node **d_triples, uint64_t *n)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->wide) return shard_list(h, h->wstore.ovf, &h->wstore.ovf_n[0], h->wstore.ovf_cap, d_triples, n);
	return shard_list(h, h->store.ovf, &h->store.ovf_n[0], h->store.ovf_cap, d_triples, n);
}

extern "C" int dbgk_shard_heavy(dbgk_handle *h, dbgk_node **d_table, uint64_t *n_slots)
{
	if (!h || !d_table || !n_slots) return DBGK_ERR_ARG;
	if (h->wide) { // no side table of aggregated surplus in the wide path: a full overflow list is DBGK_ERR_CAPACITY at finalize
		*d_table = nullptr;
		*n_slots = 0;
		return (h->wmulti && h->finalized) ? DBGK_OK : DBGK_ERR_STATE;
	}
	if (!h->part || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	unsigned long long v = 0;
	HIPCHK(hipMemcpyAsync(&v, &h->store.ovf_n[0], 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	*d_table = reinterpret_cast<dbgk_node *>(h->store.hh);
	*n_slots = (h->store.hh && v > h->store.ovf_cap) ? h->store.hh_size : 0; // unused unless the overflow list ran full
	return DBGK_OK;
}

extern "C" int dbgk_shard_merge(dbgk_handle *h, const dbgk_node *d_nodes, uint64_t n, int is_triple, int from_previous_shard)
{
	if (!h || (n && !d_nodes)) return DBGK_ERR_ARG;
	if (!(h->sharded || (h->wide && h->wmulti)) || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	if (n == 0) return DBGK_OK;
	TimedSpan sp;
	rc = span_begin(h, PH_FIXUP, sp);
	if (rc) return rc;
	if (h->wide) {
		hipLaunchKernelGGL(k_wide_merge_sharded, dim3(grid_for(h, n)), dim3(kBlock), 0, h->stream, reinterpret_cast<const dbgk_node32 *>(d_nodes),
		                   (const unsigned long long *)nullptr, n, n, is_triple ? 1 : 0, from_previous_shard ? 1 : 0, h->wgeom, h->wstore, h->wnodes, h->d_ctr);
		HIPCHK(hipGetLastError());
		return span_end(h, sp);
	}
	hipLaunchKernelGGL(k_merge_sharded, dim3(grid_for(h, n)), dim3(kBlock), 0, h->stream, reinterpret_cast<const Node *>(d_nodes),
	                   (const unsigned long long *)nullptr, n, n, is_triple ? 1 : 0, from_previous_shard ? 1 : 0, h->geom, h->store, h->table,
	                   h->d_ctr);
	HIPCHK(hipGetLastError());
	return span_end(h, sp);
}

extern "C" int dbgk_wide_pass_info(dbgk_handle *h, uint32_t *n_passes, uint32_t *passes_done)
{
	if (!h) return DBGK_ERR_ARG;
	if (!h->wide) return DBGK_ERR_STATE;
	if (n_passes) *n_passes = h->wpart ? h->wgeom.n_passes : 1u;
	if (passes_done) *passes_done = h->wpart ? h->wpasses_done : 0u;
	return DBGK_OK;
}

extern "C" int dbgk_wide_begin_pass(dbgk_handle *h, uint32_t pass)
{
	if (!h) return DBGK_ERR_ARG;
	if (!h->wide || !h->wpart || h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	if (pass == 0 && h->wpass_open && h->wpasses_done == 0 && h->pending_kmers == 0) return DBGK_OK; // pass 0 is open after create / reset
	return wide_begin_pass(h, pass);
}

extern "C" int dbgk_wide_end_pass(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	if (!h->wide || !h->wpart || !h->wmulti || h->finalized || !h->wpass_open) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	return wide_end_pass(h);
}

extern "C" int dbgk_shard_side_export(dbgk_handle *h, dbgk_node32 **d_nodes, uint64_t *n)
{
	if (!h || !d_nodes || !n) return DBGK_ERR_ARG;
	if (!h->wide || !h->wmulti || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipMemsetAsync(h->w_side_n, 0, 8, h->stream));
	hipLaunchKernelGGL(k_wide_side_export, dim3(16), dim3(kBlock), 0, h->stream, h->wside, h->d_ctr, h->w_side_out, h->w_side_n, (uint64_t)kWideSideSlots + 1);
	HIPCHK(hipGetLastError());
	unsigned long long v = 0;
	HIPCHK(hipMemcpyAsync(&v, h->w_side_n, 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	*d_nodes = h->w_side_out;
	*n = v + 1; // + the key-0 node in front
	return DBGK_OK;
}

extern "C" int dbgk_shard_side_clear(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	if (!h->wide || !h->wmulti || !h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	// the claims in the side table were counted as new keys of this handle: take them back with the table
	HIPCHK(hipMemsetAsync(h->w_side_n, 0, 8, h->stream));
	hipLaunchKernelGGL(k_wide_side_export, dim3(16), dim3(kBlock), 0, h->stream, h->wside, h->d_ctr, h->w_side_out, h->w_side_n, (uint64_t)kWideSideSlots + 1);
	HIPCHK(hipGetLastError());
	unsigned long long v = 0;
	HIPCHK(hipMemcpyAsync(&v, h->w_side_n, 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	Counters c;
	HIPCHK(hipMemcpyAsync(&c, h->d_ctr, sizeof c, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	c.n_new -= v;
	c.polyA_links = 0;
	HIPCHK(hipMemcpyAsync(h->d_ctr, &c, sizeof c, hipMemcpyHostToDevice, h->stream));
	HIPCHK(hipMemsetAsync(h->wside, 0, kWideSideSlots * sizeof(WNode), h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return DBGK_OK;
}
