// dbgk_host_push.h -- part of libdbgk.so's host side (one translation unit: included by dbgk.hip, in this order).
// the push entry points: staging slots, host-to-device copies on the copy stream, ASCII and 2-bit packed input, zero-copy hand-over
#pragma once

// pageable host buffer -> pinned staging buffer.  One memcpy thread moves ~12 GB/s, less than a third of
// what the H2D copy behind it can take, so large batches are cut over a few threads (DBGK_COPY_THREADS,
// default 8; the offset bookkeeping of the batch runs on the calling thread meanwhile).
static void staged_copy(char *dst, const char *src, size_t n, std::vector<std::thread> &workers)
{
	static const int want = getenv("DBGK_COPY_THREADS") ? atoi(getenv("DBGK_COPY_THREADS")) : 8;
	const size_t min_piece = 8u << 20;
	size_t pieces = want > 1 ? std::min<size_t>((size_t)want, n / min_piece) : 1;
	if (pieces <= 1) {
		memcpy(dst, src, n);
		return;
	}
	const size_t per = ((n + pieces - 1) / pieces + 4095) & ~(size_t)4095;
	for (size_t p = 1; p < pieces; p++) {
		const size_t lo = p * per, hi = std::min(n, lo + per);
		if (lo < hi) workers.emplace_back([=]() { memcpy(dst + lo, src + lo, hi - lo); });
	}
	memcpy(dst, src, std::min(n, per));
}

// Is the caller's buffer page-locked memory the GPU reads directly (hipHostMalloc / hipHostRegister -- a torch pinned tensor, a
// parser's own pinned arena)?  Then dbgk_push_reads copies host-to-device straight out of it: the staging copy, which is what
// bounds the pageable path (~30 GB/s against the link's 57), does not happen.
static bool device_readable_host(const char *p, size_t n)
{
	static const bool off = DBGK_EXPERIMENT_ENV("DBGK_NO_PINNED_SOURCE") && atoi(DBGK_EXPERIMENT_ENV("DBGK_NO_PINNED_SOURCE"));
	if (off || !p || !n) return false;
	void *dev[2] = {nullptr, nullptr};
	int i = 0;
	for (const char *q : {p, p + n - 1}) {
		hipPointerAttribute_t a;
		if (hipPointerGetAttributes(&a, q) != hipSuccess) {
			(void)hipGetLastError(); // plain malloc'ed memory: "invalid value", not an error of ours
			return false;
		}
		if (a.type != hipMemoryTypeHost) return false;
		dev[i++] = a.devicePointer;
	}
	// both ends page-locked is not enough: two registrations with pageable memory between them would pass.  One mapping means
	// one contiguous range of device addresses: the two ends must lie exactly n - 1 bytes apart there as well.
	if (dev[0] && dev[1] && (const char *)dev[1] - (const char *)dev[0] != (ptrdiff_t)(n - 1)) return false;
	return true;
}

static int ensure_slot(dbgk_handle *h, StageSlot &s)
{
	if (s.d_bases) return DBGK_OK;
	const uint64_t words = bitmap_words(h->cap_bases);
	if (hipHostMalloc(&s.h_bases, h->cap_bases, hipHostMallocDefault) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipHostMalloc(&s.h_offsets, (h->cap_reads + 1) * 8, hipHostMallocDefault) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&s.d_bases, h->cap_bases + 64) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&s.d_offsets, (h->cap_reads + 1) * 8) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&s.d_start, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
	if (hipMalloc(&s.d_dead, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
	HIPCHK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
	return DBGK_OK;
}

// The host-to-device copies of a batch (sequences from `src`, offsets from the slot's pinned array) on the handle's COPY stream; the
// compute stream waits for them.  The copy of batch i+1 so overlaps the kernels of batch i (and the table reset in front of the
// first batch) instead of queueing behind them.  The slot's device buffers are free: the caller has waited for s.done.
static int h2d_batch(dbgk_handle *h, StageSlot &s, const char *src, uint64_t nb, uint64_t n_offsets, bool last_of_pinned_source)
{
	static const bool serial = DBGK_EXPERIMENT_ENV("DBGK_COPY_ON_COMPUTE_STREAM") && atoi(DBGK_EXPERIMENT_ENV("DBGK_COPY_ON_COMPUTE_STREAM")); // measurements
	if (!serial && !h->copy_stream) HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
	hipStream_t cs = serial ? h->stream : h->copy_stream;
	if (nb) HIPCHK(hipMemcpyAsync(s.d_bases, src, nb, hipMemcpyHostToDevice, cs));
	if (n_offsets) HIPCHK(hipMemcpyAsync(s.d_offsets, s.h_offsets, n_offsets * 8, hipMemcpyHostToDevice, cs)); // (0: reads of one length, no offsets travel)
	if (last_of_pinned_source) { // the caller's buffer is free again once the LAST copy out of it has run: waited for on return
		if (!h->source_read) HIPCHK(hipEventCreateWithFlags(&h->source_read, hipEventDisableTiming));
		HIPCHK(hipEventRecord(h->source_read, cs));
	}
	if (!serial) {
		HIPCHK(hipEventRecord(s.copied, cs));
		// level 2 of what the batches before this one stored goes onto the compute stream BEFORE that stream is told to wait for this
		// batch's copy: it runs while the batch is on the link
		int rc = early_l2(h);
		if (rc) return rc;
		HIPCHK(hipStreamWaitEvent(h->stream, s.copied, 0));
	}
	return DBGK_OK;
}

// Zero-copy hand-over of a batch: the caller writes the sequences and offsets straight into the handle's pinned staging
// buffers (dbgk_push_acquire) and commits them (dbgk_push_commit) -- what dbgk_push_reads does minus its copy of the batch.
extern "C" int dbgk_push_acquire(dbgk_handle *h, char **bases, uint64_t **offsets, uint64_t *cap_bases, uint64_t *cap_reads)
{
	if (!h || !bases || !offsets) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	StageSlot &s = h->slots[h->next_slot];
	rc = ensure_slot(h, s);
	if (rc) return rc;
	if (s.busy) {
		HIPCHK(hipEventSynchronize(s.done));
		s.busy = false;
	}
	s.acquired = true;
	*bases = s.h_bases;
	*offsets = s.h_offsets;
	if (cap_bases) *cap_bases = h->cap_bases;
	if (cap_reads) *cap_reads = h->cap_reads;
	return DBGK_OK;
}

static int push_commit_impl(dbgk_handle *h, uint64_t n_reads, bool packed)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (packed && h->seed) return DBGK_ERR_ARG;
	if (n_reads == 0) return DBGK_OK;
	if (n_reads > h->cap_reads) return DBGK_ERR_ARG;
	if (h->seed && h->total_reads + n_reads > 0xFFFFFFFFull) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	StageSlot &s = h->slots[h->next_slot];
	if (!s.h_offsets || s.busy || !s.acquired) return DBGK_ERR_STATE; // dbgk_push_acquire first, one commit per acquire
	s.acquired = false;
	const uint64_t K = (uint64_t)h->cfg.kmer_size, max_len = (uint64_t)h->cfg.max_read_len;
	if (s.h_offsets[0] != 0) return DBGK_ERR_ARG;
	uint64_t batch_windows = 0, len_max = 0;
	int has_long = 0;
	int64_t uniform_len = (int64_t)(s.h_offsets[1] - s.h_offsets[0]);
	for (uint64_t i = 1; i <= n_reads; i++) {
		if (s.h_offsets[i] < s.h_offsets[i - 1] || s.h_offsets[i] > h->cap_bases) return DBGK_ERR_ARG;
		const uint64_t len = s.h_offsets[i] - s.h_offsets[i - 1], rl = len > max_len ? max_len : len;
		if (rl >= K) batch_windows += rl - K + 1;
		if (len > max_len) has_long = 1;
		if ((int64_t)len != uniform_len) uniform_len = 0;
		len_max = std::max(len_max, len);
		if (h->seed && len >= (1ull << 30)) return DBGK_ERR_ARG;
	}
	const uint64_t nb = s.h_offsets[n_reads];
	const bool streaming = (h->part && !h->sharded) || (h->wpart && !h->wbuilt);
	if (streaming && h->pending_kmers > 0 && h->pending_kmers + batch_windows > h->store_capacity) { // the store is full: records -> table first
		rc = flush_records(h);
		if (rc) return rc;
	}
	rc = h2d_batch(h, s, s.h_bases, packed ? ((nb + 15) >> 4) * 4 : nb, n_reads + 1, false);
	if (rc) return rc;
	rc = launch_batch(h, packed ? nullptr : s.d_bases, s.d_offsets, n_reads, nb, s.d_start, s.d_dead, has_long, uniform_len, len_max,
	                  packed ? reinterpret_cast<const uint32_t *>(s.d_bases) : nullptr);
	if (rc) return rc;
	h->pending_kmers += batch_windows;
	HIPCHK(hipEventRecord(s.done, h->stream));
	s.busy = true;
	h->next_slot ^= 1;
	return DBGK_OK;
}

extern "C" int dbgk_push_commit(dbgk_handle *h, uint64_t n_reads) { return push_commit_impl(h, n_reads, false); }

extern "C" int dbgk_push_commit_packed(dbgk_handle *h, uint64_t n_reads, uint64_t other_bytes)
{
	const int rc = push_commit_impl(h, n_reads, true);
	if (rc == DBGK_OK) h->host_other_bytes += other_bytes;
	return rc;
}

extern "C" int dbgk_push_reads(dbgk_handle *h, const char *bases, const uint64_t *offsets, uint64_t n_reads)
{
	if (!h || !offsets || (n_reads && !bases && offsets[n_reads] != offsets[0])) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (h->seed && h->total_reads + n_reads > 0xFFFFFFFFull) return DBGK_ERR_ARG; // id is a 32-bit field
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t K = (uint64_t)h->cfg.kmer_size, max_len = (uint64_t)h->cfg.max_read_len;
	// PARTITION engine: the record store holds store_capacity occurrences; a batch that would not fit is
	// preceded by a flush (records -> table, dbgk_flush).  Batches are cut to the room that is left only when
	// the store is at least one staging batch large; a smaller store (expected_kmers a gross under-estimate)
	// takes whole batches and sends the excess through its overflow lists as before.
	const bool streaming = (h->part && !h->sharded) || (h->wpart && !h->wbuilt);
	const bool cut_to_room = streaming && h->store_capacity >= h->cap_bases && !h->wpart; // (a wide store is built once: no point in filling it to the brim)
	const bool pinned_source = n_reads && device_readable_host(bases + offsets[0], offsets[n_reads] - offsets[0]);
	bool source_in_flight = false;
	struct WaitSource { // EVERY exit path waits for the copies that still read the caller's buffer (the header: it may be reused on return)
		dbgk_handle *h; bool &on;
		~WaitSource() { if (on && h->source_read) (void)hipEventSynchronize(h->source_read); }
	} wait_source{h, source_in_flight};
	uint64_t r0 = 0;
	while (r0 < n_reads) {
		// largest [r0, r1) that fits the staging buffers (and the record store).  ONE read-only pass over the offsets validates them
		// and gathers everything the launch needs (windows, longest read, equal lengths); the rebased copy for the device is
		// written later, while the sequences are on their way
		uint64_t r1 = r0, batch_windows = 0, len_max = 0;
		const uint64_t base0 = offsets[r0];
		const uint64_t room = h->store_capacity > h->pending_kmers ? h->store_capacity - h->pending_kmers : 0;
		const uint64_t r_end = std::min<uint64_t>(n_reads, r0 + h->cap_reads);
		const uint64_t first_len = offsets[r0 + 1] >= base0 ? offsets[r0 + 1] - base0 : 0;
		bool uniform = true;
		for (uint64_t prev = base0; r1 < r_end; r1++) {
			const uint64_t next = offsets[r1 + 1];
			if (next < prev) return DBGK_ERR_ARG;
			if (next - base0 > h->cap_bases) break;
			const uint64_t len = next - prev, rl = len > max_len ? max_len : len, w = rl >= K ? rl - K + 1 : 0ull;
			if (cut_to_room && batch_windows + w > room && (r1 > r0 || h->pending_kmers > 0)) break;
			batch_windows += w;
			len_max = len > len_max ? len : len_max;
			uniform = uniform && len == first_len;
			prev = next;
		}
		if (streaming && !(h->wpart && h->wbuilt) && h->pending_kmers > 0 &&
		    (r1 == r0 || (!cut_to_room && h->pending_kmers + batch_windows > h->store_capacity))) {
			rc = flush_records(h);
			if (rc) return rc;
			if (r1 == r0) continue; // cut again with the whole store free
		}
		if (r1 == r0) return DBGK_ERR_ARG; // a single read larger than max_batch_bases
		if (h->seed && len_max >= (1ull << 30)) return DBGK_ERR_ARG; // pos is a 30-bit field
		StageSlot &s = h->slots[h->next_slot];
		rc = ensure_slot(h, s);
		if (rc) return rc;
		if (s.busy) {
			HIPCHK(hipEventSynchronize(s.done));
			s.busy = false;
		}
		s.acquired = false; // (the slot is overwritten: a batch acquired before this call and not committed is gone)
		const uint64_t nb = offsets[r1] - base0, nr = r1 - r0;
		std::vector<std::thread> copiers;
		struct Join { // every exit path below waits for the copy threads
			std::vector<std::thread> &w;
			~Join() { for (auto &t : w) if (t.joinable()) t.join(); }
		} join_copiers{copiers};
		if (nb && !pinned_source) staged_copy(s.h_bases, bases + base0, nb, copiers);
		const int has_long = len_max > max_len ? 1 : 0;
		const int64_t uniform_len = uniform ? (int64_t)first_len : 0;
		{
			const uint64_t *src = offsets + r0;
			uint64_t *dst = s.h_offsets;
			for (uint64_t i = 0; i <= nr; i++) dst[i] = src[i] - base0;
		}
		for (auto &t : copiers) t.join();
		rc = h2d_batch(h, s, pinned_source ? bases + base0 : s.h_bases, nb, nr + 1, pinned_source);
		if (rc) return rc;
		source_in_flight = source_in_flight || pinned_source;
		rc = launch_batch(h, s.d_bases, s.d_offsets, nr, nb, s.d_start, s.d_dead, has_long, uniform_len, len_max);
		if (rc) return rc;
		h->pending_kmers += batch_windows;
		HIPCHK(hipEventRecord(s.done, h->stream));
		s.busy = true;
		h->next_slot ^= 1;
		r0 = r1;
	}
	return DBGK_OK; // (wait_source: as with the staged path, `bases` may be reused on return)
}

extern "C" int dbgk_push_reads_device(dbgk_handle *h, const char *d_bases, const uint64_t *d_offsets,
                                      uint64_t n_reads, uint64_t n_bases)
{
	if (!h || !d_offsets || (n_bases && !d_bases)) return DBGK_ERR_ARG;
	if (((uintptr_t)d_bases & 15u) || ((uintptr_t)d_offsets & 7u)) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t words = bitmap_words(n_bases);
	if (words > h->dev_bits_words) {
		HIPCHK(hipStreamSynchronize(h->stream));
		if (h->dev_start) (void)hipFree(h->dev_start);
		if (h->dev_dead) (void)hipFree(h->dev_dead);
		h->dev_start = h->dev_dead = nullptr;
		h->dev_bits_words = 0;
		if (hipMalloc(&h->dev_start, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		if (hipMalloc(&h->dev_dead, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		h->dev_bits_words = words;
	}
	// the record store takes what it was sized for; the number of windows of a device batch is only known as
	// an upper bound here (one per base)
	if (((h->part && !h->sharded) || (h->wpart && !h->wbuilt)) && h->pending_kmers > 0 && h->pending_kmers + n_bases > h->store_capacity) {
		rc = flush_records(h);
		if (rc) return rc;
	}
	rc = launch_batch(h, d_bases, d_offsets, n_reads, n_bases, h->dev_start, h->dev_dead, -1);
	if (rc == DBGK_OK) h->pending_kmers += n_bases;
	return rc;
}

// dbgk_push_reads for a batch that is already 2 bits per base (include/dbgk.h).  Same cutting into staging batches; a batch
// must start on a word of its own on the device, so one that starts in the middle of a source word is shifted into place while
// it is copied into the pinned staging buffer (host threads; the copy is a quarter of the ASCII one).  A page-locked source is
// read by the copy engine directly whenever the batch starts on a word boundary -- later batches are cut where that holds.
extern "C" void dbgk_internal_shift_packed(const uint32_t *src, uint64_t first_base, uint64_t n_words, uint64_t src_words, uint32_t *dst);

// one pass over offsets[r0 .. r1]: monotone?  windows the reads hold, longest and shortest read.  Ten million reads are 80 MB of
// offsets -- a single thread needs longer for them than the packed sequences need for the PCIe link, so large ranges are cut over threads.
namespace {
struct OffsetScan {
	bool ok = true;
	uint64_t windows = 0, len_max = 0, len_min = ~0ull;
};
OffsetScan scan_offsets(const uint64_t *off, uint64_t r0, uint64_t r1, uint64_t K, uint64_t max_len)
{
	auto part = [=](uint64_t a, uint64_t b) {
		OffsetScan o;
		uint64_t prev = off[a];
		for (uint64_t i = a; i < b; i++) {
			const uint64_t next = off[i + 1];
			if (next < prev) { o.ok = false; break; }
			const uint64_t len = next - prev, rl = len > max_len ? max_len : len;
			o.windows += rl >= K ? rl - K + 1 : 0ull;
			o.len_max = len > o.len_max ? len : o.len_max;
			o.len_min = len < o.len_min ? len : o.len_min;
			prev = next;
		}
		return o;
	};
	const uint64_t n = r1 - r0;
	static const int want = getenv("DBGK_COPY_THREADS") ? atoi(getenv("DBGK_COPY_THREADS")) : 8;
	const uint64_t pieces = want > 1 ? std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)want, n >> 19)) : 1;
	if (pieces <= 1) return part(r0, r1);
	std::vector<OffsetScan> res(pieces);
	std::vector<std::thread> th;
	const uint64_t per = (n + pieces - 1) / pieces;
	for (uint64_t p = 0; p < pieces; p++) {
		const uint64_t a = r0 + p * per, b = std::min(r1, a + per);
		if (a >= b) continue;
		if (p + 1 < pieces) th.emplace_back([&res, part, p, a, b]() { res[p] = part(a, b); });
		else res[p] = part(a, b);
	}
	for (auto &t : th) t.join();
	OffsetScan o;
	for (const OffsetScan &x : res) {
		o.ok = o.ok && x.ok;
		o.windows += x.windows;
		o.len_max = std::max(o.len_max, x.len_max);
		o.len_min = std::min(o.len_min, x.len_min);
	}
	return o;
}
// dst[i] = src[i] - base for i in [0, n]: the offsets of a batch as the device sees them
void rebase_offsets(uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t base)
{
	auto part = [=](uint64_t a, uint64_t b) { for (uint64_t i = a; i < b; i++) dst[i] = src[i] - base; };
	static const int want = getenv("DBGK_COPY_THREADS") ? atoi(getenv("DBGK_COPY_THREADS")) : 8;
	const uint64_t pieces = want > 1 ? std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)want, (n + 1) >> 19)) : 1;
	if (pieces <= 1) { part(0, n + 1); return; }
	std::vector<std::thread> th;
	const uint64_t per = (n + 1 + pieces - 1) / pieces;
	for (uint64_t p = 1; p < pieces; p++) th.emplace_back(part, p * per, std::min(n + 1, (p + 1) * per));
	part(0, std::min(n + 1, per));
	for (auto &t : th) t.join();
}
} // namespace

extern "C" int dbgk_push_reads_packed(dbgk_handle *h, const uint32_t *packed, const uint64_t *offsets, uint64_t n_reads, uint64_t other_bytes)
{
	if (!h || !offsets || (n_reads && !packed && offsets[n_reads] != offsets[0])) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (h->seed) return DBGK_ERR_ARG; // windows of the seed index are cut at 'N'
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t K = (uint64_t)h->cfg.kmer_size, max_len = (uint64_t)h->cfg.max_read_len;
	const bool streaming = (h->part && !h->sharded) || (h->wpart && !h->wbuilt);
	const bool cut_to_room = streaming && h->store_capacity >= h->cap_bases && !h->wpart;
	const OffsetScan all = scan_offsets(offsets, 0, n_reads, K, max_len); // validates every offset before anything is queued
	if (!all.ok) return DBGK_ERR_ARG;
	const bool all_equal = n_reads && all.len_min == all.len_max;
	const uint64_t src_words = n_reads ? (offsets[n_reads] + 15) >> 4 : 0;
	const bool pinned_source = n_reads && offsets[n_reads] > offsets[0] &&
	                           device_readable_host(reinterpret_cast<const char *>(packed + (offsets[0] >> 4)), (src_words - (offsets[0] >> 4)) * 4);
	bool source_in_flight = false;
	struct WaitSource { // every exit path waits for the copies that still read the caller's buffer
		dbgk_handle *h; bool &on;
		~WaitSource() { if (on && h->source_read) (void)hipEventSynchronize(h->source_read); }
	} wait_source{h, source_in_flight};
	uint64_t r0 = 0;
	while (r0 < n_reads) {
		const uint64_t base0 = offsets[r0];
		const uint64_t room = h->store_capacity > h->pending_kmers ? h->store_capacity - h->pending_kmers : 0;
		const uint64_t r_end = std::min<uint64_t>(n_reads, r0 + h->cap_reads);
		// the longest [r0, r1) whose bases fit the staging buffers (the offsets are known to be monotone)
		uint64_t r1 = (uint64_t)(std::upper_bound(offsets + r0, offsets + r_end + 1, base0 + h->cap_bases) - offsets) - 1;
		if (r1 < n_reads && r1 > r0 + 64 && (offsets[r1] & 15u)) { // end the batch where the next one starts on a word boundary, if that is near
			for (uint64_t back = 1; back <= 64; back++)
				if ((offsets[r1 - back] & 15u) == 0) { r1 -= back; break; }
		}
		OffsetScan st;
		if (all_equal) { // (no second pass over the offsets)
			const uint64_t rl = all.len_max > max_len ? max_len : all.len_max;
			st.windows = (r1 - r0) * (rl >= K ? rl - K + 1 : 0ull);
			st.len_max = st.len_min = all.len_max;
		} else {
			st = scan_offsets(offsets, r0, r1, K, max_len);
		}
		if (cut_to_room && st.windows > room && (r1 > r0 + 1 || h->pending_kmers > 0)) { // the record store takes only part of it
			uint64_t w = 0, r = r0;
			st = OffsetScan();
			for (; r < r1; r++) {
				const uint64_t len = offsets[r + 1] - offsets[r], rl = len > max_len ? max_len : len, wr = rl >= K ? rl - K + 1 : 0ull;
				if (w + wr > room && (r > r0 || h->pending_kmers > 0)) break;
				w += wr;
				st.len_max = std::max(st.len_max, len);
				st.len_min = std::min(st.len_min, len);
			}
			st.windows = w;
			r1 = r;
		}
		const uint64_t batch_windows = st.windows, len_max = st.len_max;
		if (streaming && !(h->wpart && h->wbuilt) && h->pending_kmers > 0 &&
		    (r1 == r0 || (!cut_to_room && h->pending_kmers + batch_windows > h->store_capacity))) {
			rc = flush_records(h);
			if (rc) return rc;
			if (r1 == r0) continue;
		}
		if (r1 == r0) return DBGK_ERR_ARG; // a single read larger than max_batch_bases
		StageSlot &s = h->slots[h->next_slot];
		rc = ensure_slot(h, s);
		if (rc) return rc;
		if (s.busy) {
			HIPCHK(hipEventSynchronize(s.done));
			s.busy = false;
		}
		s.acquired = false;
		const uint64_t nb = offsets[r1] - base0, nr = r1 - r0, n_words = (nb + 15) >> 4;
		const bool direct = pinned_source && (base0 & 15u) == 0;
		std::vector<std::thread> copiers;
		struct Join {
			std::vector<std::thread> &w;
			~Join() { for (auto &t : w) if (t.joinable()) t.join(); }
		} join_copiers{copiers};
		if (n_words && !direct) {
			uint32_t *dst = reinterpret_cast<uint32_t *>(s.h_bases);
			static const int want = getenv("DBGK_COPY_THREADS") ? atoi(getenv("DBGK_COPY_THREADS")) : 8;
			const uint64_t min_piece = 1u << 20; // words
			const uint64_t pieces = want > 1 ? std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)want, n_words / min_piece)) : 1;
			const uint64_t per = (n_words + pieces - 1) / pieces;
			for (uint64_t pc = 1; pc < pieces; pc++) {
				const uint64_t lo = pc * per, hi = std::min(n_words, lo + per);
				if (lo < hi) copiers.emplace_back([=]() { dbgk_internal_shift_packed(packed, base0 + 16 * lo, hi - lo, src_words, dst + lo); });
			}
			dbgk_internal_shift_packed(packed, base0, std::min(n_words, per), src_words, dst);
		}
		const int has_long = len_max > max_len ? 1 : 0;
		const int64_t uniform_len = st.len_min == st.len_max ? (int64_t)st.len_max : 0;
		rebase_offsets(s.h_offsets, offsets + r0, nr, base0);
		for (auto &t : copiers) t.join();
		rc = h2d_batch(h, s, direct ? reinterpret_cast<const char *>(packed + (base0 >> 4)) : s.h_bases, n_words * 4, nr + 1, direct);
		if (rc) return rc;
		source_in_flight = source_in_flight || direct;
		rc = launch_batch(h, nullptr, s.d_offsets, nr, nb, s.d_start, s.d_dead, has_long, uniform_len, len_max, reinterpret_cast<const uint32_t *>(s.d_bases));
		if (rc) return rc;
		h->pending_kmers += batch_windows;
		HIPCHK(hipEventRecord(s.done, h->stream));
		s.busy = true;
		h->next_slot ^= 1;
		r0 = r1;
	}
	h->host_other_bytes += other_bytes;
	return DBGK_OK;
}

extern "C" int dbgk_push_reads_packed_device(dbgk_handle *h, const uint32_t *d_packed, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases)
{
	if (!h || !d_offsets || (n_bases && !d_packed)) return DBGK_ERR_ARG;
	if (((uintptr_t)d_packed & 15u) || ((uintptr_t)d_offsets & 7u)) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (h->seed) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	const uint64_t words = bitmap_words(n_bases);
	if (words > h->dev_bits_words) {
		HIPCHK(hipStreamSynchronize(h->stream));
		if (h->dev_start) (void)hipFree(h->dev_start);
		if (h->dev_dead) (void)hipFree(h->dev_dead);
		h->dev_start = h->dev_dead = nullptr;
		h->dev_bits_words = 0;
		if (hipMalloc(&h->dev_start, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		if (hipMalloc(&h->dev_dead, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		h->dev_bits_words = words;
	}
	if (((h->part && !h->sharded) || (h->wpart && !h->wbuilt)) && h->pending_kmers > 0 && h->pending_kmers + n_bases > h->store_capacity) {
		rc = flush_records(h);
		if (rc) return rc;
	}
	rc = launch_batch(h, nullptr, d_offsets, n_reads, n_bases, h->dev_start, h->dev_dead, -1, -1, 0, d_packed);
	if (rc == DBGK_OK) h->pending_kmers += n_bases;
	return rc;
}

// Reads of ONE length (what a sequencer writes before anything trims them), 2 bits per base, back to back: no offsets travel and no
// statistics pass runs in front of level 1 (launch_batch: no_offsets).  Batches are cut at reads where a word begins.
extern "C" int dbgk_push_reads_packed_uniform(dbgk_handle *h, const uint32_t *packed, uint64_t n_reads, uint32_t read_len, uint64_t other_bytes)
{
	if (!h || (n_reads && read_len && !packed)) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (h->seed) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (n_reads == 0) return DBGK_OK;
	if (read_len == 0) { // records without a sequence count as reads (DBGgraph.cpp:274)
		h->total_reads += n_reads;
		return DBGK_OK;
	}
	const uint64_t L = read_len, K = (uint64_t)h->cfg.kmer_size, max_len = (uint64_t)h->cfg.max_read_len;
	const uint64_t rl = L > max_len ? max_len : L, w_read = rl >= K ? rl - K + 1 : 0ull;
	uint64_t align_reads = 16; // a batch begins where a word begins: at a multiple of 16 / gcd(L, 16) reads
	for (uint64_t g = 16; g >= 1; g >>= 1)
		if (L % g == 0) { align_reads = 16 / g; break; }
	uint64_t per_batch = h->cap_bases / L;
	per_batch -= per_batch % align_reads;
	if (per_batch == 0) return DBGK_ERR_ARG; // reads larger than max_batch_bases
	// (large batches in whole level-1 tiles -- 1024 / Q reads, Q a power of two: no second launch for the reads behind the last whole tile)
	if (per_batch >= 64 * 1024 && align_reads <= 16) align_reads = 1024;
	per_batch -= per_batch % align_reads;
	const bool streaming = (h->part && !h->sharded) || (h->wpart && !h->wbuilt);
	const uint64_t src_words = (n_reads * L + 15) >> 4;
	const bool pinned_source = device_readable_host(reinterpret_cast<const char *>(packed), src_words * 4);
	bool source_in_flight = false;
	struct WaitSource {
		dbgk_handle *h; bool &on;
		~WaitSource() { if (on && h->source_read) (void)hipEventSynchronize(h->source_read); }
	} wait_source{h, source_in_flight};
	int ramp = h->pending_kmers == 0 ? 0 : 4; // index into the opening batch sizes of a fresh job (below); 4: full batches
	for (uint64_t r0 = 0; r0 < n_reads;) {
		uint64_t nr = std::min(per_batch, n_reads - r0);
		// the kernels of a batch cannot start before its copy has ended, and they take about 1.5 times as long as the copy: a job of
		// several batches opens with batches of 1/8, 1/4, 1/2 and 3/4 of the full size, so that the GPU waits for an eighth of a
		// batch's copy before it has work and hardly again (profiles/r05_h2d_region_timeline.txt)
		if (ramp < 4 && n_reads > per_batch) {
			static const uint64_t kEighths[4] = {1, 2, 4, 6};
			const uint64_t want = per_batch / 8 * kEighths[ramp++];
			if (want >= align_reads) nr = std::min(nr, want - want % align_reads);
		}
		if (streaming && h->pending_kmers > 0) {
			const uint64_t room = h->store_capacity > h->pending_kmers ? h->store_capacity - h->pending_kmers : 0;
			if (w_read && nr * w_read > room) { // what the record store still takes, in whole alignment groups; else flush first
				uint64_t fit = room / w_read;
				fit -= fit % align_reads;
				if (fit == 0 || !(h->store_capacity >= h->cap_bases && !h->wpart)) {
					rc = flush_records(h);
					if (rc) return rc;
					continue;
				}
				nr = std::min(nr, fit);
			}
		}
		StageSlot &s = h->slots[h->next_slot];
		rc = ensure_slot(h, s);
		if (rc) return rc;
		if (s.busy) {
			HIPCHK(hipEventSynchronize(s.done));
			s.busy = false;
		}
		s.acquired = false;
		const uint64_t base0 = r0 * L, nb = nr * L, n_words = (nb + 15) >> 4;
		const uint32_t *src = packed + (base0 >> 4); // (base0 is a multiple of 16)
		if (!pinned_source) {
			std::vector<std::thread> copiers;
			staged_copy(s.h_bases, reinterpret_cast<const char *>(src), n_words * 4, copiers);
			for (auto &t : copiers) t.join();
		}
		rc = h2d_batch(h, s, pinned_source ? reinterpret_cast<const char *>(src) : s.h_bases, n_words * 4, 0, pinned_source);
		if (rc) return rc;
		source_in_flight = source_in_flight || pinned_source;
		rc = launch_batch(h, nullptr, nullptr, nr, nb, s.d_start, s.d_dead, L > max_len ? 1 : 0, (int64_t)L, L, reinterpret_cast<const uint32_t *>(s.d_bases));
		if (rc) return rc;
		h->pending_kmers += nr * w_read;
		HIPCHK(hipEventRecord(s.done, h->stream));
		s.busy = true;
		h->next_slot ^= 1;
		r0 += nr;
	}
	h->host_other_bytes += other_bytes;
	return DBGK_OK;
}

extern "C" int dbgk_push_reads_packed_uniform_device(dbgk_handle *h, const uint32_t *d_packed, uint64_t n_reads, uint32_t read_len)
{
	if (!h || (n_reads && read_len && !d_packed)) return DBGK_ERR_ARG;
	if ((uintptr_t)d_packed & 15u) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	if (h->seed) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (n_reads == 0) return DBGK_OK;
	if (read_len == 0) {
		h->total_reads += n_reads;
		return DBGK_OK;
	}
	const uint64_t L = read_len, K = (uint64_t)h->cfg.kmer_size, max_len = (uint64_t)h->cfg.max_read_len, n_bases = n_reads * L;
	const uint64_t rl = L > max_len ? max_len : L, windows = (rl >= K ? rl - K + 1 : 0ull) * n_reads;
	const uint64_t words = bitmap_words(n_bases);
	if (words > h->dev_bits_words) {
		HIPCHK(hipStreamSynchronize(h->stream));
		if (h->dev_start) (void)hipFree(h->dev_start);
		if (h->dev_dead) (void)hipFree(h->dev_dead);
		h->dev_start = h->dev_dead = nullptr;
		h->dev_bits_words = 0;
		if (hipMalloc(&h->dev_start, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		if (hipMalloc(&h->dev_dead, words * 4) != hipSuccess) return DBGK_ERR_NOMEM;
		h->dev_bits_words = words;
	}
	if (((h->part && !h->sharded) || (h->wpart && !h->wbuilt)) && h->pending_kmers > 0 && h->pending_kmers + windows > h->store_capacity) {
		rc = flush_records(h);
		if (rc) return rc;
	}
	rc = launch_batch(h, nullptr, nullptr, n_reads, n_bases, h->dev_start, h->dev_dead, L > max_len ? 1 : 0, (int64_t)L, L, d_packed);
	if (rc == DBGK_OK) h->pending_kmers += windows; // (exact here: the lengths are known)
	return rc;
}

// ASCII -> 2-bit on the device (the host twin is dbgk_pack_bases): d_packed gets (n_bases + 15) / 16 words; bytes outside
// ACGTNacgtn become 'A' and are added to the handle's stats.other_bytes
extern "C" int dbgk_pack_bases_device(dbgk_handle *h, const char *d_bases, uint64_t n_bases, uint32_t *d_packed)
{
	if (!h || (n_bases && (!d_bases || !d_packed))) return DBGK_ERR_ARG;
	if (((uintptr_t)d_bases & 15u) || ((uintptr_t)d_packed & 3u)) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	if (n_bases == 0) return DBGK_OK;
	const uint64_t n_chunks = (n_bases + 15) >> 4;
	hipLaunchKernelGGL(k_pack_bases, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, d_bases, n_bases, d_packed, h->d_ctr);
	hipLaunchKernelGGL(k_count_other_bytes, dim3(grid_for(h, n_chunks)), dim3(kBlock), 0, h->stream, d_bases, n_bases, h->d_ctr);
	HIPCHK(hipGetLastError());
	return DBGK_OK;
}

extern "C" int dbgk_flush(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	if (h->finalized) return DBGK_ERR_STATE;
	int rc = use_device(h);
	if (rc) return rc;
	return flush_records(h);
}

extern "C" int dbgk_store_room(dbgk_handle *h, uint64_t *pending_kmers, uint64_t *capacity_kmers)
{
	if (!h) return DBGK_ERR_ARG;
	const bool records = h->part || (h->wpart && !h->wbuilt);
	if (pending_kmers) *pending_kmers = records ? h->pending_kmers : 0;
	if (capacity_kmers) *capacity_kmers = records ? h->store_capacity : 0;
	return DBGK_OK;
}
