// dbgk_host_create.h -- part of libdbgk.so's host side (one translation unit: included by dbgk.hip, in this order).
// dbgk_create / destroy / reset / sync, flush of the record store, table resize
#pragma once

extern "C" int dbgk_create(const dbgk_config *cfg, dbgk_handle **out)
{
	if (!cfg || !out) return DBGK_ERR_ARG;
	*out = nullptr;
	const bool wide = cfg->engine == DBGK_ENGINE_WIDE;
	if (cfg->kmer_size < 1 || cfg->kmer_size > (wide ? 63 : 32)) return DBGK_ERR_ARG; // 64-bit keys: the reference's "max 31" (+32, main.cpp:100); WIDE: 128-bit keys
	if (wide && (cfg->flags & ~DBGK_FLAG_PREALLOC_STAGING)) return DBGK_ERR_ARG;
	if (cfg->max_read_len < cfg->kmer_size) return DBGK_ERR_ARG;
	const bool kfreq = cfg->engine == DBGK_ENGINE_KFREQ;
	if (kfreq && cfg->kmer_size > 18) return DBGK_ERR_ARG; // 4^18 bytes = 64 GiB
	if (!kfreq && cfg->table_slots < 3) return DBGK_ERR_ARG;
	const bool seed = cfg->engine == DBGK_ENGINE_SEEDIDX;
	if (cfg->engine != DBGK_ENGINE_AUTO && cfg->engine != DBGK_ENGINE_DIRECT && cfg->engine != DBGK_ENGINE_PARTITION && !kfreq && !seed && !wide)
		return DBGK_ERR_ARG;
	if (seed && (cfg->shard_count || (cfg->flags & DBGK_FLAG_TRACK_FIRST_SEEN))) return DBGK_ERR_ARG;

	// DBGK_TIMINGS: where the time of creating a handle goes (stderr, one line)
	static const bool lap_wanted = getenv("DBGK_TIMINGS") != nullptr;
	double laps[6] = {0, 0, 0, 0, 0, 0};
	auto clock_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double lap_t = clock_s();
	auto lap = [&](int i) { const double t = clock_s(); laps[i] += t - lap_t; lap_t = t; };
	int n_dev = 0;
	hipError_t e = hipGetDeviceCount(&n_dev);
	if (e != hipSuccess || n_dev <= 0) {
		g_last_error = "no HIP device visible (this library has no CPU fallback)";
		return DBGK_ERR_HIP;
	}
	if (cfg->device_id < 0 || cfg->device_id >= n_dev) return DBGK_ERR_ARG;
	lap(0);

	dbgk_handle *h = new (std::nothrow) dbgk_handle();
	if (!h) return DBGK_ERR_NOMEM;
	h->cfg = *cfg;
	h->device = cfg->device_id;
	h->kfreq = kfreq;
	h->seed = seed;
	h->wide = wide;
	if (seed) h->cfg.max_read_len = 0x7FFFFFFF; // contigs are never trimmed (the pos field bounds them, see push)
	h->size = kfreq ? 3 : cfg->table_slots;
	// KFREQ with a known input size runs through the PARTITION engine: occurrences are partitioned by
	// hash_code(key) % size like graph records and aggregated per key in LDS; `size` is only the modulus
	// (no node table exists), chosen so that the LDS regions stay about half empty even if every second
	// occurrence were a new key.  Without expected_kmers: direct atomics on the byte table.
	static const bool kf_direct = DBGK_EXPERIMENT_ENV("DBGK_KFREQ_DIRECT") != nullptr;
	const bool kf_part = kfreq && cfg->expected_kmers > 0 && cfg->shard_count == 0 && !kf_direct;
	// k >= 13 (a table of 2^26 bytes and more): the direct-block form -- regions ARE 64-KiB blocks of the table
	// (dbgk_partition.h, kf_slot_of_key); smaller k (or DBGK_KFREQ_HASHED=1, measurements): the hashed form
	const bool kf_hashed = dbgk_hook("kfreq_hashed") && atoi(dbgk_hook("kfreq_hashed"));
	h->kf_blocks = kf_part && cfg->kmer_size >= 13 && !kf_hashed;
	if (h->kf_blocks) {
		h->size = 1ull << (2 * cfg->kmer_size);
	} else if (kf_part) {
		const uint64_t want = std::max<uint64_t>(1ull << 26, cfg->expected_kmers / 2);
		h->size = std::min<uint64_t>(want, (1ull << 32) - (1ull << 23)) | 1ull;
	}
	h->magic = make_mod_magic(h->size);
	h->tslots = h->size;
	if ((!kfreq || kf_part) && !wide) {
		const int prc = plan_partition(h); // geometry first: a sharded handle holds only its slot range
		if (prc != DBGK_OK) {
			delete h;
			return prc;
		}
		if (kfreq) {
			if (!h->part) { // infeasible geometry: fall back to the direct table
				h->size = 3;
				h->magic = make_mod_magic(h->size);
			}
			if (!h->part) h->kf_blocks = false;
			h->geom.kf = h->part ? (h->kf_blocks ? 2u : 1u) : 0u;
			h->tslots = 0;
		}
	}

	auto fail = [&](int rc) {
		free_handle(h);
		return rc;
	};
	if (hipSetDevice(h->device) != hipSuccess) return fail(DBGK_ERR_HIP);
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) return fail(DBGK_ERR_HIP);
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		g_last_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
		return fail(DBGK_ERR_HIP);
	}
	h->n_cu = prop.multiProcessorCount;
	h->grid = h->n_cu * 8; // 8 x 256-thread blocks per CU = 32 waves/CU, the residency limit

	if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(DBGK_ERR_HIP);
	lap(1);
	if (kfreq) {
		h->n_counts = 1ull << (2 * cfg->kmer_size);
		if (h->n_counts < 64) h->n_counts = 64; // whole dwords / 8-byte groups for the scan kernels (k < 3)
		if (hipMalloc(&h->counts, h->n_counts) != hipSuccess) {
			g_last_error = "hipMalloc of the 4^k count table failed";
			return fail(DBGK_ERR_NOMEM);
		}
	} else if (wide) {
		bool werr = false;
		h->wpart = plan_wide_partition(h, &werr); // input size known: records first, the table region by region (a shard: its slot range only)
		if (werr) return fail(DBGK_ERR_ARG);
		if (hipMalloc(&h->wnodes, h->tslots * sizeof(WNode)) != hipSuccess || hipMalloc(&h->wside, kWideSideSlots * sizeof(WNode)) != hipSuccess) {
			g_last_error = "hipMalloc of the wide k-mer table failed";
			return fail(DBGK_ERR_NOMEM);
		}
		if (h->wpart) {
			const int wrc = setup_wide_partition(h);
			if (wrc != DBGK_OK) return fail(wrc);
		}
	} else if (hipMalloc(&h->table, h->tslots * sizeof(Node)) != hipSuccess) {
		g_last_error = "hipMalloc of the k-mer table failed";
		return fail(DBGK_ERR_NOMEM);
	}
	if (cfg->flags & DBGK_FLAG_TRACK_FIRST_SEEN) {
		if (h->part || h->kfreq) {
			g_last_error = "DBGK_FLAG_TRACK_FIRST_SEEN needs the DIRECT engine";
			return fail(DBGK_ERR_ARG);
		}
		if (hipMalloc(&h->first_pos, h->tslots * 8) != hipSuccess) return fail(DBGK_ERR_NOMEM);
		h->track = true;
	}
	if (hipMalloc(&h->d_ctr, sizeof(Counters)) != hipSuccess) return fail(DBGK_ERR_NOMEM);
	if (hipHostMalloc(&h->h_ctr, sizeof(Counters), hipHostMallocDefault) != hipSuccess) return fail(DBGK_ERR_NOMEM);
	h->cap_bases = cfg->max_batch_bases ? cfg->max_batch_bases : (256ull << 20);
	h->cap_reads = h->cap_bases / 16 + 1024;
	lap(2);
	{
		const int prc = setup_partition(h);
		if (prc != DBGK_OK) return fail(prc);
	}
	lap(3);
	int rc = reset_state(h);
	if (rc != DBGK_OK) return fail(rc);
	if (lap_wanted && hipStreamSynchronize(h->stream) != hipSuccess) return fail(DBGK_ERR_HIP);
	lap(4);
	if (cfg->flags & DBGK_FLAG_PREALLOC_STAGING) { // page-locking is most of it, and two threads lock two buffers in about the time of one
		int rc1 = DBGK_OK;
		std::thread second([&]() { rc1 = hipSetDevice(h->device) == hipSuccess ? ensure_slot(h, h->slots[1]) : DBGK_ERR_HIP; });
		rc = ensure_slot(h, h->slots[0]);
		second.join();
		if (rc != DBGK_OK || rc1 != DBGK_OK) return fail(rc != DBGK_OK ? rc : rc1);
	}
	if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(DBGK_ERR_HIP);
	lap(5);
	if (lap_wanted)
		fprintf(stderr, "dbgk_create (s): runtime start %.4f device+stream %.4f table %.4f record store %.4f first reset %.4f staging %.4f\n", laps[0], laps[1],
		        laps[2], laps[3], laps[4], laps[5]);
	*out = h;
	return DBGK_OK;
}

extern "C" int dbgk_destroy(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	free_handle(h);
	return DBGK_OK;
}

extern "C" int dbgk_reset(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(h->stream));
	rc = collect_spans(h);
	if (rc) return rc;
	return reset_state(h);
}

extern "C" int dbgk_sync(dbgk_handle *h)
{
	if (!h) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(h->stream));
	for (auto &s : h->slots) s.busy = false;
	return collect_spans(h);
}

extern "C" void *dbgk_stream(dbgk_handle *h) { return h ? (void *)h->stream : nullptr; }

static int build_from_records(dbgk_handle *h);
static int plan_partition(dbgk_handle *h);
static int setup_partition(dbgk_handle *h);

// PARTITION engine, streaming use: records -> table now (regions that already hold nodes are loaded back
// into LDS first), record store emptied.
static int flush_records(dbgk_handle *h)
{
	if (h->wpart && !h->finalized) return h->pending_kmers ? wide_build_from_records(h) : DBGK_OK;
	if (!h->part || h->finalized) return DBGK_OK;
	if (h->sharded) {
		g_last_error = "dbgk_flush: a sharded handle is flushed by its communicator (dbgk_comm_flush)";
		return DBGK_ERR_STATE;
	}
	if (h->pending_kmers == 0) return DBGK_OK;
	int rc = build_from_records(h);
	if (rc) return rc;
	h->incr = true;
	return clear_record_store(h);
}

// the PARTITION engine's bucket geometry is a function of the table size: a resize re-seats the nodes
// (k_rehash, global atomics) and re-plans / re-allocates the record stores.  Pending records are flushed first.
static int resize_partition_table(dbgk_handle *h, uint64_t new_slots)
{
	if (h->kfreq || h->sharded) {
		g_last_error = "dbgk_resize_table: not available for KFREQ and sharded handles";
		return DBGK_ERR_STATE;
	}
	int rc = flush_records(h);
	if (rc) return rc;
	dbgk_handle probe_cfg;           // feasibility first: nothing is touched if the new size does not fit the engine
	probe_cfg.cfg = h->cfg;
	probe_cfg.size = new_slots;
	probe_cfg.magic = make_mod_magic(new_slots);
	rc = plan_partition(&probe_cfg);
	if (rc) return rc;
	if (!probe_cfg.part) {
		g_last_error = "dbgk_resize_table: the new size does not fit the PARTITION engine's geometry";
		return DBGK_ERR_ARG;
	}
	Node *fresh = nullptr;
	if (hipMalloc(&fresh, new_slots * sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
	const TableRef dst{fresh, new_slots, make_mod_magic(new_slots)};
	if (h->incr) { // the old table holds nodes
		hipError_t e = hipMemsetAsync(fresh, 0, new_slots * sizeof(Node), h->stream);
		if (e == hipSuccess) {
			hipLaunchKernelGGL(k_rehash, dim3(grid_for(h, h->size)), dim3(kBlock), 0, h->stream, h->table, h->size, dst, h->d_ctr,
			                   (const unsigned long long *)nullptr, (unsigned long long *)nullptr);
			e = hipGetLastError();
		}
		if (e == hipSuccess) e = hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
		if (e != hipSuccess) {
			(void)hipFree(fresh);
			return hip_fail(e, "resize_table(partition)", __LINE__);
		}
		if (h->h_ctr->error & 1u) {
			(void)hipFree(fresh);
			HIPCHK(hipMemsetAsync(&h->d_ctr->error, 0, sizeof(unsigned int), h->stream));
			HIPCHK(hipStreamSynchronize(h->stream));
			return DBGK_ERR_TABLE_FULL;
		}
	}
	HIPCHK(hipStreamSynchronize(h->stream));
	if (h->stream2) HIPCHK(hipStreamSynchronize(h->stream2));
	(void)hipFree(h->table);
	h->table = fresh;
	h->size = new_slots;
	h->magic = dst.magic;
	h->cfg.table_slots = new_slots;
	free_partition_stores(h);
	rc = plan_partition(h);
	if (rc == DBGK_OK && !h->part) rc = DBGK_ERR_STATE;
	if (rc == DBGK_OK) rc = setup_partition(h);
	if (rc) return rc;
	h->zero_pending = !h->incr; // nothing built yet: the first region build writes every slot
	return clear_record_store(h);
}

extern "C" int dbgk_resize_table(dbgk_handle *h, uint64_t new_slots)
{
	if (h && h->wide) return DBGK_ERR_STATE;  // WIDE handles: dbgk_wide_export_*
	if (h && h->kfreq) return DBGK_ERR_STATE; // KFREQ handles have no node table
	if (!h || new_slots < 3) return DBGK_ERR_ARG;
	int rc = use_device(h);
	if (rc) return rc;
	rc = dbgk_sync(h);
	if (rc) return rc;
	if (new_slots == h->size) return DBGK_OK;
	if (h->part) return resize_partition_table(h, new_slots);
	Node *fresh = nullptr;
	unsigned long long *fresh_first = nullptr;
	if (hipMalloc(&fresh, new_slots * sizeof(Node)) != hipSuccess) return DBGK_ERR_NOMEM;
	if (h->track) {
		if (hipMalloc(&fresh_first, new_slots * 8) != hipSuccess) {
			(void)hipFree(fresh);
			return DBGK_ERR_NOMEM;
		}
		if (hipMemsetAsync(fresh_first, 0xFF, new_slots * 8, h->stream) != hipSuccess) {
			(void)hipFree(fresh);
			(void)hipFree(fresh_first);
			return DBGK_ERR_HIP;
		}
	}
	TableRef dst{fresh, new_slots, make_mod_magic(new_slots)};
	hipError_t e = hipMemsetAsync(fresh, 0, new_slots * sizeof(Node), h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_rehash, dim3(grid_for(h, h->size)), dim3(kBlock), 0, h->stream, h->table, h->size, dst, h->d_ctr,
		                   (const unsigned long long *)h->first_pos, fresh_first);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) {
		(void)hipFree(fresh);
		if (fresh_first) (void)hipFree(fresh_first);
		return hip_fail(e, "resize_table", __LINE__);
	}
	if (h->h_ctr->error & 1u) { // new table too small for the existing nodes: keep the old one
		(void)hipFree(fresh);
		if (fresh_first) (void)hipFree(fresh_first);
		HIPCHK(hipMemsetAsync(&h->d_ctr->error, 0, sizeof(unsigned int), h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		return DBGK_ERR_TABLE_FULL;
	}
	(void)hipFree(h->table);
	if (h->track) {
		(void)hipFree(h->first_pos);
		h->first_pos = fresh_first;
	}
	h->table = fresh;
	h->size = new_slots;
	h->tslots = new_slots;
	h->magic = dst.magic;
	h->cfg.table_slots = new_slots;
	return DBGK_OK;
}

// ---------------------------------------------------------------------------------------------
// the hot path
// ---------------------------------------------------------------------------------------------
