#!/usr/bin/env python3
"""bench.py -- k-mer hashing throughput of the MI355X path on BASELINE.json's metric.

  python bench.py --gpus 1 --steps K --warmup W            (single process)
  python bench.py --gpus N ...                              (starts its own N ranks: a torch.distributed.run child)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one whole pass of the hot path over one batch of synthetic reads that is already
resident in HBM -- as 2-bit packed read blocks, the payload SURVEY 8(d)'s timed region starts from (--input ascii: as the
ASCII bytes a FASTA reader holds; reported next to it as `value_ascii_resident`) --: reset (init_kmerset_parallel), mark read boundaries, extract + canonicalise +
hash every k-mer into 8-byte records partitioned by final table slot range, second-level
partition, build every 4096-slot region of the reference-layout table in LDS and write it out,
finalize (counters, key-0 node) -- the PARTITION engine; --engine 1 selects the DIRECT engine
(fused extract + global-atomic insert).  For N > 1 additionally the exchange of SURVEY.md section
8(e): all-reduce of the per-bucket record counts, all-to-all of the level-1 record buckets over
RCCL/xGMI to the ranks that own their slot range, every rank builds its part of ONE global table.
Workload (--config): cfg2 (default) = BASELINE.json configs[1] PER GPU at every N (10 M x 150 bp
reads, k = 31, 30x of a 50 Mb genome per GPU => weak scaling); cfg3 = configs[2], 200 M reads of a
1 Gb genome into one table of 8.6 G slots at N = 8 (25 M reads per GPU; fewer GPUs run their share).

After the timed loop at N > 1 the ranks' results (node digest, DepthStat, counts) are added up and compared with the WHOLE
job rebuilt on rank 0 by the atomic engine (`verified`; --no-verify skips it).  --backend gloo --one-gpu runs the same N-rank
flows with all ranks on GPU 0 and host-staged collectives: a correctness rehearsal for one-GPU boxes, not a measurement.

Prints ONE JSON line on rank 0 (contract in the task description).  `roofline` is the WHOLE step:
algorithmic bytes of the step (33.25 B per k-mer) / ms_per_step against the 8 TB/s
spec peak (`frac`) and against the copy bandwidth measured in the same run (`frac_of_measured`: the best of the runtime's
DtoD memcpy and the library's own 16-byte-per-lane copy kernels over 2 GiB buffers);
per kernel: its measured time (HIP events on the library's own streams), its OWN bytes and its PMC
traffic -- measured in this run at N = 1 (`traffic_source`: two child processes of this bench, one step each, under
rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE), else taken from the committed profiles/traffic_r*.json.  After the timed loop the graph of the last step is checked against the full-size golden
record of the CPU oracle (`verified`); `value_from_first_h2d` is SURVEY 8(d)'s region as written -- the same job timed from
the first host-to-device copy of the packed read blocks (page-locked host buffer) to the final table, median of REPS
repetitions, with the host packer's own rate next to it (`host_pack`) and the older ASCII / pageable variants; `also`
carries three-step runs of the other BASELINE configs (cfg3 share, cfg4, cfg5 share), each checked against the atomic
engine; and `cpu_baseline` times the real reference (oracle/_ref; the oracle port if
that binary is absent) on a bounded sample -- a reported number, not something the GPU path calls.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 33.25          # algorithmic bytes per k-mer, k=31 L=150: 1.25 B bases + 16 B node read + 16 B node write;
                       # cfg5 (k=63, 32-byte nodes): 150/88 + 64 = 65.70 B, set in main().  With the reads resident as 2-bit
                       # blocks the bases are a quarter of that: 150/120/4 + 32 = 32.31 B -- the figure used for such a step
H2D_REPS = 5           # repetitions of the region "first host-to-device copy -> final table" (SURVEY 8(d)); the median is reported
PROBE_BYTES = 2 << 30  # the copy-bandwidth probe moves this much per copy (far beyond the 256 MiB Infinity Cache)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


CONFIGS = {"cfg2": dict(reads_per_gpu=10_000_000, genome_per_gpu=50_000_000, table_slots=600_000_000, synth_cfg=2,
                        workload="cfg2: synthetic 10 M x 150 bp reads per GPU (30x of 50 Mb/GPU genome, 0.5% subst, 0.01% N), k=31"),
           "cfg2t": dict(reads_per_gpu=10_000_000, genome_per_gpu=50_000_000, table_slots=600_000_000, synth_cfg=2, trimmed=True,
                         workload="cfg2t: cfg2's reads with MIXED lengths -- 30 % of the 10 M reads trimmed to a length uniform in [60, 149] (quality-trimmed "
                                  "input as debruijn_contig really gets it; dbg_assembly_amd/workloads.py), k=31"),
           "cfg5": dict(reads_per_gpu=75_000_000, genome_per_gpu=375_000_000, table_slots=1_600_000_000, synth_cfg=5, kmer=63, sub_rate=0.001,
                        workload="cfg5 share of one GPU: synthetic 75 M x 150 bp reads (30x of 375 Mb; the stated job is 600 M reads of a 3 Gb "
                                 "genome on 8 GPUs), 0.1% subst, 0.01% N, k=63, 128-bit keys, 32-byte nodes (WIDE engine, PARITY UNPINNED: the "
                                 "reference stops at k=31)"),
           "cfg4": dict(reads_per_gpu=10_000_000, genome_per_gpu=50_000_000, table_slots=0, synth_cfg=2, kmer=17,
                        workload="cfg4: correct_error k-mer frequency table, k=17 (4^17 saturating byte counters, 16 GiB), of cfg2's reads "
                                 "(10 M x 150 bp per GPU, 0.5% subst, 0.01% N); counted through the partitioned records (KFREQ engine)"),
           "cfg3": dict(reads_per_gpu=25_000_000, genome_per_gpu=125_000_000, table_slots=1_075_000_000, synth_cfg=3,
                        workload="cfg3: synthetic 200 M x 150 bp reads of a 1 Gb genome at N=8 (25 M reads, 125 Mb, 1.075 G slots per GPU; "
                                 "fewer GPUs run that share of it), 0.5% subst, 0.01% N, k=31, ONE table over all GPUs")}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)   # (the first steps run before the clocks have settled: 14.9 against 14.4 ms with one warm-up step)
    ap.add_argument("--config", choices=["cfg2", "cfg2t", "cfg3", "cfg4", "cfg5"], default="cfg2",
                    help="cfg2 (default, the weak-scaling line): BASELINE configs[1] PER GPU -- 10 M x 150 bp reads, 50 Mb of genome "
                         "and 600 M table slots per GPU; cfg3: BASELINE configs[2] split over the GPUs that are there -- 25 M reads, "
                         "125 Mb of genome and 1.075 G slots per GPU, i.e. at N = 8 the stated job: 200 M reads of a 1 Gb genome "
                         "into ONE table of 8.6 G slots (~5.6 G nodes); cfg5: one GPU's share of BASELINE configs[4] (k = 63, WIDE engine, "
                         "single GPU only); cfg4: BASELINE configs[3], the correct_error k-mer frequency table (k = 17) of cfg2's reads, "
                         "KFREQ engine; N > 1: every GPU counts its reads, the tables are combined by a saturating reduce-scatter")
    ap.add_argument("--reads-per-gpu", type=int, default=None)
    ap.add_argument("--genome-per-gpu", type=int, default=None)
    ap.add_argument("--kmer", type=int, default=None)
    ap.add_argument("--table-slots", type=int, default=None, help="per GPU; rounded up by find_next_prime")
    ap.add_argument("--engine", type=int, default=2, help="1 = DIRECT (global atomics), 2 = PARTITION (default)")
    ap.add_argument("--exchange", choices=["records", "nodes"], default="records",
                    help="N > 1: 'records' = slot-range ownership, level-1 buckets exchanged (default); "
                         "'nodes' = local tables, aggregated nodes exchanged by hash owner")
    ap.add_argument("--passes", type=int, default=0,
                    help="cfg5, N > 1: passes over the input (0 = as few as the level-1 fan-out needs: 3 at N = 8, where the 12.8 G-slot "
                         "table has 3052 level-1 buckets and a pass's records, 35 GB sent + 35 GB received, fit beside the 51 GB shard)")
    ap.add_argument("--exchange-chunks", type=int, default=8,
                    help="N > 1, records flow: pieces the level-1 buckets travel in (the build of a piece overlaps "
                         "the transfer of the next); 1 = one all-to-all, then the build")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N > 1 code path (process group, slot-range shard, exchange in pieces) with the ranks "
                         "there are, even one: rehearsal of the multi-GPU flow on a one-GPU box")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="N > 1 transport: nccl = RCCL over xGMI (the product path); gloo = the same flow with every collective staged "
                         "through host memory (multigpu.py 'transport') -- with --one-gpu the rehearsal of N ranks on a one-GPU box")
    ap.add_argument("--one-gpu", action="store_true",
                    help="every rank uses GPU 0 (needs --backend gloo: RCCL refuses two ranks on one device); correctness rehearsal, "
                         "the throughput printed is meaningless")
    ap.add_argument("--no-verify", action="store_true",
                    help="N > 1: skip the check after the timed loop (rank 0 rebuilds the WHOLE job with the atomic engine and compares)")
    ap.add_argument("--cpu-sample-reads", type=int, default=1_000_000)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-buffer steps after the timed loop (profiling passes)")
    ap.add_argument("--input", choices=["packed", "ascii"], default="packed",
                    help="form of the reads resident in HBM when the timed region starts: 2-bit packed blocks (default; what SURVEY 8(d)'s "
                         "region hands to the device) or ASCII bytes")
    ap.add_argument("--no-also", action="store_true", help="default cfg2 run at N = 1: skip the three-step runs of cfg3 / cfg4 / cfg5")
    ap.add_argument("--brief", action="store_true", help="the timed loop and its check only (no H2D legs, probes, CPU baseline, also-runs)")
    ap.add_argument("--plan-only", action="store_true",
                    help="print the geometry `--gpus N` would run with -- global table, level-1 buckets and their owners, records and bytes per "
                         "rank -- as one JSON line and leave: no GPU, no process group (dbgk_plan_partition); what the first run on a real "
                         "multi-GPU node starts from")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: do not measure roofline.traffic in this run (two child processes of this bench under rocprofv3 --pmc, "
                         "one step each); the committed profiles/traffic_r*.json is used instead")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="per-launch HBM bytes of the dominant kernel from a separate rocprofv3 --pmc run")
    args = ap.parse_args()
    for key in ("reads_per_gpu", "genome_per_gpu", "table_slots"):
        if getattr(args, key) is None:
            setattr(args, key, CONFIGS[args.config][key])
    if args.kmer is None:
        args.kmer = CONFIGS[args.config].get("kmer", 31)
    if args.config == "cfg5":
        args.engine = 5   # capi.ENGINE_WIDE
    if args.config == "cfg4":
        args.engine = 3   # capi.ENGINE_KFREQ
    if args.one_gpu and args.backend != "gloo":
        ap.error("--one-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
    return args


def measured_traffic(args, size, kernel):
    """HBM bytes per launch of `kernel` (None = sum over all kernels of the step) from the committed PMC
    passes (profiles/traffic_r*.json: one workload per file, or a list under "workloads"), only when this run is the
    workload -- reads, k, table, engine, form of the resident input -- those passes were taken on; else None."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json"))):
        with open(path) as fh:
            t = json.load(fh)
        for entry in t.get("workloads", [t]):
            w = entry.get("workload", {})
            per = entry.get("bytes_per_launch", {})
            if (w.get("config", "cfg2"), w.get("reads_per_gpu"), w.get("kmer"), w.get("table_slots"), w.get("engine"), w.get("input", "ascii")) == \
                    (args.config, args.reads_per_gpu, args.kmer, size, args.engine, args.input) and (kernel is None or kernel in per):
                best = sum(per.values()) if kernel is None else per[kernel]
    return best


TRAFFIC_KERNELS = ("k_extract_scatter", "k_scatter_l2", "k_build_regions", "k_kf_build_blocks", "k_wide_scatter_l1", "k_wide_scatter_l2",
                   "k_wide_build_regions", "k_prefix")   # the kernels of a step (k_pack_bases packs the resident input once, before the timed steps)


def live_traffic(args):
    """HBM bytes of ONE step, measured in this run: two child processes of this bench (--steps 1 --warmup 0 --brief, same workload)
    under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- counters alone, separate passes, the program itself behind `--`,
    as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes; values are KiB, FETCH_SIZE doubled for these kernels'
    coalesced streaming reads, WRITE_SIZE as is (the same arithmetic as profiles/make_traffic.py).  -> {kernel: bytes} or None."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None
    # this process itself runs under a profiler (rocprofv3 -- python3 bench.py ...): its children would inherit the tool -- no nesting
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    work = tempfile.mkdtemp(prefix="dbgk_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    sums = {"FETCH_SIZE": collections.defaultdict(float), "WRITE_SIZE": collections.defaultdict(float)}
    try:
        for counter in sums:
            out_dir = os.path.join(work, counter)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--steps", "1", "--warmup", "1", "--brief", "--no-live-traffic", "--config", args.config, "--input", args.input,
                   "--engine", str(args.engine), "--reads-per-gpu", str(args.reads_per_gpu), "--genome-per-gpu", str(args.genome_per_gpu),
                   "--table-slots", str(args.table_slots), "--kmer", str(args.kmer)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                print("live traffic: the %s pass failed (rc %d): %s" % (counter, r.returncode, r.stderr[-300:]), file=sys.stderr)
                return None
            # the child runs one warm-up step and one timed step (first-step one-off work is excluded): only the launches of the LAST
            # step count -- everything from the last dbgk_reset (k_zero_list) on
            rows = []
            for f in files:
                rows += [row for row in csv.DictReader(open(f)) if row.get("Counter_Name") == counter]
            rows.sort(key=lambda row: int(row.get("Start_Timestamp") or row.get("Dispatch_Id") or 0))
            resets = [i for i, row in enumerate(rows) if "k_zero_list" in row["Kernel_Name"]]
            for row in rows[(resets[-1] if resets else 0):]:
                k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("dbgk::", "")
                if any(k.startswith(w) for w in TRAFFIC_KERNELS):
                    sums[counter]["k_prefix_*" if k.startswith("k_prefix") else k] += float(row["Counter_Value"])
    except Exception as e:  # noqa: BLE001  (a profiler that is missing or refuses to run must not take the bench line with it)
        print("live traffic: %s" % e, file=sys.stderr)
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)
    kernels = sorted(set(sums["FETCH_SIZE"]) | set(sums["WRITE_SIZE"]))
    if not kernels:
        return None
    return {k: 2.0 * sums["FETCH_SIZE"][k] * 1024 + sums["WRITE_SIZE"][k] * 1024 for k in kernels}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, genome_len):
    """Time the reference's own pthreaded CPU path (oracle/_ref/ref_dbg) on the first
    --cpu-sample-reads reads of rank 0's workload, at -t = the box's cores and at the reference's default
    -t 10, and the oracle port with the same reads pre-loaded in memory (no file parsing); falls back to the
    port alone when the binary did not travel.  Checker code: used here only as the thing timed NEXT TO the
    GPU path.  Returns (main record, list of variants)."""
    import tempfile
    from oracle import oracle_py as O
    import ctypes as C
    # threads actually used: the box's CPU share for one GPU (16), never more than the cores visible
    visible = len(os.sched_getaffinity(0))
    cores = min(visible, args.cpu_threads)
    n = args.cpu_sample_reads
    P = O.synth_params(genome_len, 150, sub_rate=CONFIGS[args.config].get("sub_rate", 0.005), cfg=CONFIGS[args.config]["synth_cfg"])
    init = max(2.2 * n * (150 - args.kmer + 1) / 1e9 * 0.75, 0.001)  # distinct <= kmers; load <= ~0.6
    host = "%s, %d logical CPUs on the box, %d visible to this process" % (cpu_model(), os.cpu_count() or 0, visible)
    sample = "first %d reads of the N=1 workload (%d k-mers), -i %.3f -b 10000" % (n, n * (150 - args.kmer + 1), init)

    def port(threads):
        bases, offsets = O.synth_reads(P, 0, n)
        t0 = time.perf_counter()
        res = O.build_graph(files_mem=[(bases, offsets)], k=args.kmer, threads=threads, init_hash_size=init)
        dt = time.perf_counter() - t0
        return {"value": res.total_kmers / dt / 1e6, "unit": "M k-mers/s", "cores": threads, "kind": "port", "host": host,
                "sample": sample + ", -t %d, reads pre-loaded in memory (no file parsing), wall %.2f s" % (threads, dt)}

    if args.kmer > 32:
        # the reference stops at k = 31: the CPU figure next to the WIDE engine is this build's own single-threaded
        # restatement of the 128-bit rules (oracle/wide_oracle.cpp; PARITY UNPINNED), reads pre-loaded in memory
        n_w = n
        bases, offsets = O.synth_reads(P, 0, n_w)
        t0 = time.perf_counter()
        nodes, total = O.wide_build(bases, offsets, args.kmer, 250)
        dt = time.perf_counter() - t0
        return {"value": n_w * (150 - args.kmer + 1) / dt / 1e6, "unit": "M k-mers/s", "cores": 1, "kind": "port", "host": host,
                "sample": "first %d reads of the N=1 workload (%d k-mers), 128-bit restatement, one thread, reads pre-loaded in memory, wall %.2f s"
                          % (n_w, n_w * (150 - args.kmer + 1), dt)}, []
    if not O.have_ref():
        return port(cores), []
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "sample.fa")
        O.lib().orc_synth_write_file(C.byref(P), 0, n, os.fsencode(fa), 2, 0)
        libf = os.path.join(tmp, "reads.lib")
        open(libf, "w").write(fa + "\n")
        for threads in dict.fromkeys([cores, min(10, visible)]):  # the box's share, and the reference's default -t 10
            js = O.ref_build(libf, k=args.kmer, max_read_len=250, threads=threads, init_hash_size=init,
                             buffer_num=10000, fmt=2, timeout=900)
            out.append({"value": js["kmers"] / js["wall_s"] / 1e6, "unit": "M k-mers/s", "cores": threads, "kind": "reference", "host": host,
                        "sample": sample + ", -t %d, one-line FASTA from local disk, wall %.2f s" % (threads, js["wall_s"])})
        if visible > cores:
            # -t = EVERY core visible on the box ("the GPU box's host cores"), on a fifth of the sample: the reference's update phase
            # scans every block once per thread (DBGgraph.cpp:130-150), so more threads than ~16 make it slower, not faster
            n_all = max(n // 5, 10000)
            fa2, lib2 = os.path.join(tmp, "sample_all.fa"), os.path.join(tmp, "reads_all.lib")
            O.lib().orc_synth_write_file(C.byref(P), 0, n_all, os.fsencode(fa2), 2, 0)
            open(lib2, "w").write(fa2 + "\n")
            js = O.ref_build(lib2, k=args.kmer, max_read_len=250, threads=visible, init_hash_size=max(init / 5, 0.001), buffer_num=10000, fmt=2, timeout=900)
            out.append({"value": js["kmers"] / js["wall_s"] / 1e6, "unit": "M k-mers/s", "cores": visible, "kind": "reference", "host": host,
                        "sample": "first %d reads of the N=1 workload (%d k-mers), -t %d = all cores visible, one-line FASTA from local disk, wall %.2f s"
                                  % (n_all, js["kmers"], visible, js["wall_s"])})
    out.append(port(cores))
    return out[0], out[1:]


def golden_cfg2(args, world, n_reads, genome_len):
    """tests/golden/cfg2_full.json (made by the CPU oracle at FULL size) when this run is that workload"""
    path = os.path.join(ROOT, "tests", "golden", "cfg2_full.json")
    if world != 1 or not os.path.exists(path):
        return None
    with open(path) as fh:
        gold = json.load(fh)
    if args.config != "cfg2" or (gold["n_reads"], gold["genome_len"], gold["k"]) != (n_reads, genome_len, args.kmer):
        return None
    return gold


def verify_whole_job(args, torch, capi, g, res, P, world, rank, dev_index, device, n_reads, size, sharded):
    """N > 1, after the timed loop: what the ranks hold TOGETHER against a rebuild of the WHOLE job on rank 0 by a different
    code path -- the atomic engine (DIRECT; k > 32: the WIDE engine without records): global atomics on one table, no records,
    no partition, no exchange; itself pinned to the oracle / the reference's dumps in tests/.  Compared: node count, k-mer
    total, the order-independent node digest (a sum over nodes, so the ranks' digests add up) and DepthStat (contig.cpp:119-181).
    A mismatch ends the run with an error; a rebuild that cannot run (memory) is reported as such."""
    wide = args.engine == capi.ENGINE_WIDE
    ls = g.link_stats(2)
    depth = [int(x) for x in ls.depth_stat]
    got = sum_u64_over_ranks([g.digest()] + depth, torch, device)
    if not sharded:
        # hash-ownership flow: every rank's table carries a key-0 node and only rank 0's is real -- the placeholders are empty
        # nodes (eight zero counters each in DepthStat; their digest term is not known here, so the digest is left out)
        got[1] -= 8 * (world - 1)
    if rank != 0:
        return None
    v_size = size if sharded else capi.find_next_prime_ref(size * world)
    node_bytes = 32 if wide else 16
    free = torch.cuda.mem_get_info()[0]
    need = v_size * node_bytes + n_reads * 200 + (6 << 30)
    if free < need:
        return "not run: %.1f GB free on rank 0, the rebuild of the whole job needs %.1f" % (free / 1e9, need / 1e9)
    try:
        with capi.Graph(k=args.kmer, table_slots=v_size, max_read_len=250, device=dev_index,
                        engine=capi.ENGINE_WIDE if wide else capi.ENGINE_DIRECT, expected_kmers=0) as v:
            for r in range(world):
                rb, ro, rnb = v.synth_reads_device(P, r * n_reads, n_reads)
                v.push_reads_device(rb.ptr, ro.ptr, n_reads, rnb)
                v.sync()
                rb.free()
                ro.free()
            st = v.finalize()
            want = [v.digest()] + [int(x) for x in v.link_stats(2).depth_stat]
            want_scalars = (int(st.count), int(st.total_kmers), int(st.total_reads))
    except capi.DbgkError as e:   # the check could not be made (memory): say so, the measurement stands
        return "not run: the rebuild of the whole job on rank 0 failed (%s)" % e
    got_scalars = (int(res["count"]), int(res["total_kmers"]), int(res["total_reads"]))
    same = got_scalars == want_scalars and got[1:] == want[1:] and (got[0] == want[0] or not sharded)
    if not same:
        sys.exit("bench.py: the graph the %d ranks hold together differs from the whole job rebuilt on rank 0 with the atomic engine "
                 "(count / k-mers / reads %r, expected %r; digest %x, expected %x; DepthStat %s)"
                 % (world, got_scalars, want_scalars, got[0], want[0], "equal" if got[1:] == want[1:] else "differs"))
    return ("node count, k-mer total, %sDepthStat of the last timed step, added over the %d ranks == the WHOLE job (%d reads) rebuilt on "
            "rank 0 with the atomic engine (global atomics on one table: no records, no exchange)"
            % ("node digest and " if sharded else "", world, n_reads * world))


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: start the N ranks ourselves --
    a fresh `python -m torch.distributed.run` child (one process per GPU over RCCL), its stdout (rank 0's one JSON
    line) relayed, its exit status returned.  This process never imports torch or touches HIP, so nothing that
    initialised a GPU is ever replaced by another program."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def start_ranks(args, torch, dist, world, local_rank, single=False):
    """device of this rank + the process group (RCCL, or gloo for the host-staged rehearsal)"""
    dev_index = 0 if args.one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 or single:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = dict(rank=0, world_size=1) if world == 1 else {}
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, **kw)
        else:
            dist.init_process_group("gloo", **kw)
    return dev_index, device


def sum_u64_over_ranks(values, torch, device):
    """all-reduce of unsigned 64-bit sums (mod 2^64): as two 32-bit halves each, so that nothing overflows on the way"""
    from dbg_assembly_amd.multigpu import all_reduce
    import torch.distributed as dist
    halves = []
    for v in values:
        v = int(v) & ((1 << 64) - 1)
        halves += [v & 0xFFFFFFFF, v >> 32]
    t = torch.tensor(halves, dtype=torch.int64, device=device)
    all_reduce(t, op=dist.ReduceOp.SUM)
    h = [int(x) for x in t.cpu().numpy()]
    return [(h[2 * i] + (h[2 * i + 1] << 32)) & ((1 << 64) - 1) for i in range(len(values))]


def cpu_baseline_kfreq(args, genome_len):
    """The producer of the correct_error frequency table (`kmerfreq`) is not part of the reference, so the CPU figure next
    to cfg4 is the oracle's restatement of the counting (extraction pinned to the reference; numpy accumulation), one
    thread, on a bounded sample.  Checker code, timed NEXT TO the GPU path."""
    from oracle import oracle_py as O
    n = min(args.cpu_sample_reads, 2_000_000)
    P = O.synth_params(genome_len, 150, cfg=CONFIGS["cfg4"]["synth_cfg"])
    bases, offsets = O.synth_reads(P, 0, n)
    t0 = time.perf_counter()
    import numpy as np
    keys = []
    raw = bases.tobytes()
    for i in range(n):
        km, _, _ = O.parse_read(raw[int(offsets[i]):int(offsets[i + 1])], args.kmer, 250)
        keys.append(km)
    uniq, cnt = np.unique(np.concatenate(keys), return_counts=True)
    dt = time.perf_counter() - t0
    return {"value": n * (150 - args.kmer + 1) / dt / 1e6, "unit": "M k-mers/s", "cores": 1, "kind": "port",
            "host": "%s, %d logical CPUs" % (cpu_model(), os.cpu_count() or 0),
            "sample": "first %d reads of the N=1 workload (%d k-mers, %d distinct canonical), oracle extraction + numpy.unique, reads pre-loaded, "
                      "wall %.2f s (the reference holds no producer of this table: SURVEY 8(c))" % (n, n * (150 - args.kmer + 1), len(uniq), dt)}


class Ctx:
    """what every run of this process shares: the libraries, this rank's place in the job, its device"""


def setup_process(args):
    import torch
    import torch.distributed as dist
    from dbg_assembly_amd import capi
    c = Ctx()
    c.torch, c.dist, c.capi = torch, dist, capi
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    c.rank = int(os.environ.get("RANK", "0"))
    c.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if c.world != args.gpus:
        if c.world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = c.world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    c.multi = c.world > 1 or args.force_sharded
    c.dev_index, c.device = start_ranks(args, torch, dist, c.world, c.local_rank, single=args.force_sharded)
    c.copy_bw = None
    c.copy_bw_memcpy = None
    return c


def copy_bandwidth(ctx, g):
    """the measured-HBM denominator, once per process (N = 1): best of the runtime's DtoD memcpy and the library's own copy kernels"""
    if ctx.copy_bw is None and ctx.world == 1:
        try:
            ctx.copy_bw, ctx.copy_bw_memcpy = g.copy_bandwidth_detail(PROBE_BYTES, 5)
        except Exception as e:  # noqa: BLE001
            print("copy bandwidth probe failed: %s" % e, file=sys.stderr)
    return ctx.copy_bw


def rccl_record(ctx, args):
    """N > 1: who ran where -- a run that put two ranks on one device, or fell back to gloo, is visible in the line"""
    torch, dist = ctx.torch, ctx.dist
    if not (ctx.world > 1 or args.force_sharded):
        return None
    props = torch.cuda.get_device_properties(ctx.dev_index)
    mine = {"rank": ctx.rank, "local_rank": ctx.local_rank, "device_ordinal": ctx.dev_index, "name": props.name,
            "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
            "uuid": str(getattr(props, "uuid", ""))}
    everyone = [None] * ctx.world
    if ctx.world > 1:
        dist.all_gather_object(everyone, mine)
    else:
        everyone = [mine]
    backend = dist.get_backend()
    rec = {"world": ctx.world, "backend": backend + (" (RCCL over xGMI)" if backend == "nccl" else " (host-staged rehearsal, NOT a measurement)"),
           "ranks": everyone, "distinct_devices": len({(r["pci_bus_id"], r["uuid"]) for r in everyone})}
    try:
        rec["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        pass
    # what the collectives of the timed steps cost, by stage (multigpu.stage: device events for RCCL, wall clock for host-staged
    # transports; sums over the steps, bytes per rank): one SCALE line explains its own efficiency.  Rank 0's view.
    from dbg_assembly_amd.multigpu import stage_summary
    torch.cuda.synchronize()
    st = stage_summary()
    rec["stages"] = {k: {"calls": v["calls"], "ms": round(v["ms"], 3), "bytes_out": v["bytes_out"], "bytes_in": v["bytes_in"], "peers": v["peers"]}
                     for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])}
    rec["stages_note"] = "summed over warm-up and timed steps; a collective's ms include the wait for the slowest rank to arrive"
    rec["collective_timeout_s"] = float(os.environ.get("DBGK_COLLECTIVE_TIMEOUT_S", "120"))
    return rec


def run_kfreq(args, ctx, brief=False):
    """--config cfg4: one step = extract every k-mer of the resident reads into records, partition them by 64-KiB block of the
    table (two levels), add every block up in LDS and write it out (every block: nothing zeroes the 4^k counters beforehand),
    table summary kept on the way.  N > 1: every rank counts its own
    reads into a whole table, then the tables are combined by the saturating reduce-scatter of multigpu.kfreq_reduce
     inside the timed step.  Returns the result record on rank 0."""
    torch, dist, capi = ctx.torch, ctx.dist, ctx.capi
    from dbg_assembly_amd.multigpu import all_reduce, kfreq_reduce, kfreq_slice_bounds, wrap_device_memory
    world, rank, dev_index, device = ctx.world, ctx.rank, ctx.dev_index, ctx.device
    multi = world > 1
    n_reads, k = args.reads_per_gpu, args.kmer
    kpr = 150 - k + 1
    genome_len = args.genome_per_gpu * world
    P = capi.synth_params(genome_len, 150, cfg=CONFIGS["cfg4"]["synth_cfg"])
    g = capi.Graph(k=k, table_slots=0, max_read_len=250, device=dev_index, engine=capi.ENGINE_KFREQ, expected_kmers=n_reads * kpr)
    d_bases, d_off, nb = g.synth_reads_device(P, rank * n_reads, n_reads)
    packed_in = args.input == "packed"
    d_packed = g.pack_bases_device(d_bases.ptr, nb) if packed_in else None

    def step():
        g.reset()
        if packed_in:
            g.push_reads_packed_uniform_device(d_packed.ptr, n_reads, 150)
        else:
            g.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
        st = g.finalize()
        if multi:
            kfreq_reduce(g, device)
        return st

    def fence():
        g.sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    g.sync()
    g.reset_timings()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = step()
    fence()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tm = g.timings()
    ms_per_step = dt / args.steps * 1e3
    kmers_step = int(st.stored_kmers)
    distinct = int(st.count)
    # after the timed region (N = 1): the same reads through the OTHER counting path (atomics on the byte table) must give the
    # same table -- both are pinned against the oracle at smaller sizes (tests/test_kfreq.py); here at full size
    verified = None
    if not multi:
        ptr, n_counts = g.kfreq_device_counts()
        mine = wrap_device_memory(ptr, n_counts, device).view(torch.int64)
        sum_a = int(mine.sum().item()) & ((1 << 64) - 1)
        with capi.Graph(k=k, table_slots=0, max_read_len=250, device=dev_index, engine=capi.ENGINE_KFREQ, expected_kmers=0) as g2:
            g2.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
            st2 = g2.finalize()
            ptr2, _ = g2.kfreq_device_counts()
            other = wrap_device_memory(ptr2, n_counts, device).view(torch.int64)
            same = bool(torch.equal(mine, other))
            sum_b = int(other.sum().item()) & ((1 << 64) - 1)
        if not same or (int(st2.count), int(st2.stored_kmers)) != (distinct, kmers_step) or sum_a != sum_b:
            sys.exit("bench.py cfg4: the partitioned count differs from the atomic count of the same reads")
        verified = ("the 4^%d-byte table of the last timed step == the table the atomic kernel builds from the same reads (ASCII input; byte for byte), "
                    "%d distinct canonical k-mers; both paths are pinned to the oracle in tests/test_kfreq.py" % (k, distinct))
        del mine, other
    elif not args.no_verify:
        # N > 1: after the reduce-scatter rank d owns the counters of the k-mer values [bounds[d], bounds[d+1]) of the WHOLE job.
        # Two 64-bit sums over the 8-byte words of every owned range (plain, and of a mixed word), added over the ranks, against
        # the same sums over the table rank 0 builds from ALL ranks' reads with the atomic kernel (no records, no exchange)
        def sums(words):
            mixed = (words * -7046029254386353131) ^ (words >> 29)
            return [int(words.sum().item()), int(mixed.sum().item())]
        ptr, n_counts = g.kfreq_device_counts()
        bounds = kfreq_slice_bounds(n_counts, world)
        table = wrap_device_memory(ptr, n_counts, device)
        got = sum_u64_over_ranks(sums(table[bounds[rank]:bounds[rank + 1]].view(torch.int64)), torch, device)
        if rank == 0:
            free = torch.cuda.mem_get_info()[0]
            if free < n_counts + (4 << 30):
                verified = "not run: %.1f GB free on rank 0, the whole-job table needs %.1f" % (free / 1e9, n_counts / 1e9)
            else:
                try:
                    with capi.Graph(k=k, table_slots=0, max_read_len=250, device=dev_index, engine=capi.ENGINE_KFREQ, expected_kmers=0) as g2:
                        for r in range(world):
                            rb, ro, rnb = g2.synth_reads_device(P, r * n_reads, n_reads)
                            g2.push_reads_device(rb.ptr, ro.ptr, n_reads, rnb)
                            g2.sync()
                            rb.free()
                            ro.free()
                        st2 = g2.finalize()
                        ptr2, _ = g2.kfreq_device_counts()
                        want = [v & ((1 << 64) - 1) for v in sums(wrap_device_memory(ptr2, n_counts, device).view(torch.int64))]
                        whole_distinct = int(st2.count)
                except capi.DbgkError as e:   # the check could not be made (memory): say so, the measurement stands
                    want = None
                    verified = "not run: the whole-job table on rank 0 failed (%s)" % e
                if want is not None and got != want:
                    sys.exit("bench.py cfg4: the reduce-scattered tables of the %d ranks differ from the table of the whole job "
                             "(checksums %r, expected %r)" % (world, got, want))
                if want is not None:
                    verified = ("two 64-bit checksums over the counters every rank owns after the reduce-scatter, added over the %d ranks == the same "
                            "checksums of the table the atomic kernel builds on rank 0 from ALL ranks' reads (%d distinct canonical k-mers)"
                            % (world, whole_distinct))
    rccl = rccl_record(ctx, args)
    out = None
    if rank == 0:
        base_bytes = (150.0 / kpr) / (4.0 if packed_in else 1.0)
        b_alg = base_bytes + 2.0   # bases + one counter byte read + one written
        achieved = kmers_step * b_alg / (ms_per_step * 1e-3) / 1e9
        l1_ms, l2_ms, build_ms, wall_ms = tm.insert_ms / args.steps, tm.partition_ms / args.steps, tm.build_ms / args.steps, tm.l2_build_wall_ms / args.steps
        # k >= 13: the build's regions are 64-KiB blocks of the table itself (k_kf_build_blocks): it reads every record and writes the
        # whole table once, nothing zeroes or summarises the table separately
        blocks = k >= 13 and not "kfreq_hashed=1" in os.environ.get("DBGK_TEST_HOOKS", "")
        bname = "k_kf_build_blocks" if blocks else "k_build_regions(KF)"
        own = {"k_extract_scatter_uniform": kmers_step * (base_bytes + (4.0 if blocks else 8.0)), "k_scatter_l2": kmers_step * (6.0 if blocks else 16.0),
               bname: kmers_step * (2.0 if blocks else 8.0) + (4.0 ** k if blocks else distinct * 1.0)}   # (blocks: 32-bit level-1 records, level 2 leaves 16-bit ones)
        ms = {"k_extract_scatter_uniform": l1_ms, "k_scatter_l2": l2_ms, bname: build_ms}
        copy_bw = copy_bandwidth(ctx, g)
        own_total = sum(own.values()) + (0.0 if blocks else 2.0 * 4 ** k)   # hashed form: + zeroing the table at reset and the summary pass over it
        traffic = measured_traffic(args, 0, None)
        out = {"metric": "M k-mers/s counted (k=%d, 150 bp)" % k, "value": kmers_step * world / (dt / args.steps) / 1e6, "unit": "M k-mers/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": CONFIGS["cfg4"]["workload"], "reads_per_gpu": n_reads, "kmers_per_gpu": kmers_step, "input": args.input,
                          "table_bytes": 4 ** k, "distinct_canonical_kmers": distinct if not multi else None, "engine": "kfreq (partitioned records, 64-KiB table blocks in LDS)",
                          "parallelism": "single GPU" if not multi else
                                         "reads sharded by record x%d, whole tables per GPU, saturating reduce-scatter of the counters" % world},
               "roofline": {"bound": "hbm", "kernel": "whole step: " + ("" if blocks else "reset -> ") + " -> ".join(ms) + ("" if blocks else " -> table summary"), "achieved": achieved,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                            "frac_of_measured": achieved / copy_bw if copy_bw else None, "copy_bandwidth_GBs": copy_bw,
                         "copy_bandwidth_hipmemcpy_GBs": ctx.copy_bw_memcpy,
                         "frac_of_measured_definition": "achieved / copy_bandwidth_GBs; the denominator is the best of the library's own copy kernels (16 B per lane, "
                                                        "2 GiB buffers, default and non-temporal, 1 and 2 workgroups per CU) and hipMemcpyDtoD, read + written bytes, same run",
                            "bytes_per_kmer": b_alg, "kmers_per_step": kmers_step, "step_ms": ms_per_step, "traffic": traffic,
                            "own_bytes_per_step": own_total,
                            "own_traffic_ratio": own_total / (kmers_step * b_alg),
                            "note": "the algorithmic figure (a byte counter read and written + the bases) is what a table small enough to "
                                    "stay cached would cost; the 16 GiB table is random-access, so the path moves 8-byte records instead and "
                                    "writes the table block by block (own_bytes_per_step): own_traffic_ratio times the algorithmic bytes",
                            "l2_build_wall_ms": wall_ms,
                            "kernels": {kn: {"ms_per_step": ms[kn], "own_bytes_per_step": own[kn],
                                             "own_GBs": own[kn] / (ms[kn] * 1e-3) / 1e9 if ms[kn] > 0 else None} for kn in ms}},
               "phases_ms_per_step": {"mark": tm.mark_ms / args.steps, "insert": l1_ms, "partition": l2_ms, "build": build_ms,
                                      "partition_and_build_wall": wall_ms, "merge": tm.fixup_ms / args.steps, "finalize": tm.finalize_ms / args.steps},
               "verified": verified}
        if rccl:
            out["rccl"] = rccl
        if world == 1:
            out["copy_bandwidth_GBs"] = copy_bw
            if not args.no_cpu_baseline and not brief:
                out["cpu_baseline"] = cpu_baseline_kfreq(args, genome_len)
    d_bases.free()
    d_off.free()
    if d_packed is not None:
        d_packed.free()
    g.close()
    return out


def verify_single(args, ctx, g, res, d_bases, d_off, n_reads, nb, size):
    """N = 1, a workload without a full-size oracle record (cfg3 share, cfg5 share): the graph of the last timed step against the
    SAME reads (their ASCII form) through the atomic engine -- DIRECT, or WIDE without records: global atomics on one table, no
    records, no partition; itself pinned to the oracle / the reference's dumps in tests/.  Node count, k-mer total, node digest, DepthStat."""
    capi, torch = ctx.capi, ctx.torch
    wide = args.engine == capi.ENGINE_WIDE
    got = (int(res["count"]), int(res["stored_kmers"]), g.digest(), [int(x) for x in g.link_stats(2).depth_stat])
    free = torch.cuda.mem_get_info()[0]
    need = size * (32 if wide else 16) + (4 << 30)
    if free < need:
        return "not run: %.1f GB free, the atomic engine's table needs %.1f" % (free / 1e9, need / 1e9)
    try:
        with capi.Graph(k=args.kmer, table_slots=size, max_read_len=250, device=ctx.dev_index,
                        engine=capi.ENGINE_WIDE if wide else capi.ENGINE_DIRECT, expected_kmers=0) as v:
            v.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)
            st = v.finalize()
            want = (int(st.count), int(st.stored_kmers), v.digest(), [int(x) for x in v.link_stats(2).depth_stat])
    except capi.DbgkError as e:
        return "not run: the atomic engine's rebuild failed (%s)" % e
    if got != want:
        sys.exit("bench.py: the graph built in the timed loop differs from the atomic engine's graph of the same reads "
                 "(count/kmers/digest %r, expected %r)" % (got[:3], want[:3]))
    return ("count, k-mer total, node digest and DepthStat of the last timed step == the same reads (ASCII) through the atomic engine "
            "(global atomics on one table; pinned to the oracle in tests/)")


def pack_on_host(capi, h_bases, threads):
    """ASCII -> 2-bit words with dbgk_pack_bases on `threads` host threads (ranges cut at word boundaries); -> (words, other, seconds)"""
    import threading
    import numpy as np
    nb = len(h_bases)
    words = np.zeros((nb + 15) // 16, dtype=np.uint32)
    per = ((nb + threads - 1) // threads + 15) & ~15
    other = [0] * threads

    def work(i):
        a, b = i * per, min(nb, (i + 1) * per)
        if a < b:
            other[i] = capi.pack_bases(h_bases[a:b], out=words, first_base=a)[1]
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(1, threads)]
    for t in th:
        t.start()
    work(0)
    for t in th:
        t.join()
    return words, sum(other), time.perf_counter() - t0


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def run_graph(args, ctx, brief=False):
    """the graph configurations (cfg2, cfg3, cfg5): the timed loop, the check of what it built, the legs around it.  Returns the
    result record on rank 0."""
    torch, dist, capi = ctx.torch, ctx.dist, ctx.capi
    from dbg_assembly_amd.multigpu import HipEngine, WideHipEngine, all_reduce, exchange_and_merge, sharded_finalize, wide_sharded_build
    world, rank, dev_index, device, multi = ctx.world, ctx.rank, ctx.dev_index, ctx.device, ctx.multi

    n_reads = args.reads_per_gpu
    genome_len = args.genome_per_gpu * world
    kpr = 150 - args.kmer + 1
    P = capi.synth_params(genome_len, 150, sub_rate=CONFIGS[args.config].get("sub_rate", 0.005), cfg=CONFIGS[args.config]["synth_cfg"])
    # N > 1: ONE global table of world * slots_per_gpu slots, every rank owns a contiguous slot range
    # (PARTITION engine, slot-range ownership; up to 2^34 slots in total).  --exchange nodes selects
    # the older flow (local tables, aggregated nodes shipped to hash owners).
    # cfg5 (WIDE, k = 63): the same slot-range ownership for 16-byte records; a job whose level-1 buckets (counted over all ranks)
    # exceed the kernel's fan-out, or whose records would not fit beside the table, reads its input in several passes
    wide_sharded = multi and args.engine == capi.ENGINE_WIDE and args.exchange == "records"
    sharded = multi and (args.engine == capi.ENGINE_PARTITION or wide_sharded) and args.exchange == "records"
    # cfg2 (weak scaling): the global table stays below 2^32 slots where that still leaves room (N = 8: 536 M slots per
    # GPU, load 0.4), because level 2 is fastest with <= 1024 final buckets per level-1 bucket; cfg3 IS the big table
    per_gpu_slots = args.table_slots
    if sharded and args.config == "cfg2":
        per_gpu_slots = min(per_gpu_slots, (2 ** 32 - 2 ** 22) // world)
    size = capi.find_next_prime_ref(per_gpu_slots * world if sharded else per_gpu_slots)

    g = capi.Graph(k=args.kmer, table_slots=size, max_read_len=250, device=dev_index, engine=args.engine,
                   expected_kmers=n_reads * kpr if args.engine in (capi.ENGINE_PARTITION, capi.ENGINE_WIDE) and not "wide_direct=" in os.environ.get("DBGK_TEST_HOOKS", "")
                   else 0,  # exact for fixed-length reads; WIDE: records first, the table in one pass (dbgk_wide_partition.h)
                   shard_count=world if sharded else 0, shard_index=rank if sharded else 0,
                   n_passes=args.passes if wide_sharded else 0, max_batch_bases=int(os.environ.get("DBGK_BENCH_BATCH_BASES", "0")))
    d_bases, d_off, nb = g.synth_reads_device(P, rank * n_reads, n_reads)  # inputs resident in HBM before timing
    if CONFIGS[args.config].get("trimmed"):   # mixed lengths: the reads are trimmed on the host once (setup, untimed) and go back to the device
        import numpy as np
        from dbg_assembly_amd import workloads
        t_bases, t_off = workloads.trim_reads(d_bases.to_host(np.uint8, nb), 150, workloads.trimmed_lengths(rank * n_reads, n_reads, 150))
        d_bases.free()
        d_off.free()
        nb = len(t_bases)
        d_bases, d_off = g.malloc(nb + 64), g.malloc(t_off.nbytes)
        d_bases.from_host(t_bases)
        d_off.from_host(t_off)
        kpr = float(np.maximum(t_off[1:] - t_off[:-1], args.kmer - 1).sum() - (args.kmer - 1) * n_reads) / n_reads   # mean windows per read
        del t_bases, t_off
    packed_in = args.input == "packed"
    d_packed = g.pack_bases_device(d_bases.ptr, nb) if packed_in else None  # ... as 2-bit blocks ("packed read blocks")
    g.sync()
    # WIDE (cfg5) on several GPUs: every rank builds the graph of its reads, the aggregated 32-byte nodes go to their owners
    engine = WideHipEngine(g, device) if args.engine == capi.ENGINE_WIDE else HipEngine(g, device)

    debug_mode = int(os.environ.get("DBGK_DEBUG_MODE", "0"))  # kernel timing experiments: results are wrong
    debug_l2 = int(os.environ.get("DBGK_DEBUG_L2", "0")) or int(os.environ.get("DBGK_DEBUG_BUILD", "0"))

    state = {"verify": True, "packed": packed_in}   # the first step run checksums what every rank sent against what its peers received

    fixed_len = not CONFIGS[args.config].get("trimmed")   # reads of one length travel without offsets (dbgk_push_reads_packed_uniform*)

    def push(h):
        if state["packed"] and fixed_len and not os.environ.get("DBGK_BENCH_OFFSETS"):
            h.push_reads_packed_uniform_device(d_packed.ptr, n_reads, 150)
        elif state["packed"]:
            h.push_reads_packed_device(d_packed.ptr, d_off.ptr, n_reads, nb)
        else:
            h.push_reads_device(d_bases.ptr, d_off.ptr, n_reads, nb)

    def step():
        g.reset()
        if wide_sharded:   # (pushes happen inside: once per pass)
            return wide_sharded_build(g, device, push, exchange_chunks=args.exchange_chunks, verify_exchange=state["verify"])
        push(g)
        if debug_mode:  # level-1 timing experiments leave garbage records: never run the later phases on them
            g.sync()
            return {"stored_kmers": int(n_reads * kpr), "count": 0}
        if debug_l2:    # level-2 timing experiment: the library refuses to build regions afterwards
            try:
                g.finalize()
            except capi.DbgkError:
                pass
            g.sync()
            return {"stored_kmers": int(n_reads * kpr), "count": 0}
        if sharded:
            return sharded_finalize(g, device, exchange_chunks=args.exchange_chunks, verify_exchange=state["verify"])
        st = g.finalize()
        if multi:
            return exchange_and_merge(engine)
        return {"stored_kmers": int(st.stored_kmers), "count": int(st.count)}

    def fence():
        g.sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
        state["verify"] = False
    if not args.warmup and multi:   # no warm-up step: verify the exchange once, untimed
        step()
        state["verify"] = False
    g.sync()
    g.reset_timings()
    fence()
    t0 = time.perf_counter()
    res = None
    marks = [t0]
    for _ in range(args.steps):
        res = step()
        marks.append(time.perf_counter())   # (a step ends with the counters read back: the stream has drained)
    fence()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tm = g.timings()
    per_step_ms = [(b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:])]

    total_kmers = res["stored_kmers"]  # all ranks
    ms_per_step = dt / args.steps * 1e3
    value = total_kmers / (dt / args.steps) / 1e6

    # ---- after the timed region: correctness of what was just measured, the H2D-inclusive rate ------
    verified = None
    gold = golden_cfg2(args, world, n_reads, genome_len) if not (debug_mode or debug_l2) else None
    if gold is not None and not multi:
        got = (int(res["count"]), int(res["stored_kmers"]), g.digest(), [int(x) for x in g.link_stats(2).depth_stat])
        want = (gold["count"], gold["total_kmers"], gold["digest"], gold["depth_stat"])
        if got != want:
            sys.exit("bench.py: the graph built in the timed loop differs from tests/golden/cfg2_full.json "
                     "(count/kmers/digest %r, expected %r)" % (got[:3], want[:3]))
        verified = ("count, k-mer total, node digest and DepthStat of the last timed step == tests/golden/cfg2_full.json (full size; CPU oracle"
                    + (", confirmed by the real reference" if gold.get("confirmed_by") else "") + ")")
    elif not multi and not args.no_verify and not (debug_mode or debug_l2):
        verified = verify_single(args, ctx, g, res, d_bases, d_off, n_reads, nb, size)
    if multi and not args.no_verify and not (debug_mode or debug_l2):
        verified = verify_whole_job(args, torch, capi, g, res, P, world, rank, dev_index, device, n_reads, size, sharded)

    # ---- the other form of the resident input, three steps (N = 1) ----
    other_resident = None
    if world == 1 and not multi and not brief and not (debug_mode or debug_l2):
        state["packed"] = not packed_in
        if d_packed is None:
            d_packed = g.pack_bases_device(d_bases.ptr, nb)
        step()
        g.sync()
        t1 = time.perf_counter()
        for _ in range(3):
            r2 = step()
        g.sync()
        dt1 = (time.perf_counter() - t1) / 3
        assert r2["count"] == res["count"]
        other_resident = {"input": "ascii" if packed_in else "packed", "value": n_reads * kpr / dt1 / 1e6, "unit": "M k-mers/s", "ms_per_step": dt1 * 1e3, "steps": 3}
        state["packed"] = packed_in

    # ---- SURVEY 8(d)'s region: from the first host-to-device copy of the packed read blocks to the final table (N = 1) ----
    from_h2d, host_pack = None, None
    if world == 1 and not multi and not args.no_h2d and not brief and args.config != "cfg5" and not (debug_mode or debug_l2):
        import numpy as np
        h_bases = d_bases.to_host(np.uint8, nb)
        h_off = d_off.to_host(np.uint64)
        threads = min(16, len(os.sched_getaffinity(0)))
        _, _, t_one = pack_on_host(capi, h_bases[:nb // 8], 1)
        words, other, t_all = pack_on_host(capi, h_bases, threads)
        host_pack = {"GBs_one_thread": (nb // 8) / t_one / 1e9, "GBs": nb / t_all / 1e9, "threads": threads, "ms": t_all * 1e3, "other_bytes": other,
                     "note": "dbgk_pack_bases (AVX2), ASCII bytes in -> 2-bit words out; host work OUTSIDE the timed region, like the FASTA parsing it "
                             "belongs to (SURVEY 8(d)); Python threads calling the C function"}
        p_words = torch.empty(len(words), dtype=torch.int32).pin_memory()
        p_words.numpy()[:] = words.view(np.int32)
        p_off = torch.empty(len(h_off), dtype=torch.int64).pin_memory()
        p_off.numpy()[:] = h_off.view(np.int64)

        def region(push_fn, reps):
            times = []
            for it in range(reps + 1):    # the first pass (untimed) allocates the pinned staging buffers
                g.sync()
                t1 = time.perf_counter()
                g.reset()
                push_fn()
                st2 = g.finalize()
                g.sync()
                if it:
                    times.append(time.perf_counter() - t1)
                assert int(st2.count) == int(res["count"]), "the graph built from host buffers differs"
            return times

        def rec(times, note):
            m = median(times)
            return {"value": n_reads * kpr / m / 1e6, "unit": "M k-mers/s", "ms_per_step": m * 1e3, "reps_ms": [t * 1e3 for t in times], "note": note}
        with_offsets = None
        if fixed_len:
            with_offsets = region(lambda: g.push_reads_packed_ptr(p_words.data_ptr(), p_off.data_ptr(), n_reads, other), 3)
            t_pp = region(lambda: g.push_reads_packed_uniform(p_words.data_ptr(), n_reads, 150, other), H2D_REPS)
        else:
            t_pp = region(lambda: g.push_reads_packed_ptr(p_words.data_ptr(), p_off.data_ptr(), n_reads, other), H2D_REPS)
        if gold is not None:
            assert g.digest() == gold["digest"]
        from_h2d = rec(t_pp, "SURVEY 8(d)'s timed region: reset, first host-to-device copy of the 2-bit packed read blocks (page-locked host buffer, "
                             "read by the copy engine directly), level 1 batch by batch while the next batch travels, level 2 + build, counters read back; "
                             "median of %d repetitions" % H2D_REPS)
        from_h2d["h2d_bytes_per_step"] = int(words.nbytes + (0 if fixed_len else h_off.nbytes))
        from_h2d["variants"] = {
            "packed_pinned_with_offsets": rec(with_offsets, "the same with the 64-bit offsets of the reads travelling too (reads of mixed lengths need them): +80 MB") if with_offsets else None,
            "packed_pageable": rec(region(lambda: g.push_reads_packed(words, h_off, other), 3), "packed words in pageable memory: shifted / copied into the pinned staging buffers by host threads"),
            "ascii_pinned": None, "ascii_pageable": None}
        p_bases = torch.empty(nb, dtype=torch.uint8).pin_memory()
        p_bases.numpy()[:] = h_bases
        hp = p_bases.numpy()
        from_h2d["variants"]["ascii_pinned"] = rec(region(lambda: g.push_reads(hp, h_off), 3), "ASCII bytes, page-locked caller buffer (round 3's `value_incl_h2d.pinned_source`)")
        from_h2d["variants"]["ascii_pageable"] = rec(region(lambda: g.push_reads(h_bases, h_off), 3), "ASCII bytes, pageable caller buffer (round 3's `value_incl_h2d`)")
        del p_bases, hp, p_words, p_off, words, h_bases

    rccl = rccl_record(ctx, args)
    out = None
    if rank == 0:
        kmers_step = int(round(n_reads * kpr))
        base_bytes = (float(nb) / kmers_step) / (4.0 if packed_in else 1.0)   # bases per k-mer as they are read from HBM
        # Per-kernel figures (HIP events on the library's own streams), each kernel with ITS OWN bytes:
        # level 1 reads the bases and writes one 8-byte record per k-mer, level 2 reads and writes every
        # record, the region build reads every record and writes every 16-byte table slot once.
        l1_name = "k_wide_extract_insert" if args.engine == capi.ENGINE_WIDE else "k_extract_insert" if args.engine != capi.ENGINE_PARTITION else \
                  ("k_extract_scatter_prefix" if tm.prefix_launches else "k_extract_scatter_uniform" if tm.uniform_launches else "k_extract_scatter")  # mixed lengths / equal lengths / the flat kernel
        l1_ms = tm.insert_ms / args.steps
        l2_ms, build_ms, wall_ms = tm.partition_ms / args.steps, tm.build_ms / args.steps, tm.l2_build_wall_ms / args.steps
        slots_local = size // world if sharded else size
        wide_records = args.engine == capi.ENGINE_WIDE and l2_ms > 0   # the WIDE handle went through 16-byte records
        if args.engine == capi.ENGINE_PARTITION:
            own_bytes = {l1_name: kmers_step * (base_bytes + 8.0), "k_scatter_l2": kmers_step * 16.0,
                         "k_build_regions": kmers_step * 8.0 + slots_local * 16.0}
            kernel_ms = {l1_name: l1_ms, "k_scatter_l2": l2_ms, "k_build_regions": build_ms}
        elif wide_records:
            l1_name = "k_wide_scatter_l1_uniform" if tm.uniform_launches else "k_wide_scatter_l1"
            own_bytes = {l1_name: kmers_step * (base_bytes + 16.0), "k_wide_scatter_l2": kmers_step * 32.0,
                         "k_wide_build_regions": kmers_step * 16.0 + slots_local * 32.0}
            kernel_ms = {l1_name: l1_ms, "k_wide_scatter_l2": l2_ms, "k_wide_build_regions": build_ms}
        else:
            own_bytes = {l1_name: kmers_step * (base_bytes + (32.0 if args.kmer <= 32 else 64.0))}
            kernel_ms = {l1_name: l1_ms}
        live = None
        if world == 1 and not multi and not brief and not args.no_live_traffic and args.traffic_bytes is None:
            live = live_traffic(args)
        kernels = {}
        for kname, ms in kernel_ms.items():
            traffic = live.get(kname) if live else measured_traffic(args, size, kname)
            kernels[kname] = {"ms_per_step": ms, "own_bytes_per_step": own_bytes[kname],
                              "own_GBs": own_bytes[kname] / (ms * 1e-3) / 1e9 if ms > 0 else None,
                              "pmc_traffic_bytes_per_step": traffic,
                              "pmc_GBs": traffic / (ms * 1e-3) / 1e9 if (traffic and ms > 0) else None}
        chunks = max(int(tm.partition_launches) // max(args.steps, 1), 1)
        pipeline_ms = l1_ms + (wall_ms if wall_ms > 0 else l2_ms + build_ms)   # kernels only: the concurrent pair at its wall time
        # THE roofline figure of this path: algorithmic bytes of one step (bases + one node read + one node written per
        # k-mer; 33.25 B with ASCII bases, 32.31 B when the reads are resident as 2-bit blocks) over the WHOLE step time
        # (ms_per_step: reset, mark, all kernels, finalize), against the 8 TB/s spec peak
        b_alg = base_bytes + (32.0 if args.kmer <= 32 else 64.0)   # 32-byte nodes: one read + one write = 64 B (65.70 B at k=63, ASCII)
        achieved = kmers_step * b_alg / (ms_per_step * 1e-3) / 1e9   # per GPU (every rank processes kmers_step per step)
        copy_bw = copy_bandwidth(ctx, g) if not (brief and ctx.copy_bw is None) else None
        out = {
            "metric": "M k-mers/s hashed (k=%d, 150 bp)" % args.kmer, "value": value, "unit": "M k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64" if args.kmer <= 32 else "u128",
            "data": "synthetic",
            "config": {"workload": CONFIGS[args.config]["workload"],
                       "input": ("reads resident in HBM as 2-bit packed blocks (16 bases per 32-bit word)" + (", one length: no offsets" if fixed_len else " + 64-bit offsets")) if packed_in
                                else "reads resident in HBM as ASCII bytes + 64-bit offsets",
                       "reads_per_gpu": n_reads, "kmers_per_gpu": kmers_step, "bases_per_gpu": nb, "table_slots": size,
                       "nodes": res["count"], "engine": {capi.ENGINE_PARTITION: "partition", capi.ENGINE_WIDE: "wide"}.get(args.engine, "direct"),
                       "parallelism": ("reads sharded by record x%d, k-mers owned by slot range of one global table "
                                       "(all-to-all of level-1 record buckets)" % world) if sharded else
                                      ("reads sharded by record x%d, keys owned by hash (aggregated nodes exchanged)" % world
                                       if world > 1 else "single GPU")},
            "step_ms": {"median": median(per_step_ms), "min": min(per_step_ms), "max": max(per_step_ms),
                        "note": "per step, host clock between the counter read-backs that end the steps; ms_per_step is the K-step bracket / K"},
            "roofline": {"bound": "hbm", "kernel": "whole step: " + " -> ".join(kernel_ms), "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_of_measured": achieved / copy_bw if copy_bw else None, "copy_bandwidth_GBs": copy_bw,
                         "copy_bandwidth_hipmemcpy_GBs": ctx.copy_bw_memcpy,
                         "frac_of_measured_definition": "achieved / copy_bandwidth_GBs; the denominator is the best of the library's own copy kernels (16 B per lane, "
                                                        "2 GiB buffers, default and non-temporal, 1 and 2 workgroups per CU) and hipMemcpyDtoD, read + written bytes, same run",
                         "bytes_per_kmer": b_alg, "kmers_per_step": kmers_step, "step_ms": ms_per_step,
                         "traffic": args.traffic_bytes if args.traffic_bytes is not None else (sum(live.values()) if live else measured_traffic(args, size, None)),
                         "traffic_source": "--traffic-bytes" if args.traffic_bytes is not None else
                                           ("measured in this run: two child processes of this bench (one step each) under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, "
                                            "KiB, FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM)" if live else "profiles/traffic_r*.json (committed PMC passes of this workload)"),
                         "traffic_measured_in_run": bool(live) and args.traffic_bytes is None,
                         "traffic_committed": measured_traffic(args, size, None),
                         "kernels_only_ms": pipeline_ms, "kernels_only_frac": kmers_step * b_alg / (pipeline_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "l2_build_wall_ms": wall_ms, "l2_build_chunks": chunks, "kernels": kernels},
            "phases_ms_per_step": {"mark": tm.mark_ms / args.steps, "insert": tm.insert_ms / args.steps,
                                   "partition": tm.partition_ms / args.steps, "build": tm.build_ms / args.steps,
                                   "partition_and_build_wall": tm.l2_build_wall_ms / args.steps,
                                   "merge": tm.fixup_ms / args.steps},
            "verified": verified,
        }
        if other_resident:
            out["value_%s_resident" % other_resident["input"]] = other_resident
        if from_h2d:
            out["value_from_first_h2d"] = from_h2d
            out["host_pack"] = host_pack
        if rccl:
            out["rccl"] = rccl
        if world == 1 and not brief:
            out["copy_bandwidth_GBs"] = copy_bw
            try:   # the practical random-access ceiling, next to the streaming one
                gb, ga = g.gather_bandwidth(16 << 30, 1 << 29)
                out["random_gather_64B"] = {"GBs": gb, "G_sectors_per_s": ga, "buffer_bytes": 16 << 30,
                                            "note": "random 64-byte sectors of a 16 GiB buffer, four lanes per sector"}
            except Exception as e:  # noqa: BLE001
                print("random gather probe failed: %s" % e, file=sys.stderr)

    d_bases.free()
    d_off.free()
    if d_packed is not None:
        d_packed.free()
    g.close()
    return out


def also_runs(args, ctx):
    """default run (cfg2, N = 1): three steps each of the other BASELINE configs on this GPU -- the cfg3 share (25 M reads into a
    1.075 G-slot table), cfg4 (k = 17 frequency table) and the cfg5 share (k = 63) -- so that every config's number is in the one
    line the driver records.  Each is checked against the atomic engine on the same reads."""
    import copy
    out = {}
    for cfg in ("cfg3", "cfg4", "cfg5"):
        a = copy.copy(args)
        a.config = cfg
        for key in ("reads_per_gpu", "genome_per_gpu", "table_slots"):
            setattr(a, key, CONFIGS[cfg][key])
        a.kmer = CONFIGS[cfg].get("kmer", 31)
        a.engine = {"cfg5": 5, "cfg4": 3}.get(cfg, 2)
        a.steps, a.warmup = 3, 1
        t0 = time.perf_counter()
        try:
            r = run_kfreq(a, ctx, brief=True) if cfg == "cfg4" else run_graph(a, ctx, brief=True)
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001 -- (memory on a box shared with something else): the headline stands, the gap is visible
            out[cfg] = {"error": "%s: %s" % (type(e).__name__, e)}
            continue
        rf = r["roofline"]
        out[cfg] = {"workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                    "bytes_per_kmer": rf["bytes_per_kmer"], "frac": rf["frac"], "frac_of_measured": rf["frac_of_measured"], "traffic": rf["traffic"],
                    "own_traffic_ratio": rf.get("own_traffic_ratio"), "verified": r["verified"],
                    "kernels_ms": {k: v["ms_per_step"] for k, v in rf["kernels"].items()}, "l2_build_wall_ms": rf.get("l2_build_wall_ms"),
                    "wall_s_incl_setup_and_check": time.perf_counter() - t0}
    return out


def plan_only(args):
    """the sharded geometry of run_graph for world = --gpus, computed without a device"""
    from dbg_assembly_amd import capi
    world = max(1, args.gpus)
    sharded = world > 1 and args.engine == capi.ENGINE_PARTITION and args.exchange == "records"
    per_gpu_slots = args.table_slots
    if sharded and args.config == "cfg2":
        per_gpu_slots = min(per_gpu_slots, (2 ** 32 - 2 ** 22) // world)
    size = capi.find_next_prime_ref(per_gpu_slots * world if sharded else per_gpu_slots)
    kpr = 150 - args.kmer + 1
    plans = [capi.plan_partition(size, args.reads_per_gpu * kpr, world if sharded else 0, r if sharded else 0) for r in range(world if sharded else 1)]
    p0 = plans[0]
    worst = max(plans, key=lambda p: p.table_bytes + p.level1_store_bytes + p.inbox_bytes + p.final_store_bytes)
    return {"config": args.config, "world": world, "sharded": bool(sharded), "reads_per_gpu": args.reads_per_gpu, "records_per_rank": args.reads_per_gpu * kpr,
            "table_slots_global": int(p0.table_slots), "r": int(p0.r), "level1_buckets": int(p0.level1_buckets), "final_per_level1": int(p0.final_per_level1),
            "three_level": bool(p0.three_level), "buckets_per_rank": int(p0.buckets_per_rank), "own_buckets": [int(p.own_buckets) for p in plans],
            "slot_ranges": [[int(p.slot_lo), int(p.slot_hi)] for p in plans],
            "records_per_level1_bucket": int(p0.records_per_level1_bucket), "records_per_final_bucket": int(p0.records_per_final_bucket),
            "bytes_per_rank": {"table": int(worst.table_bytes), "level1_store": int(worst.level1_store_bytes), "inbox": int(worst.inbox_bytes),
                               "final_buckets": int(worst.final_store_bytes), "reads_packed": args.reads_per_gpu * 150 // 4,
                               "total": int(worst.table_bytes + worst.level1_store_bytes + worst.inbox_bytes + worst.final_store_bytes) + args.reads_per_gpu * 150 // 4},
            "exchange_bytes_out_per_rank": int(args.reads_per_gpu * kpr * 8 * (world - 1) // world) if sharded else 0,
            "exchange_chunks": args.exchange_chunks}


def main():
    args = parse_args()
    if args.plan_only:
        print(json.dumps(plan_only(args)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    # stdout carries ONE line, the result.  RCCL prints a version banner to stdout when a communicator is
    # created (any time up to the first point-to-point transfer), so file descriptor 1 is pointed at
    # stderr for the whole run and the JSON line is written to the saved original.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ctx = setup_process(args)
    out = run_kfreq(args, ctx, brief=args.brief) if args.config == "cfg4" else run_graph(args, ctx, brief=args.brief)
    if ctx.rank == 0:
        if ctx.world == 1 and not ctx.multi and not args.brief:
            defaults = all(getattr(args, key) == CONFIGS["cfg2"][key] for key in ("reads_per_gpu", "genome_per_gpu", "table_slots"))
            if args.config == "cfg2" and defaults and args.engine == 2 and not args.no_also and not os.environ.get("DBGK_DEBUG_MODE"):
                out["also"] = also_runs(args, ctx)
            if not args.no_cpu_baseline and args.config != "cfg4":   # (k > 32: the 128-bit restatement, the reference has no such path)
                out["cpu_baseline"], out["cpu_baseline_variants"] = cpu_baseline(args, args.genome_per_gpu)
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if ctx.multi:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
