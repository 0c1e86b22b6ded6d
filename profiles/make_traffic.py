#!/usr/bin/env python3
"""profiles/traffic_r05.json (TRAFFIC_OUT) from the PMC summaries of profiles/run_profile.sh (gpurun_out/prof_<tag>/pmc_summary.csv):
HBM bytes per STEP and kernel = FETCH_SIZE (KiB; doubled for these kernels' coalesced streaming reads, MI355X_MICROARCH.md section HBM)
+ WRITE_SIZE (KiB, as is), summed over the launches of the one step each PMC pass runs.
    python profiles/make_traffic.py tag:config:reads:kmer:table_slots:engine:input[:kernel-prefixes] ..."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (profiles/run_profile.sh <tag> [--config ...]: bench.py --steps 1 "
                      "--warmup 0 --brief under the profiler), kernels of the round that made the file.  PER STEP; kernels launched once per bucket chunk are summed.  "
                      "Counter values are KiB; FETCH_SIZE doubled, WRITE_SIZE as is (MI355X_MICROARCH.md, section HBM).  Made by profiles/make_traffic.py.",
       "workloads": []}
for spec in sys.argv[1:]:
    f = spec.split(":")
    tag, config, reads, kmer, slots, engine, inp = f[0], f[1], int(f[2]), int(f[3]), int(f[4]), int(f[5]), f[6]
    want = f[7].split(",") if len(f) > 7 else ["k_extract_scatter", "k_scatter_l2", "k_build_regions", "k_kf_build_blocks", "k_wide_scatter_l1", "k_wide_scatter_l2",
                                               "k_wide_build_regions", "k_prefix"]   # (k_pack_bases packs the resident input once, before the timed steps)
    fetch, write = collections.defaultdict(float), collections.defaultdict(float)
    for r in csv.DictReader(open(os.path.join(ROOT, "gpurun_out", "prof_" + tag, "pmc_summary.csv"))):
        k = r["kernel"].split("<")[0].replace("dbgk::", "")
        if not any(k.startswith(w) for w in want):
            continue
        if k.startswith("k_prefix"):
            k = "k_prefix_* (lane prefix of the batch)"
        if r["counter"] == "FETCH_SIZE":
            fetch[k] += float(r["value"])
        elif r["counter"] == "WRITE_SIZE":
            write[k] += float(r["value"])
    per = {k: 2.0 * fetch[k] * 1024 + write[k] * 1024 for k in sorted(set(fetch) | set(write))}
    out["workloads"].append({"workload": {"config": config, "reads_per_gpu": reads, "kmer": kmer, "table_slots": slots, "engine": engine, "input": inp, "tag": tag},
                             "bytes_per_launch": per,
                             "fetch_KiB_as_counted": dict(fetch), "write_KiB": dict(write)})
json.dump(out, open(os.path.join(ROOT, "profiles", os.environ.get("TRAFFIC_OUT", "traffic_r05.json")), "w"), indent=1)
for w in out["workloads"]:
    print(w["workload"]["tag"], {k: "%.2f GB" % (v / 1e9) for k, v in w["bytes_per_launch"].items()}, "sum %.2f GB" % (sum(w["bytes_per_launch"].values()) / 1e9))
