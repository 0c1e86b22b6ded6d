#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  profiles/run_profile.sh <tag> <bench args...>
#   1. rocprofv3 --kernel-trace --stats   -> per-kernel average duration (+ the trace: start/end of every dispatch)
#   2. separate --pmc passes              -> FETCH_SIZE, WRITE_SIZE, SQ counters (never combined with tracing)
# Summaries land in gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 9 --warmup 1 --brief "$@" > $OUT/kt.log 2>&1 || echo "kt failed"
if [ "${PMC:-1}" != "0" ]; then
# PMC=traffic: the two HBM-byte passes only (profiles/traffic_r*.json)
if [ "${PMC:-1}" = "traffic" ]; then SETS=("FETCH_SIZE" "WRITE_SIZE"); else SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE"); fi
for C in "${SETS[@]}"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --steps 1 --warmup 0 --brief "$@" > $OUT/pmc_$N.log 2>&1 || echo "pmc $N failed"
done
# one summary file: kernel, counter, value, duration_ms
{
  echo "kernel,counter,value,duration_ms"
  for f in $OUT/pmc_*/*/*counter_collection.csv; do
    python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "dbgk" not in k: continue
    k = k.split("(")[0].replace("void ", "").replace(", ", ";")  # template arguments would break the CSV
    print("%s,%s,%s,%.4f" % (k, row["Counter_Name"], row["Counter_Value"], (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))
PY
  done
} > $OUT/pmc_summary.csv
fi
cp $OUT/kt/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
# timeline of the LAST timed step of the trace: every dispatch with its start relative to the step's first memset
python3 - $OUT/kt/*/*kernel_trace.csv > $OUT/last_step_timeline.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST timed step: from the memsets in front of its level-1 kernel (its k_mark / k_add_totals, if any) to the last kernel of the
# step's tail (k_merge_nodes / k_kf_* / the wide merges); what follows (digest, link statistics, copies) is the check, not the step
l1 = [i for i, r in enumerate(rows) if any(x in r["Kernel_Name"] for x in ("k_extract_scatter", "k_wide_scatter_l1", "k_extract_insert<", "k_wide_extract_insert", "k_extract_count"))]
tail = [i for i, r in enumerate(rows) if "k_build_regions" in r["Kernel_Name"] or "k_kf_build_blocks" in r["Kernel_Name"] or "k_wide_build_regions" in r["Kernel_Name"]]
if l1:
    last_build = tail[-1] if tail else l1[-1]
    first_l1 = max(i for i in l1 if i <= last_build)
    while first_l1 - 1 in l1: first_l1 -= 1   # (a regular-tile launch + the launch for the rest of the reads)
    rows = rows[max(first_l1 - 12, 0):last_build + 6]
t0 = int(rows[0]["Start_Timestamp"])
print("kernel,start_us,end_us,dur_us,vgpr,lds")
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(", ", ";")[:70]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%s,%.1f,%.1f,%.1f,%s,%s" % (k, s / 1e3, e / 1e3, (e - s) / 1e3, r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")))
PY
cat $OUT/kernel_stats.csv | cut -c1-160
cat $OUT/last_step_timeline.csv | head -60
if [ -f $OUT/pmc_summary.csv ]; then cat $OUT/pmc_summary.csv; fi
