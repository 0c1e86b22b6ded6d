#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  profiles/run_profile.sh <tag> <bench args...>
#   1. rocprofv3 --kernel-trace --stats   -> per-kernel average duration
#   2. separate --pmc passes              -> FETCH_SIZE, WRITE_SIZE, SQ counters (never combined with tracing)
# Summaries land in gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 9 --warmup 1 --no-cpu-baseline "$@" > $OUT/kt.log 2>&1 || echo "kt failed"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/pmc_$N.log 2>&1 || echo "pmc $N failed"
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
cp $OUT/kt/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
cat $OUT/kernel_stats.csv | cut -c1-160
cat $OUT/pmc_summary.csv
