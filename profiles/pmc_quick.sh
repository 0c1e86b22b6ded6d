#!/bin/bash
# profiles/pmc_quick.sh <tag> "<counters>" [env settings ...] -- one rocprofv3 --pmc pass of one bench step (never combined with tracing);
# prints per kernel: counter sums and duration.  The environment settings are exported before the profiler starts
# (the program after `--` is python3 itself).
TAG=$1; CTRS=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-h2d > $OUT/log.txt 2>&1 || echo "pmc failed"
python3 - $OUT/*/*counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for f in sys.argv[1:]:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "dbgk" not in k: continue
        k = k.split("(")[0].replace("void ", "")[:60]
        d = acc.setdefault(k, collections.OrderedDict())
        d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k, d in acc.items():
    print(k, " ".join("%s=%.4g" % kv for kv in d.items()))
PY
