#!/bin/bash
# pmc_sets.sh <tag> "<set1>" "<set2>" ...: one rocprofv3 --pmc pass of ONE cfg2 bench step per counter set (counters only, never
# combined with tracing; python3 itself behind `--`); prints per kernel the summed counters.  DBGK_* settings come from the environment;
# PMC_BENCH_ARGS="--config cfg5" profiles another configuration.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "$@"; do
  OUT=$R/gpurun_out/pmcs_${TAG}_$i; i=$((i+1))
  mkdir -p $OUT
  rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 0 --brief --no-live-traffic ${PMC_BENCH_ARGS:-} > $OUT/log.txt 2>&1 || echo "pmc failed: $CTRS"
  python3 - $OUT/*/*counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for f in sys.argv[1:]:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(x in k for x in ("k_extract_scatter_uniform", "k_scatter_l2", "k_build_regions", "k_wide_scatter_l1", "k_wide_scatter_l2", "k_wide_build_regions", "k_kf_", "k_extract_scatter")): continue
        k = k.split("(")[0].replace("void ", "").replace("dbgk::", "")[:44]
        d = acc.setdefault(k, collections.OrderedDict())
        d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k, d in acc.items():
    print("%-44s" % k, " ".join("%s=%.4g" % kv for kv in d.items()))
PY
done
