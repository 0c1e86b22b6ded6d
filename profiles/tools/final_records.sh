#!/bin/bash
# final_records.sh <tag>: everything the round's records come from, in one gpurun call -- run_profile.sh (kernel trace + PMC passes of the
# cfg2 bench), the plain bench line, the first-H2D region under a kernel + memory-copy trace.  Progress goes to gpurun_out/.
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash profiles/run_profile.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/h2d_$TAG -- python3 $R/profiles/tools/h2d_region.py 5 > $R/gpurun_out/h2d_$TAG.log 2>&1)
python3 profiles/tools/timeline.py gpurun_out/h2d_$TAG/*/*kernel_trace.csv gpurun_out/h2d_$TAG/*/*memory_copy_trace.csv > gpurun_out/h2d_${TAG}_timeline.txt 2>&1
tail -3 gpurun_out/h2d_$TAG.log
cat gpurun_out/bench_$TAG.json | cut -c1-600
