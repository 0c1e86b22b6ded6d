#!/bin/bash
# level-2 workgroups per CU (DBGK_L2_WG_PER_CU); meaningful with a library built with -DDBGK_L2_THREADS=512
for w in 1 2; do
  DBGK_L2_WG_PER_CU=$w timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/l2wg_$w.json 2> gpurun_out/l2wg_$w.err || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/l2wg_$w.json'));print($w, round(d['ms_per_step'],3), d['roofline']['all_kernels_ms'], d['config']['nodes'])"
done
