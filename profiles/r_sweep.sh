#!/bin/bash
# level-1 bucket width sweep (DBGK_PART_R): n1 = size >> r level-1 buckets, n2 = 2^(r-12) final buckets per level-1 bucket
for r in 20 21 22; do
  DBGK_PART_R=$r timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r_$r.json 2> gpurun_out/r_$r.err || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/r_$r.json'));print($r, round(d['ms_per_step'],3), d['roofline']['all_kernels_ms'], d['config']['nodes'])"
done
