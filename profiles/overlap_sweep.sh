#!/bin/bash
# level 2 || region build: number of bucket chunks (DBGK_OVERLAP_CHUNKS; 1 = level 2 first, then the build)
for c in 1 4 8 16; do
  DBGK_OVERLAP_CHUNKS=$c timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ovl_$c.json 2> gpurun_out/ovl_$c.err || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/ovl_$c.json'));print($c, round(d['ms_per_step'],3), d['roofline']['all_kernels_ms'], d['config']['nodes'])"
done
