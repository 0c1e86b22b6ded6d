#!/bin/bash
# start-delay sweep for the persistent scatter kernels (DBGK_STAGGER, units of s_sleep(16) ~ 0.4 us)
for st in 0 16 32 64 128; do
  DBGK_STAGGER=$st timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/stagger_$st.json 2> gpurun_out/stagger_$st.err || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/stagger_$st.json'));print($st, round(d['ms_per_step'],3), d['roofline']['all_kernels_ms'], d['config']['nodes'])"
done
