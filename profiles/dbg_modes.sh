#!/bin/bash
# (round 5) the DBGK_DEBUG_* / tile-geometry / schedule switches used here exist only in a library built with -DDBGK_EXPERIMENTS:
#   profiles/tools/build_variant.sh exp dbg_assembly_amd/csrc -DDBGK_EXPERIMENTS  &&  export DBGK_LIB=$PWD/dbg_assembly_amd/_variants/exp.so
# timing experiments for the level-1 kernel: DBGK_DEBUG_MODE 0 = production, 1 = extraction only, 2 = no copy-out, 3 = copy-out into a 32 KiB window
for m in 0 1 2 3; do
  DBGK_DEBUG_MODE=$m timeout -k 10 200 python bench.py --steps 2 --warmup 1 --engine 2 --no-cpu-baseline > gpurun_out/dbg_$m.json 2> gpurun_out/dbg_$m.err
  python -c "
import json;d=json.load(open('gpurun_out/dbg_$m.json'));print($m, d['roofline']['all_kernels_ms'])"
done
