#!/bin/bash
# (round 5) the DBGK_DEBUG_* / tile-geometry / schedule switches used here exist only in a library built with -DDBGK_EXPERIMENTS:
#   profiles/tools/build_variant.sh exp dbg_assembly_amd/csrc -DDBGK_EXPERIMENTS  &&  export DBGK_LIB=$PWD/dbg_assembly_amd/_variants/exp.so
# timing experiments for the build kernel: DBGK_DEBUG_BUILD 0 = production, 1 = clear + load only, 2 = no emit,
# 3 = emit without recomputing the keys
for m in 0 1 2 3; do
  DBGK_DEBUG_BUILD=$m timeout -k 10 200 python bench.py --steps 2 --warmup 1 --engine 2 --no-cpu-baseline > gpurun_out/dbgb_$m.json 2> gpurun_out/dbgb_$m.err || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/dbgb_$m.json'));print($m, d['roofline']['all_kernels_ms'])"
done
