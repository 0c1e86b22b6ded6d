#!/bin/bash
# (round 5) the DBGK_DEBUG_* / tile-geometry / schedule switches used here exist only in a library built with -DDBGK_EXPERIMENTS:
#   profiles/tools/build_variant.sh exp dbg_assembly_amd/csrc -DDBGK_EXPERIMENTS  &&  export DBGK_LIB=$PWD/dbg_assembly_amd/_variants/exp.so
# level-2 grid size while it shares the chip with the region build (DBGK_L2_GRID workgroups of 512 threads)
for g in 64 128 192 256 512; do
  DBGK_L2_GRID=$g timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/l2g_$g.json 2> gpurun_out/l2g_$g.err || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/l2g_$g.json'));r=d['roofline'];print($g, round(d['ms_per_step'],3), r['all_kernels_ms'], round(r['l2_build_wall_ms'],3))"
done
