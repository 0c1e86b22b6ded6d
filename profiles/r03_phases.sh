#!/bin/bash
# (round 5) the DBGK_DEBUG_* / tile-geometry / schedule switches used here exist only in a library built with -DDBGK_EXPERIMENTS:
#   profiles/tools/build_variant.sh exp dbg_assembly_amd/csrc -DDBGK_EXPERIMENTS  &&  export DBGK_LIB=$PWD/dbg_assembly_amd/_variants/exp.so
# Phase timings of the three PARTITION kernels from their debug builds (results of those runs are wrong by design):
#   DBGK_DEBUG_MODE  (level 1) 1 = extraction only, 2 = no copy-out, 3 = copy-out into a 32 KiB window
#   DBGK_DEBUG_L2    (level 2) 1 = loads + ranking only, 2 = no copy-out, 3 = copy-out into a window
#   DBGK_DEBUG_BUILD (build)   1 = clear + load only, 2 = no emit, 3 = emit without recomputing the keys
# DBGK_OVERLAP_CHUNKS=1 runs level 2 and the build one after the other (each kernel's time alone).
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  env $1 python3 $R/bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline --no-h2d "${@:2}" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
p=j['phases_ms_per_step']
print('%-44s step %.3f ms  mark %.3f l1 %.3f l2 %.2f build %.2f wall %.3f merge %.3f' % ('$1', j['ms_per_step'], p['mark'], p['insert'], p['partition'], p['build'], p['partition_and_build_wall'], p['merge']))
"
}
run "DBGK_NONE=0" "$@"
run "DBGK_OVERLAP_CHUNKS=1" "$@"
for m in 1 2 3; do run "DBGK_DEBUG_MODE=$m" "$@"; done
for m in 1 2 3; do run "DBGK_DEBUG_L2=$m DBGK_OVERLAP_CHUNKS=1" "$@"; done
for m in 1 2 3; do run "DBGK_DEBUG_BUILD=$m DBGK_OVERLAP_CHUNKS=1" "$@"; done
