#!/bin/bash
# (round 5) the DBGK_DEBUG_* / tile-geometry / schedule switches used here exist only in a library built with -DDBGK_EXPERIMENTS:
#   profiles/tools/build_variant.sh exp dbg_assembly_amd/csrc -DDBGK_EXPERIMENTS  &&  export DBGK_LIB=$PWD/dbg_assembly_amd/_variants/exp.so
# ab_serial.sh lib...: each build with level 2 and the region build one after the other (DBGK_OVERLAP_CHUNKS=1: each kernel's time alone) and overlapped (default)
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    for ch in 1 12; do
    DBGK_OVERLAP_CHUNKS=$ch DBGK_LIB=$lib python3 $R/bench.py --steps ${STEPS:-10} --warmup 2 --brief 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
p=j['phases_ms_per_step']
print('%-20s chunks %2d step %.3f ms  l1 %.3f l2 %.2f build %.2f wall %.3f %s' % ('$(basename $lib)', $ch, j['ms_per_step'], p['insert'], p['partition'], p['build'], p['partition_and_build_wall'], 'ok' if j['verified'] else 'UNVERIFIED'))
"
    done
  done
done
