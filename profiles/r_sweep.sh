#!/bin/bash
# (round 5) the DBGK_DEBUG_* / tile-geometry / schedule switches used here exist only in a library built with -DDBGK_EXPERIMENTS:
#   profiles/tools/build_variant.sh exp dbg_assembly_amd/csrc -DDBGK_EXPERIMENTS  &&  export DBGK_LIB=$PWD/dbg_assembly_amd/_variants/exp.so
# level-1 bucket width r (DBGK_PART_R): level-1 fan-out size >> r, level-2 fan-out 2^(r-12)
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in ${RS:-20 21 22}; do
  DBGK_PART_R=$r python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print('r', $r, 'ms_per_step %.3f' % j['ms_per_step'], 'l1 %.2f' % j['phases_ms_per_step']['insert'], 'l2 %.2f build %.2f wall %.2f' % (j['phases_ms_per_step']['partition'], j['phases_ms_per_step']['build'], j['phases_ms_per_step']['partition_and_build_wall']), 'verified' if j['verified'] else 'UNVERIFIED')
"
done
