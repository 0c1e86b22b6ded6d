#!/bin/bash
# A/B of library builds inside ONE gpurun call (boxes differ by a few per cent): profiles/ab_bench.sh libA.so libB.so ...
# each build is benchmarked in turn, ROUNDS times round robin
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in $(seq 1 ${ROUNDS:-3}); do
  for lib in "$@"; do
    DBGK_LIB=$lib python3 $R/bench.py --steps ${STEPS:-20} --warmup 2 --brief 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
p=j['phases_ms_per_step']
print('%-28s step %.3f ms  mark %.3f l1 %.3f l2 %.2f build %.2f wall %.3f %s' % ('$(basename $lib)', j['ms_per_step'], p['mark'], p['insert'], p['partition'], p['build'], p['partition_and_build_wall'], 'ok' if j['verified'] else 'UNVERIFIED'))
"
  done
done
