#!/bin/bash
# ab_cfg.sh <config> lib...: ab_bench.sh for another bench configuration (cfg4: level 1 / level 2 / blocks, cfg5, cfg3)
CFG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    DBGK_LIB=$lib python3 $R/bench.py --config $CFG --steps ${STEPS:-10} --warmup 2 --brief 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
p=j.get('phases_ms_per_step',{})
print('%-20s %s step %.3f ms  %s  %s' % ('$(basename $lib)', '$CFG', j['ms_per_step'], ' '.join('%s %.2f' % (k, v) for k, v in p.items() if isinstance(v,(int,float))), 'ok' if j.get('verified') else 'UNVERIFIED'))
"
  done
done
