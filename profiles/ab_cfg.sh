#!/bin/bash
# A/B of environment settings on another bench workload: CONFIG=cfg4 profiles/ab_cfg.sh - "VAR=1" ...   ("-" = no setting)
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in $(seq 1 ${ROUNDS:-2}); do
  for setting in "$@"; do
    if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
    env $envs python3 $R/bench.py --config ${CONFIG:-cfg4} --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-also --no-h2d 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
k=j['roofline'].get('kernels') or {}
print('%-34s step %.3f ms  %s  %s' % ('$setting', j['ms_per_step'], ' '.join('%s %.2f' % (n.replace('k_extract_scatter','l1').replace('k_wide_scatter_','w'), v['ms_per_step']) for n, v in k.items()), 'ok' if j.get('verified') else 'UNVERIFIED'))
"
  done
done
