#!/bin/bash
# A/B of environment settings inside ONE gpurun call: profiles/ab_env.sh "VAR=1" "VAR2=x VAR3=y" ...  ("-" = no setting)
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in $(seq 1 ${ROUNDS:-2}); do
  for setting in "$@"; do
    if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
    env $envs python3 $R/bench.py --steps ${STEPS:-10} --warmup 2 --brief 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
p=j['phases_ms_per_step']
print('%-34s step %.3f ms  mark %.3f l1 %.3f l2 %.2f build %.2f wall %.3f %s' % ('$setting', j['ms_per_step'], p['mark'], p['insert'], p['partition'], p['build'], p['partition_and_build_wall'], 'ok' if j['verified'] else 'UNVERIFIED'))
"
  done
done
