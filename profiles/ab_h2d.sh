#!/bin/bash
# the region "first host-to-device copy -> final table" (bench.py value_from_first_h2d) for several batch sizes: profiles/ab_h2d.sh 0 134217728 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for b in "$@"; do
  DBGK_BENCH_BATCH_BASES=$b python3 $R/bench.py --steps 5 --warmup 2 --no-also --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
h=j['value_from_first_h2d']
print('batch bases %-12s resident %.3f ms  from first H2D %.3f ms (%s)  with offsets %.3f' % ('$b', j['ms_per_step'], h['ms_per_step'], ' '.join('%.2f' % x for x in h['reps_ms']), h['variants']['packed_pinned_with_offsets']['ms_per_step']))
"
done
