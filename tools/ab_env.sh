#!/bin/bash
# A/B timing of environment knobs on the GPU box: tools/ab_env.sh <steps> "<VAR=val ...>|none" ...
steps=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = none ]; then envs=""; else envs="$v"; fi
  env $envs python3 $R/bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-host-to-host 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('[$v]', 'ms/step %.3f' % d['ms_per_step'], 'median %.3f' % d['ms_per_step_median'], {k: round(x, 3) for k, x in s.items() if k in ('voxelize','adjacency','localcut','merge','localcut_bulk','localcut_kernel')})"
done
