#!/bin/bash
# A/B timing of library variants on one BASELINE configuration (tools/run_config.py): tools/ab_cfg.sh <cfg> <steps> <lib name or "base"> ...
cfg=$1; steps=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = base ]; then unset VGS_LIB; else export VGS_LIB=libvgs_hip_$v.so; fi
  python3 $R/tools/run_config.py $cfg 0 $steps 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('$cfg', '$v', 'ms/step %.3f' % d['ms_per_step'], 'median %.3f' % d['ms_per_step_median'], {k: round(x, 3) for k, x in s.items() if k in ('voxelize','adjacency','localcut','merge','localcut_bulk','localcut_kernel','supervoxel')})"
done
