#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1800 python3 -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_gpu_vccs.py tests/test_gpu_svgs.py tests/test_golden.py tests/test_gpu_classes.py 2>&1 | tail -3
bash tools/ab_env.sh 20 none none
for cfg in c4 c4p; do python3 tools/run_config.py $cfg 0 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items() if k in ('features','supervoxel','total')})"; done
