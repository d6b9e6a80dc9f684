#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/ab_env.sh 10 none VGS_NO_DENSE_TO_PG=1 none VGS_NO_DENSE_TO_PG=1
python3 tools/run_config.py c3 0 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: v for k, v in d['schedule'].items() if v})"
timeout 900 python3 -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_gpu_schedules.py 2>&1 | tail -4
