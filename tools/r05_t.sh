#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1800 python3 -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_gpu_schedules.py tests/test_gpu_pairlists.py tests/test_gpu_classes.py tests/test_gpu_fuzz_cases.py 2>&1 | tail -3
bash tools/ab_env.sh 20 none none
for cfg in c2 c3n; do python3 tools/run_config.py $cfg 0 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['ms_per_step'],3))"; done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt_b; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_b -o kt -- python3 $R/tools/nl_time.py > /dev/null 2>&1
python3 $R/tools/kstats.py /tmp/kt_b/kt_kernel_stats.csv 4 30 | grep -E "k_near_lists"
