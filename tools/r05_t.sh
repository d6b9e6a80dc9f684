#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python3 -m pytest -x -q -m gpu tests/test_gpu_schedules.py 2>&1 | tail -3
bash tools/ab_env.sh 20 none none
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt_b; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_b -o kt -- python3 $R/tools/nl_time.py > /dev/null 2>&1
python3 $R/tools/kstats.py /tmp/kt_b/kt_kernel_stats.csv 4 30 | grep -E "k_near_lists|k_adjacency_masks"
