#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05_try
timeout 1200 python3 -m pytest -x -q -m gpu tests/test_gpu_schedules.py tests/test_gpu_classes.py tests/test_gpu_edge.py tests/test_gpu_parity.py 2>&1 | tail -5
bash tools/ab_env.sh 10 none VGS_NO_VOTE=1 VGS_NO_PAIRLISTS=1 none VGS_NO_VOTE=1 VGS_NO_PAIRLISTS=1
