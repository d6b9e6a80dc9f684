#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05_try
timeout 1200 python3 -m pytest -x -q -m gpu $TESTS > gpurun_out/r05_try/pytest_full.log 2>&1
grep -n "PASSED\|FAILED\|passed\|failed\|Fatal\|fault\|Error" gpurun_out/r05_try/pytest_full.log | head -20
head -60 gpurun_out/r05_try/pytest_full.log | cut -c1-220
