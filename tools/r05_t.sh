#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python3 -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_gpu_classes.py tests/test_gpu_edge.py 2>&1 | tail -3
for v in base as2 as3 as4 base; do
  if [ "$v" = base ]; then unset VGS_LIB; else export VGS_LIB=libvgs_hip_$v.so; fi
  python3 tools/adj_time.py 2>&1 | tail -1
done
