#!/bin/bash
# c3n with the prof build + kernel stats + default timing (no tests)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r05_try
mkdir -p $out
cd $R
VGS_DEBUG=1 VGS_LIB=libvgs_hip_prof.so python3 tools/run_config.py c3n 0 1 > /dev/null 2> $out/c3n_prof_stderr.txt; grep "k_localcut_pg\|handed-over\|classes" $out/c3n_prof_stderr.txt | tail -3
TESTS="${TESTS:-none}" CFGS="${CFGS:-c3n}" KSTATS=1 bash tools/r05_try.sh
