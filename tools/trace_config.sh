#!/bin/bash
# usage: tools/trace_config.sh <config> : kernel trace of tools/run_config.py <config>, prints the top kernels and the last local-cut timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$1 -o kt -- python3 $R/tools/run_config.py $1 2>&1 | grep "run 1" | cut -c1-120
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$R/gpurun_out/kt_$1/kt_kernel_stats.csv")))[:12]:
    print(r["Name"][:70].ljust(70), r["Calls"], r["AverageNs"], r["Percentage"])
PY
python3 $R/tools/prof_timeline.py $R/gpurun_out/kt_$1/kt_kernel_trace.csv k_classify 10
