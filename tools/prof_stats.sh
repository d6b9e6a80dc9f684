#!/bin/bash
# rocprofv3 kernel stats of one bench run on the GPU box: tools/prof_stats.sh <tag> [env assignments...]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof
cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof/${tag}_stdout.txt 2>&1
cp /tmp/kt_$tag/kt_kernel_stats.csv $R/gpurun_out/prof/${tag}_kernel_stats.csv
python3 $R/tools/kstats.py $R/gpurun_out/prof/${tag}_kernel_stats.csv 7
