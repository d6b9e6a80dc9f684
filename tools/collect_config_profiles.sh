#!/bin/bash
# rocprofv3 kernel stats + a bench-style JSON line for BASELINE configs 2 and 4 (bench.py itself times config 3).
# usage (GPU box): tools/collect_config_profiles.sh r03    -> gpurun_out/profiles/<tag>_c2_*, <tag>_c4_*
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/profiles
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for cfg in c2 c4 c4p; do
  python3 $R/tools/run_config.py $cfg 0 5 > $out/${tag}_${cfg}_line.json 2> $out/${tag}_${cfg}_stderr.txt
  rm -rf /tmp/kt_$cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$cfg -o kt -- python3 $R/tools/run_config.py $cfg 0 5 > /dev/null 2>&1
  cp /tmp/kt_$cfg/kt_kernel_stats.csv $out/${tag}_${cfg}_kernel_stats.csv
done
ls -la $out
