#!/bin/bash
# round-5 baseline numbers on the GPU box: c3n / c2 / xl(small) lines, kernel stats of c3n, phase cycles (prof build) of c3n and c2
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r05_base
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for cfg in c3n c2; do
  python3 $R/tools/run_config.py $cfg 0 3 > $out/${cfg}_line.json 2> $out/${cfg}_stderr.txt
  rm -rf /tmp/kt_$cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$cfg -o kt -- python3 $R/tools/run_config.py $cfg 0 3 > /dev/null 2>&1
  cp /tmp/kt_$cfg/kt_kernel_stats.csv $out/${cfg}_kernel_stats.csv
  python3 $R/tools/kstats.py $out/${cfg}_kernel_stats.csv 4 16 > $out/${cfg}_kstats.txt
  VGS_DEBUG=1 VGS_LIB=libvgs_hip_prof.so python3 $R/tools/run_config.py $cfg 0 1 > $out/${cfg}_prof_line.json 2> $out/${cfg}_prof_stderr.txt
done
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-to-host > $out/bench_line.json 2> $out/bench_stderr.txt
tail -c 1500 $out/c3n_line.json; echo; cat $out/c3n_kstats.txt; grep -v "^$" $out/c3n_prof_stderr.txt | tail -30
