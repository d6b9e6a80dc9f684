#!/bin/bash
# quick GPU check of a change: a few parity test files (TESTS, twice: default schedule and FORCE env), then timing lines of CFGS and the bench
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r05_try
mkdir -p $out
cd $R
T=${TESTS:-tests/test_gpu_parity.py tests/test_gpu_classes.py tests/test_gpu_schedules.py tests/test_gpu_fuzz_cases.py tests/test_gpu_edge.py}
if [ "$T" != none ]; then
  timeout 900 python3 -m pytest -x -q -m gpu $T 2>&1 | tail -8
  if [ -n "$FORCE" ]; then echo "== again with $FORCE"; env $FORCE timeout 900 python3 -m pytest -x -q -m gpu $T 2>&1 | tail -8; fi
fi
for cfg in ${CFGS:-c3n c2}; do
  python3 tools/run_config.py $cfg 0 3 2> $out/${cfg}_stderr.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('$cfg', 'ms/step %.3f' % d['ms_per_step'], {k: round(x, 3) for k, x in s.items() if k in ('voxelize','adjacency','localcut','merge','localcut_bulk','localcut_kernel')}, {k: v for k, v in d['schedule'].items() if v})"
  tail -3 $out/${cfg}_stderr.txt
  if [ -n "$KSTATS" ]; then
    ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt_$cfg; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$cfg -o kt -- python3 $R/tools/run_config.py $cfg 0 3 > /dev/null 2>&1; cp /tmp/kt_$cfg/kt_kernel_stats.csv $out/${cfg}_kernel_stats.csv; python3 $R/tools/kstats.py $out/${cfg}_kernel_stats.csv 4 14 )
  fi
done
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-to-host 2>$out/bench_stderr.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('bench ms/step %.3f' % d['ms_per_step'], {k: round(x, 3) for k, x in s.items()})"
