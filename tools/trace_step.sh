#!/bin/bash
# kernel + memory-copy timeline of one bench step (device-resident loop): rocprofv3 --kernel-trace --memory-copy-trace of a short bench run,
# then tools/step_timeline.py on the last step.  usage (GPU box): tools/trace_step.sh <tag>  -> gpurun_out/<tag>_step_timeline.txt
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt_step
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kt_step -o kt -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
python3 $R/tools/step_timeline.py /tmp/kt_step/kt_kernel_trace.csv > $R/gpurun_out/${tag}_step_timeline.txt 2>&1
python3 - /tmp/kt_step/kt_memory_copy_trace.csv >> $R/gpurun_out/${tag}_step_timeline.txt 2>&1 <<'PY'
import csv, sys
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except OSError:
    rows = []
print("memory copies in the trace:", len(rows))
import collections
c = collections.Counter((r.get("Direction", "?")) for r in rows)
print(dict(c))
PY
tail -5 $R/gpurun_out/${tag}_step_timeline.txt
