#!/bin/bash
# usage: tools/pmc_run.sh <out-subdir> <counter list...>   (one rocprofv3 --pmc pass over a 2-step bench run)
# Writes gpurun_out/<out-subdir>/ and prints the per-kernel sums for the local-cut kernels.
out=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$out -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - "$R/gpurun_out/$out/pmc_counter_collection.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(k, r["Counter_Name"])] += 1
for k in acc:
    if "localcut" in k or "adjacency" in k:
        print(k, {c: f"{v / calls[(k, c)]:.4g}" for c, v in acc[k].items()})
PY
