#!/bin/bash
# Instruction counts of the bulk local-cut kernel phase by phase: the same bench run with the kernel leaving after the
# gather (VGS_SHELL0=-1), the enumeration (VGS_DBG_STOP=1), evaluation+sort (2, 3) and merge (4) of the first shell, and in full.
# usage (GPU box): tools/pmc_phases.sh   -> gpurun_out/pmc_phases.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/pmc_phases.txt; : > $out
run() {
  tag=$1; shift
  rm -rf /tmp/pp_$tag
  env "$@" true   # (validates the assignments)
  for kv in "$@"; do export "$kv"; done
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES --output-format csv -d /tmp/pp_$tag -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
  for kv in "$@"; do unset "${kv%%=*}"; done
  python3 - /tmp/pp_$tag/pmc_counter_collection.csv $tag >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
for k in acc:
    if "k_localcut_wave<96, 448, 1, false" in k:
        print(sys.argv[2], {c: f"{v / calls[(k, c)]:.4g}" for c, v in sorted(acc[k].items())})
PY
}
run gather VGS_SHELL0=-1
run enum VGS_DBG_STOP=1
run evalsort VGS_DBG_STOP=2
run sort VGS_DBG_STOP=3
run merge VGS_DBG_STOP=4
run full VGS_NOTHING=1
cat $out
