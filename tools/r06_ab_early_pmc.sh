#!/bin/bash
# the bulk kernel's wave-cycle counters with and without the early hand-over (one --pmc pass each, own runs: no tracing beside it)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for n in 0 256; do
  rm -rf /tmp/pmc_eh_$n
  VGS_EARLY_HO=$n timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_eh_$n -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
  python3 - /tmp/pmc_eh_$n $n <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_localcut_wave<96" in k or "k_localcut_dense<128" in k:
            acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k[:60], r["Counter_Name"])] += 1
for k in acc:
    per = {c: v / calls[(k, c)] for c, v in acc[k].items()}
    share = per.get("SQ_WAIT_ANY", 0) / per["SQ_WAVE_CYCLES"] if per.get("SQ_WAVE_CYCLES") else float("nan")
    print(f"VGS_EARLY_HO={sys.argv[2]} {k}: launches {max(calls[(k, c)] for c in acc[k])}, per launch " + ", ".join(f"{c} {v:.4g}" for c, v in sorted(per.items())) + f", SQ_WAIT_ANY / SQ_WAVE_CYCLES = {share:.3f}")
PY
done
