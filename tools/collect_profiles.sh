#!/bin/bash
# Collect the round's rocprofv3 evidence for `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline` on the GPU box:
#   1. --kernel-trace --stats            -> <tag>_bench_kernel_stats.csv (per-kernel calls / total / average duration)
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, TCC slots)  -> per-kernel averages per launch
#   3. --pmc SQ instruction mix           -> per-kernel averages per launch
# and <tag>_traffic.json = HBM bytes per launch of the dominant kernel (FETCH_SIZE + WRITE_SIZE, both reported in KB).
# Output: gpurun_out/profiles/ (copy into profiles/ to commit).  usage: tools/collect_profiles.sh r01
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/profiles
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
cmd="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- $cmd > $out/${tag}_bench_stdout.txt 2>&1
cp $out/kt/kt_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc_$name -o pmc -- $cmd > /dev/null 2>&1
done
python3 - "$out" "$tag" <<'PY'
import csv, sys, collections, json, glob, os
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(os.path.join(out, "pmc_*", "pmc_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
names = sorted({c for k in acc for c in acc[k]})
with open(os.path.join(out, f"{tag}_bench_pmc_per_launch.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "launches"] + names)
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
        w.writerow([k, max(calls[(k, c)] for c in acc[k])] + [f"{acc[k][c] / calls[(k, c)]:.6g}" if c in acc[k] else "" for c in names])
dom = [k for k in acc if k.startswith("void k_localcut_wave<96")]
if dom:
    k = dom[0]
    fetch_kb = acc[k]["FETCH_SIZE"] / calls[(k, "FETCH_SIZE")]; write_kb = acc[k]["WRITE_SIZE"] / calls[(k, "WRITE_SIZE")]
    valu = acc[k]["SQ_INSTS_VALU"] / calls[(k, "SQ_INSTS_VALU")] if (k, "SQ_INSTS_VALU") in calls else None
    json.dump({"kernel": k[:40], "points": 10000000, "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb,
               "hbm_bytes_per_launch": (fetch_kb + write_kb) * 1024.0, "valu_wave_instructions_per_launch": valu,
               "note": "FETCH_SIZE + WRITE_SIZE (KB) from separate rocprofv3 --pmc passes of the bench command; raw counters, no 2x wide-load correction (the kernel gathers 8-byte row entries and 16-byte record quarters, not 16 B/lane streams)"},
              open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
PY
rm -rf $out/kt $out/pmc_*
ls -la $out
