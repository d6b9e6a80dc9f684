#!/bin/bash
# rocprofv3 evidence for one BASELINE configuration other than the bench's (tools/run_config.py c2|c4|c4p): kernel stats, then the SQ
# instruction / wait / LDS counters in separate --pmc passes (per-kernel averages per launch).
# usage (GPU box): tools/collect_cfg_pmc.sh r04 c2   -> gpurun_out/profiles/r04_c2_{line.json,kernel_stats.csv,pmc_per_launch.csv}
tag=${1:-r04}; cfg=${2:-c2}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/profiles
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/run_config.py $cfg 0 5 > $out/${tag}_${cfg}_line.json 2> $out/${tag}_${cfg}_stderr.txt
rm -rf /tmp/kt_$cfg /tmp/pmc_${cfg}_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$cfg -o kt -- python3 $R/tools/run_config.py $cfg 0 5 > /dev/null 2>&1
cp /tmp/kt_$cfg/kt_kernel_stats.csv $out/${tag}_${cfg}_kernel_stats.csv
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_${cfg}_$i -o pmc -- python3 $R/tools/run_config.py $cfg 0 3 > /dev/null 2> /tmp/pmc_${cfg}_$i.err || echo "pass $i ($pass) failed" >> $out/${tag}_${cfg}_stderr.txt
done
python3 - "$out" "$tag" "$cfg" <<'PY'
import csv, sys, collections, glob, os
out, tag, cfg = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(f"/tmp/pmc_{cfg}_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
names = sorted({c for k in acc for c in acc[k]})
with open(os.path.join(out, f"{tag}_{cfg}_pmc_per_launch.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "launches"] + names)
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", acc[k].get("SQ_INSTS_VALU", 0))):
        w.writerow([k[:90], max(calls[(k, c)] for c in acc[k])] + [f"{acc[k][c] / calls[(k, c)]:.6g}" if c in acc[k] else "" for c in names])
PY
ls -la $out
