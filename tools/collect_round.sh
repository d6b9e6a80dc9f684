#!/bin/bash
# The round's whole evidence set in ONE call on the GPU box, every file stamped with the commit it was taken at:
#   bench (config 3):  kernel stats, PMC per launch (traffic + instruction + cycle passes), traffic.json, step timeline, stage bandwidths,
#                      instruction counts of the bulk kernel phase by phase -> valu_mix.json
#   c2, c3n:           line, kernel stats, PMC per launch            xl: line (the solid block at r = 10)
#   c4, c4s:           line, kernel stats (c4 = PCL order, the default; c4s = the synchronous variant)
# The box holds no .git: the commit comes in as an argument (the caller's `git rev-parse --short HEAD`, clean tree) and the script
# REFUSES a tag directory that already holds files of another commit (profiles/<tag>_MANIFEST.json lists file -> commit).
# usage (from the container):  gpurun --timeout 2400 -- 'tools/collect_round.sh r05 <commit> [parts]'     parts default: all
#        parts: bench pmc timeline phases c2 c3n xl c4 c4s c5
tag=${1:?tag}; commit=${2:?commit}; shift 2
parts=${*:-bench pmc timeline phases c2 c3n xl c4 c4s c5}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/profiles
mkdir -p $out
man=$R/profiles/${tag}_MANIFEST.json
if [ -f $man ] && ! python3 - "$man" "$commit" "$parts" <<'PY'
import json, sys
m, commit, parts = json.load(open(sys.argv[1])), sys.argv[2], sys.argv[3].split()
# a part may be re-collected at a new commit only together with everything that shares its kernels: all or nothing
other = sorted({c for f, c in m.items() if c != commit})
if other and set(parts) != set("bench pmc timeline phases c2 c3n xl c4 c4s c5".split()):
    print(f"profiles of {other} are in the manifest: collect ALL parts at {commit}, not a subset", file=sys.stderr); sys.exit(1)
PY
then exit 2; fi
cd /tmp; export TMPDIR=/tmp
has() { [[ " $parts " == *" $1 "* ]]; }
bench="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host"

if has bench; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_b -o kt -- $bench > $out/${tag}_bench_stdout.txt 2>&1
  cp /tmp/kt_b/kt_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
  python3 $R/bench.py --steps 20 --warmup 3 > $out/${tag}_bench_line.json 2> $out/${tag}_bench_stderr.txt
  # the native tiled driver with the world one GPU can form (multi-GPU pre-flight: what the driver costs beside the plain engine)
  python3 $R/bench.py --gpus 1 --native --steps 10 --warmup 3 --no-cpu-baseline --no-host-to-host > $out/${tag}_bench_native_line.json 2> $out/${tag}_bench_native_stderr.txt
fi
if has pmc; then
  i=0
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES" \
              "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    i=$((i+1)); rm -rf /tmp/pmc_b_$i
    timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_b_$i -o pmc -- $bench > /dev/null 2>&1
  done
  python3 $R/tools/pmc_table.py "/tmp/pmc_b_*" $out/${tag}_bench_pmc_per_launch.csv $out/${tag}_traffic.json $commit
fi
if has timeline; then
  rm -rf /tmp/kt_step
  timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kt_step -o kt -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
  python3 $R/tools/step_timeline.py /tmp/kt_step/kt_kernel_trace.csv > $out/${tag}_step_timeline.txt 2>&1
  python3 - /tmp/kt_step/kt_memory_copy_trace.csv >> $out/${tag}_step_timeline.txt 2>&1 <<'PY'
import csv, sys, collections
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except OSError:
    rows = []
print("memory copies in the trace (6 steps + set-up):", len(rows), dict(collections.Counter(r.get("Direction", "?") for r in rows)))
PY
fi
if has phases; then
  bash $R/tools/pmc_phases.sh > /dev/null 2>&1
  cp $R/gpurun_out/pmc_phases.txt $out/${tag}_pmc_phases.txt
fi
for cfg in c2 c3n; do
  if has $cfg; then
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
    python3 $R/tools/pmc_table.py "/tmp/pmc_${cfg}_*" $out/${tag}_${cfg}_pmc_per_launch.csv "" $commit
  fi
done
if has xl; then
  python3 $R/tools/run_config.py xl 0 3 > $out/${tag}_xl_line.json 2> $out/${tag}_xl_stderr.txt
  rm -rf /tmp/kt_xl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_xl -o kt -- python3 $R/tools/run_config.py xl 0 3 > /dev/null 2>&1
  cp /tmp/kt_xl/kt_kernel_stats.csv $out/${tag}_xl_kernel_stats.csv
fi
for cfg in c4 c4s; do
  if has $cfg; then
    python3 $R/tools/run_config.py $cfg 0 5 > $out/${tag}_${cfg}_line.json 2> $out/${tag}_${cfg}_stderr.txt
    rm -rf /tmp/kt_$cfg
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$cfg -o kt -- python3 $R/tools/run_config.py $cfg 0 5 > /dev/null 2>&1
    cp /tmp/kt_$cfg/kt_kernel_stats.csv $out/${tag}_${cfg}_kernel_stats.csv
  fi
done
if has c5; then
  # config 5 at its real size on this one GPU: the test writes its evidence itself (tests/test_gpu_config5.py)
  (cd $R && python3 -m pytest tests/test_gpu_config5.py -x -q -p no:cacheprovider > $out/${tag}_c5_stdout.txt 2>&1)
  cp $R/gpurun_out/c5_onegpu.json $out/${tag}_c5_onegpu.json
fi
# empty stderr files say nothing
find $out -name "${tag}_*_stderr.txt" -size 0 -delete
# the manifest: every file of this call -> the commit
python3 - "$out" "$tag" "$commit" <<'PY'
import json, os, sys
out, tag, commit = sys.argv[1:4]
m = {f: commit for f in sorted(os.listdir(out)) if f.startswith(tag + "_") and not f.endswith("MANIFEST.json")}
json.dump(m, open(os.path.join(out, f"{tag}_MANIFEST.json"), "w"), indent=1)
PY
ls -la $out
