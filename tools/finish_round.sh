#!/bin/bash
# after tools/collect_round.sh came back (container side): adopt the files, make the derived ones, list what the manifest holds
# usage: tools/finish_round.sh r05 <commit>
tag=${1:?tag}; commit=${2:?commit}
cd "$(dirname "$0")/.."
python3 tools/adopt_profiles.py $tag || exit 1
kms=$(python3 -c "
import json
d=json.loads(open('profiles/${tag}_bench_line.json').read().strip().splitlines()[-1])
print(d['roofline']['kernel_ms'])")
python3 tools/valu_mix.py profiles/${tag}_pmc_phases.txt $kms profiles/${tag}_valu_mix.json $commit > /dev/null || exit 1
python3 tools/stage_bw.py profiles/${tag}_bench_kernel_stats.csv profiles/${tag}_bench_line.json profiles/${tag}_stage_bw.json profiles/${tag}_bench_pmc_per_launch.csv > /dev/null || echo "stage_bw failed"
python3 - "$tag" "$commit" <<'PY'
import json, sys
tag, commit = sys.argv[1:3]
p = f"profiles/{tag}_MANIFEST.json"
m = json.load(open(p))
m[f"{tag}_valu_mix.json"] = commit; m[f"{tag}_stage_bw.json"] = commit
json.dump(dict(sorted(m.items())), open(p, "w"), indent=1)
print({c: sum(1 for x in m.values() if x == c) for c in set(m.values())})
PY
