#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in 1 2; do
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-to-host 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('plain', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()})"
python3 bench.py --gpus 1 --native --steps 10 --warmup 3 --no-cpu-baseline --no-host-to-host 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('native', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()}, d['driver']['per_rank'][0].get('tiles_ms'))"
done
