#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python3 -m pytest -x -q -m gpu tests/test_gpu_tiles.py tests/test_gpu_tiles_native.py tests/test_gpu_bench.py tests/test_cpp_driver.py tests/test_gpu_parity.py 2>&1 | tail -3
timeout 300 python3 tools/fuzz_tiles.py 150 5202 2>&1 | tail -1
for k in 1 2; do
python3 bench.py --gpus 1 --native --steps 10 --warmup 3 --no-cpu-baseline --no-host-to-host 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('native', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['driver']['per_rank'][0].get('tiles_ms').items()})"
done
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-to-host 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('plain', round(d['ms_per_step'],3))"
