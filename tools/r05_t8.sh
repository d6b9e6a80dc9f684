#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for e in "VGS_X=1" "VGS_NO_VOTE=1" "VGS_NO_PAIRLISTS=1"; do
echo "== $e"; env $e python3 tools/run_config.py c3 0 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('ms/step %.3f' % d['ms_per_step'], {k: round(x, 3) for k, x in s.items() if k in ('localcut','merge','localcut_bulk','localcut_kernel')}, {k: v for k, v in d['schedule'].items() if v}, d['counts']['handed_over'])"
done
TESTS=none CFGS=c3 KSTATS=1 bash tools/r05_try.sh | head -24
