#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for e in "VGS_X=1" "VGS_VOTE_FORCE=1" "VGS_NO_VOTE=1"; do
echo "== $e"; env $e TESTS=none CFGS=c3n KSTATS=1 bash tools/r05_try.sh 2>&1 | grep -v "^bench" | head -8
done
TESTS=none CFGS=none bash tools/r05_try.sh | tail -1
