#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/ab_cfg.sh c3n 3 base occ6 base occ6
TESTS=none CFGS=c3n KSTATS=1 bash tools/r05_try.sh | head -12
