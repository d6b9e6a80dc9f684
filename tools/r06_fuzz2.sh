#!/bin/bash
# fuzz campaigns on the code with the chain merge of the pair-list kernel and the capped hand-over grids (seeds of their own)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/fuzz2
timeout 300 python3 tools/fuzz_parity.py 200 6311 2 2>&1 | tail -2 | tee gpurun_out/fuzz2/vgs.txt
timeout 400 python3 tools/fuzz_parity.py 300 6312 2 wide 2>&1 | tail -2 | tee gpurun_out/fuzz2/vgs_wide.txt
timeout 250 python3 tools/fuzz_parity.py 150 6313 3 2>&1 | tail -2 | tee gpurun_out/fuzz2/svgs.txt
