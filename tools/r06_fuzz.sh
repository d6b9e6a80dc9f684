#!/bin/bash
# the round's closing fuzz campaigns on the final code (seeds of their own): VGS plain + wide balls, SVGS, tiles
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/fuzz
timeout 400 python3 tools/fuzz_parity.py 300 6301 2 2>&1 | tail -2 | tee gpurun_out/fuzz/vgs.txt
timeout 400 python3 tools/fuzz_parity.py 300 6302 2 wide 2>&1 | tail -2 | tee gpurun_out/fuzz/vgs_wide.txt
timeout 300 python3 tools/fuzz_parity.py 200 6303 3 2>&1 | tail -2 | tee gpurun_out/fuzz/svgs.txt
timeout 300 python3 tools/fuzz_tiles.py 200 6304 2>&1 | tail -2 | tee gpurun_out/fuzz/tiles.txt
timeout 300 python3 tools/fuzz_vccs.py 200 6305 1 2>&1 | tail -2 | tee gpurun_out/fuzz/vccs_pcl.txt
