#!/bin/bash
# fuzz campaigns on the code with the chain merge of the pair-list kernel and the capped hand-over grids (seeds of their own)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/fuzz2
if [ -z "$1" ]; then
timeout 300 python3 tools/fuzz_parity.py 200 6311 2 2>&1 | tail -2 | tee gpurun_out/fuzz2/vgs.txt
timeout 400 python3 tools/fuzz_parity.py 300 6312 2 wide 2>&1 | tail -2 | tee gpurun_out/fuzz2/vgs_wide.txt
timeout 250 python3 tools/fuzz_parity.py 150 6313 3 2>&1 | tail -2 | tee gpurun_out/fuzz2/svgs.txt
fi
# (second call) the tiled driver and the PCL-order supervoxels on the same code
if [ "$1" = "tiles" ]; then
timeout 400 python3 tools/fuzz_tiles.py 300 6314 2>&1 | tail -2 | tee gpurun_out/fuzz2/tiles.txt
timeout 300 python3 tools/fuzz_vccs.py 200 6315 1 2>&1 | tail -2 | tee gpurun_out/fuzz2/vccs_pcl.txt
fi
