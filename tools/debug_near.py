#!/usr/bin/env python3
"""Debug helper: one test_gpu_schedules scene through the GPU path against the oracle; prints the rows that differ."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np
import vgs_svgs_segmentation_amd as v
import refcpu_py as oracle
from helpers import oracle_params
rng = np.random.default_rng(1)
n = 120_000
side = np.sqrt(n / 6000.0)
x, y = rng.random(n) * side, rng.random(n) * side
z = 0.3 * np.sin(2.0 * x) * np.cos(1.5 * y) + rng.normal(0, 0.03, n) + 2.0
xyz = np.stack([x + 0.011, y + 0.017, z], axis=1).astype(np.float32)
p = v.default_params(2, voxel_size=0.1, graph_size=0.4)
e = v.Engine(p); e.set_points(xyz); e.run()
ref = oracle.run_vgs(xyz, oracle_params(oracle, p))
off, idx = e.lists("connect_cut"); roff, ridx = ref.lists("connect_cut")
ao, ai = e.lists("adjacency")
bad = [i for i in range(len(off) - 1) if set(idx[off[i]:off[i + 1]]) != set(ridx[roff[i]:roff[i + 1]])]
print(os.environ.get("VGS_NO_NEAR"), "rows", len(off) - 1, "bad", len(bad), e.schedule_counters())
for i in bad[:12]:
    print(i, "n", ao[i + 1] - ao[i], "gpu", len(idx[off[i]:off[i + 1]]), "ref", len(ridx[roff[i]:roff[i + 1]]), "extra", sorted(set(idx[off[i]:off[i + 1]]) - set(ridx[roff[i]:roff[i + 1]]))[:6], "missing", sorted(set(ridx[roff[i]:roff[i + 1]]) - set(idx[off[i]:off[i + 1]]))[:6])
