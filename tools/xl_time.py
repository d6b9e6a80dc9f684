#!/usr/bin/env python3
"""Time of the extra-large local-cut class on the solid-block scene of tests/test_gpu_edge.py::test_neighbourhoods_above_2048_voxels
(300 k points, voxel 0.0625 m, ball of eight voxels: 275 neighbourhoods above 2048 used voxels) and on a bigger block at config 2's own
ball (voxel 0.05 m, graph 0.5 m: up to 4189 neighbours).  usage (GPU box): tools/xl_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vgs_svgs_segmentation_amd as v

for name, n, side, kw in (("block r=8", 300_000, 1.15, dict(voxel_size=0.0625, graph_size=0.5)),
                          ("block r=10", 500_000, 1.30, dict(voxel_size=0.05, graph_size=0.5))):
    rng = np.random.default_rng(12)
    xyz = (rng.uniform(0, 1, (n, 3)) * side + np.array([1.0, -2.0, 0.2])).astype(np.float32)
    eng = v.Engine(v.default_params(2, **kw))
    eng.set_points(xyz)
    for it in range(2):
        t = time.perf_counter(); eng.run(); dt = (time.perf_counter() - t) * 1e3
    nn = eng.adjacency_counts()
    print(name, "points", n, "voxels", eng.counts()["voxels"], "n max", int(nn.max()), "n > 2048:", int((nn > 2048).sum()), f"step {dt:.1f} ms", eng.stage_times(), eng.schedule_counters(), flush=True)
