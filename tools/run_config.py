#!/usr/bin/env python3
"""Run one BASELINE.json configuration on the GPU and print counts + stage times (not the bench contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vgs_svgs_segmentation_amd as v

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else None
if cfg == "c2":
    xyz, p = v.scenes.pc_scene(n or 1_000_000), v.default_params(2, voxel_size=0.05)
elif cfg == "c3":
    xyz, p = v.scenes.urban_scene(n or 10_000_000), v.default_params(2, voxel_size=0.1)
elif cfg == "c4":
    xyz, p = v.scenes.urban_scene(n or 10_000_000), v.default_params(3)
elif cfg == "c1":
    xyz, p = v.scenes.town_scene(n or 500_000), v.default_params(2)
eng = v.Engine(p)
eng.set_points(xyz)
for it in range(3):
    t = time.perf_counter()
    if cfg == "c4":
        eng.supervoxels()   # run() keeps supervoxel labels once they exist
    eng.run(); dt = time.perf_counter() - t
    print(cfg, "run", it, f"{dt*1e3:.1f} ms", eng.counts(), {k: round(x, 2) for k, x in eng.stage_times().items()})
