import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import vgs_svgs_segmentation_amd as v
xyz = v.scenes.urban_scene(10_000_000)
p = v.default_params(2, voxel_size=0.1)
eng = v.Engine(p)
for it in range(6):
    t0 = time.perf_counter(); eng.set_points(xyz); t1 = time.perf_counter(); eng.run(); t2 = time.perf_counter(); lab = eng.point_labels(); t3 = time.perf_counter()
    print(f"H2D {1e3*(t1-t0):.2f} ms, run {1e3*(t2-t1):.2f} ms, D2H labels {1e3*(t3-t2):.2f} ms, total {1e3*(t3-t0):.2f} ms -> {1e-6*xyz.shape[0]/(t3-t0):.0f} Mpts/s")
