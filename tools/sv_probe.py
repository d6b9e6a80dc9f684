"""Supervoxel stage probe (GPU box): runs the SVGS pipeline on a small urban scene and prints the supervoxel count.
usage: python tools/sv_probe.py [points]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vgs_svgs_segmentation_amd as v

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
xyz = v.scenes.urban_scene(n)
p = v.default_params(3)
eng = v.Engine(p)
eng.set_points(xyz)
print("points", len(xyz), flush=True)
eng.supervoxels() if hasattr(eng, "supervoxels") else None
print("supervoxel stage done", flush=True)
eng.run()
labels, mx = eng.supervoxel_labels()
print("max label", mx, "labelled", int((labels > 0).sum()), flush=True)
