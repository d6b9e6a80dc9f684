#!/usr/bin/env python3
"""Adjacency stage alone on URB10M (for A/B of library variants: VGS_LIB=libvgs_hip_<name>.so).  usage: tools/adj_time.py [points]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vgs_svgs_segmentation_amd as v
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
xyz = v.scenes.urban_scene(n)
eng = v.Engine(v.default_params(2, voxel_size=0.1))
eng.set_points(xyz)
ts = []
for it in range(6):
    eng.set_points(xyz); eng.voxelize(); eng.features(); eng.adjacency()
    ts.append(eng.stage_times()["adjacency"])
print(os.environ.get("VGS_LIB", "base"), "adjacency ms", " ".join(f"{t:.3f}" for t in ts[1:]))
