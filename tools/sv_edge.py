#!/usr/bin/env python3
"""Supervoxel stage on parameter corners the campaigns do not draw (GPU box): dense tiles (coarse voxels on a small scene), voxels of
several metres (the tile-relative 32-bit sums fall back to 64-bit atomics), seeds as small as a voxel, both modes, against the oracle.
usage: python tools/sv_edge.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np
import vgs_svgs_segmentation_amd as v
import refcpu_py as oracle
from helpers import oracle_params

cases = [
    ("urban", 150_000, dict(voxel_size=0.3, seed_size=1.2, graph_size=1.5)),
    ("urban", 150_000, dict(voxel_size=4.0, seed_size=16.0, graph_size=20.0)),
    ("urban", 150_000, dict(voxel_size=6.0, seed_size=12.0, graph_size=30.0)),
    ("pc", 120_000, dict(voxel_size=0.02, seed_size=0.06, graph_size=0.2)),
    ("town", 200_000, dict(voxel_size=0.1, seed_size=0.1, graph_size=0.5)),
    ("urban", 200_000, dict(voxel_size=0.05, seed_size=1.0, graph_size=0.5)),
    ("urban*40", 200_000, dict(voxel_size=4.0, seed_size=12.0, graph_size=20.0)),     # a scene of kilometres: tile-relative positions above 2^21 units
    ("urban*100", 200_000, dict(voxel_size=10.0, seed_size=40.0, graph_size=50.0)),
]
bad = 0
for mode in (0, 1):
    for kind, n, kw in cases:
        base, _, scale = kind.partition("*")
        xyz = {"urban": v.scenes.urban_scene, "town": v.scenes.town_scene, "pc": v.scenes.pc_scene}[base](n, seed=11)
        if scale: xyz = (xyz * np.float32(float(scale))).astype(np.float32)
        p = v.default_params(3, vccs_mode=mode, **kw)
        try:
            e = v.Engine(p); e.set_points(xyz); e.supervoxels()
        except v.VgsError as ex:
            print("skip", mode, kind, kw, str(ex)[:80], flush=True)
            continue
        labels, mx = e.supervoxel_labels()
        ref_labels, ref_max = (oracle.vccs_pcl if mode == 1 else oracle.vccs)(xyz, oracle_params(oracle, p))
        ok = mx == ref_max and np.array_equal(labels, ref_labels)
        bad += 0 if ok else 1
        print("ok " if ok else "MISMATCH", "mode", mode, kind, n, kw, "supervoxels", mx, flush=True)
print("cases with mismatches:", bad)
sys.exit(1 if bad else 0)
