#!/usr/bin/env python3
"""FINAL segments of SVGS with the supervoxels of vccs_mode 0 (synchronous rounds) against vccs_mode 1 (PCL's order), P2 of SURVEY 8c on points."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vgs_svgs_segmentation_amd as v
from helpers import p2_protocol
out = {}
for name, fn, n in (("urban", v.scenes.urban_scene, 200_000), ("pc", v.scenes.pc_scene, 120_000), ("town", v.scenes.town_scene, 150_000), ("urban2M", v.scenes.urban_scene, 2_000_000)):
    xyz = fn(n)
    lab = []
    for mode in (0, 1):
        e = v.Engine(v.default_params(3, vccs_mode=mode)); e.set_points(xyz); e.run(); lab.append(e.point_labels())
    r = p2_protocol(lab[0], lab[1], np.arange(xyz.shape[0]), min_voxels=2000)
    out[name] = r
    print(name, json.dumps(r), flush=True)
