#!/usr/bin/env python3
"""One small scene through the library named by VGS_LIB against the oracle (connect lists after the cut, labels); for A/B builds."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np
import vgs_svgs_segmentation_amd as v
import refcpu_py as oracle
from helpers import oracle_params, ragged_sets
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
xyz = v.scenes.urban_scene(n, seed=3)
p = v.default_params(2)
e = v.Engine(p); e.set_points(xyz); e.run()
ref = oracle.run_vgs(xyz, oracle_params(oracle, p))
off, idx = e.lists("connect_cut"); roff, ridx = ref.lists("connect_cut")
same = np.array_equal(off, roff) and ragged_sets(off, idx) == ragged_sets(roff, ridx)
print(os.environ.get("VGS_LIB", "base"), "connect_cut", "same" if same else "DIFFERENT", "labels", "same" if np.array_equal(e.point_labels(), ref.labels()[0]) else "DIFFERENT", e.schedule_counters())
