#!/usr/bin/env python3
"""Randomised differential test of the supervoxel stage (csrc/vccs.hip) and the SVGS pipeline behind it against the oracle:
random scenes, voxel / seed sizes, importances.  usage: fuzz_vccs.py [seconds] [seed] [vccs_mode]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np
import vgs_svgs_segmentation_amd as v
import refcpu_py as oracle
from helpers import oracle_params

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
MODE = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 1: supervoxels in PCL's own order (vccs_mode 1) against the sequential restatement
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
runs = bad = 0
while time.time() < t_end:
    kind = rng.choice(["urban", "town", "pc"])
    n = int(rng.integers(20_000, 120_000))
    seed = int(rng.integers(0, 1 << 30))
    xyz = {"urban": v.scenes.urban_scene, "town": v.scenes.town_scene, "pc": v.scenes.pc_scene}[kind](n, seed=seed)
    kw = dict(voxel_size=float(rng.choice([0.04, 0.05, 0.08])), seed_size=float(rng.choice([0.2, 0.25, 0.4])),
              graph_size=float(rng.choice([0.4, 0.5, 0.7])), spatial_impt=float(rng.choice([0.25, 0.5, 1.0])),
              normal_impt=float(rng.choice([0.25, 0.75, 1.0])), cut_thred=float(rng.choice([0.3, 0.5, 0.7])))
    p = v.default_params(3, vccs_mode=MODE, **kw)
    print("start", kind, n, seed, kw, flush=True)
    try:
        e = v.Engine(p); e.set_points(xyz); e.run()
    except v.VgsError as ex:
        print("skip", str(ex)[:90], flush=True)
        continue
    labels, max_label = e.supervoxel_labels()
    ref_labels, ref_max = (oracle.vccs_pcl if MODE == 1 else oracle.vccs)(xyz, oracle_params(oracle, p))
    ok = max_label == ref_max and np.array_equal(labels, ref_labels)
    if ok:
        ref = oracle.run_svgs_from_labels(xyz, ref_labels, ref_max, oracle_params(oracle, p))
        ok = bool(np.array_equal(e.point_labels(), ref.labels()[0]))
        if not ok:
            print("MISMATCH segmentation", flush=True)
    else:
        print("MISMATCH supervoxels", max_label, ref_max, int((labels != ref_labels).sum()), flush=True)
    runs += 1; bad += 0 if ok else 1
    print(f"run {runs} {kind} n={n} supervoxels={max_label} {'ok' if ok else 'BAD'}", flush=True)
print(f"{runs} runs, {bad} mismatches")
sys.exit(1 if bad else 0)
