#!/usr/bin/env python3
"""Randomised differential test, GPU path against the oracle (DevMath + lean): random scenes, sizes and parameters for a time
budget; every run compares the connect lists after the local cut and the point labels.  usage: fuzz_parity.py [seconds] [seed]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np
import vgs_svgs_segmentation_amd as v
import refcpu_py as oracle
from helpers import oracle_params, ragged_sets

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
WIDE = len(sys.argv) > 4 and sys.argv[4] == "wide"   # search balls of 6-10 voxels: the multi-wavefront classes and their hand-overs
METHOD = int(sys.argv[3]) if len(sys.argv) > 3 else 2   # 3: SVGS from a grid labelling (everything behind pcl::SupervoxelClustering)


def grid_supervoxels(xyz, seed_size, r):
    cell = np.floor(xyz.astype(np.float64) / seed_size).astype(np.int64)
    cell -= cell.min(0)
    code = (cell[:, 0] * 4096 + cell[:, 1]) * 4096 + cell[:, 2]
    _, inv = np.unique(code, return_inverse=True)
    labels = (inv + 1).astype(np.int32)
    labels[r.random(labels.size) < 0.01] = 0
    return labels, int(labels.max())

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def fuzzy(n, seed, sigma):
    r = np.random.default_rng(seed)
    side = np.sqrt(n / 6000.0)
    x, y = r.random(n) * side, r.random(n) * side
    z = 0.3 * np.sin(2.0 * x) * np.cos(1.5 * y) + r.normal(0, sigma, n) + 2.0
    return np.stack([x + 0.011, y + 0.017, z], axis=1).astype(np.float32)


def slab(n, seed, thick):
    r = np.random.default_rng(seed)
    side = np.sqrt(n / (2500.0 * max(thick / 0.1, 0.3)))
    x = r.random(n) * side + 0.013; y = r.random(n) * side + 0.027
    z = r.random(n) * thick + 0.02 * np.sin(3.0 * x) + 1.0
    return np.stack([x, y, z], axis=1).astype(np.float32)


t_end = time.time() + budget
runs = bad = 0
while time.time() < t_end:
    kind = rng.choice(["urban", "town", "pc", "fuzzy", "slab"])
    n = int(rng.integers(20_000, 90_000))
    seed = int(rng.integers(0, 1 << 30))
    if kind == "urban": xyz = v.scenes.urban_scene(n, seed=seed)
    elif kind == "town": xyz = v.scenes.town_scene(n, seed=seed)
    elif kind == "pc": xyz = v.scenes.pc_scene(n, seed=seed)
    elif kind == "fuzzy": xyz = fuzzy(n, seed, float(rng.choice([0.01, 0.03, 0.06, 0.1])))
    else: xyz = slab(n, seed, float(rng.choice([0.05, 0.2, 0.35])))
    kw = dict(voxel_size=float(rng.choice([0.06, 0.08, 0.1, 0.15])), graph_size=float(rng.choice([0.3, 0.4, 0.5, 0.6])),
              cut_thred=float(rng.choice([0.1, 0.3, 0.5, 0.7, 0.9])), sig_w=float(rng.choice([1.0, 2.0])),
              sig_n=float(rng.choice([0.2, 0.5])), sig_p=float(rng.choice([0.1, 0.2, 0.4])))
    if WIDE and METHOD == 2:
        kw["voxel_size"] = float(rng.choice([0.04, 0.05, 0.06]))
        kw["graph_size"] = float(rng.choice([0.3, 0.4, 0.5]))
        n = min(n, 50_000)
        xyz = xyz[: n]
    if kw["graph_size"] / kw["voxel_size"] > (10.5 if WIDE else 8.0):
        kw["graph_size"] = (10.0 if WIDE else 8.0) * kw["voxel_size"]
    if METHOD == 2 and rng.random() < 0.5:   # the size filters, and inputs with holes: non-finite points, repeated points
        kw.update(points_min=int(rng.choice([3, 5, 10, 20])), voxels_min=int(rng.choice([1, 3, 8])), adjacency_min=int(rng.choice([1, 3, 6])))
        if rng.random() < 0.5:
            xyz = xyz.copy()
            xyz[rng.random(xyz.shape[0]) < 0.002] = np.nan
            xyz = np.concatenate([xyz, xyz[rng.integers(0, xyz.shape[0], xyz.shape[0] // 50)]])
    if METHOD == 3:
        kw = dict(graph_size=float(rng.choice([0.4, 0.5, 0.8, 1.2])), cut_thred=kw["cut_thred"], sig_w=kw["sig_w"], sig_n=kw["sig_n"], sig_p=kw["sig_p"])
        seed_size = float(rng.choice([0.15, 0.25, 0.4]))
    p = v.default_params(METHOD, **kw)
    print("start", kind, n, seed, kw, flush=True)
    try:
        e = v.Engine(p); e.set_points(xyz)
        if METHOD == 3:
            labels, max_label = grid_supervoxels(xyz, seed_size, rng)
            e.set_supervoxel_labels(labels, max_label); e.svgs_segment()
        else:
            e.run()
    except v.VgsError as ex:
        print("skip", kind, n, kw, str(ex)[:80], flush=True)
        continue
    ref = oracle.run_svgs_from_labels(xyz, labels, max_label, oracle_params(oracle, p)) if METHOD == 3 else oracle.run_vgs(xyz, oracle_params(oracle, p))
    ok = True
    for which in ("connect_cut", "connect_final"):
        off, idx = e.lists(which); roff, ridx = ref.lists(which)
        if not (np.array_equal(off, roff) and ragged_sets(off, idx) == ragged_sets(roff, ridx)):
            ok = False
            print("MISMATCH", which, kind, n, seed, kw, flush=True)
    if not np.array_equal(e.point_labels(), ref.labels()[0]):
        ok = False
        print("MISMATCH labels", kind, n, seed, kw, flush=True)
    # element order (round 3): connect lists in merge-history order, getClusterIdx in the reference's DFS order; full adjacency lists
    if ok and runs % 2 == 0:
        for which in ("connect_cut", "connect_final"):
            off, idx = e.lists(which, "reference"); roff, ridx = ref.lists(which)
            if not (np.array_equal(off, roff) and np.array_equal(idx, ridx)):
                ok = False
                print("MISMATCH order", which, kind, n, seed, kw, flush=True)
        co, ci = e.clusters("reference"); rco, rci = ref.lists("clusters_points")
        if not (np.array_equal(co, rco) and np.array_equal(ci, rci)):
            ok = False
            print("MISMATCH cluster order", kind, n, seed, kw, flush=True)
        ao, ai = e.lists("adjacency"); rao, rai = ref.lists("adjacency")
        if not (np.array_equal(ao, rao) and np.array_equal(ai, rai)):
            ok = False
            print("MISMATCH adjacency (all voxels)", kind, n, seed, kw, flush=True)
    sc = e.schedule_counters()
    runs += 1; bad += 0 if ok else 1
    print(f"run {runs} {kind} n={n} used={e.counts()['used']} handed={sc['handed_over']}/{sc['handed_over_large']} banded={sc['banded']} sent_on={sc['dense_sent_on']} {'ok' if ok else 'BAD'}", flush=True)
print(f"{runs} runs, {bad} mismatches")
sys.exit(1 if bad else 0)
