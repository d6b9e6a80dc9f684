#!/usr/bin/env python3
"""Randomised test of the tiled (multi-GPU) path on one GPU: two simulated ranks against one engine over the whole scene,
random tile contents, sizes and parameters.  Compared: the part of the labelling that does not depend on closestCheck's scan
order (tests/test_gpu_tiles.py: exact part) must be identical up to renaming, the rest must agree to 99.9 %.
usage: fuzz_tiles.py [seconds] [seed]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
torch.zeros(1, device="cuda:0")   # torch first: it must see the GPU before the library's runtime is up
import vgs_svgs_segmentation_amd as v
from helpers import canonical_labels, partition_agreement
from test_gpu_tiles import _run_tiled, _single

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
runs = bad = 0
while time.time() < t_end:
    world = 2
    n_per = int(rng.integers(40_000, 220_000))
    seed0 = int(rng.integers(0, 1 << 30))
    kw = dict(voxel_size=float(rng.choice([0.08, 0.1, 0.15])), graph_size=float(rng.choice([0.3, 0.4, 0.5])),
              cut_thred=float(rng.choice([0.2, 0.3, 0.5])), sig_w=float(rng.choice([1.0, 2.0])))
    pitch = 50.0 * np.sqrt(n_per / 10_000_000)
    gen = np.concatenate([v.scenes.tiled_urban_scene(n_per * world, tiles=(world, 1), seed0=seed0, tile_index=r) for r in range(world)])
    # every rank loads the points of ITS region (objects of a generated tile may reach over the border into the neighbour's)
    tiles = [gen[gen[:, 0] < 0.0], gen[gen[:, 0] >= 0.0]]
    print("start", n_per, seed0, kw, flush=True)
    whole = np.concatenate(tiles)
    eng = _single(v, whole, kw)
    ref = eng.point_labels()
    out = _run_tiled(v, tiles, kw, pitch)
    tiled = np.concatenate([out[r][0] for r in range(world)])
    assert tiled.shape[0] == whole.shape[0]
    agree = partition_agreement(tiled, ref)
    # exact part: voxels of the components the mutual connect lists form BEFORE closestCheck, if such a component alone passes
    # the size filter with room to spare -- re-attachments (whose choice depends on the scan order) only ever hang
    # single voxels onto these components, never join two of them
    off, idx = eng.lists("connect_cross")
    V = off.size - 1
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    rows = np.repeat(np.arange(V), np.diff(off))
    _, core = connected_components(coo_matrix((np.ones(idx.size, np.int8), (rows, idx)), shape=(V, V)), directed=False)
    core_size = np.bincount(core, minlength=V)[core]
    pv = eng.point_voxel()
    ok_vox = core_size >= 8
    m = (pv >= 0) & ok_vox[np.maximum(pv, 0)]
    a, b = canonical_labels(tiled[m]), canonical_labels(ref[m])
    exact = bool(np.array_equal(a, b))
    ok = exact and out[0][1] == out[1][1] and (agree >= 0.999 or m.mean() < 0.9)   # the agreement bar is for scenes closestCheck hardly touches
    runs += 1; bad += 0 if ok else 1
    print(f"run {runs} n_per={n_per} agree={agree:.5f} exact_part={m.mean():.3f} identical={exact} kept={out[0][1]}/{eng.counts()['kept']} {'ok' if ok else 'BAD'}", flush=True)
print(f"{runs} runs, {bad} mismatches")
sys.exit(1 if bad else 0)
