#!/usr/bin/env python3
"""Re-run cases of tools/fuzz_tiles.py by their printed parameters: repro_tiles.py n_per seed0 voxel graph cut sig_w [...more cases]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
torch.zeros(1, device="cuda:0")
import vgs_svgs_segmentation_amd as v
from helpers import canonical_labels, partition_agreement
from test_gpu_tiles import _run_tiled, _single
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

a = sys.argv[1:]
for k in range(0, len(a), 6):
    n_per, seed0 = int(a[k]), int(a[k + 1])
    kw = dict(voxel_size=float(a[k + 2]), graph_size=float(a[k + 3]), cut_thred=float(a[k + 4]), sig_w=float(a[k + 5]))
    world = 2
    pitch = 50.0 * np.sqrt(n_per / 10_000_000)
    gen = np.concatenate([v.scenes.tiled_urban_scene(n_per * world, tiles=(world, 1), seed0=seed0, tile_index=r) for r in range(world)])
    tiles = [gen[gen[:, 0] < 0.0], gen[gen[:, 0] >= 0.0]]
    whole = np.concatenate(tiles)
    eng = _single(v, whole, kw)
    ref = eng.point_labels()
    out = _run_tiled(v, tiles, kw, pitch)
    tiled = np.concatenate([out[r][0] for r in range(world)])
    off, idx = eng.lists("connect_cross")
    V = off.size - 1
    rows = np.repeat(np.arange(V), np.diff(off))
    _, core = connected_components(coo_matrix((np.ones(idx.size, np.int8), (rows, idx)), shape=(V, V)), directed=False)
    core_size = np.bincount(core, minlength=V)[core]
    pv = eng.point_voxel()
    m = (pv >= 0) & (core_size >= 8)[np.maximum(pv, 0)]
    ca, cb = canonical_labels(tiled[m]), canonical_labels(ref[m])
    diff = np.nonzero(ca != cb)[0]
    print(f"case {n_per} {seed0} {kw}: agree={partition_agreement(tiled, ref):.5f} identical={diff.size == 0} differing points {diff.size} kept={out[0][1]}/{eng.counts()['kept']}", flush=True)
    if diff.size:
        pts = whole[m][diff]
        vox = np.unique(pv[m][diff])
        cen = eng.voxel_centers()
        print("  voxels", vox[:12], "centres x", np.round(cen[vox[:12], 0], 3), "tiled labels", np.unique(tiled[m][diff])[:8], "ref labels", np.unique(ref[m][diff])[:8])
        print("  x range of differing points", float(pts[:, 0].min()), float(pts[:, 0].max()), " border at 0; halo", 2 * kw["graph_size"] + kw["voxel_size"])
