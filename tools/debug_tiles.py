#!/usr/bin/env python3
"""Diagnose a failing case of tools/fuzz_tiles.py: which connection of the single engine is missing in the tiled run?
usage: debug_tiles.py n_per seed0 voxel graph cut sig_w"""
import os, sys, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
torch.zeros(1, device="cuda:0")
import vgs_svgs_segmentation_amd as v
from vgs_svgs_segmentation_amd.dist import TiledSegmenter
from helpers import canonical_labels
from test_gpu_tiles import FakeDist, _single

a = sys.argv[1:]
n_per, seed0 = int(a[0]), int(a[1])
kw = dict(voxel_size=float(a[2]), graph_size=float(a[3]), cut_thred=float(a[4]), sig_w=float(a[5]))
world = 2
pitch = 50.0 * np.sqrt(n_per / 10_000_000)
gen = np.concatenate([v.scenes.tiled_urban_scene(n_per * world, tiles=(world, 1), seed0=seed0, tile_index=r) for r in range(world)])
tiles = [gen[gen[:, 0] < 0.0], gen[gen[:, 0] >= 0.0]]
whole = np.concatenate(tiles)
eng = _single(v, whole, kw)
ref = eng.point_labels()
fd = FakeDist(world)
segs = [None] * world
def work(r):
    fd.tls.rank = r
    d = torch.from_numpy(tiles[r]).to("cuda:0")
    seg = TiledSegmenter(v.default_params(2, **kw), fd, tiles=(world, 1), rank=r, world=world, pitch=pitch)
    seg.set_points_device(d, tiles[r]); seg.run(); segs[r] = seg
th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
tiled = np.concatenate([segs[r].point_labels() for r in range(world)])
pv = eng.point_voxel()
cen = eng.voxel_centers()
V = cen.shape[0]
# voxel-level labels of both runs (single-engine voxel ids)
vt = np.full(V, -2); vr = np.full(V, -2)
ok = pv >= 0
vt[pv[ok]] = tiled[ok]; vr[pv[ok]] = ref[ok]
# segments of the single engine that the tiled run MERGES
merged = {}
for l in np.unique(vt[vt >= 0]):
    rl = np.unique(vr[(vt == l) & (vr >= 0)])
    if rl.size > 1:
        merged[int(l)] = rl.tolist()
print("tiled segments that hold several segments of the single engine:", {k: x for k, x in list(merged.items())[:6]})
def key(c):
    return tuple(np.round(c / kw["voxel_size"] * 2).astype(np.int64).tolist())
gmap = {key(cen[i]): i for i in range(V)}
for r in range(world):
    e = segs[r].engine
    cc = e.voxel_centers()
    g_of = np.array([gmap.get(key(cc[i]), -1) for i in range(cc.shape[0])])
    o, x = e.lists("connect_final")
    oc, xc = e.lists("connect_cross")
    lo, hi = segs[r].regions[r]
    owned = (cc[:, 0] >= lo[0]) & (cc[:, 0] < hi[0]) & (cc[:, 1] >= lo[1]) & (cc[:, 1] < hi[1])
    shown = 0
    for a_ in range(cc.shape[0]):
        ga = g_of[a_]
        if ga < 0: continue
        for b_ in x[o[a_]:o[a_ + 1]]:
            gb = g_of[b_]
            if gb < 0 or a_ >= b_: continue
            if vr[ga] >= 0 and vr[gb] >= 0 and vr[ga] != vr[gb] and (owned[a_] or owned[b_]):
                in_cross = b_ in xc[oc[a_]:oc[a_ + 1]].tolist()
                o0, x0 = off_f, idx_f = eng.lists("connect_final")
                print(f"  rank {r}: trusted final edge local {a_}-{b_} (global {ga}-{gb}) owned ({owned[a_]},{owned[b_]}) in_cross {in_cross} "
                      f"x=({cc[a_,0]:.3f},{cc[b_,0]:.3f}) y=({cc[a_,1]:.3f},{cc[b_,1]:.3f}) z=({cc[a_,2]:.3f},{cc[b_,2]:.3f}) ref labels ({vr[ga]},{vr[gb]}) "
                      f"single final has it: {gb in x0[o0[ga]:o0[ga+1]].tolist()} single lists: {x0[o0[ga]:o0[ga+1]].tolist()[:6]} / {x0[o0[gb]:o0[gb+1]].tolist()[:6]} "
                      f"rank lists: {x[o[a_]:o[a_+1]].tolist()[:6]} / {x[o[b_]:o[b_+1]].tolist()[:6]}")
                shown += 1
                if shown >= 6: break
        if shown >= 6: break
sys.exit(0)
