"""BASELINE config 5 at its real size on ONE GPU: URB80M = 4 x 2 tiles of 10 M points (seeds 20260110..17), eight ranks of the native
tiled driver (libvgs_tiles.so, include/vgs_tiles.h) as threads of one process meeting through the thread communicator, eight engine
contexts on the one device (8 x 8 GB of 288 GB).  An 8-GPU node is the driver's to run; everything in csrc/tiles.cpp and csrc/multigpu.hip
that depends on SIZE -- halo strips of half a million points, 10^4..10^5 boundary records per rank (beyond the fixed-size payload: the
variable-length exchange), the grid replay over eight real bounding boxes, the per-rank tables -- is exercised here.

What is checked (the reference is single process, so the contract is "what ONE engine over the whole scene returns"):
  * every rank ends on one shared grid, and it is the octree box a single engine gives the 80 M points;
  * the 80 M labels equal the single engine's partition: exactly (up to renaming, dropped points included) outside closestCheck's
    candidates and the tiny segments one re-attached voxel decides about, >= 99.9 % of all points with them (P2);
  * the same number of kept segments on every rank, within 1 % of the single engine's;
  * a voxel that holds points of several ranks carries one label on all of them; the ground spans all eight ranks under one label;
  * no point is left unlabelled because its voxel is owned elsewhere (the labelled share per rank equals the single
    engine's share on that rank's points up to closestCheck);
  * the second run returns the first run's labels.
The evidence (per-rank points / halo / records / HBM, bytes through the ONE exchange, per-phase driver times) goes to
gpurun_out/c5_onegpu.json; the committed copy is profiles/r06_c5_onegpu.json."""
import ctypes as C
import json
import os
import threading
import time

import numpy as np
import pytest

from helpers import canonical_labels, partition_agreement

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILES = (4, 2)
N_PER = 10_000_000
PITCH = 50.0


def _hbm_in_use():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return total.value - free.value, total.value


def _rank_threads(gpu, world, tiles, pitch, parts, body, timeout=900.0):
    """`world` ranks of the native driver as threads of this process (VGS_TILES_COMM_LOCAL); body(rank, driver, points) -> anything."""
    from vgs_svgs_segmentation_amd import tiles_native as tn
    grp = tn.LocalGroup(world)
    out = [None] * world

    def rank_main(r):
        try:
            t = tn.NativeTiles(gpu.default_params(2, voxel_size=0.1), tn.COMM_LOCAL, grp.handle, r, world, tiles, pitch)
            try:
                out[r] = body(r, t, parts[r])
            finally:
                t.close()
        except Exception as ex:  # noqa: BLE001
            out[r] = ex
            grp.abort()
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout)
    hung = [r for r, t in enumerate(th) if t.is_alive()]
    if hung:
        grp.abort()
        pytest.fail(f"rank(s) {hung} still inside the driver after {timeout} s")
    grp.close()
    for r, o in enumerate(out):
        if isinstance(o, Exception):
            raise AssertionError(f"rank {r}: {o!r}")
    return out


def test_config5_urb80m_eight_ranks_on_one_gpu(gpu):
    world = TILES[0] * TILES[1]
    t0 = time.time()
    parts = [gpu.scenes.tiled_urban_scene(N_PER * world, tiles=TILES, tile_index=r) for r in range(world)]
    t_scene = time.time() - t0
    hbm0, hbm_total = _hbm_in_use()
    hbm_peak = [hbm0]
    all_up = threading.Barrier(world)

    def body(r, t, xyz):
        t.set_points(xyz)
        t.run()
        labels, kept = t.point_labels()
        first = dict(labels=labels.copy(), kept=kept, info=t.info(), times=t.times(), exchange=t.exchange(), counts=t.counts(), bbox=t.bbox(),
                     stage_ms=t.stage_times())
        all_up.wait()                      # every rank's tables are allocated: the device holds config 5 whole
        if r == 0:
            hbm_peak[0] = _hbm_in_use()[0]
        all_up.wait()
        t.run()                            # second run = first (same context: grow-only tables reused, atomics, stream timing)
        labels2, kept2 = t.point_labels()
        first["second_equal"] = bool(np.array_equal(labels2, first["labels"]) and kept2 == kept)
        first["times2"] = t.times()
        return first
    out = _rank_threads(gpu, world, TILES, PITCH, parts, body)
    tiled = np.concatenate([o["labels"] for o in out])
    n_own = [p.shape[0] for p in parts]
    rank_of_point = np.repeat(np.arange(world, dtype=np.int8), n_own)

    # ---- what only tiles have ----
    assert all(o["second_equal"] for o in out)
    assert len({o["kept"] for o in out}) == 1
    kept = out[0]["kept"]
    for o in out:
        np.testing.assert_array_equal(o["bbox"], out[0]["bbox"])                 # one shared grid
        assert o["info"]["n_outside"] <= 2          # (the generator rounds a point or two of a tile onto its neighbour's side of the edge)
        assert o["exchange"]["collectives"] == 3                                  # beyond the fixed 8192-record payload: the variable-length path
    records = [o["info"]["n_boundary_records"] for o in out]
    # SURVEY 8e estimated 10^4 .. 10^5 per rank; measured: 7.7 k on the four corner tiles (two inner edges), 11.7 k on the four others (three)
    assert min(records) > 2000 and 8192 < max(records) < 200_000, records
    halo = [o["info"]["n_local"] - n for o, n in zip(out, n_own)]
    redundancy = sum(o["info"]["n_local"] for o in out) / float(sum(n_own))
    assert 1.0 < redundancy < 1.12, redundancy                                    # SURVEY 8e estimates 1.07 for 60 m tiles, h = 1.1 m
    assert tiled.min() >= -1 and tiled.max() == kept - 1
    assert np.array_equal(np.unique(tiled[tiled >= 0]), np.arange(kept))           # global labels are dense over the ranks together
    lab_ranks = {}
    for r in range(world):
        for lab in np.unique(out[r]["labels"][out[r]["labels"] >= 0]).tolist():
            lab_ranks.setdefault(lab, set()).add(r)
    assert max(len(v) for v in lab_ranks.values()) == world                       # the ground: one label on all eight ranks

    # ---- against ONE engine over the 80 M points (rank order = the insertion order of a single octree) ----
    whole = np.concatenate(parts)
    del parts
    eng = gpu.Engine(gpu.default_params(2, voxel_size=0.1))
    eng.set_points(whole)
    eng.run()
    ref = eng.point_labels()
    c = eng.counts()
    assert c["points"] == N_PER * world
    np.testing.assert_array_equal(out[0]["bbox"], eng.bbox())                    # ... and it is the single engine's octree box
    agree = partition_agreement(tiled, ref)
    assert agree >= 0.999, agree
    assert abs(kept - c["kept"]) <= max(2, 0.01 * c["kept"]), (kept, c["kept"])
    pv = eng.point_voxel()
    off, _ = eng.lists("connect_cross")
    used = eng.attributes()["used"] != 0
    cand = used & (np.diff(off) == 1)
    del off
    root, _ = eng.node_labels()
    seg_size = np.bincount(root, minlength=root.size)[root]
    ok_vox = (~cand) & ((seg_size >= 8) | (seg_size <= 1))
    m = (pv >= 0) & ok_vox[np.maximum(pv, 0)]
    assert m.mean() > 0.85
    a, b = canonical_labels(tiled[m]), canonical_labels(ref[m])
    assert np.array_equal(a, b), f"{int((a != b).sum())} of {int(m.sum())} points differ outside closestCheck"
    del a, b
    # a voxel with points of several ranks (cut by a border, or at a four-owner corner): one label on all of them, the single engine's
    order = np.argsort(pv, kind="stable")
    pvs, rks, tl, rl = pv[order], rank_of_point[order], tiled[order], ref[order]
    start = np.flatnonzero(np.r_[True, pvs[1:] != pvs[:-1]])
    vox_of = pvs[start]
    rmin, rmax = np.minimum.reduceat(rks, start), np.maximum.reduceat(rks, start)
    lmin, lmax = np.minimum.reduceat(tl, start), np.maximum.reduceat(tl, start)
    shared = (rmin != rmax) & (vox_of >= 0)
    assert shared.sum() > 1000, int(shared.sum())
    assert np.array_equal(lmin[shared], lmax[shared])
    ref_first = rl[start]
    assert np.array_equal(lmin[shared] >= 0, ref_first[shared] >= 0) or \
        ((lmin[shared] >= 0) != (ref_first[shared] >= 0)).mean() < 0.01              # (closestCheck candidates among them)
    # nobody is left unlabelled because its voxel is owned elsewhere: per rank, the labelled share equals the single engine's on those points
    shares = []
    for r in range(world):
        sel = rank_of_point == r
        s_t, s_r = float((tiled[sel] >= 0).mean()), float((ref[sel] >= 0).mean())
        shares.append((s_t, s_r))
        assert abs(s_t - s_r) < 2e-3 and s_t > 0.7, (r, s_t, s_r)

    evidence = {
        "config": "BASELINE configs[4]: URB80M, 4 x 2 tiles x 10 M points, VGS, voxel 0.1 m, graph 0.5 m -- eight ranks of libvgs_tiles.so as threads "
                  "(VGS_TILES_COMM_LOCAL), eight contexts on ONE MI355X",
        "points": int(N_PER * world), "kept_segments_tiled": int(kept), "kept_segments_single_engine": int(c["kept"]),
        "partition_agreement_with_single_engine": float(agree), "points_outside_closestcheck_identical": int(m.sum()),
        "voxels_shared_by_ranks": int(shared.sum()), "halo_redundancy_factor": redundancy, "halo_redundancy_survey_8e": 1.07,
        "hbm_in_use_all_ranks_gb": (hbm_peak[0] - hbm0) / 1e9, "hbm_per_rank_gb": (hbm_peak[0] - hbm0) / 1e9 / world, "hbm_total_gb": hbm_total / 1e9,
        "scene_generation_s": t_scene,
        "per_rank": [{"rank": r, "own_points": int(n_own[r]), "halo_points": int(halo[r]), "boundary_records": int(records[r]),
                      "voxels": out[r]["counts"]["voxels"], "used_voxels": out[r]["counts"]["used"],
                      "exchange": out[r]["exchange"], "driver_ms_first_run": out[r]["times"], "driver_ms_second_run": out[r]["times2"],
                      "stage_ms": out[r]["stage_ms"], "labelled_share_tiled_vs_single": shares[r]} for r in range(world)],
        "note": "driver times are host wall time with eight ranks SHARING one GPU (the stages of eight tiles run side by side on one device): "
                "they bound nothing about an 8-GPU node; the exchange and merge phases are host work and do carry over",
    }
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "c5_onegpu.json"), "w") as f:
        json.dump(evidence, f, indent=1)
