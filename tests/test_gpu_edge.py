"""Edge cases of the hot path against the oracle: empty and degenerate clouds, non-finite points, padded points,
clouds far from the origin whose bounding box grows in every direction, and the documented limits."""
import numpy as np
import pytest

from helpers import canonical_labels, oracle_params, ragged_sets

pytestmark = pytest.mark.gpu


def _both(gpu, oracle, xyz, **kw):
    p = gpu.default_params(2, **kw)
    eng = gpu.Engine(p)
    eng.set_points(xyz)
    eng.run()
    ref = oracle.run_vgs(np.ascontiguousarray(xyz[:, :3]), oracle_params(oracle, p))
    return eng, ref


def _same(eng, ref):
    c = eng.counts()
    assert (c["voxels"], c["clusters"], c["kept"]) == (ref.V, ref.clusters_num, ref.kept_clusters)
    pl_ref, nc_ref = ref.labels()
    np.testing.assert_array_equal(eng.point_labels(), pl_ref)
    root, _ = eng.node_labels()
    np.testing.assert_array_equal(canonical_labels(root), canonical_labels(nc_ref))


def test_empty_cloud(gpu):
    eng = gpu.Engine(gpu.default_params(2))
    eng.set_points(np.zeros((0, 3), np.float32))
    eng.run()
    c = eng.counts()
    assert (c["points"], c["voxels"], c["clusters"], c["kept"]) == (0, 0, 0, 0)
    assert eng.point_labels().size == 0


def test_only_non_finite_points(gpu):
    xyz = np.full((100, 3), np.nan, np.float32)
    xyz[::3, 1] = np.inf
    eng = gpu.Engine(gpu.default_params(2))
    eng.set_points(xyz)
    eng.run()
    c = eng.counts()
    assert (c["points"], c["finite"], c["voxels"], c["kept"]) == (100, 0, 0, 0)
    assert (eng.point_labels() == -1).all()


@pytest.mark.parametrize("n", [1, 5, 11, 40])
def test_tiny_clouds(gpu, oracle, n):
    rng = np.random.default_rng(n)
    xyz = (rng.standard_normal((n, 3)) * 0.05 + np.array([2.0, -1.0, 0.5])).astype(np.float32)
    _same(*_both(gpu, oracle, xyz))


def test_no_voxel_with_enough_points(gpu, oracle):
    """A cloud so sparse that no voxel reaches points_min: every stage runs on zero used voxels, and every getter must cope
    (vgs_get_lists(3) read the re-attachment table of a merge stage that never allocated it: found by tools/fuzz_parity.py)."""
    xyz = gpu.scenes.urban_scene(21_170, seed=206792296)
    p = gpu.default_params(2, voxel_size=0.08, cut_thred=0.9)
    eng = gpu.Engine(p); eng.set_points(xyz); eng.run()
    c = eng.counts()
    assert c["used"] == 0 and c["kept"] == 0 and c["clusters"] == c["voxels"] > 0
    for which in ("connect_cut", "connect_cross", "connect_final"):
        off, idx = eng.lists(which)
        assert off[-1] == 0 and idx.size == 0
    # findAllVoxelAdjacency builds a list for every voxel whether it is used or not (VS:236-263): the same as the oracle's
    ref = oracle.run_vgs(xyz, oracle_params(oracle, p))
    off, idx = eng.lists("adjacency")
    roff, ridx = ref.lists("adjacency")
    np.testing.assert_array_equal(off, roff)
    np.testing.assert_array_equal(idx, ridx)
    assert off[-1] >= c["voxels"]          # every voxel is its own first neighbour
    assert (eng.point_labels() == -1).all()
    root, kept = eng.node_labels()
    assert (root == np.arange(c["voxels"])).all() and (kept == -1).all()
    eng.attributes(); eng.voxel_table(); eng.clusters(); eng.schedule_counters()


def test_duplicate_points_and_sparse_voxels(gpu, oracle):
    rng = np.random.default_rng(9)
    base = (rng.standard_normal((300, 3)) * np.array([3.0, 3.0, 0.02])).astype(np.float32)
    xyz = np.concatenate([base, base[:150], base[:150], np.tile(base[:1], (50, 1))])   # repeated points, one voxel with 50 copies
    _same(*_both(gpu, oracle, xyz))


def test_non_finite_points_are_skipped(gpu, oracle):
    xyz = gpu.scenes.town_scene(30_000).copy()
    xyz[::97, 0] = np.nan
    xyz[5::211, 2] = np.inf
    xyz[0] = np.nan                       # the first point defines the octree box: here the first FINITE one does
    eng, ref = _both(gpu, oracle, xyz)
    _same(eng, ref)
    bad = ~np.isfinite(xyz).all(axis=1)
    assert (eng.point_labels()[bad] == -1).all() and eng.counts()["finite"] == int((~bad).sum())


def test_padded_points_give_the_same_result(gpu):
    xyz = gpu.scenes.town_scene(40_000)
    a = gpu.Engine(gpu.default_params(2)); a.set_points(xyz); a.run()
    xyz4 = np.concatenate([xyz, np.ones((xyz.shape[0], 1), np.float32)], axis=1)      # pcl::PointXYZ: 16-byte points
    b = gpu.Engine(gpu.default_params(2)); b.set_points(xyz4); b.run()
    np.testing.assert_array_equal(a.point_labels(), b.point_labels())


@pytest.mark.parametrize("shift", [(-140.0, 120.0, 3.0), (149.0, -149.0, -2.0)])
def test_far_from_origin_and_growth_in_all_directions(gpu, oracle, shift):
    xyz = gpu.scenes.town_scene(40_000).copy()
    rng = np.random.default_rng(4)
    xyz = xyz[rng.permutation(xyz.shape[0])]      # the box grows towards every side as points arrive
    xyz += np.array(shift, np.float32)
    eng, ref = _both(gpu, oracle, xyz)
    t, rt = eng.voxel_table(), ref.voxel_table()
    assert np.array_equal(t["key"], rt["key"]) and np.array_equal(t["start"], rt["start"])
    _same(eng, ref)


def test_limits_are_reported_not_silently_wrong(gpu):
    xyz = gpu.scenes.town_scene(5_000)
    rng = np.random.default_rng(1)
    dense = (rng.uniform(0, 1, (20_000, 3)) * np.array([0.3, 0.3, 0.04])).astype(np.float32)   # 44 points per 0.02 m voxel
    eng = gpu.Engine(gpu.default_params(2, voxel_size=0.02, graph_size=0.5))          # ball of radius 25 voxels > 8192 offsets
    eng.set_points(dense)
    with pytest.raises(gpu.VgsError) as e:
        eng.run()
    assert "UNSUPPORTED" in str(e.value)
    far = xyz.copy(); far[-1] = (3.0e6, 0, 0)                                         # octree depth would exceed 21 at 0.15 m
    eng2 = gpu.Engine(gpu.default_params(2)); eng2.set_points(far)
    with pytest.raises(gpu.VgsError):
        eng2.run()


def test_tables_that_cannot_fit_are_refused_with_an_estimate(gpu):
    """The adjacency / connect tables are dense (used voxels x lattice offsets of the ball x 12 B): a sheet of ten million used voxels seen through a
    ball of ten voxels would need half a terabyte.  The adjacency stage says so -- VGS_E_NOMEM with the numbers and what to change -- before
    it allocates anything, and the context stays usable."""
    #  sized from the device that is there, not from the 288 GB the round was developed on: a sheet whose tables at a ball of ten
    # voxels (4189 lattice offsets x 12 B per used voxel) need 1.5 x the device's memory
    import torch
    total_b = torch.cuda.mem_get_info()[1]
    side = int(np.ceil(np.sqrt(1.5 * total_b / (4189.0 * 12.0) / 0.65)))   # (about 0.7 of the sheet's cells come out as used voxels: float cell edges)
    assert 500 < side < 6000, side
    ij = np.stack(np.meshgrid(np.arange(side, dtype=np.float32), np.arange(side, dtype=np.float32), indexing="ij"), -1).reshape(-1, 2)
    cell = np.concatenate([ij * 0.05 + 0.0125, ij * 0.05 + 0.0375])                  # two points in every voxel of a 160 m sheet
    xyz = np.concatenate([cell, np.full((cell.shape[0], 1), 1.02, np.float32)], axis=1).astype(np.float32)
    eng = gpu.Engine(gpu.default_params(2, voxel_size=0.05, graph_size=0.5, points_min=1))
    eng.set_points(xyz)
    eng.voxelize(); eng.features()
    assert eng.counts()["used"] * 4189.0 * 12.0 > 1.2 * total_b
    with pytest.raises(gpu.VgsError) as e:
        eng.adjacency()
    assert "VGS_E_NOMEM" in str(e.value) and "tables need" in str(e.value) and "GB" in str(e.value), str(e.value)
    p2 = gpu.default_params(2, voxel_size=0.05, graph_size=0.15, points_min=1)       # a ball of three voxels fits
    eng.set_params(p2)
    eng.run()                                  # (two points per voxel carry no normal: every voxel stays alone, none is kept)
    c = eng.counts()
    assert c["adj"] > c["used"] and c["clusters"] >= c["used"], c


_BLOCK_REF = {}


@pytest.mark.parametrize("wide", ["pair_lists", "general_kernel"])
def test_neighbourhoods_above_2048_voxels(gpu, oracle, monkeypatch, wide):
    """A solid block seen through a ball of eight voxels: 275 voxels have more than 2048 used neighbours.  The reference sizes its
    matrix to any n (voxel_segmentation.h:1815-1818, 1913-1933); until round 3 one such voxel ended the run with VGS_E_UNSUPPORTED.
    Round 4 cut them with the extra-large instantiation of the general kernel (every pair weight again in every histogram round:
    3 ms a voxel); round 5 cuts them from the pair lists (k_localcut_pg<4224, ...>: whole balls of up to ten voxels).  Both ways:
    connect lists and labels identical to the oracle (DevMath + lean; its local cuts on every core -- 1.2e10 pair weights)."""
    import os
    if wide == "general_kernel":
        monkeypatch.setenv("VGS_PG_WIDE", "0")
    rng = np.random.default_rng(12)
    xyz = (rng.uniform(0, 1, (300_000, 3)) * 1.15 + np.array([1.0, -2.0, 0.2])).astype(np.float32)
    p = gpu.default_params(2, voxel_size=0.0625, graph_size=0.5)
    eng = gpu.Engine(p)
    eng.set_points(xyz)
    eng.run()
    n = eng.adjacency_counts()
    sc = eng.schedule_counters()
    n_xl = int((n > 2048).sum())
    assert n.max() > 2048 and sc["outside_limits"] == 0, (n.max(), sc)
    if wide == "general_kernel":
        assert sc["extra_large"] == n_xl and sc["pair_list_cut"] == 0, sc
    else:
        assert sc["extra_large"] == 0 and sc["pair_list_cut"] >= n_xl, sc
    if "ref" not in _BLOCK_REF:
        _BLOCK_REF["ref"] = oracle.run_vgs(xyz, oracle_params(oracle, p, threads=os.cpu_count() or 1))
    ref = _BLOCK_REF["ref"]
    for which in ("connect_cut", "connect_cross", "connect_final"):
        off, idx = eng.lists(which)
        roff, ridx = ref.lists(which)
        assert np.array_equal(off, roff)
        assert ragged_sets(off, idx) == ragged_sets(roff, ridx)
    np.testing.assert_array_equal(eng.point_labels(), ref.labels()[0])
    assert eng.counts()["kept"] == ref.kept_clusters


_BALL10_REF = {}


@pytest.mark.parametrize("wide", ["pair_lists", "no_extra_large_pair_lists", "general_kernel"])
def test_ball_of_ten_voxels_solid_block(gpu, oracle, monkeypatch, wide):
    """BASELINE config 2's own ratio (graph 0.5 / voxel 0.05) on a solid block: the ball's offsets reach ten voxels per axis while the loop
    that enumerates them runs to eleven.  Round 5 gated the extra-large pair-list instantiation on the loop bound, so it was never launched at
    this ratio and the voxels class D's instantiation had queued for it (every neighbourhood above 1024 voxels) were never cut.
    Three ways through the wide classes, each identical to the oracle: the extra-large pair-list kernel; without it (VGS_NO_PG_XL: class D's
    instantiation hands on to the dense kernel and the general kernels behind it, as for a wider ball); no pair lists for the wide classes."""
    import os
    if wide == "general_kernel":
        monkeypatch.setenv("VGS_PG_WIDE", "0")
    if wide == "no_extra_large_pair_lists":
        monkeypatch.setenv("VGS_NO_PG_XL", "1")
    rng = np.random.default_rng(13)
    xyz = (rng.uniform(0, 1, (100_000, 3)) * 0.8 + np.array([1.0, -2.0, 0.2])).astype(np.float32)
    p = gpu.default_params(2, voxel_size=0.05, graph_size=0.5)
    eng = gpu.Engine(p)
    eng.set_points(xyz)
    eng.run()
    n = eng.adjacency_counts()
    sc = eng.schedule_counters()
    n_d = int((n > 1024).sum())
    assert n.max() > 2048 and n_d > 1000 and sc["outside_limits"] == 0, (n.max(), n_d, sc)
    if wide == "pair_lists":
        assert sc["extra_large"] == 0 and sc["pair_list_cut"] >= n_d, sc      # (rows cut, not work items visited)
    elif wide == "no_extra_large_pair_lists":
        assert sc["handed_over_large"] > 0.9 * n_d and sc["pair_list_cut"] < sc["handed_over_large"], sc   # (n counts unused neighbours too)
    else:
        assert sc["pair_list_cut"] == 0, sc
    if "ref" not in _BALL10_REF:
        _BALL10_REF["ref"] = oracle.run_vgs(xyz, oracle_params(oracle, p, threads=os.cpu_count() or 1))
    ref = _BALL10_REF["ref"]
    for which in ("connect_cut", "connect_cross", "connect_final"):
        off, idx = eng.lists(which)
        roff, ridx = ref.lists(which)
        assert np.array_equal(off, roff)
        assert ragged_sets(off, idx) == ragged_sets(roff, ridx)
    np.testing.assert_array_equal(eng.point_labels(), ref.labels()[0])
    assert eng.counts()["kept"] == ref.kept_clusters


def test_dense_volume_adjacency_second_pass(gpu, oracle):
    """A solid block of points: the search ball (radius 8 voxels, 2109 lattice cells) is full, more neighbours than the
    first adjacency pass holds (2048), so the rows go through the second pass; the lists must equal the oracle's."""
    rng = np.random.default_rng(12)
    xyz = (rng.uniform(0, 1, (400_000, 3)) * np.array([1.3, 1.3, 1.3]) + np.array([1.0, -2.0, 0.2])).astype(np.float32)
    p = gpu.default_params(2, voxel_size=0.0625, graph_size=0.5)
    eng = gpu.Engine(p)
    eng.set_points(xyz)
    eng.voxelize(); eng.features(); eng.adjacency()
    ref = oracle.run_vgs_adjacency(xyz, oracle_params(oracle, p)) if hasattr(oracle, "run_vgs_adjacency") else None
    off, idx = eng.lists("adjacency")
    n = np.diff(off)
    assert n.max() > 2048                     # the case is what it claims to be
    if ref is not None:
        roff, ridx = ref.lists("adjacency")
        assert np.array_equal(off, roff) and np.array_equal(idx, ridx)
    else:
        # independent check with a KD-tree on the voxel centres (float32 centres, FLANN's d2 < float(r*r) predicate)
        from scipy.spatial import cKDTree
        cen = eng.voxel_centers().astype(np.float64)
        used = eng.attributes()["used"].astype(bool)
        tree = cKDTree(cen)
        probe = np.nonzero(used)[0][:: max(1, used.sum() // 300)]
        for v in probe:
            cand = np.array(tree.query_ball_point(cen[v], 0.5 + 1e-4), dtype=np.int64)
            d = (cen[cand].astype(np.float32) - cen[v].astype(np.float32))
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            want = cand[d2 < np.float32(0.25)]
            got = idx[off[v]:off[v + 1]]
            assert set(got.tolist()) == set(want.tolist())
            gd = (cen[got].astype(np.float32) - cen[v].astype(np.float32))
            gd2 = (gd[:, 0] * gd[:, 0] + gd[:, 1] * gd[:, 1]) + gd[:, 2] * gd[:, 2]
            assert (np.diff(gd2) >= 0).all()  # getOneVoxelAdjacency order: ascending distance

