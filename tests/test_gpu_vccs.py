"""svgs_supervoxels (csrc/vccs.hip): the VCCS-style supervoxel stage that stands in for pcl::SupervoxelClustering
(supervoxel_segmentation.h:265-284; PCL is unavailable -> parity with PCL is unpinned).  Checked (a) label for label
against the oracle's CPU restatement of the same algorithm, (b) through invariants of the published algorithm
(SURVEY.md B.4), (c) end to end: vgs_run on an SVGS context equals the oracle pipeline fed with those labels."""
import numpy as np
import pytest
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

from helpers import oracle_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def run(gpu, oracle):
    xyz = gpu.scenes.urban_scene(250_000)
    p = gpu.default_params(3, vccs_mode=0)    # the synchronous variant (the default is PCL's order since round 6: run_pcl below)
    eng = gpu.Engine(p)
    eng.set_points(xyz)
    eng.run()                      # svgs_supervoxels + svgs_segment (segmentationSVGS, test:138-160)
    labels, max_label = eng.supervoxel_labels()
    return dict(eng=eng, xyz=xyz, labels=labels, max_label=max_label, p=p)


def test_labels_match_oracle_restatement(run, oracle):
    ref_labels, ref_max = oracle.vccs(run["xyz"], oracle_params(oracle, run["p"]))
    assert run["max_label"] == ref_max
    np.testing.assert_array_equal(run["labels"], ref_labels)


def test_supervoxels_vs_the_independent_leg(run, oracle):
    """SURVEY 8 row a12's independent leg: the same published steps in double precision with libm and an eigen-solver of their own
    (oracle/refcpu_vccs_ref.cpp -- no vccs_common.h, no vgs_math.h).  Supervoxel decisions sit on float edges (a voxel equidistant
    from two seeds on a lattice) and one flip moves centroids and re-seeding for 54 rounds, so the bar is a stated tolerance: the
    same number of supervoxels, >= 80 % of the voxels in matching supervoxels, and -- what the path is for -- >= 93 % of the points
    in matching FINAL segments when the oracle's RefMath + faithful SVGS chain runs on the independent labels."""
    from helpers import p2_protocol, partition_agreement
    ref_labels, ref_max = oracle.vccs_refmath(run["xyz"], oracle_params(oracle, run["p"]))
    assert ref_max == run["max_label"]
    pv = oracle.voxelize(run["xyz"], run["p"].voxel_size).voxel_table()["point_voxel"]
    r = p2_protocol(run["labels"].astype(np.int64) - 1, ref_labels.astype(np.int64) - 1, pv, min_voxels=10 ** 9)
    assert r["agreement"] >= 0.80 and r["kept_test"] == r["kept_ref"], r
    ref = oracle.run_svgs_from_labels(run["xyz"], ref_labels, ref_max, oracle_params(oracle, run["p"], math=0, flavour=0))
    agree = partition_agreement(run["eng"].point_labels(), ref.labels()[0])
    assert agree >= 0.93, agree


def test_vccs_invariants(run, oracle):
    xyz, labels, p = run["xyz"], run["labels"], run["p"]
    assert labels.min() >= 0 and labels.max() <= run["max_label"]
    assert (labels == 0).mean() < 0.01                       # nearly every point is reached by some supervoxel
    t = oracle.voxelize(xyz, p.voxel_size).voxel_table()     # the VCCS voxels (same binning as the engine)
    pv = t["point_voxel"]
    V = t["key"].shape[0]
    vlab = np.zeros(V, dtype=np.int64)
    vlab[pv] = labels                                        # all points of a voxel share its label
    assert (vlab[pv] == labels).all()
    # number of supervoxels ~ occupied seed cells (one seed per occupied seed_res cell)
    cells = np.unique(np.floor((xyz - xyz.min(0)) / p.seed_size).astype(np.int64), axis=0).shape[0]
    assert 0.5 * cells <= run["max_label"] <= 1.5 * cells
    # each supervoxel is 26-connected and compact
    key = t["key"].astype(np.int64)
    code = {tuple(k): i for i, k in enumerate(key.tolist())}
    src, dst = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if (dx, dy, dz) <= (0, 0, 0):
                    continue
                nb = key + np.array([dx, dy, dz])
                for i, k in enumerate(map(tuple, nb.tolist())):
                    j = code.get(k)
                    if j is not None and vlab[i] == vlab[j] and vlab[i] > 0:
                        src.append(i); dst.append(j)
    g = coo_matrix((np.ones(len(src), np.int8), (src, dst)), shape=(V, V))
    _, comp = connected_components(g, directed=False)
    lab_v, comp_v = vlab[vlab > 0], comp[vlab > 0]
    pairs, cnt = np.unique(np.stack([lab_v, comp_v], 1), axis=0, return_counts=True)
    tot = np.bincount(lab_v)
    big = np.zeros_like(tot)
    np.maximum.at(big, pairs[:, 0], cnt)
    # stealing (as in PCL's expand) can cut voxels off in sparse clutter, where voxels hold ~1 point and normals are
    # noise; on surfaces supervoxels are connected.  Bars calibrated on this scene: 0.92 overall, 0.996 / 0.96 on the ground
    assert big.sum() / tot.sum() >= 0.85
    ground = np.unique(vlab[(t["center"][:, 2] < 0.06) & (vlab > 0)])
    assert big[ground].sum() / tot[ground].sum() >= 0.98
    assert (big[ground] == tot[ground]).mean() >= 0.90
    lab_ids = np.unique(lab_v)
    ext = np.array([np.ptp(t["center"][vlab == l], axis=0).max() for l in lab_ids[:400]])
    assert np.percentile(ext, 95) <= 2.5 * p.seed_size


def test_end_to_end_matches_oracle_pipeline(run, oracle):
    ref = oracle.run_svgs_from_labels(run["xyz"], run["labels"], run["max_label"], oracle_params(oracle, run["p"]))
    pl, _ = ref.labels()
    np.testing.assert_array_equal(run["eng"].point_labels(), pl)
    c = run["eng"].counts()
    assert c["clusters"] == ref.clusters_num


# ---- vccs_mode 1: pcl::SupervoxelClustering's own order (sequential owners, 2-ring normals, seed rejection; round 5: its own lattice,
# refineNormals, re-seeding by the nearest of ALL voxels) ----------------------------------------------------------------------
@pytest.fixture(scope="module", params=[("urban", 200_000), ("pc", 120_000), ("town", 150_000)], ids=["urban", "pc", "town"])
def run_pcl(request, gpu, oracle):
    name, n = request.param
    xyz = {"urban": gpu.scenes.urban_scene, "pc": gpu.scenes.pc_scene, "town": gpu.scenes.town_scene}[name](n)
    p = gpu.default_params(3)
    assert p.vccs_mode == 1                   # PCL's order is what SVGS means (vgs_params_default_svgs)
    eng = gpu.Engine(p)
    eng.set_points(xyz)
    eng.run()
    labels, max_label = eng.supervoxel_labels()
    return dict(eng=eng, xyz=xyz, labels=labels, max_label=max_label, p=p)


def test_pcl_order_vs_its_independent_leg(run_pcl, oracle):
    """SURVEY 8 row a12 for the DEFAULT mode (round 6): the PCL-order steps once more in double precision with libm, a Jacobi eigen-solver,
    two-pass covariances and plain means (oracle/refcpu_vccs_ref.cpp: vccs_pcl_supervoxels_refmath; no vccs_common.h, no vgs_math.h).  The
    sequential owner order amplifies any flipped decision (tests/test_vccs_sensitivity.py: the two legs differ from each other exactly as much
    as each differs from itself under a micrometre of jitter, 0.73 of the points), so the bar is a stated tolerance: the seed count within
    1 %, >= 65 % of the points in matching supervoxels and >= 85 % of the points in matching FINAL segments when the oracle's RefMath +
    faithful SVGS chain runs on the independent labels."""
    from helpers import partition_agreement
    ref_labels, ref_max = oracle.vccs_pcl_refmath(run_pcl["xyz"], oracle_params(oracle, run_pcl["p"]))
    assert abs(ref_max - run_pcl["max_label"]) <= 0.01 * ref_max, (ref_max, run_pcl["max_label"])
    assert partition_agreement(run_pcl["labels"].astype(np.int64) - 1, ref_labels.astype(np.int64) - 1) >= 0.65
    ref = oracle.run_svgs_from_labels(run_pcl["xyz"], ref_labels, ref_max, oracle_params(oracle, run_pcl["p"], math=0, flavour=0))
    assert partition_agreement(run_pcl["eng"].point_labels(), ref.labels()[0]) >= 0.85


def test_synchronous_variant_is_not_within_p2_of_pcl_order(run_pcl, gpu):
    """how far are the FINAL segments with mode 0's supervoxels from those with mode 1's?  P2 of SURVEY 8c on points
    (>= 99.5 % in matching segments, IoU >= 0.98 for every large segment, kept count within 1 %).  Measured in round 6 (tools/sv_modes_p2.py):
    agreement 0.92 / 0.96 / 0.93, smallest IoU 0.76 / 0.28 / 0.67 on these three scenes -- the synchronous variant is an approximation, not
    a faster route to the same result, which is why PCL's order is the default.  The test pins that finding: should mode 0 ever come within
    P2, the default and config 4's quoted mode are to be reconsidered."""
    from helpers import p2_protocol
    e0 = gpu.Engine(gpu.default_params(3, vccs_mode=0))
    e0.set_points(run_pcl["xyz"])
    e0.run()
    r = p2_protocol(e0.point_labels(), run_pcl["eng"].point_labels(), np.arange(run_pcl["xyz"].shape[0]), min_voxels=2000)
    assert 0.85 <= r["agreement"] < 0.995, r            # the same scene structure, not the same segments
    assert r["min_iou"] < 0.98, r


def test_pcl_order_labels_match_the_sequential_restatement(run_pcl, oracle):
    """The GPU resolves PCL's sequential owner order with a fixed point over 'live' leaves; the oracle simply takes the
    supervoxels' turns one after the other (refcpu_vccs.cpp: vccs_pcl_supervoxels).  Same labels, point for point."""
    ref_labels, ref_max = oracle.vccs_pcl(run_pcl["xyz"], oracle_params(oracle, run_pcl["p"]))
    assert run_pcl["max_label"] == ref_max
    np.testing.assert_array_equal(run_pcl["labels"], ref_labels)
    assert (run_pcl["labels"] == 0).mean() < 0.02
    # the two modes are different algorithms: they must not agree by accident of falling through to the same code
    sync_labels, _ = oracle.vccs(run_pcl["xyz"], oracle_params(oracle, run_pcl["p"]))
    assert (sync_labels != ref_labels).mean() > 0.05


def test_pcl_order_bins_on_the_adjacency_octrees_own_lattice(run_pcl, oracle):
    """vccs_mode 1 (round 5): pcl::SupervoxelClustering's OctreePointCloudAdjacency defines its box from the cloud's bounding box (padded
    symmetrically to 2^depth voxels) before the first point goes in; the class's own octree grows from the first point.  Supervoxels are
    unions of voxels: on the bbox-first lattice (the oracle's restatement of it) every voxel's points carry ONE label, on the grown
    lattice -- the same cells shifted by a fraction of a voxel -- many voxels are split between two supervoxels."""
    xyz, labels, res = run_pcl["xyz"], run_pcl["labels"], run_pcl["p"].voxel_size

    def mixed(pv):
        ok = pv >= 0
        order = np.lexsort((labels[ok], pv[ok]))
        v, l = pv[ok][order], labels[ok][order]
        return int(((v[1:] == v[:-1]) & (l[1:] != l[:-1])).sum())
    own = oracle.voxelize(xyz, res, bbox_first=True).voxel_table()["point_voxel"]
    grown = oracle.voxelize(xyz, res).voxel_table()["point_voxel"]
    assert mixed(own) == 0
    assert mixed(grown) > 100


def test_pcl_order_end_to_end(run_pcl, oracle):
    ref = oracle.run_svgs_from_labels(run_pcl["xyz"], run_pcl["labels"], run_pcl["max_label"], oracle_params(oracle, run_pcl["p"]))
    pl, _ = ref.labels()
    np.testing.assert_array_equal(run_pcl["eng"].point_labels(), pl)


def test_pcl_order_switch_rolls_the_labels_back(gpu):
    """vccs_mode is one of the VCCS parameters: changing it invalidates supervoxel labels made by the other mode."""
    xyz = gpu.scenes.urban_scene(60_000)
    p = gpu.default_params(3, vccs_mode=0)
    eng = gpu.Engine(p)
    eng.set_points(xyz)
    eng.run()
    a, _ = eng.supervoxel_labels()
    p.vccs_mode = 1
    eng.set_params(p)
    eng.run()
    b, _ = eng.supervoxel_labels()
    assert (a != b).any()
    e2 = gpu.Engine(gpu.default_params(3, vccs_mode=1))
    e2.set_points(xyz)
    e2.run()
    np.testing.assert_array_equal(b, e2.supervoxel_labels()[0])


@pytest.mark.parametrize("knob", ["VGS_VCCS_NBR_NORMALS", "VGS_VCCS_PINGPONG", "VGS_NO_VCCS_TILES"])
def test_pcl_order_schedule_knobs_change_nothing(gpu, knob):
    """The two-ring normals over the tiles' LDS arrays (round 6) against the [26][V] neighbour table, the live sweeps in place against two
    flag arrays, tiles against whole-cloud kernels: the same supervoxels, label for label."""
    import os
    xyz = gpu.scenes.town_scene(90_000)
    p = gpu.default_params(3)
    a = gpu.Engine(p); a.set_points(xyz); a.supervoxels()
    la, ma = a.supervoxel_labels()
    old = os.environ.get(knob)
    os.environ[knob] = "1"
    try:
        b = gpu.Engine(p); b.set_points(xyz); b.supervoxels()
        lb, mb = b.supervoxel_labels()
    finally:
        if old is None:
            del os.environ[knob]
        else:
            os.environ[knob] = old
    assert ma == mb
    np.testing.assert_array_equal(la, lb)
