"""SVGS on the GPU against the oracle (supervoxel_segmentation.h:279-421 from a given supervoxel labelling).
The labelling here is a plain seed-size grid (what matters for these rows is everything downstream of
pcl::SupervoxelClustering: SURVEY.md 8 rows a13-a15; the VCCS-style clustering itself is row a12)."""
import numpy as np
import pytest

from helpers import assert_p2, canonical_labels, oracle_params, ragged_lists, ragged_sets

pytestmark = pytest.mark.gpu


def grid_supervoxels(xyz, seed):
    cell = np.floor(xyz.astype(np.float64) / seed).astype(np.int64)
    cell -= cell.min(0)
    code = (cell[:, 0] * 4096 + cell[:, 1]) * 4096 + cell[:, 2]
    _, inv = np.unique(code, return_inverse=True)
    labels = (inv + 1).astype(np.int32)
    rng = np.random.default_rng(5)
    labels[rng.random(labels.size) < 0.01] = 0          # some unassigned points (label 0, SS:303)
    return labels, int(labels.max())                     # getMaxLabel(): the supervoxel with this label is dropped (SS:313)


@pytest.fixture(scope="module", params=[("urban", 150_000), ("pc", 60_000)], ids=["urban", "pc"])
def run(request, gpu, oracle):
    name, n = request.param
    xyz = {"urban": gpu.scenes.urban_scene, "pc": gpu.scenes.pc_scene}[name](n)
    labels, max_label = grid_supervoxels(xyz, 0.25)
    p = gpu.default_params(3)
    eng = gpu.Engine(p)
    eng.set_points(xyz)
    eng.set_supervoxel_labels(labels, max_label)
    eng.svgs_segment()
    ref = oracle.run_svgs_from_labels(xyz, labels, max_label, oracle_params(oracle, p))
    return dict(eng=eng, ref=ref, xyz=xyz, labels=labels, max_label=max_label, p=p)


def test_supervoxel_table(run):
    eng, ref = run["eng"], run["ref"]
    c = eng.counts()
    assert c["voxels"] == ref.V == c["supervoxels"]
    assert ref.V == len(np.unique(run["labels"][(run["labels"] > 0) & (run["labels"] < run["max_label"])]))
    off, idx = ref.lists("sv_points")
    t = eng.voxel_table()
    np.testing.assert_array_equal(t["start"], off.astype(np.int32))
    np.testing.assert_array_equal(t["point_idx"], idx)


def test_attributes_bit_exact(run):
    g, r = run["eng"].attributes(), run["ref"].nodes()
    assert g["used"].all() and r["used"].all()
    for k in ("centroid", "normal", "eigen"):
        np.testing.assert_array_equal(g[k].view(np.uint32), r[k].view(np.uint32))


def test_neighbours_exact_order(run):
    go, gi = run["eng"].lists("adjacency")
    ro, ri = run["ref"].lists("adjacency")
    assert ragged_lists(go, gi) == ragged_lists(ro, ri)


@pytest.mark.parametrize("which", ["connect_cut", "connect_cross", "connect_final"])
def test_connect_lists_exact(run, which):
    go, gi = run["eng"].lists(which)
    ro, ri = run["ref"].lists(which)
    gs, rs = ragged_sets(go, gi), ragged_sets(ro, ri)
    bad = [v for v in range(len(rs)) if gs[v] != rs[v]]
    assert not bad, f"{which}: {len(bad)} of {len(rs)} supervoxels differ, first {bad[:5]}"


def test_partition_vs_refmath_faithful(run, oracle):
    """P2 (all three clauses) against the oracle in the reference's own arithmetic and data flow (SS:1565-2305)."""
    ref = oracle.run_svgs_from_labels(run["xyz"], run["labels"], run["max_label"], oracle_params(oracle, run["p"], math=0, flavour=0))
    lab, mx = run["labels"].astype(np.int64), run["max_label"]
    # supervoxel (node) of a point: rank of its label among the kept labels 1 .. max_label-1 (SS:303-313)
    keep = (lab >= 1) & (lab < mx)
    node = np.full(lab.size, -1, np.int64)
    node[keep] = np.unique(lab[keep], return_inverse=True)[1]
    assert_p2(run["eng"].point_labels(), ref.labels()[0], node)


def test_labels_identical(run):
    eng, ref = run["eng"], run["ref"]
    pl, nc = ref.labels()
    root, _ = eng.node_labels()
    np.testing.assert_array_equal(canonical_labels(root), canonical_labels(nc))
    np.testing.assert_array_equal(eng.point_labels(), pl)
    c = eng.counts()
    assert c["clusters"] == ref.clusters_num and c["kept"] == ref.kept_clusters   # SVGS keeps every cluster (SS:2113)


def test_cluster_index_lists_in_reference_order(run):
    """getClusterIdx element for element (supervoxel_segmentation.h:2079-2126): DFS order of the supervoxels with the seed
    last, each supervoxel's points in ascending index; no sorting on either side."""
    eng, ref = run["eng"], run["ref"]
    for which in ("connect_cut", "connect_cross", "connect_final"):   # merge-history order of the local cut, kept by the later steps
        go, gi = eng.lists(which, "reference")
        ro, ri = ref.lists(which)
        np.testing.assert_array_equal(go, ro)
        np.testing.assert_array_equal(gi, ri)
    co, ci = eng.clusters("reference")
    rco, rci = ref.lists("clusters_points")
    np.testing.assert_array_equal(co, rco)
    np.testing.assert_array_equal(ci, rci)


def test_class_mirror(run, gpu):
    """segmentationSVGS (reference test:138-160) through the class mirror."""
    sv = gpu.SuperVoxelBasedSegmentation(0.05)
    sv.setInputCloud(run["xyz"])
    sv.getCloudPointNum(run["xyz"])
    sv.addPointsFromInputCloud()
    sv.setVoxelSize(0.05, 10)
    sv.setSupervoxelSize(0.25, 3, 10, 3)
    sv.setGraphSize(0.5, 0.5)
    sv.setSupervoxelLabels(run["labels"], run["max_label"])
    sv.segmentSupervoxelCloudWithGraphModel(0.0, 0.25, 0.75, 0.5, 0.2, 0.2, 0.2, 0.2, 0.2, 1.0)
    np.testing.assert_array_equal(sv.drawColorMapofPointsinClusters(), run["eng"].point_labels())
    assert sv.getClusterNum() == run["eng"].counts()["clusters"]
    assert len(sv.getClusterIdx()) == run["eng"].counts()["kept"]


def test_labels_do_not_outlive_their_cloud(gpu):
    """a supervoxel labelling belongs to the cloud (and, when it came from svgs_supervoxels, to the VCCS
    parameters) it was made for.  A new cloud or new voxel/seed sizes must trigger a new clustering, never a run over
    stale labels (with a larger cloud that was an out-of-bounds read)."""
    from vgs_svgs_segmentation_amd._lib import VgsError
    small = gpu.scenes.urban_scene(40_000)
    big = gpu.scenes.urban_scene(90_000, seed=77)
    p = gpu.default_params(3)

    def fresh(xyz, params):
        e = gpu.Engine(params)
        e.set_points(xyz)
        e.run()
        return e.point_labels(), e.supervoxel_labels()

    eng = gpu.Engine(p)
    eng.set_points(small)
    eng.run()
    eng.set_points(big)                       # larger cloud: the old labels must be gone
    with pytest.raises(VgsError):
        eng.svgs_segment()
    with pytest.raises(VgsError):
        eng.supervoxel_labels()
    eng.run()
    lab, (sv, mx) = fresh(big, p)
    np.testing.assert_array_equal(eng.point_labels(), lab)
    np.testing.assert_array_equal(eng.supervoxel_labels()[0], sv)
    # new seed size: labels from svgs_supervoxels are recomputed ...
    q = gpu.default_params(3, seed_size=0.4)
    eng.set_params(q)
    eng.run()
    lab2, (sv2, mx2) = fresh(big, q)
    assert mx2 != mx
    np.testing.assert_array_equal(eng.supervoxel_labels()[0], sv2)
    np.testing.assert_array_equal(eng.point_labels(), lab2)
    # ... while a caller's own labelling survives parameter changes but not a new cloud
    labels, max_label = grid_supervoxels(big, 0.25)
    eng.set_supervoxel_labels(labels, max_label)
    eng.set_params(p)
    eng.run()
    np.testing.assert_array_equal(eng.supervoxel_labels()[0], labels)
    eng.set_points(small)
    with pytest.raises(VgsError):
        eng.svgs_segment()

    # the class mirror asks the context, it keeps no flag of its own
    sv_cls = gpu.SuperVoxelBasedSegmentation(0.05)
    for cloud in (small, big):
        sv_cls.setInputCloud(cloud)
        sv_cls.addPointsFromInputCloud()
        sv_cls.segmentSupervoxelCloudWithGraphModel(0.0, 0.25, 0.75, 0.5, 0.2, 0.2, 0.2, 0.2, 0.2, 1.0)
        np.testing.assert_array_equal(sv_cls.drawColorMapofPointsinClusters(), fresh(cloud, p)[0])
