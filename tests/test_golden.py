"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py with the CPU oracle).
CPU: the oracle still reproduces them (pins the oracle).  GPU: the HIP path reproduces the integer tables
exactly, the RefMath attributes within the P1 tolerances and the DevMath labels bit for bit."""
import glob
import os

import numpy as np
import pytest

from helpers import assert_p2, canonical_labels, partition_agreement

_ALL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
GOLD = [p for p in _ALL if not os.path.basename(p).startswith("svgs_")]       # method 2 (VGS)
GOLD_SVGS = [p for p in _ALL if os.path.basename(p).startswith("svgs_")]      # method 3 (SVGS from a supervoxel labelling)


def _params(g):
    return {str(k): float(v) for k, v in zip(g["params_keys"], g["params_vals"])}


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    kw = _params(g)
    for math, tag in ((0, "ref"), (1, "dev")):
        r = oracle.run_vgs(g["xyz"], oracle.vgs_params(math=math, flavour=1, **kw))
        t = r.voxel_table()
        np.testing.assert_array_equal(t["key"], g["key"])
        np.testing.assert_array_equal(t["start"], g["start"])
        np.testing.assert_array_equal(t["point_voxel"], g["point_voxel"])
        np.testing.assert_array_equal(r.bbox(), g["bbox"])
        nd = r.nodes()
        np.testing.assert_array_equal(nd["used"], g["used"])
        np.testing.assert_array_equal(nd["centroid"].view(np.uint32), g[f"centroid_{tag}"].view(np.uint32))
        if math == 1:  # DevMath is libm-free: bit-reproducible everywhere
            np.testing.assert_array_equal(nd["normal"].view(np.uint32), g["normal_dev"].view(np.uint32))
            np.testing.assert_array_equal(nd["eigen"].view(np.uint32), g["eigen_dev"].view(np.uint32))
            np.testing.assert_array_equal(r.labels()[0], g["point_label_dev"])
        else:          # RefMath goes through libm: allow ulp-level drift between libm builds
            np.testing.assert_allclose(nd["eigen"], g["eigen_ref"], atol=1e-5)
            assert partition_agreement(r.labels()[0], g["point_label_ref"]) > 0.999
            assert_p2(g["point_label_dev"], r.labels()[0], g["point_voxel"], g["used"])   # DevMath vectors vs the reference's arithmetic
        off, _ = r.lists("adjacency")
        np.testing.assert_array_equal(np.diff(off).astype(np.int32), g["adj_len"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_gpu_reproduces_golden(gpu, path):
    g = np.load(path)
    kw = _params(g)
    p = gpu.default_params(2, **kw)
    eng = gpu.Engine(p)
    eng.set_points(g["xyz"])
    eng.run()
    t = eng.voxel_table()
    np.testing.assert_array_equal(t["key"], g["key"])
    np.testing.assert_array_equal(t["start"], g["start"])
    np.testing.assert_array_equal(eng.point_voxel(), g["point_voxel"])
    np.testing.assert_array_equal(eng.bbox(), g["bbox"])
    a = eng.attributes()
    np.testing.assert_array_equal(a["used"], g["used"])
    # bit-exact against the DevMath vectors
    for k in ("centroid", "normal", "eigen"):
        np.testing.assert_array_equal(a[k].view(np.uint32), g[f"{k}_dev"].view(np.uint32))
    np.testing.assert_array_equal(eng.point_labels(), g["point_label_dev"])
    root, _ = eng.node_labels()
    np.testing.assert_array_equal(canonical_labels(root), canonical_labels(g["node_cluster_dev"]))
    c = eng.counts()
    assert [c["clusters"], c["kept"]] == g["clusters_dev"].tolist()
    # P1 against the RefMath vectors: centroid 1e-4 m, normal angle 1e-3 rad, eigen features 1e-3
    used = g["used"].astype(bool)
    np.testing.assert_allclose(a["centroid"], g["centroid_ref"], atol=1e-4)
    cosang = (a["normal"][used] * g["normal_ref"][used]).sum(1)
    assert (np.arccos(np.clip(cosang, -1, 1)) < 1e-3).mean() > 0.999
    assert np.nanmax(np.abs(a["eigen"] - g["eigen_ref"])) < 2e-3
    # P2 against the RefMath labels
    assert_p2(eng.point_labels(), g["point_label_ref"], g["point_voxel"], g["used"])
    off, _ = eng.lists("adjacency")
    np.testing.assert_array_equal(np.diff(off).astype(np.int32), g["adj_len"])   # every voxel, used or not (VS:236-263)


def _sv_node(g):
    """Supervoxel (node) of every point: labels 1 .. max_label-1 are kept (SS:313), everything else belongs to no node."""
    sv, mx = g["sv_label"].astype(np.int64), int(g["max_label"])
    return np.where((sv >= 1) & (sv < mx), sv - 1, -1)


# ---- SVGS (method 3): supervoxel labelling -> attributes -> neighbours -> local cuts -> merge (SS:279-421) ----------------
@pytest.mark.parametrize("path", GOLD_SVGS, ids=[os.path.basename(p)[:-4] for p in GOLD_SVGS])
def test_oracle_reproduces_svgs_golden(oracle, path):
    g = np.load(path)
    for math, tag in ((0, "ref"), (1, "dev")):
        r = oracle.run_svgs_from_labels(g["xyz"], g["sv_label"], int(g["max_label"]), oracle.svgs_params(math=math, flavour=1, **_params(g)))
        off, idx = r.lists("sv_points")
        np.testing.assert_array_equal(off, g["sv_start"])
        np.testing.assert_array_equal(idx, g["sv_point_idx"])
        nd = r.nodes()
        np.testing.assert_array_equal(nd["centroid"].view(np.uint32), g[f"centroid_{tag}"].view(np.uint32))
        aoff, _ = r.lists("adjacency")
        np.testing.assert_array_equal(np.diff(aoff).astype(np.int32), g["adj_len"])
        if math == 1:
            np.testing.assert_array_equal(nd["normal"].view(np.uint32), g["normal_dev"].view(np.uint32))
            np.testing.assert_array_equal(nd["eigen"].view(np.uint32), g["eigen_dev"].view(np.uint32))
            np.testing.assert_array_equal(r.labels()[0], g["point_label_dev"])
            assert [r.clusters_num, r.kept_clusters] == g["clusters_dev"].tolist()
        else:
            np.testing.assert_allclose(nd["eigen"], g["eigen_ref"], atol=1e-5)
            assert partition_agreement(r.labels()[0], g["point_label_ref"]) > 0.999
    # the two data flows of the oracle agree on the partition in the reference's arithmetic
    assert partition_agreement(g["point_label_ref"], g["point_label_ref_faithful"]) > 0.995
    for leg in ("point_label_ref", "point_label_ref_faithful"):     # full P2 of the DevMath vectors against both RefMath data flows
        assert_p2(g["point_label_dev"], g[leg], _sv_node(g))
    if "vccs" in os.path.basename(path):   # the labelling itself is the oracle's restatement of this repo's supervoxel stage (vccs0: the synchronous variant)
        lab, mx = (oracle.vccs if "vccs0" in os.path.basename(path) else oracle.vccs_pcl)(g["xyz"], oracle.svgs_params(**_params(g)))
        np.testing.assert_array_equal(lab, g["sv_label"])
        assert mx == int(g["max_label"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD_SVGS, ids=[os.path.basename(p)[:-4] for p in GOLD_SVGS])
def test_gpu_reproduces_svgs_golden(gpu, path):
    g = np.load(path)
    p = gpu.default_params(3, **_params(g))
    eng = gpu.Engine(p)
    eng.set_points(g["xyz"])
    eng.set_supervoxel_labels(g["sv_label"], int(g["max_label"]))
    eng.svgs_segment()
    t = eng.voxel_table()
    np.testing.assert_array_equal(t["start"], g["sv_start"].astype(np.int32))
    np.testing.assert_array_equal(t["point_idx"], g["sv_point_idx"])
    a = eng.attributes()
    for k in ("centroid", "normal", "eigen"):       # bit-exact against the DevMath vectors
        np.testing.assert_array_equal(a[k].view(np.uint32), g[f"{k}_dev"].view(np.uint32))
    np.testing.assert_array_equal(eng.point_labels(), g["point_label_dev"])
    root, _ = eng.node_labels()
    np.testing.assert_array_equal(canonical_labels(root), canonical_labels(g["node_cluster_dev"]))
    c = eng.counts()
    assert [c["clusters"], c["kept"]] == g["clusters_dev"].tolist()
    off, _ = eng.lists("adjacency")
    np.testing.assert_array_equal(np.diff(off).astype(np.int32), g["adj_len"])
    # P1 against the RefMath vectors, P2 against the RefMath labels of both data flows (lean, faithful)
    np.testing.assert_allclose(a["centroid"], g["centroid_ref"], atol=1e-4)
    cosang = (a["normal"] * g["normal_ref"]).sum(1)
    assert (np.arccos(np.clip(cosang, -1, 1)) < 1e-3).mean() > 0.999
    assert np.nanmax(np.abs(a["eigen"] - g["eigen_ref"])) < 2e-3
    for leg in ("point_label_ref", "point_label_ref_faithful"):
        assert_p2(eng.point_labels(), g[leg], _sv_node(g))
    if "vccs" in os.path.basename(path):             # the engine's own supervoxel stage gives this labelling
        e2 = gpu.Engine(gpu.default_params(3, vccs_mode=0, **_params(g)) if "vccs0" in os.path.basename(path) else p)
        e2.set_points(g["xyz"])
        e2.supervoxels()
        lab, mx = e2.supervoxel_labels()
        np.testing.assert_array_equal(lab, g["sv_label"])
        assert mx == int(g["max_label"])
