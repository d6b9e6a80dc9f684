"""The round-2 schedules of the local cut and the adjacency search must not show in any result: near-pair lists
(nearlist.hip: shells inside two lattice steps read stored weights), the split of the bulk class, and the brick-mask
candidate search (k_adjacency_masks) are compared with the oracle (DevMath, lean: bit-exact bar) on inputs chosen to reach
their corners -- lists that overflow (NL_NONE), balls wider than the lists' offset map, list ends behind entry 16,
rows handed to the general adjacency kernel -- and against the same engine with each schedule switched off.  The dense
hand-over kernel (localcut_dense.hpp) gets the scenes it exists for: fuzzy surfaces where the lazy schedule gives up on nine
voxels out of ten, neighbourhoods whose heavy edges overflow its list (taken in bands of descending weight), and a million-point
urban scene that holds the few dozen voxels whose phase B runs in bands."""
import os

import numpy as np
import pytest

from helpers import canonical_labels, oracle_params, ragged_lists, ragged_sets

pytestmark = pytest.mark.gpu


def _slab_scene(n, seed, thickness=0.35):
    """A thick slab with a gentle ripple: voxels filled in three dimensions with consistent normals -- every voxel has far more
    than 32 neighbours within two lattice steps (near-pair lists overflow when the cut keeps them all) and adjacency rows
    longer than the mask kernel's list."""
    rng = np.random.default_rng(seed)
    side = np.sqrt(n / (2500.0 * (thickness / 0.1)))
    x = rng.random(n) * side - side / 2 + 0.013
    y = rng.random(n) * side - side / 2 + 0.027
    z = rng.random(n) * thickness + 0.02 * np.sin(3.0 * x) + 1.0
    return np.stack([x, y, z], axis=1).astype(np.float32)


def _fuzzy_scene(n, seed, sigma):
    """An undulating surface under range noise as thick as the voxels: normals scatter, no segment freezes early."""
    rng = np.random.default_rng(seed)
    side = np.sqrt(n / 6000.0)
    x, y = rng.random(n) * side, rng.random(n) * side
    z = 0.3 * np.sin(2.0 * x) * np.cos(1.5 * y) + rng.normal(0, sigma, n) + 2.0
    return np.stack([x + 0.011, y + 0.017, z], axis=1).astype(np.float32)


CASES = [
    # name, scene, n, params
    ("urban_loose_cut", "urban", 90_000, dict(voxel_size=0.1, cut_thred=0.6)),          # thr0 = 0.4: long lists, entries behind 16
    ("urban_tight_cut", "urban", 90_000, dict(voxel_size=0.1, cut_thred=0.1)),          # thr0 = 0.9: short lists, many rounds
    ("urban_sig_w1", "urban", 90_000, dict(voxel_size=0.1, sig_w=1.0)),
    ("town_r3", "town", 70_000, dict()),                                                # voxel 0.15: ball of 3 voxels, 3x3x3 bricks
    ("urban_r6", "urban", 90_000, dict(voxel_size=0.08)),                               # ball of 6.25 voxels: beyond the offset map, lists off
    ("slab_overflow", "slab", 70_000, dict(voxel_size=0.1, cut_thred=0.9)),             # thr0 = 0.1: > 32 heavy near pairs -> no list
    ("slab_default", "slab", 70_000, dict(voxel_size=0.1)),
    # coordinates whose float spacing (1.2e-4 m, 2e-3 m) puts centroids visibly outside their voxel's cube: the lists' reach
    # argument needs every listed voxel's centroid inside its cube, so such voxels must end up without a list
    ("urban_2km_away", "urban", 90_000, dict(voxel_size=0.1, shift=(2000.0, -1500.0, 30.0))),
    ("urban_20km_away", "urban", 90_000, dict(voxel_size=0.1, shift=(20000.0, 100.0, 5.0))),
    # the dense hand-over kernel's workloads (schedule_counters says which path a case must reach, REACHES below)
    ("fuzzy_dense", "fuzzy", 120_000, dict(voxel_size=0.1, graph_size=0.4)),
    ("fuzzy_banded", "fuzzy", 120_000, dict(voxel_size=0.1, graph_size=0.45, cut_thred=0.4)),
    ("urban_1M_bands", "urban", 1_000_000, dict(voxel_size=0.1)),
]
REACHES = {"fuzzy_dense": ("handed_over",), "fuzzy_banded": ("handed_over", "banded"), "slab_overflow": ("banded", "handed_over_large"),
           "urban_1M_bands": ("handed_over", "banded", "handed_over_large", "cross_put_off"), "urban_loose_cut": ("handed_over",)}


def _scene(gpu, kind, n):
    if kind == "slab":
        return _slab_scene(n, 5)
    if kind == "fuzzy":
        return _fuzzy_scene(n, 1, 0.03)
    return {"urban": gpu.scenes.urban_scene, "town": gpu.scenes.town_scene}[kind](n)


@pytest.fixture(scope="module", params=CASES, ids=[c[0] for c in CASES])
def case(request, gpu, oracle):
    name, kind, n, kw = request.param
    kw = dict(kw)
    shift = kw.pop("shift", None)
    xyz = _scene(gpu, kind, n)
    if shift is not None:
        xyz = (xyz + np.array(shift, np.float32)).astype(np.float32)
    p = gpu.default_params(2, **kw)
    eng = gpu.Engine(p)
    eng.set_points(xyz)
    eng.run()
    ref = oracle.run_vgs(xyz, oracle_params(oracle, p))
    return dict(name=name, xyz=xyz, p=p, eng=eng, ref=ref)


def test_case_reaches_its_path(case):
    sc = case["eng"].schedule_counters()
    assert sc["outside_limits"] == 0
    for key in REACHES.get(case["name"], ()):
        # (round 5: where hand-overs are many they are cut from the pair lists -- bands of descending weight by construction -- instead
        # of the dense kernel's banded phases)
        got = sc[key] + (sc["pair_list_cut"] if key in ("banded", "cross_put_off") else 0)   # (many hand-overs: crossValidation waits, no row is put off)
        assert got > 0, f"{case['name']} was built to reach {key}: {sc}"


def test_adjacency_exact_order(case):
    eng, ref = case["eng"], case["ref"]
    used = np.nonzero(ref.nodes()["used"])[0]
    gl, rl = ragged_lists(*eng.lists("adjacency")), ragged_lists(*ref.lists("adjacency"))
    bad = [int(v) for v in used if gl[v] != rl[v]]
    assert not bad, f"{len(bad)} adjacency rows differ, first {bad[:5]}"


@pytest.mark.parametrize("which", ["connect_cut", "connect_cross", "connect_final"])
def test_connect_lists_exact(case, which):
    gs, rs = ragged_sets(*case["eng"].lists(which)), ragged_sets(*case["ref"].lists(which))
    bad = [v for v in range(len(rs)) if gs[v] != rs[v]]
    assert not bad, f"{which}: {len(bad)} of {len(rs)} voxels differ, first {bad[:5]}"


def test_labels_identical(case):
    eng, ref = case["eng"], case["ref"]
    pl_ref, nc_ref = ref.labels()
    root, _ = eng.node_labels()
    np.testing.assert_array_equal(canonical_labels(root), canonical_labels(nc_ref))
    np.testing.assert_array_equal(eng.point_labels(), pl_ref)


@pytest.mark.parametrize("knob", ["VGS_NO_NEAR", "VGS_NO_ADJMASKS", "VGS_A1MAX", "VGS_NO_DENSE", "VGS_HO_GRID=0", "VGS_HO_GRID=3"])
def test_same_result_with_the_schedule_off(case, gpu, knob):
    """The knobs only schedule: rows, connect lists and labels are identical bit for bit (pair-evaluation counts may differ)."""
    knob, _, value = knob.partition("=")   # (VGS_HO_GRID: a workgroup per handed-over row as before round 6 / three workgroups striding over them all)
    old = os.environ.get(knob)
    os.environ[knob] = value or ("-1" if knob == "VGS_A1MAX" else "1")
    try:
        e2 = gpu.Engine(case["p"])
        e2.set_points(case["xyz"])
        e2.run()
        for which in ("adjacency", "connect_cut", "connect_final"):
            a, b = e2.lists(which), case["eng"].lists(which)
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(e2.point_labels(), case["eng"].point_labels())
    finally:
        if old is None:
            del os.environ[knob]
        else:
            os.environ[knob] = old


@pytest.mark.parametrize("kind,n,seed,kw", [
    ("urban", 88_651, 804343104, dict(voxel_size=0.15, graph_size=0.3, cut_thred=0.7, sig_w=1.0, sig_n=0.5, sig_p=0.4)),
    ("town", 68_733, 1010320390, dict(voxel_size=0.15, graph_size=0.3, cut_thred=0.1, sig_w=2.0, sig_n=0.5, sig_p=0.4)),
    ("town", 60_000, 7, dict(voxel_size=0.15, graph_size=0.5, cut_thred=0.5, sig_p=0.4)),       # the reference's own ratio, 3.33 voxels
    ("urban", 60_000, 11, dict(voxel_size=0.1, graph_size=0.18)),                                  # below sqrt(3) voxels: no lists
])
def test_near_lists_are_complete_only_inside_the_search_ball(gpu, oracle, kind, n, seed, kw):
    """Regression (round 2, found by tools/fuzz_parity.py): the near-pair lists are built from the adjacency rows, so they hold
    the partners inside the search ball only.  With graph_size = 2 voxels the lattice offset (2, 1, 0) is outside the ball while
    centroids that far apart on the lattice can be one voxel apart; the cut read its first shells from lists it took for
    complete up to two voxels and missed such pairs.  The reach now follows the ball (nearlist.hip)."""
    xyz = {"urban": gpu.scenes.urban_scene, "town": gpu.scenes.town_scene}[kind](n, seed=seed)
    p = gpu.default_params(2, **kw)
    eng = gpu.Engine(p); eng.set_points(xyz); eng.run()
    ref = oracle.run_vgs(xyz, oracle_params(oracle, p))
    for which in ("connect_cut", "connect_final"):
        gs, rs = ragged_sets(*eng.lists(which)), ragged_sets(*ref.lists(which))
        bad = [v for v in range(len(rs)) if gs[v] != rs[v]]
        assert not bad, f"{which}: {len(bad)} of {len(rs)} voxels differ, first {bad[:5]}"
    np.testing.assert_array_equal(eng.point_labels(), ref.labels()[0])


def test_labels_stable_across_fresh_engines(gpu):
    """Regression (round 2): k_flatten used to walk with path halving; a late halving store of another thread could leave a
    voxel pointing at an ancestor below its root, and the labels read parent[v] as the root -- about one run in twenty
    dropped a voxel from its segment, depending on what the freshly allocated buffers held.  Fresh contexts over scenes of
    different sizes shuffle the allocations; every run must give the labels of the first."""
    scenes = [(gpu.scenes.urban_scene(90_000), dict(voxel_size=0.1, cut_thred=0.1)), (gpu.scenes.town_scene(70_000), dict())]
    want = []
    for it in range(24):
        for k, (xyz, kw) in enumerate(scenes):
            e = gpu.Engine(gpu.default_params(2, **kw))
            e.set_points(xyz)
            e.run()
            lab = e.point_labels()
            root, _ = e.node_labels()
            assert (root[root] == root).all(), "a voxel's parent is not a root"
            if it == 0:
                want.append(lab)
            else:
                np.testing.assert_array_equal(lab, want[k])


def test_one_context_many_parameter_sets(gpu):
    """A context keeps parameter-only tables (lattice offsets, ball masks, length groups) and per-run state across runs:
    walking one context through parameter changes -- graph size (new tables), voxel size (new lattice), cut and sigmas (new
    near-pair lists), back to the start -- must give what a fresh context gives for every set."""
    xyz = gpu.scenes.urban_scene(80_000)
    sets = [dict(voxel_size=0.1), dict(voxel_size=0.1, graph_size=0.35), dict(voxel_size=0.1, graph_size=0.65), dict(voxel_size=0.12),
            dict(voxel_size=0.1, cut_thred=0.5, sig_n=0.3), dict(voxel_size=0.1, points_min=5), dict(voxel_size=0.1)]
    eng = gpu.Engine(gpu.default_params(2, **sets[0]))
    eng.set_points(xyz)
    for kw in sets:
        p = gpu.default_params(2, **kw)
        eng.set_params(p)
        eng.run()
        fresh = gpu.Engine(p)
        fresh.set_points(xyz)
        fresh.run()
        np.testing.assert_array_equal(eng.point_labels(), fresh.point_labels())
        for which in ("adjacency", "connect_final"):
            a, b = eng.lists(which), fresh.lists(which)
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
        assert eng.counts()["kept"] == fresh.counts()["kept"]
