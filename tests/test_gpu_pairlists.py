"""The pair lists (csrc/pairlist.hip) and the local-cut class that reads them (csrc/localcut_pg.hpp), round 5: every way a neighbourhood
can reach them, each against the oracle (DevMath + lean: connect lists after the cut / crossValidation / closestCheck and the point
labels IDENTICAL), and against the kernels they replaced (same engine, pair lists switched off).

* many hand-overs: a surface under centimetres of range noise -- the samples of the one-wavefront classes vote the lazy schedule
  off, every row is built, k_localcut_pg<128> cuts the hand-overs, crossValidation waits for them (LcGate: LC_MANY);
* few hand-overs forced through the lists (VGS_PG_MINFRAC): clutter of an urban scene, the rims of a planar scene;
* wide neighbourhoods as classes of their own (129-320, -512, -1024 voxels and whole balls): BASELINE config 2's parameters and a
  slab seen through balls of ten and six voxels, with the shell classes of round 4 (VGS_PG_WIDE=0) beside them."""
import numpy as np
import pytest

from helpers import oracle_params, ragged_sets

pytestmark = pytest.mark.gpu


def _noisy(n, seed, sigma):
    rng = np.random.default_rng(seed)
    side = np.sqrt(n / 6000.0)
    x, y = rng.random(n) * side, rng.random(n) * side
    z = 0.3 * np.sin(2.0 * x) * np.cos(1.5 * y) + rng.normal(0, sigma, n) + 2.0
    return np.stack([x + 0.011, y + 0.017, z], axis=1).astype(np.float32)


def _slab(n, size, thick, seed):
    rng = np.random.default_rng(seed)
    xyz = np.empty((n, 3), dtype=np.float64)
    xyz[:, 0] = rng.uniform(-size / 2, size / 2, n)
    xyz[:, 1] = rng.uniform(-size / 2, size / 2, n)
    xyz[:, 2] = 1.0 + rng.uniform(0, thick, n) + 0.01 * np.sin(7.0 * xyz[:, 0])
    return xyz.astype(np.float32)


def _compare(eng, ref):
    for which in ("connect_cut", "connect_cross", "connect_final"):
        off, idx = eng.lists(which)
        roff, ridx = ref.lists(which)
        assert np.array_equal(off, roff), which
        gs, rs = ragged_sets(off, idx), ragged_sets(roff, ridx)
        bad = [v for v in range(len(rs)) if gs[v] != rs[v]]
        assert not bad, f"{which}: {len(bad)} of {len(rs)} voxels differ, first {bad[:5]}"
    np.testing.assert_array_equal(eng.point_labels(), ref.labels()[0])
    assert eng.counts()["kept"] == ref.kept_clusters


# name, cloud, parameters, environment, what the schedule counters must show
CASES = [
    ("noisy_many", lambda v: _noisy(100_000, 1, 0.03), dict(voxel_size=0.1), {},
     lambda sc, c: sc["pair_list_cut"] > 0.8 * c["used"] and sc["cross_put_off"] == 0),
    ("noisy_sigma_1cm", lambda v: _noisy(100_000, 2, 0.01), dict(voxel_size=0.1), {}, lambda sc, c: sc["outside_limits"] == 0),
    ("noisy_lists_off", lambda v: _noisy(100_000, 1, 0.03), dict(voxel_size=0.1), {"VGS_NO_PAIRLISTS": "1"},
     lambda sc, c: sc["pair_list_cut"] == 0 and sc["handed_over"] > 0.8 * c["used"]),
    ("noisy_no_vote", lambda v: _noisy(100_000, 1, 0.03), dict(voxel_size=0.1), {"VGS_NO_VOTE": "1"},
     lambda sc, c: sc["voted_over"] == 0 and sc["pair_list_cut"] > 0.8 * c["used"]),
    ("urban_few_forced", lambda v: v.scenes.urban_scene(120_000), dict(voxel_size=0.1), {"VGS_PG_MINFRAC": "1000000000"},
     lambda sc, c: sc["pair_list_cut"] == sc["handed_over"] > 0),
    ("pc_few_forced", lambda v: v.scenes.pc_scene(60_000), dict(voxel_size=0.05, graph_size=0.25), {"VGS_PG_MINFRAC": "1000000000"},
     lambda sc, c: sc["outside_limits"] == 0),
    ("wide_c2", lambda v: v.scenes.pc_scene(110_000), dict(voxel_size=0.05, graph_size=0.5), {},
     lambda sc, c: sc["pair_list_cut"] >= c["class_bc"] - 200 > 3000),
    ("wide_c2_shell_classes", lambda v: v.scenes.pc_scene(110_000), dict(voxel_size=0.05, graph_size=0.5), {"VGS_PG_WIDE": "0"},
     lambda sc, c: sc["pair_list_cut"] < 200 and c["class_bc"] > 3000),   # (the few neighbourhoods the dense kernel passes on still read lists)
    ("wide_slab_r10", lambda v: _slab(60_000, 0.85, 0.11, 7), dict(voxel_size=0.05, graph_size=0.5), {},
     lambda sc, c: sc["pair_list_cut"] > 100 and c["class_d"] > 100),
    ("wide_slab_r6", lambda v: _slab(120_000, 2.0, 0.25, 8), dict(voxel_size=0.08, graph_size=0.5), {}, lambda sc, c: sc["pair_list_cut"] > 100),
    # a tight cut (thr0 = 0.9: short lists, the ring stage early) and a loose one (thr0 = 0.4: long lists, phase B)
    ("noisy_tight_cut", lambda v: _noisy(80_000, 3, 0.03), dict(voxel_size=0.1, cut_thred=0.1), {}, lambda sc, c: sc["outside_limits"] == 0),
    ("noisy_loose_cut", lambda v: _noisy(80_000, 4, 0.03), dict(voxel_size=0.1, cut_thred=0.6), {}, lambda sc, c: sc["outside_limits"] == 0),
    # a ball of 3.3 voxels (the reference's own defaults): the ring of pairs outside each other's ball is wide, the floor w_ring high
    ("noisy_town_defaults", lambda v: _noisy(120_000, 5, 0.04), dict(voxel_size=0.15), {}, lambda sc, c: sc["outside_limits"] == 0),
]


@pytest.fixture(scope="module", params=CASES, ids=[c[0] for c in CASES])
def run(request, gpu, oracle):
    import os
    name, make, kw, env, pred = request.param
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        xyz = make(gpu)
        p = gpu.default_params(2, **kw)
        eng = gpu.Engine(p)       # (the knobs are read when the context is created)
        eng.set_points(xyz)
        eng.run()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    ref = oracle.run_vgs(xyz, oracle_params(oracle, p, threads=os.cpu_count() or 1))
    return dict(name=name, eng=eng, ref=ref, pred=pred, xyz=xyz, p=p)


def test_reaches_its_path(run):
    sc, c = run["eng"].schedule_counters(), run["eng"].counts()
    assert sc["outside_limits"] == 0 and sc["pair_list_pool_full"] == 0, sc
    assert run["pred"](sc, c), (run["name"], sc, c)


def test_identical_to_the_oracle(run):
    _compare(run["eng"], run["ref"])


def test_second_run_is_the_first(run):
    """Idempotence: the pool, the marks and the gate word are reset per run; a second run on the same context gives the same lists."""
    eng = run["eng"]
    a = [eng.lists(w) for w in ("connect_cut", "connect_final")]
    lab = eng.point_labels().copy()
    eng.run()
    b = [eng.lists(w) for w in ("connect_cut", "connect_final")]
    for (ao, ai), (bo, bi) in zip(a, b):
        assert np.array_equal(ao, bo) and np.array_equal(ai, bi)
    np.testing.assert_array_equal(eng.point_labels(), lab)
