"""The native tiled driver (include/vgs_tiles.h: C++ host code, RCCL collectives) on the GPU box.
  * emulated ranks (threads of examples/vgs_tiles_run meeting in shared memory, one GPU): the labels of every rank equal the
    labels of the Python twin (TiledSegmenter over the in-process FakeDist) element for element, for 2x1 and 2x2 layouts;
  * the RCCL path with the communicator of a one-rank world (what a 1-GPU box can run): ncclCommInitRank, ncclAllGather and
    ncclBroadcast are the calls a multi-GPU run makes, and the labels equal a plain engine's."""
import os
import subprocess

import numpy as np
import pytest

from test_gpu_tiles import _run_tiled

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vgs-svgs-segmentation_amd", "csrc")
EXE = os.path.join(ROOT, "examples", "vgs_tiles_run")


def _native(tmp_path, parts, mode, tiles, pitch, env=None):
    subprocess.check_call(["make", "-C", CSRC, "-s", "example"])
    prefix = str(tmp_path / "t")
    for r, p in enumerate(parts):
        np.ascontiguousarray(p, dtype=np.float32).tofile(f"{prefix}.{r}.f32")
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.check_output([EXE, mode, f"{tiles[0]}x{tiles[1]}", "--pitch", repr(float(pitch)), "--voxel", "0.1", prefix], text=True, env=e)
    world, kept, n0, nrec = (int(x) for x in out.strip().splitlines()[-1].split())   # RCCL may print a banner first
    labels = [np.fromfile(f"{prefix}.{r}.labels.i32", dtype=np.int32) for r in range(len(parts))]
    return world, kept, labels, nrec


@pytest.mark.parametrize("tiles,n_per", [((2, 1), 150_000), ((2, 2), 120_000), ((4, 2), 60_000)], ids=["2x1", "2x2", "4x2"])
def test_native_driver_equals_the_python_twin(gpu, tmp_path, tiles, n_per):
    world = tiles[0] * tiles[1]
    pitch = 50.0 * np.sqrt(n_per / 10_000_000)
    parts = [gpu.scenes.tiled_urban_scene(n_per * world, tiles=tiles, tile_index=r) for r in range(world)]
    w, kept, labels, nrec = _native(tmp_path, parts, "--emulate", tiles, pitch)
    assert w == world and nrec > 0
    out = _run_tiled(gpu, parts, dict(voxel_size=0.1), pitch, world=world, tiles=tiles)
    assert kept == out[0][1]
    for r in range(world):
        np.testing.assert_array_equal(labels[r], out[r][0])
    # a segment that crosses a border carries one label on both sides
    assert set(labels[0][labels[0] >= 0].tolist()) & set(labels[1][labels[1] >= 0].tolist())


def test_native_driver_over_rccl_one_rank(gpu, tmp_path):
    xyz = gpu.scenes.urban_scene(200_000)
    env = {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29611"}
    w, kept, labels, nrec = _native(tmp_path, [xyz], "--rccl", (1, 1), 1000.0, env=env)
    eng = gpu.Engine(gpu.default_params(2, voxel_size=0.1))
    eng.set_points(xyz)
    eng.run()
    assert w == 1 and kept == eng.counts()["kept"]
    np.testing.assert_array_equal(labels[0], eng.point_labels())


# ---- failures must not leave a peer waiting inside a collective: agreed status words --------------------------------
def _two_rank_threads(gpu, parts, pitch, body, timeout=120.0):
    """Two ranks of the native driver as threads of this process (VGS_TILES_COMM_LOCAL); body(rank, driver) -> anything.
    Returns the per-rank results or exceptions; fails if a rank is still waiting after `timeout` seconds."""
    import threading
    from vgs_svgs_segmentation_amd import tiles_native as tn
    grp = tn.LocalGroup(2)
    out = [None, None]

    def rank_main(r):
        try:
            t = tn.NativeTiles(gpu.default_params(2, voxel_size=0.1), tn.COMM_LOCAL, grp.handle, r, 2, (2, 1), pitch)
            try:
                out[r] = body(r, t, parts[r])
            finally:
                t.close()
        except Exception as ex:  # noqa: BLE001
            out[r] = ex
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout)
    hung = [r for r, t in enumerate(th) if t.is_alive()]
    if hung:
        grp.abort()
        pytest.fail(f"rank(s) {hung} still inside the driver after {timeout} s: a peer's failure left them waiting")
    grp.close()
    return out


@pytest.mark.parametrize("phase", ["points", "grid", "stages"])
def test_a_failing_rank_takes_its_peers_out_with_it(gpu, monkeypatch, phase):
    """Rank 1 fails locally (injected: VGS_TILES_FAIL_RANK / VGS_TILES_FAIL_AT) before the collective of the named phase.  It still
    takes part in that collective with its status in the payload; it returns its own error, rank 0 returns VGS_E_PEER naming it,
    and nobody waits."""
    monkeypatch.setenv("VGS_TILES_FAIL_RANK", "1")
    monkeypatch.setenv("VGS_TILES_FAIL_AT", phase)
    n_per = 60_000
    pitch = 50.0 * np.sqrt(n_per / 10_000_000)
    parts = [gpu.scenes.tiled_urban_scene(n_per * 2, tiles=(2, 1), tile_index=r) for r in range(2)]

    def body(r, t, xyz):
        t.set_points(xyz)
        t.run()
        return "finished"
    out = _two_rank_threads(gpu, parts, pitch, body)
    assert isinstance(out[1], gpu.VgsError) and "VGS_E_STATE" in str(out[1]) and "failure requested" in str(out[1]), out[1]
    assert isinstance(out[0], gpu.VgsError) and "VGS_E_PEER" in str(out[0]) and "rank 1" in str(out[0]), out[0]


def test_a_failed_upload_travels_with_the_next_collective(gpu, monkeypatch):
    """Rank 1's upload fails BEHIND the last collective of set_points (injected: VGS_TILES_FAIL_AT=upload).  set_points returns the
    error there; a caller that goes on to run() all the same joins the grid's collective with that status in its word: it gets its
    own error again, rank 0 gets VGS_E_PEER, nobody waits."""
    monkeypatch.setenv("VGS_TILES_FAIL_RANK", "1")
    monkeypatch.setenv("VGS_TILES_FAIL_AT", "upload")
    n_per = 60_000
    pitch = 50.0 * np.sqrt(n_per / 10_000_000)
    parts = [gpu.scenes.tiled_urban_scene(n_per * 2, tiles=(2, 1), tile_index=r) for r in range(2)]
    seen = [None, None]

    def body(r, t, xyz):
        try:
            t.set_points(xyz)
        except gpu.VgsError as ex:
            seen[r] = str(ex)
        t.run()
        return "finished"
    out = _two_rank_threads(gpu, parts, pitch, body)
    assert seen[0] is None and seen[1] is not None and "upload" in seen[1], seen
    assert isinstance(out[1], gpu.VgsError) and "VGS_E_STATE" in str(out[1]), out[1]
    assert isinstance(out[0], gpu.VgsError) and "VGS_E_PEER" in str(out[0]) and "rank 1" in str(out[0]), out[0]


def test_strict_region_refuses_points_outside_the_rank(gpu):
    """VGS_TILES_OPT_STRICT_REGION: a rank that holds points beyond its region makes every rank refuse the cloud (by default it is
    a warning on stderr and a count in vgs_tiles_get_info)."""
    from vgs_svgs_segmentation_amd import tiles_native as tn
    n_per = 60_000
    pitch = 50.0 * np.sqrt(n_per / 10_000_000)
    parts = [gpu.scenes.tiled_urban_scene(n_per * 2, tiles=(2, 1), tile_index=r) for r in range(2)]
    parts = [parts[0][parts[0][:, 0] < 0.0], parts[1][parts[1][:, 0] >= 0.0]]   # loaded by region (2 x 1 tiles about x = 0) ...
    parts[1] = np.concatenate([parts[1], parts[0][:100]])        # ... except that rank 1 also holds a few of rank 0's points

    def strict(r, t, xyz):
        t.set_option(tn.OPT_STRICT_REGION, 1)
        t.set_points(xyz)
        return "accepted"
    out = _two_rank_threads(gpu, parts, pitch, strict)
    assert isinstance(out[1], gpu.VgsError) and "VGS_E_ARG" in str(out[1]) and "outside" in str(out[1]), out[1]
    assert isinstance(out[0], gpu.VgsError) and "VGS_E_PEER" in str(out[0]), out[0]

    def lenient(r, t, xyz):
        t.set_points(xyz)
        t.run()
        return t.info()["n_outside"], t.times()
    out = _two_rank_threads(gpu, parts, pitch, lenient)
    assert out[1][0] == 100 and out[0][0] == 0 and out[0][1]["total"] > 0 and out[0][1]["exchange"] >= 0, out
