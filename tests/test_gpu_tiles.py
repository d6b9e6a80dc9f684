"""GPU test of the tiled path: two simulated ranks (threads, one Engine each on cuda:0, an in-process stand-in
for torch.distributed) against one Engine over the whole scene.  Same voxel lattice (the chained grid starts
from the same first point), so the two partitions must agree except where closestCheck's scan order matters."""
import threading

import numpy as np
import pytest

from helpers import partition_agreement

pytestmark = pytest.mark.gpu


class FakeDist:
    class ReduceOp:
        MAX = "max"

    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.tls = threading.local()

    def get_backend(self):
        return "gloo"

    def get_world_size(self):
        return self.world

    def all_gather(self, outs, t):
        self.slots[self.tls.rank] = t.clone()
        self.bar.wait()
        for i in range(self.world):
            outs[i].copy_(self.slots[i])
        self.bar.wait()

    def broadcast(self, t, src):
        if self.tls.rank == src:
            self.slots[src] = t.clone()
        self.bar.wait()
        t.copy_(self.slots[src])
        self.bar.wait()

    def all_reduce(self, t, op=None):
        self.slots[self.tls.rank] = t.clone()
        self.bar.wait()
        m = self.slots[0].clone()
        for i in range(1, self.world):
            m = m.maximum(self.slots[i])
        self.bar.wait()
        t.copy_(m)


def test_two_tiles_match_single_engine(gpu):
    import torch
    from vgs_svgs_segmentation_amd.dist import TiledSegmenter
    world, n_per = 2, 150_000
    pitch = 50.0 * np.sqrt(n_per / 10_000_000)
    tiles = [gpu.scenes.tiled_urban_scene(n_per * world, tiles=(world, 1), tile_index=r) for r in range(world)]
    p = gpu.default_params(2, voxel_size=0.1)
    whole = np.concatenate(tiles)
    eng = gpu.Engine(p)
    eng.set_points(whole)
    eng.run()
    ref = eng.point_labels()

    fd = FakeDist(world)
    out, errs = [None] * world, []

    def work(r):
        try:
            fd.tls.rank = r
            d = torch.from_numpy(tiles[r]).to("cuda:0")
            seg = TiledSegmenter(gpu.default_params(2, voxel_size=0.1), fd, tiles=(world, 1), rank=r, world=world, pitch=pitch)
            seg.set_points_device(d, tiles[r])
            seg.run()
            out[r] = (seg.point_labels(), seg.kept, seg.engine.counts())
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            fd.bar.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    tiled = np.concatenate([out[r][0] for r in range(world)])
    assert out[0][1] == out[1][1]
    agree = partition_agreement(tiled, ref)
    assert agree >= 0.999, agree
    kept_ref = eng.counts()["kept"]
    assert abs(out[0][1] - kept_ref) <= max(2, 0.01 * kept_ref), (out[0][1], kept_ref)
    # at least one segment spans both tiles and carries ONE label
    left = set(tiled[: n_per][tiled[: n_per] >= 0].tolist())
    right = set(tiled[n_per:][tiled[n_per:] >= 0].tolist())
    assert left & right
