"""world_size-2 gloo tests of the tiled path's host logic (vgs-svgs-segmentation_amd/dist.py): the all-gather of
boundary records and the per-rank union-find must reproduce the single-process segmentation.  The per-rank
engine results are emulated from one oracle run over the whole scene (the HIP engine needs a GPU): a rank sees
every voxel but trusts only connections with an owned endpoint, exactly what the engine does with its halo."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _emulate_rank(r, regions, centers, used, off, idx, voxels_min):
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    lo, hi = regions[r]
    V = centers.shape[0]
    owned = (centers[:, 0] >= lo[0]) & (centers[:, 0] < hi[0]) & (centers[:, 1] >= lo[1]) & (centers[:, 1] < hi[1])
    src = np.repeat(np.arange(V), np.diff(off))
    dst = idx
    trusted = owned[src] | owned[dst]
    g = coo_matrix((np.ones(trusted.sum(), np.int8), (src[trusted], dst[trusted])), shape=(V, V))
    _, comp = connected_components(g, directed=False)
    root_of_comp = np.full(comp.max() + 1, V, dtype=np.int64)
    np.minimum.at(root_of_comp, comp, np.arange(V))
    root = root_of_comp[comp].astype(np.int32)   # smallest voxel id of the local component, like the engine
    cross = trusted & (owned[src] != owned[dst]) & (src < dst)
    codes = np.concatenate([src[cross], dst[cross]]).astype(np.uint64)   # global voxel id plays the voxel code
    roots_rec = np.concatenate([root[src[cross]], root[dst[cross]]]).astype(np.int32)
    own_roots, own_cnt = np.unique(root[owned], return_counts=True)
    return owned, root, (codes, roots_rec), (own_roots.astype(np.int32), own_cnt.astype(np.int32))


def _worker(rank, world, path, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import vgs_svgs_segmentation_amd as v
    from vgs_svgs_segmentation_amd.dist import all_gather_varlen, merge_boundary, tile_regions
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(path)
    regions = tile_regions((world, 1), float(d["pitch"]))
    owned, root, rec, roots = _emulate_rank(rank, regions, d["centers"], d["used"], d["off"], d["idx"], int(d["voxels_min"]))
    codes = all_gather_varlen(dist, rec[0].view(np.int64))
    rroots = all_gather_varlen(dist, rec[1])
    allrt = all_gather_varlen(dist, roots[0])
    alloc = all_gather_varlen(dist, roots[1])
    labels, kept = merge_boundary([(c.view(np.uint64), r) for c, r in zip(codes, rroots)], list(zip(allrt, alloc)), int(d["voxels_min"]))
    lab_of_root = dict(zip(roots[0].tolist(), labels[rank].tolist()))
    vox_label = np.array([lab_of_root[int(x)] if o else -2 for x, o in zip(root, owned)], dtype=np.int64)
    np.savez(out_path % rank, vox_label=vox_label, kept=kept)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_two_rank_merge_matches_single_process(oracle, vgs, tmp_path, world):
    import torch.multiprocessing as mp
    from helpers import canonical_labels
    n_per = 100_000
    xyz = vgs.scenes.tiled_urban_scene(n_per * world, tiles=(world, 1))
    pitch = 50.0 * np.sqrt(n_per / 10_000_000)
    P = oracle.vgs_params(voxel_size=0.1, math=1, flavour=1)
    ref = oracle.run_vgs(xyz, P)
    off, idx = ref.lists("connect_final")
    t = ref.voxel_table()
    path = str(tmp_path / "scene.npz")
    np.savez(path, centers=t["center"], used=ref.nodes()["used"], off=off, idx=idx, pitch=pitch, voxels_min=3)
    out_path = str(tmp_path / "rank%d.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, path, port, out_path), nprocs=world, join=True)
    res = [np.load(out_path % r) for r in range(world)]
    vox = np.full(ref.V, -2, dtype=np.int64)
    for r in res:
        m = r["vox_label"] != -2
        assert (vox[m] == -2).all(), "a voxel is owned by two ranks"
        vox[m] = r["vox_label"][m]
    assert (vox != -2).all(), "a voxel is owned by no rank"
    assert len({int(r["kept"]) for r in res}) == 1
    # single-process truth: clusters with > voxels_min voxels keep a label, others are dropped
    _, node_cluster = ref.labels()
    sizes = np.bincount(node_cluster)
    truth = np.where(sizes[node_cluster] > 3, node_cluster, -1)
    np.testing.assert_array_equal(canonical_labels(vox), canonical_labels(truth))
    assert int(res[0]["kept"]) == ref.kept_clusters
    # and the scene really has segments that span both tiles
    lo_side = t["center"][:, 0] < 0
    spanning = set(truth[lo_side & (truth >= 0)].tolist()) & set(truth[~lo_side & (truth >= 0)].tolist())
    assert spanning, "test scene has no segment crossing the tile border"


def test_merge_boundary_small_example(vgs):
    from vgs_svgs_segmentation_amd.dist import merge_boundary
    # rank 0: roots 5 (10 voxels), 9 (2); rank 1: roots 1 (1), 7 (3).  Code 100 links (0,9)-(1,1); code 200 links (0,5)-(1,7)
    rec0 = (np.array([100, 200], np.uint64), np.array([9, 5], np.int32))
    rec1 = (np.array([100, 200], np.uint64), np.array([1, 7], np.int32))
    roots0 = (np.array([5, 9], np.int32), np.array([10, 2], np.int32))
    roots1 = (np.array([1, 7], np.int32), np.array([1, 3], np.int32))
    labels, kept = merge_boundary([rec0, rec1], [roots0, roots1], voxels_min=3)
    assert kept == 1
    assert labels[0].tolist() == [0, -1] and labels[1].tolist() == [-1, 0]   # 10+3 kept, 2+1 dropped
