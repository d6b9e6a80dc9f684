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
    from vgs_svgs_segmentation_amd.dist import all_gather_varlen, merge_boundary_compact, tile_regions
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(path)
    regions = tile_regions((world, 1), float(d["pitch"]))
    owned, root, rec, roots = _emulate_rank(rank, regions, d["centers"], d["used"], d["off"], d["idx"], int(d["voxels_min"]))
    # the compact protocol of TiledSegmenter.run: unique boundary voxels (code, root, owned voxels of the root), the
    # number of kept local segments, ONE packed all-gather, the O(boundary) merge, labels = base + rank for the rest
    vmin = int(d["voxels_min"])
    ucode, first = np.unique(rec[0], return_index=True)
    broot = rec[1][first]
    cnt_of = dict(zip(roots[0].tolist(), roots[1].tolist()))
    bcnt = np.array([cnt_of.get(int(x), 0) for x in broot], dtype=np.int64)
    is_b = np.isin(roots[0], broot)
    local_kept = (~is_b) & (roots[1] > vmin)
    payload = np.concatenate([[ucode.size, int(local_kept.sum())], ucode.view(np.int64), broot.astype(np.int64), bcnt]).astype(np.int64)
    from vgs_svgs_segmentation_amd.dist import all_gather_records
    gathered = all_gather_records(dist, payload)   # what TiledSegmenter.run uses: one fixed-size collective when everything fits
    assert all(np.array_equal(a, b) for a, b in zip(gathered, all_gather_varlen(dist, payload)))
    # ... and the two-step fallback when some rank's payload does not fit the fixed size
    small = all_gather_records(dist, payload, cap=2 + 3 * 4)
    assert all(np.array_equal(a, b) for a, b in zip(gathered, small))
    records, kept_local = [], []
    for g in gathered:
        m = int(g[0]); kept_local.append(int(g[1]))
        body = g[2:2 + 3 * m].reshape(3, m)
        records.append((body[0].view(np.uint64), body[1].astype(np.int32), body[2].astype(np.int32)))
    base, blabels, kept = merge_boundary_compact(records, kept_local, vmin)
    lab = np.full(roots[0].size, -1, dtype=np.int64)
    lab[local_kept] = base[rank] + np.arange(int(local_kept.sum()))
    lab[np.searchsorted(roots[0], blabels[rank][0])] = blabels[rank][1]
    lab_of_root = dict(zip(roots[0].tolist(), lab.tolist()))
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


def test_compact_merge_equals_full_merge():
    """merge_boundary_compact (only boundary voxels leave the GPU, local segments are labelled on the device as
    base + rank) gives the same partition and the same kept count as merge_boundary over all roots."""
    from vgs_svgs_segmentation_amd.dist import merge_boundary, merge_boundary_compact
    rng = np.random.default_rng(5)
    world, vmin = 3, 3
    for trial in range(20):
        roots, full_records, compact_records, kept_local, local_sets = [], [], [], [], []
        n_codes = 40
        for r in range(world):
            nr = int(rng.integers(5, 30))
            rt = np.sort(rng.choice(1000, nr, replace=False)).astype(np.int32)
            oc = rng.integers(0, 6, nr).astype(np.int32)
            roots.append((rt, oc))
            # boundary voxels of this rank: a random subset of shared codes, each belonging to one of its roots
            nb = int(rng.integers(0, 15))
            codes = rng.choice(n_codes, nb, replace=False).astype(np.uint64)
            broots = rt[rng.integers(0, nr, nb)]
            dup = rng.integers(0, nb, 2 * nb) if nb else np.zeros(0, np.int64)      # the full protocol repeats records
            full_records.append((np.concatenate([codes, codes[dup]]), np.concatenate([broots, broots[dup]]).astype(np.int32)))
            cnt = oc[np.searchsorted(rt, broots)]
            compact_records.append((codes, broots.astype(np.int32), cnt.astype(np.int32)))
            is_b = np.isin(rt, broots)
            loc = (~is_b) & (oc > vmin)
            kept_local.append(int(loc.sum()))
            local_sets.append(loc)
        old_labels, old_kept = merge_boundary(full_records, roots, vmin)
        base, blabels, new_kept = merge_boundary_compact(compact_records, kept_local, vmin)
        assert new_kept == old_kept
        new_labels = []
        for r in range(world):
            rt, _ = roots[r]
            lab = np.full(rt.size, -1, dtype=np.int64)
            lab[local_sets[r]] = base[r] + np.arange(kept_local[r])          # what vgs_apply_tile_labels does on the device
            ur, ul = blabels[r]
            lab[np.searchsorted(rt, ur)] = ul
            new_labels.append(lab)
        a, b = np.concatenate(old_labels), np.concatenate(new_labels)
        assert np.array_equal(a < 0, b < 0)
        # same partition: the label pairs are a bijection
        pairs = set(zip(a[a >= 0].tolist(), b[b >= 0].tolist()))
        assert len(pairs) == len({x for x, _ in pairs}) == len({y for _, y in pairs})
