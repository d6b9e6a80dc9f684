"""Shared helpers for the parity tests: run the HIP engine and the CPU oracle on the same cloud and
compare stage by stage (P0 exact integer stages, P1 float attributes, P2 partition)."""
import numpy as np


def oracle_params(oracle, p, **kw):
    """RefParams from a VgsParams (same Task-file values); math=1 DevMath / flavour=1 lean by default."""
    d = dict(voxel_size=p.voxel_size, graph_size=p.graph_size, sig_p=p.sig_p, sig_n=p.sig_n, sig_o=p.sig_o, sig_e=p.sig_e,
             sig_c=p.sig_c, sig_w=p.sig_w, cut_thred=p.cut_thred, points_min=p.points_min, adjacency_min=p.adjacency_min,
             voxels_min=p.voxels_min, seed_size=p.seed_size, color_impt=p.color_impt, spatial_impt=p.spatial_impt,
             normal_impt=p.normal_impt, q7_count_as_index=p.q7_count_as_index, math=1, flavour=1)
    d.update(kw)
    return oracle.vgs_params(**d)


def ragged_sets(off, idx):
    return [frozenset(idx[off[i]:off[i + 1]].tolist()) for i in range(len(off) - 1)]


def ragged_lists(off, idx):
    return [idx[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]


def partition_agreement(a, b):
    """Fraction of elements whose segment in `a` is matched (best overlap) with their segment in `b`;
    labels < 0 are treated as one 'dropped' class."""
    a = np.asarray(a).astype(np.int64)
    b = np.asarray(b).astype(np.int64)
    a = np.where(a < 0, -1, a) + 1
    b = np.where(b < 0, -1, b) + 1
    key = a * (b.max() + 1) + b
    uk, cnt = np.unique(key, return_counts=True)
    ua = uk // (b.max() + 1)
    best = {}
    for x, c in zip(ua, cnt):
        if c > best.get(x, 0):
            best[x] = c
    return sum(best.values()) / len(a)


def p2_protocol(lab_test, lab_ref, point_voxel, used=None, min_voxels=20):
    """SURVEY.md 8c P2, all three clauses, of point labels `lab_test` (the path under test) against `lab_ref` (the oracle in
    the reference's arithmetic), `point_voxel` = node (voxel / supervoxel) of every point (< 0: in none):

    * agreement: share of the used nodes that lie in matching segments after best-match relabelling (every oracle
      segment is matched to the test segment it shares most nodes with; dropped nodes, label < 0, are one class);
    * min_iou:   smallest point-set IoU between an oracle segment of >= `min_voxels` nodes and its best match by points
      (1.0 if there is no such segment); `worst` names that segment;
    * kept_test / kept_ref: kept-segment counts (labels >= 0 present).
    Returns a dict; `assert_p2` applies the stated tolerances."""
    lt = np.asarray(lab_test).astype(np.int64)
    lr = np.asarray(lab_ref).astype(np.int64)
    pv = np.asarray(point_voxel).astype(np.int64)
    assert lt.shape == lr.shape == pv.shape
    ok = pv >= 0
    lt, lr, pv = np.where(lt < 0, -1, lt)[ok] + 1, np.where(lr < 0, -1, lr)[ok] + 1, pv[ok]
    nt, nr = int(lt.max(initial=0)) + 1, int(lr.max(initial=0)) + 1
    # node labels: every point of a node carries the node's label
    V = int(pv.max(initial=-1)) + 1
    vt, vr = np.zeros(V, np.int64), np.zeros(V, np.int64)
    vt[pv], vr[pv] = lt, lr
    assert np.array_equal(vt[pv], lt) and np.array_equal(vr[pv], lr), "a node's points carry different labels"
    present = np.zeros(V, bool)
    present[pv] = True
    sel = present if used is None else (present & np.asarray(used).astype(bool)[:V])
    # clause 1 on nodes
    uk, cnt = np.unique(vr[sel] * nt + vt[sel], return_counts=True)
    best = np.zeros(nr, np.int64)
    np.maximum.at(best, uk // nt, cnt)
    agreement = best.sum() / max(1, int(sel.sum()))
    # clause 2 on points, oracle segments with >= min_voxels nodes
    vox_per_ref = np.bincount(vr[present], minlength=nr)
    big = np.nonzero(vox_per_ref >= min_voxels)[0]
    big = big[big > 0]
    pk, pc = np.unique(lr * nt + lt, return_counts=True)
    size_t, size_r = np.bincount(lt, minlength=nt), np.bincount(lr, minlength=nr)
    inter = np.zeros(nr, np.int64)
    match = np.zeros(nr, np.int64)
    for k, c in zip(pk, pc):
        r, t = divmod(int(k), nt)
        if t > 0 and c > inter[r]:
            inter[r], match[r] = c, t
    min_iou, worst = 1.0, None
    for r in big:
        iou = inter[r] / (size_r[r] + size_t[match[r]] - inter[r]) if inter[r] else 0.0
        if iou < min_iou:
            min_iou, worst = float(iou), dict(ref_label=int(r) - 1, points=int(size_r[r]), nodes=int(vox_per_ref[r]),
                                              match=int(match[r]) - 1, match_points=int(size_t[match[r]]), iou=float(iou))
    return dict(agreement=float(agreement), min_iou=min_iou, worst=worst, big_segments=int(big.size),
                kept_test=int(np.unique(lt[lt > 0]).size), kept_ref=int(np.unique(lr[lr > 0]).size))


def assert_p2(lab_test, lab_ref, point_voxel, used=None, agreement=0.995, iou=0.98, count_tol=0.01):
    """The stated tolerance of SURVEY.md 8c P2: >= 99.5 % of the used nodes in matching segments, point-set IoU >= 0.98 for every
    oracle segment of >= 20 nodes, kept-segment count within +-1 %."""
    r = p2_protocol(lab_test, lab_ref, point_voxel, used)
    assert r["agreement"] >= agreement, r
    assert r["min_iou"] >= iou, r
    assert abs(r["kept_test"] - r["kept_ref"]) <= count_tol * r["kept_ref"], r
    return r


def canonical_labels(lab):
    """Relabel so that equal partitions give equal arrays: label = smallest member index of the class."""
    lab = np.asarray(lab)
    out = np.full(lab.shape, -1, dtype=np.int64)
    valid = lab >= 0
    if valid.any():
        idx = np.arange(lab.size)
        first = np.full(lab.max() + 1, lab.size, dtype=np.int64)
        np.minimum.at(first, lab[valid], idx[valid])
        out[valid] = first[lab[valid]]
    return out
