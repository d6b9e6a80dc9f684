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
