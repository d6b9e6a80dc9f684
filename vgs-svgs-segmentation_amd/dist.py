"""One process per GPU: spatial tiles, shared grid, one all-gather of boundary records.

The reference is single-process.  Everything per voxel (binning, PCA, adjacency, local cut, mutual filter) is
local to a ball of radius graph_size; only the connected components are global.  Each rank therefore segments
its tile plus a halo of raw points (2*graph_size + voxel_size wide) on ONE shared grid, trusts the mutual
connections that have an owned endpoint and the re-attachments (closestCheck) of its own voxels, and publishes one
(voxel code, local root) record per boundary voxel: the endpoints of every connection crossing the ownership
border, every owned voxel with a halo voxel in its neighbourhood (the neighbouring rank may re-attach one of its
isolated voxels to it), and every voxel whose cube reaches over the border (both ranks hold points of it; the rank
that does not own it learns its label from the owner's record).  After one all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests) every rank
runs the same small union-find over (rank, root) pairs and labels its own points.  Payloads are O(boundary
voxels): the exchange is latency-bound, so it is a single collective with no tuning.

`merge_boundary` is pure host logic (numpy/scipy) and is what the world_size-2 gloo tests exercise.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import VgsGridState
from .api import Engine, _ptr


def tile_regions(tiles, pitch, center=(0.0, 0.0)):
    """Ownership rectangles [lo, hi) of a tiles[0] x tiles[1] layout; outer edges are open ended."""
    tx, ty = tiles
    big = 1.0e30
    out = []
    for k in range(tx * ty):
        i, j = k % tx, k // tx
        x0 = center[0] + (i - tx / 2.0) * pitch
        y0 = center[1] + (j - ty / 2.0) * pitch
        lo = [x0 if i > 0 else -big, y0 if j > 0 else -big]
        hi = [x0 + pitch if i < tx - 1 else big, y0 + pitch if j < ty - 1 else big]
        out.append((np.array(lo, dtype=np.float64), np.array(hi, dtype=np.float64)))
    return out


def merge_boundary(records, roots, voxels_min):
    """Global segments from per-rank results.

    records[r] = (codes uint64[], local_roots int32[]): one entry per endpoint of a border-crossing connection.
    roots[r]   = (local_roots int32[], owned_voxel_counts int32[]): every local component with owned voxels.
    Returns labels[r] (int32, aligned with roots[r][0]): dense global label or -1 when the global segment has
    <= voxels_min voxels (the reference's filter, voxel_segmentation.h:969), and the number of kept segments.
    Deterministic: every rank computes the same table from the same gathered inputs.
    """
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    world = len(roots)
    ids, cnts, owner = [], [], []
    for r in range(world):
        rt, oc = roots[r]
        ids.append((np.int64(r) << 32) | rt.astype(np.int64))
        cnts.append(oc.astype(np.int64))
        owner.append(np.full(rt.shape, r, dtype=np.int64))
    ids = np.concatenate(ids) if ids else np.zeros(0, np.int64)
    cnts = np.concatenate(cnts) if cnts else np.zeros(0, np.int64)
    rec_code = np.concatenate([records[r][0].astype(np.uint64) for r in range(world)]) if world else np.zeros(0, np.uint64)
    rec_id = np.concatenate([(np.int64(r) << 32) | records[r][1].astype(np.int64) for r in range(world)]) if world else np.zeros(0, np.int64)
    # nodes: every (rank, root) that owns voxels or appears in a record
    nodes = np.unique(np.concatenate([ids, rec_id]))
    n = nodes.size
    if n == 0:
        return [np.zeros(0, np.int32) for _ in range(world)], 0
    pos = np.searchsorted(nodes, rec_id)
    order = np.argsort(rec_code, kind="stable")
    c_sorted, p_sorted = rec_code[order], pos[order]
    same = c_sorted[1:] == c_sorted[:-1]
    a, b = p_sorted[:-1][same], p_sorted[1:][same]
    g = coo_matrix((np.ones(a.size, dtype=np.int8), (a, b)), shape=(n, n))
    _, comp = connected_components(g, directed=False)
    total = np.zeros(comp.max() + 1, dtype=np.int64)
    np.add.at(total, comp[np.searchsorted(nodes, ids)], cnts)
    keep = total > voxels_min
    # dense labels in order of the smallest (rank, root) id of each kept component
    first = np.full(comp.max() + 1, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first, comp, nodes)
    kept_ids = np.nonzero(keep)[0]
    rank_of = np.full(comp.max() + 1, -1, dtype=np.int64)
    rank_of[kept_ids[np.argsort(first[kept_ids], kind="stable")]] = np.arange(kept_ids.size)
    labels = []
    for r in range(world):
        rt = roots[r][0]
        nid = (np.int64(r) << 32) | rt.astype(np.int64)
        labels.append(rank_of[comp[np.searchsorted(nodes, nid)]].astype(np.int32))
    return labels, int(kept_ids.size)


def merge_boundary_compact(records, kept_local, voxels_min):
    """Global segments from the compact per-rank results (vgs_get_boundary_roots).

    records[r] = (codes uint64[], local_roots int32[], owned_counts int32[]): one entry per boundary voxel of rank r,
    with the number of owned voxels of the local component it belongs to.  kept_local[r] = number of components of
    rank r that touch no boundary voxel and pass the size filter on their own (labelled on the GPU as
    base[r] + rank).  Returns (base[r] for every rank, per-rank (unique local roots, labels) of the boundary
    components, total number of kept segments).  Only the border leaves the GPUs: the work here is O(boundary voxels).
    Deterministic: every rank computes the same tables from the same gathered inputs.
    """
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    world = len(records)
    kept_local = np.asarray(kept_local, dtype=np.int64)
    base = np.concatenate([[0], np.cumsum(kept_local)[:-1]]).astype(np.int64) if world else np.zeros(0, np.int64)
    next_label = int(kept_local.sum())
    node_id, node_cnt, rec_code, rec_node = [], [], [], []
    offset = 0
    per_rank = []
    for r in range(world):
        code, root, cnt = records[r]
        uroot, first, inv = np.unique(root, return_index=True, return_inverse=True)
        per_rank.append((uroot.astype(np.int32), offset))
        node_cnt.append(cnt[first].astype(np.int64))
        rec_code.append(code.astype(np.uint64))
        rec_node.append(inv.astype(np.int64) + offset)
        offset += uroot.size
    n = offset
    if n == 0:
        return base, [(np.zeros(0, np.int32), np.zeros(0, np.int32)) for _ in range(world)], next_label
    node_cnt = np.concatenate(node_cnt)
    rec_code = np.concatenate(rec_code)
    rec_node = np.concatenate(rec_node)
    order = np.argsort(rec_code, kind="stable")
    c_sorted, p_sorted = rec_code[order], rec_node[order]
    same = c_sorted[1:] == c_sorted[:-1]
    a, b = p_sorted[:-1][same], p_sorted[1:][same]
    g = coo_matrix((np.ones(a.size, dtype=np.int8), (a, b)), shape=(n, n))
    ncomp, comp = connected_components(g, directed=False)
    total = np.bincount(comp, weights=node_cnt, minlength=ncomp).astype(np.int64)
    keep = total > voxels_min
    # dense labels after the local ones, in order of the first node of each kept component
    first_node = np.full(ncomp, n, dtype=np.int64)
    np.minimum.at(first_node, comp, np.arange(n))
    kept_ids = np.nonzero(keep)[0]
    label_of = np.full(ncomp, -1, dtype=np.int64)
    label_of[kept_ids[np.argsort(first_node[kept_ids], kind="stable")]] = next_label + np.arange(kept_ids.size)
    out = []
    for r in range(world):
        uroot, off = per_rank[r]
        out.append((uroot, label_of[comp[off:off + uroot.size]].astype(np.int32)))
    return base, out, next_label + int(kept_ids.size)


def all_gather_varlen(dist, arr, device=None):
    """all_gather of 1-D numpy arrays of different lengths (one size exchange + one padded payload exchange)."""
    import torch
    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if device is not None:
        t = t.to(device)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    outs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return [o[:s].cpu().numpy() for o, s in zip(outs, sizes)]


def all_gather_records(dist, payload, device=None, cap=3 * 8192 + 2):
    """all_gather of the per-rank record payloads (payload[0] = number of records, 3 int64 words each).  One collective
    of a fixed size when every rank's payload fits `cap` words (the usual case: a few thousand boundary voxels); if some
    rank's does not, all ranks see that in the gathered headers and fall back to the two-step variable-length gather."""
    import torch
    world = dist.get_world_size()
    buf = np.zeros(cap, dtype=np.int64)
    n = min(payload.size, cap)
    buf[:n] = payload[:n]
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    outs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    outs = [o.cpu().numpy() for o in outs]
    if all(2 + 3 * int(o[0]) <= cap for o in outs):
        return [o[:2 + 3 * int(o[0])] for o in outs]
    return all_gather_varlen(dist, payload, device)


class TiledSegmenter:
    """Drives one Engine per rank over one tile of a tiles[0] x tiles[1] layout."""

    def __init__(self, params, dist, tiles, rank, world, pitch=None, center=(0.0, 0.0)):
        self.p = params
        self.dist = dist
        self.rank, self.world = rank, world
        self.tiles = tiles
        self.halo = 2.0 * params.graph_size + params.voxel_size
        self.engine = Engine(params)
        self.pitch = pitch
        self.center = center
        self.device = None       # where the points live
        self.coll_device = None  # where collective payloads live: the GPU for RCCL ("nccl"), the host for gloo
        self._keep = None
        self.n_own = 0
        self.own_first = 0
        self.kept = 0

    # -- input: own points + halo strips of the neighbours (data loading, outside the timed region) --
    def set_points_device(self, d_xyz, xyz_host):
        import torch
        self.device = d_xyz.device
        backend = self.dist.get_backend() if hasattr(self.dist, "get_backend") else "nccl"
        self.coll_device = self.device if backend == "nccl" else torch.device("cpu")
        if self.pitch is None:
            ext = float(xyz_host[:, 0].max() - xyz_host[:, 0].min())
            t = torch.tensor([ext], device=self.coll_device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            self.pitch = float(t.item())
        self.regions = tile_regions(self.tiles, self.pitch, self.center)
        lo, hi = self.regions[self.rank]
        h = self.halo
        x, y = xyz_host[:, 0].astype(np.float64), xyz_host[:, 1].astype(np.float64)
        # Points a rank holds beyond its border (generated tiles whose objects reach over the edge) reach the owner of that ground
        # through the strips below and take part in the owner's voxels.  Their labels are read from this rank's halo computation;
        # the voxels that hold them are published as boundary voxels by this rank and by their owner (vgs_set_own_point_range),
        # so the halo voxel's local root carries the owner's label.  n_outside says how many such points there are.
        self.n_outside = int(((x < lo[0]) | (x >= hi[0]) | (y < lo[1]) | (y >= hi[1])).sum())
        near = (x < lo[0] + h) | (x >= hi[0] - h) | (y < lo[1] + h) | (y >= hi[1] - h)
        strips = all_gather_varlen(self.dist, np.ascontiguousarray(xyz_host[near]).reshape(-1), self.coll_device)
        # The local cloud is assembled in RANK ORDER -- strips of lower ranks, own points, strips of higher ranks: the order in which
        # one process would have inserted the points.  A voxel's attributes depend on the order of its points (sequential float
        # sums; the normal's flip looks at the voxel's first point), so a voxel that holds points of two ranks must see them in the
        # same order on every rank and in the single engine (found by tools/fuzz_tiles.py: with the own points always first, rank 1
        # saw the voxels on its border with the two halves swapped, and a few normals in clutter flipped the other way).
        lower, higher = [], []
        for r, s in enumerate(strips):
            if r == self.rank or s.size == 0:
                continue
            s = s.reshape(-1, 3)
            sx, sy = s[:, 0].astype(np.float64), s[:, 1].astype(np.float64)
            m = (sx >= lo[0] - h) & (sx < hi[0] + h) & (sy >= lo[1] - h) & (sy < hi[1] + h)
            (lower if r < self.rank else higher).append(s[m])
        self.n_own = xyz_host.shape[0]
        self.own_first = int(sum(e.shape[0] for e in lower))
        parts = []
        if self.own_first:
            parts.append(torch.from_numpy(np.concatenate(lower)).to(self.device))
        parts.append(d_xyz)
        if higher and sum(e.shape[0] for e in higher):
            parts.append(torch.from_numpy(np.concatenate(higher)).to(self.device))
        local = torch.cat(parts, dim=0).contiguous() if len(parts) > 1 else d_xyz
        torch.cuda.synchronize(self.device)
        self._keep = local
        self.engine.set_points_device(local.data_ptr(), local.shape[0], 12, keep=local)
        self.engine._ck(self.engine._L.vgs_set_owned_region(self.engine._h, _ptr(lo), _ptr(hi)))
        self.engine._ck(self.engine._L.vgs_set_own_point_range(self.engine._h, int(self.own_first), int(self.n_own)))

    def _chain_grid(self):
        """The shared grid: what inserting the ranks' clouds one after the other does to the octree box.
        One all-gather of the clouds' bounding boxes lets every rank replay the growth on the host wherever the box alone
        decides it (vgs_grid_advance_bbox: the usual case for tiles that lie beside the box); only a rank whose cloud leaves
        the step open scans its points on the GPU (vgs_grid_advance) and broadcasts the state -- except rank 0, whose box
        starts at its first point: it always scans, before the all-gather, and its state travels with its bounding box.
        All ranks run the same replay on the same gathered numbers."""
        import torch
        L, eng = self.engine._L, self.engine
        bb = np.zeros(6, dtype=np.float32)
        nf = C.c_int64(0)
        eng._ck(L.vgs_points_bbox(eng._h, _ptr(bb), C.byref(nf)))

        def pack(gs):
            return [gs.min[0], gs.min[1], gs.min[2], float(gs.shift[0]), float(gs.shift[1]), float(gs.shift[2]), float(gs.depth), float(gs.defined)]

        def unpack(gs, v):
            for a in range(3):
                gs.min[a] = float(v[a])
                gs.shift[a] = int(v[3 + a])
            gs.depth, gs.defined = int(v[6]), int(v[7])

        # rank 0 starts the chain from nothing, so it needs no one else's numbers for its own growth: it scans its points
        # first and sends the state it reaches along with its bounding box (no broadcast of its own)
        g0 = VgsGridState()
        L.vgs_grid_state_init(C.byref(g0))
        self.chain_scans = 0
        if self.rank == 0 and nf.value > 0:
            eng._ck(L.vgs_grid_advance(eng._h, C.byref(g0)))
        mine = torch.tensor([float(v) for v in bb] + [float(nf.value)] + pack(g0), dtype=torch.float64, device=self.coll_device)
        allbb = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(allbb, mine)
        allbb = [t.cpu().numpy() for t in allbb]
        g = VgsGridState()
        L.vgs_grid_state_init(C.byref(g))
        for r in range(self.world):
            if allbb[r][6] == 0:      # no finite point: the cloud changes nothing
                continue
            if r == 0:
                unpack(g, allbb[0][7:15])
                self.chain_scans += 1
                continue
            need = C.c_int32(0)
            box = np.ascontiguousarray(allbb[r][:6], dtype=np.float32)   # float values, exactly as gathered
            st = L.vgs_grid_advance_bbox(C.byref(g), C.c_double(float(np.float32(self.p.voxel_size))), _ptr(box), C.byref(need))
            if st != _lib.VGS_OK:
                raise _lib.VgsError(st, "vgs_grid_advance_bbox")
            if not need.value:
                continue
            self.chain_scans += 1
            buf = torch.zeros(8, dtype=torch.float64, device=self.coll_device)
            if self.rank == r:
                eng._ck(L.vgs_grid_advance(eng._h, C.byref(g)))
                buf = torch.tensor(pack(g), dtype=torch.float64, device=self.coll_device)
            self.dist.broadcast(buf, src=r)
            unpack(g, buf.cpu().numpy())
        eng._ck(L.vgs_set_grid(eng._h, C.byref(g)))

    def run(self):
        import os, time
        dbg = bool(os.environ.get("VGS_DEBUG")) and self.rank == 0
        tt = [time.perf_counter()]
        def mark(name):
            if dbg:
                tt.append(time.perf_counter())
                print(f"[tiles] {name} {1e3 * (tt[-1] - tt[-2]):.2f} ms", flush=True)
        eng, L = self.engine, self.engine._L
        self._chain_grid()
        mark("grid chain")
        eng.voxelize(); eng.features(); eng.adjacency(); eng.segment()
        mark("four stages")
        # only the border leaves the GPU: unique boundary voxels (code, local root, owned voxels of that root) and the
        # number of components that are local to this tile
        n, nkl = C.c_int64(0), C.c_int64(0)
        eng._ck(L.vgs_get_boundary_roots(eng._h, C.byref(n), None, None, None, C.byref(nkl)))
        rec = np.zeros((3, max(n.value, 1)), dtype=np.int64)
        if n.value:
            code = np.zeros(n.value, dtype=np.uint64)
            root = np.zeros(n.value, dtype=np.int32)
            cnt = np.zeros(n.value, dtype=np.int32)
            eng._ck(L.vgs_get_boundary_roots(eng._h, C.byref(n), _ptr(code), _ptr(root), _ptr(cnt), C.byref(nkl)))
            rec[0, :n.value] = code.view(np.int64); rec[1, :n.value] = root; rec[2, :n.value] = cnt
        mark(f"boundary download ({n.value} boundary voxels, {nkl.value} local segments)")
        # the one data-path exchange: a fixed-size header (record count, local segment count) and the records
        payload = np.concatenate([[n.value, nkl.value], rec[:, :n.value].reshape(-1)]).astype(np.int64)
        gathered = all_gather_records(self.dist, payload, self.coll_device)
        mark("all-gather")
        records, kept_local = [], []
        for g in gathered:
            m = int(g[0]); kept_local.append(int(g[1]))
            body = g[2:2 + 3 * m].reshape(3, m)
            records.append((body[0].view(np.uint64), body[1].astype(np.int32), body[2].astype(np.int32)))
        base, blabels, self.kept = merge_boundary_compact(records, kept_local, self.p.voxels_min)
        mark("merge_boundary")
        broot, blab = blabels[self.rank]
        eng._ck(L.vgs_apply_tile_labels(eng._h, int(base[self.rank]), _ptr(np.ascontiguousarray(broot)), _ptr(np.ascontiguousarray(blab)),
                                        broot.size))
        mark("apply labels")

    def point_labels(self):
        """Labels of this rank's own points (halo points belong to other ranks)."""
        return self.engine.point_labels()[self.own_first: self.own_first + self.n_own]
