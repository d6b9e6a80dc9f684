"""Host-side mirror of the reference's class API over the C-ABI (include/vgs.h).

`Engine` is a thin handle wrapper (one method per C entry point, numpy in/out).
`VoxelBasedSegmentation` / `SuperVoxelBasedSegmentation` keep the reference's method names, argument
meaning and call order (voxel_segmentation.h:84-421, 947-1014; supervoxel_segmentation.h:85-421), so the
driver snippets of the reference's `test` file (test:51-76, test:138-160) read the same here.
The C++ twin of this file is include/vgs_segmentation.hpp.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import VgsError, VgsGridState, VgsParams


def default_params(method=2, **kw):
    p = VgsParams()
    L = _lib.lib()
    (L.vgs_params_default_svgs if method == 3 else L.vgs_params_default_vgs)(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def parse_task_file(path):
    """inputTaskTxtFile + the line indices of segmentationVGS/SVGS (point_clouds_IO.cpp:148-169, test:25-37,108-125)."""
    p = VgsParams()
    a = C.create_string_buffer(512)
    b = C.create_string_buffer(512)
    st = _lib.lib().vgs_parse_task_file(path.encode(), C.byref(p), a, b, 512)
    if st != _lib.VGS_OK:
        raise VgsError(st, f"cannot parse task file {path}")
    return p, a.value.decode(), b.value.decode()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class _Pinned:
    """Owner of one hipHostMalloc block (freed when the last numpy view of it goes away)."""

    def __init__(self, nbytes):
        self._L = _lib.lib()
        self.p = C.c_void_p()
        st = self._L.vgs_host_alloc(C.byref(self.p), max(int(nbytes), 1))
        if st != _lib.VGS_OK:
            raise VgsError(st, f"vgs_host_alloc({nbytes})")

    def __del__(self):
        try:
            if self.p.value:
                self._L.vgs_host_free(self.p)
                self.p = C.c_void_p()
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """numpy array in pinned host memory (vgs_host_alloc): the staging buffers of stage_points / point_labels_async."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    blk = _Pinned(n)
    buf = (C.c_char * max(n, 1)).from_address(blk.p.value)
    buf._owner = blk   # the ctypes array keeps the block alive, the numpy array keeps the ctypes array alive
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class Engine:
    def __init__(self, params: VgsParams):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self.params = params
        st = self._L.vgs_create(C.byref(params), C.byref(self._h))
        if st != _lib.VGS_OK:
            raise VgsError(st, self._L.vgs_last_error_string(None).decode())
        self._keep = None
        self.n = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.vgs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        if st != _lib.VGS_OK:
            raise VgsError(st, self._L.vgs_last_error_string(self._h).decode())

    def set_params(self, params: VgsParams):
        self._ck(self._L.vgs_set_params(self._h, C.byref(params)))
        self.params = params

    # ---- input
    def set_points(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        if xyz.ndim != 2 or xyz.shape[1] not in (3, 4):
            raise ValueError("xyz must be (N,3) or (N,4) float32")
        self._keep = xyz
        self.n = xyz.shape[0]
        self._ck(self._L.vgs_set_points(self._h, _ptr(xyz), xyz.shape[0], xyz.shape[1] * 4))

    def set_points_device(self, ptr, n, stride_bytes=12, keep=None):
        self._keep = keep
        self.n = int(n)
        self._ck(self._L.vgs_set_points_device(self._h, C.c_void_p(ptr), int(n), stride_bytes))

    # ---- stages
    def voxelize(self): self._ck(self._L.vgs_voxelize(self._h))
    def features(self): self._ck(self._L.vgs_features(self._h))
    def adjacency(self): self._ck(self._L.vgs_adjacency(self._h))
    def segment(self): self._ck(self._L.vgs_segment(self._h))
    def run(self): self._ck(self._L.vgs_run(self._h))

    # ---- SVGS
    def set_supervoxel_labels(self, labels, max_label):
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        if labels.shape[0] != self.n:
            raise ValueError("one label per point required")
        self._ck(self._L.svgs_set_supervoxel_labels(self._h, _ptr(labels), int(max_label)))

    def supervoxels(self): self._ck(self._L.svgs_supervoxels(self._h))
    def svgs_segment(self): self._ck(self._L.svgs_segment(self._h))

    # ---- results
    def counts(self):
        c = np.zeros(_lib.N_COUNTS, dtype=np.int64)
        self._ck(self._L.vgs_get_counts(self._h, _ptr(c)))
        names = ["points", "finite", "voxels", "used", "adj", "clusters", "kept", "pairs", "depth", "isolated", "reattached",
                 "supervoxels", "handed_over", "class_a", "class_bc", "class_d"]  # 12..15: local-cut scheduling diagnostics
        return {k: int(c[i]) for i, k in enumerate(names)}

    def schedule_counters(self):
        c = np.zeros(13, dtype=np.int64)
        self._ck(self._L.vgs_get_schedule_counters_ex(self._h, _ptr(c), 13))
        names = ("lazy_gave_up", "list_overflow", "handed_over", "dense_sent_on", "handed_over_large", "outside_limits", "cross_put_off", "banded",
                 "extra_large", "pair_list_cut", "pair_list_entries", "voted_over", "pair_list_pool_full")
        return dict(zip(names, (int(x) for x in c)))

    def stage_times(self):
        t = np.zeros(_lib.T_COUNT, dtype=np.float64)
        self._ck(self._L.vgs_get_stage_times(self._h, _ptr(t)))
        names = ["voxelize", "features", "adjacency", "localcut", "merge", "labels", "total", "localcut_kernel", "supervoxel", "localcut_bulk"]
        return {k: float(t[i]) for i, k in enumerate(names)}

    def bbox(self):
        b = np.zeros(6, dtype=np.float64)
        self._ck(self._L.vgs_get_bbox(self._h, _ptr(b)))
        return b

    def voxel_table(self):
        c = self.counts()
        V, nf = c["voxels"], c["finite"]
        key = np.zeros((V, 3), dtype=np.uint32)
        start = np.zeros(V + 1, dtype=np.int32)
        pidx = np.zeros(nf, dtype=np.int32)
        self._ck(self._L.vgs_get_voxel_table(self._h, _ptr(key), _ptr(start), _ptr(pidx)))
        return dict(key=key, start=start, point_idx=pidx)

    def voxel_centers(self):
        V = self.counts()["voxels"]
        c = np.zeros((V, 3), dtype=np.float32)
        self._ck(self._L.vgs_get_voxel_centers(self._h, _ptr(c)))
        return c

    def point_voxel(self):
        out = np.zeros(self.n, dtype=np.int32)
        self._ck(self._L.vgs_get_point_voxel(self._h, _ptr(out)))
        return out

    def attributes(self):
        V = self.counts()["voxels"]
        cen = np.zeros((V, 3), dtype=np.float32)
        nrm = np.zeros((V, 3), dtype=np.float32)
        eig = np.zeros((V, 8), dtype=np.float32)
        used = np.zeros(V, dtype=np.uint8)
        self._ck(self._L.vgs_get_attributes(self._h, _ptr(cen), _ptr(nrm), _ptr(eig), _ptr(used)))
        return dict(centroid=cen, normal=nrm, eigen=eig, used=used)

    LISTS = {"adjacency": 0, "connect_cut": 1, "connect_cross": 2, "connect_final": 3}

    def lists(self, which, order="voxel_id"):
        """Ragged lists as (offsets, ids).  order="reference": the connect lists element for element as the reference holds them
        (merge-history order of the local cut, csrc/cutorder.hip); "voxel_id": members in adjacency-row order (the hot path's)."""
        V = self.counts()["voxels"]
        off = np.zeros(V + 1, dtype=np.int64)
        w, o = self.LISTS[which], self.ORDERS[order]
        self._ck(self._L.vgs_get_lists_ordered(self._h, w, o, _ptr(off), None))
        idx = np.zeros(max(int(off[-1]), 1), dtype=np.int32)
        self._ck(self._L.vgs_get_lists_ordered(self._h, w, o, _ptr(off), _ptr(idx)))
        return off, idx[:int(off[-1])]

    def adjacency_counts(self):
        """Neighbours inside graph_size per node, itself included (0 for unused voxels)."""
        out = np.zeros(self.counts()["voxels"], dtype=np.int32)
        self._ck(self._L.vgs_get_adjacency_counts(self._h, _ptr(out)))
        return out

    def local_weights(self, node_id):
        """(ids, W): the n x n affinity matrix of node_id's local graph over its stored adjacency row (W[a, b]: ids[a] first)."""
        n = C.c_int32(0)
        self._ck(self._L.vgs_get_local_weights(self._h, int(node_id), C.byref(n), None, None))
        ids = np.zeros(max(n.value, 1), dtype=np.int32)
        W = np.zeros((max(n.value, 1), max(n.value, 1)), dtype=np.float32)
        if n.value:
            self._ck(self._L.vgs_get_local_weights(self._h, int(node_id), C.byref(n), _ptr(ids), _ptr(W)))
        return ids[:n.value], W[:n.value, :n.value]

    def node_labels(self):
        V = self.counts()["voxels"]
        root = np.zeros(V, dtype=np.int32)
        kept = np.zeros(V, dtype=np.int32)
        self._ck(self._L.vgs_get_node_labels(self._h, _ptr(root), _ptr(kept)))
        return root, kept

    def point_labels(self):
        out = np.zeros(self.n, dtype=np.int32)
        self._ck(self._L.vgs_get_point_labels(self._h, _ptr(out)))
        return out

    def supervoxel_labels(self):
        """Per-point supervoxel labels of the last svgs_supervoxels / set_supervoxel_labels (0 = unassigned) and max_label."""
        out = np.zeros(self.n, dtype=np.int32)
        mx = C.c_int32(0)
        self._ck(self._L.svgs_get_supervoxel_labels(self._h, _ptr(out), C.byref(mx)))
        return out, int(mx.value)

    def point_labels_device_ptr(self):
        p = C.c_void_p()
        self._ck(self._L.vgs_get_point_labels_device(self._h, C.byref(p)))
        return p.value

    ORDERS = {"voxel_id": 0, "reference": 1}

    def clusters(self, order="voxel_id"):
        """getClusterIdx as (offsets, point indices).  order="reference": nodes in recursionSearch's DFS order with the seed
        last, points per node ascending (voxel_segmentation.h:2032-2080, 981-999) -- the reference's own element order."""
        K = self.counts()["kept"]
        o = self.ORDERS[order]
        off = np.zeros(K + 1, dtype=np.int64)
        self._ck(self._L.vgs_get_clusters_ordered(self._h, o, _ptr(off), None))
        idx = np.zeros(max(int(off[-1]), 1), dtype=np.int32)
        self._ck(self._L.vgs_get_clusters_ordered(self._h, o, _ptr(off), _ptr(idx)))
        return off, idx[:int(off[-1])]

    def clusters_device(self):
        """getClusterIdx left in HBM: (device pointer of the offsets [kept + 1] int64, device pointer of the point indices int32)."""
        po, pi = C.c_void_p(), C.c_void_p()
        self._ck(self._L.vgs_get_clusters_device(self._h, C.byref(po), C.byref(pi)))
        return po.value, pi.value

    # ---- a sequence of clouds: uploads of the next cloud and downloads of the last labels overlap the stages
    def stage_points(self, xyz):
        """Start the copy of the NEXT cloud (ideally a pinned array, see pinned_empty) and return at once."""
        if xyz.dtype != np.float32 or not xyz.flags["C_CONTIGUOUS"] or xyz.ndim != 2 or xyz.shape[1] not in (3, 4):
            raise ValueError("xyz must be a C-contiguous (N,3) or (N,4) float32 array")
        self._staged = xyz
        self._ck(self._L.vgs_stage_points(self._h, _ptr(xyz), xyz.shape[0], xyz.shape[1] * 4))

    def commit_points(self):
        self._ck(self._L.vgs_commit_points(self._h))
        self._keep, self._staged = self._staged, None
        self.n = self._keep.shape[0]

    def point_labels_async(self, out):
        if out.dtype != np.int32 or not out.flags["C_CONTIGUOUS"] or out.shape[0] < self.n:
            raise ValueError("out must be a C-contiguous int32 array with one entry per point")
        self._labels_out = out
        self._ck(self._L.vgs_get_point_labels_async(self._h, _ptr(out)))

    def wait_labels(self):
        self._ck(self._L.vgs_wait_point_labels(self._h))


class VoxelBasedSegmentation:
    """pcl::VoxelBasedSegmentation<PointXYZ> (voxel_segmentation.h:57-2305), same member names and call order."""

    def __init__(self, input_resolution, device=0):          # VS:84
        self._p = default_params(2, voxel_size=float(input_resolution), device=device)
        self._eng = Engine(self._p)
        self._cloud = None
        self._drawn = False
        self._adj = None

    def _push(self):
        self._eng.set_params(self._p)

    # inherited PCL surface used by the driver (test:52-56)
    def setInputCloud(self, cloud):
        self._cloud = np.ascontiguousarray(cloud, dtype=np.float32)

    def getCloudPointNum(self, cloud):                       # VS:94
        self._cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        return int(self._cloud.shape[0])

    def addPointsFromInputCloud(self):                       # test:54
        if self._cloud is None:
            raise VgsError(_lib.VGS_E_STATE, "addPointsFromInputCloud before setInputCloud")
        self._eng.set_points(self._cloud)
        self._eng.voxelize()

    def setVoxelSize(self, input_resolution, points_num_min, voxels_num_min, voxels_adj_min):  # VS:124
        # stored only (VS:127): the octree keeps binning with the constructor's resolution (VS:84)
        self.voxel_resolution_ = float(input_resolution)
        self._p.points_min = int(points_num_min)
        self._p.voxels_min = int(voxels_num_min)
        self._p.adjacency_min = int(voxels_adj_min)
        self._push()

    def getBoundingBox(self):                                # test:56
        return tuple(self._eng.bbox())

    def setBoundingBox(self, *args):                         # VS:133 (the engine keeps the octree's own box)
        pass

    def setVoxelCenters(self):                               # VS:146
        if self._eng.counts()["points"] and self._eng.stage_times()["voxelize"] == 0.0:
            self._eng.voxelize()

    def getVoxelCenters(self):                               # VS:191
        return self._eng.voxel_centers()

    def getVoxelNum(self):                                   # VS:104
        return self._eng.counts()["voxels"]

    def calcualteVoxelCloudAttributes(self, cloud=None):     # VS:290 (sic)
        self._eng.features()

    def findAllVoxelAdjacency(self, graph_size):             # VS:223
        self._p.graph_size = float(graph_size)
        self._push()
        self._eng.adjacency()
        self._adj = None

    def getOneVoxelAdjacency(self, voxel_id):                # VS:268
        if self._adj is None:    # fetched once per findAllVoxelAdjacency, not once per call
            self._adj = self._eng.lists("adjacency")
        off, idx = self._adj
        return idx[off[voxel_id]:off[voxel_id + 1]].tolist()

    def segmentVoxelCloudWithGraphModel(self, cut_thred, sig_p, sig_n, sig_o, sig_e, sig_c, sig_w):  # VS:372
        q = self._p
        q.cut_thred, q.sig_p, q.sig_n, q.sig_o, q.sig_e, q.sig_c, q.sig_w = (float(v) for v in (
            cut_thred, sig_p, sig_n, sig_o, sig_e, sig_c, sig_w))
        self._push()
        self._eng.segment()

    def drawColorMapofPointsinClusters(self, output_cloud=None):  # VS:947 "This is obligatory!"
        self._drawn = True
        return self._eng.point_labels()

    def getClusterNum(self):                                 # VS:111
        return self._eng.counts()["clusters"]

    def getClusterIdx(self):                                 # VS:117
        if not self._drawn:
            return []   # clusters_point_idx_ is only filled by drawColorMapofPointsinClusters (VS:1006)
        off, idx = self._eng.clusters("reference")
        return [idx[off[k]:off[k + 1]].tolist() for k in range(len(off) - 1)]

    @property
    def engine(self):
        return self._eng


class SuperVoxelBasedSegmentation:
    """pcl::SuperVoxelBasedSegmentation<PointXYZ> (supervoxel_segmentation.h:58-2308), same member names and call order."""

    def __init__(self, input_resolution, device=0):                                   # SS:85
        self._p = default_params(3, voxel_size=float(input_resolution), device=device)
        self._eng = Engine(self._p)
        self._cloud = None

    def setInputCloud(self, cloud):
        self._cloud = np.ascontiguousarray(cloud, dtype=np.float32)

    def getCloudPointNum(self, cloud):                                                # SS:101
        self._cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        return int(self._cloud.shape[0])

    def addPointsFromInputCloud(self):                                                # test:142 (own octree: bookkeeping only)
        self._eng.set_points(self._cloud)

    def setVoxelSize(self, input_resolution, points_num_min):                         # SS:143
        self._p.voxel_size = float(input_resolution)
        self._p.points_min = int(points_num_min)
        self._eng.set_params(self._p)

    def setSupervoxelSize(self, input_resolution, voxels_num_min, points_num_min, adjacency_num_min):  # SS:150
        self._p.seed_size = float(input_resolution)
        self._p.voxels_min = int(voxels_num_min)
        self._p.adjacency_min = int(adjacency_num_min)
        self._eng.set_params(self._p)

    def setGraphSize(self, small_resolution, large_resolution):                       # SS:159 (the small radius feeds nothing, SS:1438-1475)
        self._p.graph_size = float(large_resolution)
        self._eng.set_params(self._p)

    def setBoundingBox(self, *args):                                                  # SS:166
        pass

    def setSupervoxelCentersCentroids(self):                                          # SS:178 (own-octree bookkeeping)
        pass

    def setSupervoxelLabels(self, labels, max_label):
        """What pcl::SupervoxelClustering::getLabeledCloud / getMaxLabel return (SS:283-284), supplied by the caller."""
        self._eng.set_supervoxel_labels(labels, max_label)

    def getVoxelNum(self):                                                            # SS:111
        return self._eng.counts()["voxels"]

    def getSuperVoxelNum(self):                                                       # SS:118
        return self._eng.counts()["supervoxels"]

    def segmentSupervoxelCloudWithGraphModel(self, sig_a, sig_b, sig_l, cut_thred, sig_p, sig_n, sig_o, sig_e, sig_c, sig_w):  # SS:362
        q = self._p
        q.color_impt, q.spatial_impt, q.normal_impt = float(sig_a), float(sig_b), float(sig_l)
        q.cut_thred, q.sig_p, q.sig_n, q.sig_o, q.sig_e, q.sig_c, q.sig_w = (float(v) for v in (
            cut_thred, sig_p, sig_n, sig_o, sig_e, sig_c, sig_w))
        self._eng.set_params(q)
        self._eng.run()   # createSupervoxels unless the caller's labelling of THIS cloud is in place, then the graph stages

    def drawColorMapofPointsinClusters(self, output_cloud=None):                      # SS:613
        return self._eng.point_labels()

    def getClusterNum(self):                                                          # SS:124
        return self._eng.counts()["clusters"]

    def getClusterIdx(self):                                                          # SS:130
        off, idx = self._eng.clusters("reference")
        return [idx[off[k]:off[k + 1]].tolist() for k in range(len(off) - 1)]

    @property
    def engine(self):
        return self._eng


def segmentation_vgs(cloud, params: VgsParams):
    """segmentationVGS (test:9-86) without file IO and viewer: returns (point labels, Engine)."""
    eng = Engine(params)
    eng.set_points(cloud)
    eng.run()
    return eng.point_labels(), eng
