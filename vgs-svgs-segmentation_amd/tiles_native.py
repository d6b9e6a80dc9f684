"""ctypes binding of the native tiled driver (include/vgs_tiles.h = csrc/tiles.cpp, libvgs_tiles.so): one process per GPU, spatial
tiles, ONE all-gather of boundary records per run.  This is the product's multi-GPU path; `dist.py` is its Python
twin and stays as the test harness.

Communicators (the library's three kinds):
  * `rccl_comm(...)`            an ncclComm_t this process creates itself through librccl (ncclGetUniqueId on rank 0, the 128 id bytes
                                handed to the other ranks by the caller -- bench.py uses torch.distributed's store for that);
  * `NativeTiles.with_callbacks` the caller's own host transport (e.g. torch.distributed over gloo: two processes on ONE GPU, which
                                RCCL refuses);
  * `LocalGroup`                threads of one process (tests).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import VgsError, VgsParams

_HERE = os.path.dirname(os.path.abspath(__file__))
_TL = None
_RCCL = None

T_NAMES = ("grid", "stages", "records", "exchange", "merge", "labels", "total")
COMM_RCCL, COMM_LOCAL, COMM_CALLBACKS = 0, 1, 2
OPT_STRICT_REGION = 1

_AG_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
_BC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int)


class _Callbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", _AG_FN), ("bcast", _BC_FN)]


def lib():
    """libvgs_tiles.so (loads libvgs_hip.so first: it links against it).  No GPU is needed to load it."""
    global _TL
    if _TL is None:
        _lib.lib()
        L = C.CDLL(os.path.join(_HERE, "libvgs_tiles.so"))
        P = C.c_void_p
        L.vgs_tiles_create.restype = C.c_int
        L.vgs_tiles_create.argtypes = [P, C.c_int, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, P]
        L.vgs_tiles_destroy.restype = None
        L.vgs_tiles_destroy.argtypes = [P]
        L.vgs_tiles_last_error_string.restype = C.c_char_p
        L.vgs_tiles_last_error_string.argtypes = [P]
        L.vgs_tiles_context.restype = P
        L.vgs_tiles_context.argtypes = [P]
        L.vgs_tiles_set_option.restype = C.c_int
        L.vgs_tiles_set_option.argtypes = [P, C.c_int32, C.c_int64]
        L.vgs_tiles_set_points.restype = C.c_int
        L.vgs_tiles_set_points.argtypes = [P, P, C.c_int64, C.c_int32]
        L.vgs_tiles_run.restype = C.c_int
        L.vgs_tiles_run.argtypes = [P]
        L.vgs_tiles_get_times.restype = C.c_int
        L.vgs_tiles_get_times.argtypes = [P, P, C.c_int32]
        L.vgs_tiles_get_point_labels.restype = C.c_int
        L.vgs_tiles_get_point_labels.argtypes = [P, P, P]
        L.vgs_tiles_get_info.restype = C.c_int
        L.vgs_tiles_get_info.argtypes = [P, P, P, P]
        L.vgs_tiles_get_exchange.restype = C.c_int
        L.vgs_tiles_get_exchange.argtypes = [P, P, P, P]
        L.vgs_tiles_local_group_create.restype = C.c_int
        L.vgs_tiles_local_group_create.argtypes = [C.c_int, P]
        L.vgs_tiles_local_group_destroy.restype = None
        L.vgs_tiles_local_group_destroy.argtypes = [P]
        L.vgs_tiles_local_group_abort.restype = None
        L.vgs_tiles_local_group_abort.argtypes = [P]
        L.vgs_tiles_merge_boundary.restype = C.c_int
        L.vgs_tiles_merge_boundary.argtypes = [C.c_int, P, P, P, P, P, C.c_int, P, P, P, P, P]
        _TL = L
    return _TL


# ---- RCCL through ctypes: the three calls a caller needs to hand the driver a communicator ---------------------------------------
class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def rccl():
    global _RCCL
    if _RCCL is None:
        R = C.CDLL("librccl.so.1" if os.path.exists("/opt/rocm/lib/librccl.so.1") else "librccl.so")
        R.ncclGetUniqueId.restype = C.c_int
        R.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
        R.ncclCommInitRank.restype = C.c_int
        R.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]   # the id travels BY VALUE
        R.ncclCommCount.restype = C.c_int
        R.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        R.ncclCommDestroy.restype = C.c_int
        R.ncclCommDestroy.argtypes = [C.c_void_p]
        R.ncclCommAbort.restype = C.c_int
        R.ncclCommAbort.argtypes = [C.c_void_p]
        R.ncclGetErrorString.restype = C.c_char_p
        R.ncclGetErrorString.argtypes = [C.c_int]
        _RCCL = R
    return _RCCL


def rccl_unique_id() -> bytes:
    """rank 0: a fresh ncclUniqueId (128 bytes) to hand to the other ranks"""
    uid = _NcclUniqueId()
    st = rccl().ncclGetUniqueId(C.byref(uid))
    if st != 0:
        raise RuntimeError(f"ncclGetUniqueId: {rccl().ncclGetErrorString(st).decode()}")
    return bytes(C.string_at(C.byref(uid), 128))


class RcclComm:
    """An ncclComm_t owned by this process (the HIP device must be the current one: hipSetDevice before the call)."""

    def __init__(self, uid: bytes, rank: int, world: int, device: int):
        hip = C.CDLL("libamdhip64.so")
        if hip.hipSetDevice(int(device)) != 0:
            raise RuntimeError(f"hipSetDevice({device}) failed")
        u = _NcclUniqueId()
        C.memmove(C.byref(u), uid, 128)
        self.handle = C.c_void_p()
        st = rccl().ncclCommInitRank(C.byref(self.handle), int(world), u, int(rank))
        if st != 0:
            raise RuntimeError(f"ncclCommInitRank(rank {rank} of {world}): {rccl().ncclGetErrorString(st).decode()}")
        self.rank, self.world = rank, world

    def count(self) -> int:
        n = C.c_int(0)
        st = rccl().ncclCommCount(self.handle, C.byref(n))
        if st != 0:
            raise RuntimeError(f"ncclCommCount: {rccl().ncclGetErrorString(st).decode()}")
        return n.value

    def destroy(self):
        if self.handle and self.handle.value:
            rccl().ncclCommDestroy(self.handle)
            self.handle = C.c_void_p()

    def abort(self):
        if self.handle and self.handle.value:
            rccl().ncclCommAbort(self.handle)
            self.handle = C.c_void_p()


class LocalGroup:
    """`world` driver threads of one process meeting in shared memory (tests: several ranks on one GPU)."""

    def __init__(self, world):
        self.handle = C.c_void_p()
        st = lib().vgs_tiles_local_group_create(int(world), C.byref(self.handle))
        if st != 0:
            raise VgsError(st, "vgs_tiles_local_group_create")

    def abort(self):
        lib().vgs_tiles_local_group_abort(self.handle)

    def close(self):
        if self.handle and self.handle.value:
            lib().vgs_tiles_local_group_destroy(self.handle)
            self.handle = C.c_void_p()


class NativeTiles:
    """One rank of the native tiled driver."""

    def __init__(self, params: VgsParams, comm_kind: int, comm_handle, rank: int, world: int, tiles, pitch: float, center=(0.0, 0.0), keep=None):
        self._L = lib()
        self._h = C.c_void_p()
        self._keep = keep          # whatever the communicator handle points into (callback thunks, comm objects)
        self.rank, self.world = rank, world
        st = self._L.vgs_tiles_create(C.byref(params), comm_kind, comm_handle, rank, world, int(tiles[0]), int(tiles[1]), float(pitch),
                                      float(center[0]), float(center[1]), C.byref(self._h))
        if st != 0:
            raise VgsError(st, f"vgs_tiles_create: {_lib.lib().vgs_last_error_string(None).decode()}")
        self.n = 0

    @classmethod
    def with_callbacks(cls, params, rank, world, tiles, pitch, all_gather, bcast, center=(0.0, 0.0)):
        """all_gather(send: bytes-like numpy uint8 view, recv: numpy uint8 view of world * n bytes) and bcast(buf: numpy uint8 view,
        root) are the caller's transport over host memory; exceptions they raise become a failed collective (VGS_E_HIP)."""
        def _ag(user, send, recv, nbytes):
            try:
                s = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(int(nbytes),))
                r = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(int(nbytes) * world,))
                all_gather(s, r)
                return 0
            except Exception as ex:  # noqa: BLE001 (the C side turns this into an error status; the message goes to stderr)
                import sys
                print(f"[tiles_native] all_gather callback failed: {ex!r}", file=sys.stderr)
                return 1

        def _bc(user, buf, nbytes, root):
            try:
                b = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(int(nbytes),))
                bcast(b, int(root))
                return 0
            except Exception as ex:  # noqa: BLE001
                import sys
                print(f"[tiles_native] bcast callback failed: {ex!r}", file=sys.stderr)
                return 1

        cb = _Callbacks(None, _AG_FN(_ag), _BC_FN(_bc))
        return cls(params, COMM_CALLBACKS, C.cast(C.pointer(cb), C.c_void_p), rank, world, tiles, pitch, center, keep=cb)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.vgs_tiles_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st):
        if st != 0:
            raise VgsError(st, self._L.vgs_tiles_last_error_string(self._h).decode())

    def set_option(self, option, value):
        self._ck(self._L.vgs_tiles_set_option(self._h, int(option), int(value)))

    def set_points(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        if xyz.ndim != 2 or xyz.shape[1] not in (3, 4):
            raise ValueError("xyz must be (N,3) or (N,4) float32")
        self._xyz = xyz
        self.n = xyz.shape[0]
        self._ck(self._L.vgs_tiles_set_points(self._h, xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], xyz.shape[1] * 4))

    def run(self):
        self._ck(self._L.vgs_tiles_run(self._h))

    def times(self):
        t = np.zeros(len(T_NAMES), dtype=np.float64)
        self._ck(self._L.vgs_tiles_get_times(self._h, t.ctypes.data_as(C.c_void_p), len(T_NAMES)))
        return dict(zip(T_NAMES, (float(x) for x in t)))

    def point_labels(self):
        out = np.zeros(max(self.n, 1), dtype=np.int32)
        kept = C.c_int64(0)
        self._ck(self._L.vgs_tiles_get_point_labels(self._h, out.ctypes.data_as(C.c_void_p), C.byref(kept)))
        return out[:self.n], int(kept.value)

    def info(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._ck(self._L.vgs_tiles_get_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(n_outside=a.value, n_local=b.value, n_boundary_records=c.value)

    def exchange(self):
        """the last run's boundary exchange: bytes this rank sent / received, collectives taken"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        self._ck(self._L.vgs_tiles_get_exchange(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(bytes_sent=a.value, bytes_received=b.value, collectives=c.value)

    def bbox(self):
        from .api import Engine
        return Engine.bbox(_CtxView(self._ctx()))

    # read-only views of the rank's engine context (counts, stage times)
    def _ctx(self):
        return C.c_void_p(self._L.vgs_tiles_context(self._h))

    def counts(self):
        from .api import Engine
        return Engine.counts(_CtxView(self._ctx()))

    def stage_times(self):
        from .api import Engine
        return Engine.stage_times(_CtxView(self._ctx()))


class _CtxView:
    """Just enough of an Engine for its read-only getters, over a context the native driver owns."""

    def __init__(self, h):
        self._L = _lib.lib()
        self._h = h

    def _ck(self, st):
        if st != _lib.VGS_OK:
            raise VgsError(st, self._L.vgs_last_error_string(self._h).decode())
