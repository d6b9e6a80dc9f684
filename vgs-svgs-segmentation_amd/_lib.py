"""ctypes binding of the C-ABI in include/vgs.h (libvgs_hip.so).

The product path: there is no Python/numpy/torch fallback anywhere in this package.  If the HIP
library is missing or no GPU is present every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# VGS_LIB names another build of the same library next to this file (the -DVGS_PROF diagnostics build, `make prof`)
LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ.get("VGS_LIB", "libvgs_hip.so")))
CSRC = os.path.join(_HERE, "csrc")

VGS_OK, VGS_E_ARG, VGS_E_STATE, VGS_E_HIP, VGS_E_NOMEM, VGS_E_UNSUPPORTED, VGS_E_IO, VGS_E_PEER = range(8)
STATUS_NAMES = ["VGS_OK", "VGS_E_ARG", "VGS_E_STATE", "VGS_E_HIP", "VGS_E_NOMEM", "VGS_E_UNSUPPORTED", "VGS_E_IO", "VGS_E_PEER"]

# indices of vgs_get_counts / vgs_get_stage_times (include/vgs.h)
N_POINTS, N_FINITE, N_VOXELS, N_USED, N_ADJ, N_CLUSTERS, N_KEPT, N_PAIRS, N_DEPTH, N_ISOLATED, N_REATTACHED, N_SUPERVOXELS = range(12)
N_COUNTS = 16
T_VOXELIZE, T_FEATURES, T_ADJACENCY, T_LOCALCUT, T_MERGE, T_LABELS, T_TOTAL, T_LOCALCUT_KERNEL, T_SUPERVOXEL = range(9)
T_COUNT = 12


class VgsParams(C.Structure):
    _fields_ = [
        ("method", C.c_int32), ("voxel_size", C.c_float), ("graph_size", C.c_float),
        ("sig_p", C.c_float), ("sig_n", C.c_float), ("sig_o", C.c_float), ("sig_e", C.c_float), ("sig_c", C.c_float),
        ("sig_w", C.c_float), ("cut_thred", C.c_float),
        ("points_min", C.c_int32), ("adjacency_min", C.c_int32), ("voxels_min", C.c_int32),
        ("seed_size", C.c_float), ("color_impt", C.c_float), ("spatial_impt", C.c_float), ("normal_impt", C.c_float),
        ("q7_count_as_index", C.c_int32), ("device", C.c_int32), ("vccs_mode", C.c_int32),
    ]


class VgsGridState(C.Structure):
    _fields_ = [("min", C.c_double * 3), ("shift", C.c_uint64 * 3), ("depth", C.c_int32), ("defined", C.c_int32)]


class VgsError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS_NAMES[status] if 0 <= status < len(STATUS_NAMES) else status}: {msg}")
        self.status = status


# every symbol include/vgs.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("vgs_params_default_vgs", C.c_int, [C.POINTER(VgsParams)]),
    ("vgs_params_default_svgs", C.c_int, [C.POINTER(VgsParams)]),
    ("vgs_parse_task_file", C.c_int, [C.c_char_p, C.POINTER(VgsParams), C.c_char_p, C.c_char_p, C.c_int]),
    ("vgs_create", C.c_int, [C.POINTER(VgsParams), C.POINTER(_P)]),
    ("vgs_destroy", None, [_P]),
    ("vgs_set_params", C.c_int, [_P, C.POINTER(VgsParams)]),
    ("vgs_last_error_string", C.c_char_p, [_P]),
    ("vgs_set_points", C.c_int, [_P, _P, C.c_int64, C.c_int32]),
    ("vgs_set_points_device", C.c_int, [_P, _P, C.c_int64, C.c_int32]),
    ("vgs_stage_points", C.c_int, [_P, _P, C.c_int64, C.c_int32]),
    ("vgs_commit_points", C.c_int, [_P]),
    ("vgs_host_alloc", C.c_int, [C.POINTER(_P), C.c_uint64]),
    ("vgs_host_free", C.c_int, [_P]),
    ("vgs_host_register", C.c_int, [_P, C.c_uint64]),
    ("vgs_host_unregister", C.c_int, [_P]),
    ("vgs_voxelize", C.c_int, [_P]),
    ("vgs_features", C.c_int, [_P]),
    ("vgs_adjacency", C.c_int, [_P]),
    ("vgs_segment", C.c_int, [_P]),
    ("vgs_run", C.c_int, [_P]),
    ("svgs_set_supervoxel_labels", C.c_int, [_P, _P, C.c_int32]),
    ("svgs_supervoxels", C.c_int, [_P]),
    ("svgs_segment", C.c_int, [_P]),
    ("svgs_get_supervoxel_labels", C.c_int, [_P, _P, C.POINTER(C.c_int32)]),
    ("vgs_get_counts", C.c_int, [_P, _P]),
    ("vgs_get_schedule_counters", C.c_int, [_P, _P]),
    ("vgs_get_schedule_counters_ex", C.c_int, [_P, _P, C.c_int32]),
    ("vgs_screen_table", C.c_int, [_P, _P, _P, _P]),
    ("vgs_get_stage_times", C.c_int, [_P, _P]),
    ("vgs_get_bbox", C.c_int, [_P, _P]),
    ("vgs_get_voxel_table", C.c_int, [_P, _P, _P, _P]),
    ("vgs_get_voxel_centers", C.c_int, [_P, _P]),
    ("vgs_get_point_voxel", C.c_int, [_P, _P]),
    ("vgs_get_attributes", C.c_int, [_P, _P, _P, _P, _P]),
    ("vgs_get_lists", C.c_int, [_P, C.c_int32, _P, _P]),
    ("vgs_get_lists_ordered", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    ("vgs_get_adjacency_counts", C.c_int, [_P, _P]),
    ("vgs_get_local_weights", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), _P, _P]),
    ("vgs_get_node_labels", C.c_int, [_P, _P, _P]),
    ("vgs_get_point_labels", C.c_int, [_P, _P]),
    ("vgs_get_point_labels_async", C.c_int, [_P, _P]),
    ("vgs_wait_point_labels", C.c_int, [_P]),
    ("vgs_get_point_labels_device", C.c_int, [_P, C.POINTER(_P)]),
    ("vgs_get_clusters", C.c_int, [_P, _P, _P]),
    ("vgs_get_clusters_ordered", C.c_int, [_P, C.c_int32, _P, _P]),
    ("vgs_get_clusters_device", C.c_int, [_P, _P, _P]),
    ("vgs_grid_state_init", C.c_int, [C.POINTER(VgsGridState)]),
    ("vgs_grid_advance", C.c_int, [_P, C.POINTER(VgsGridState)]),
    ("vgs_points_bbox", C.c_int, [_P, _P, C.POINTER(C.c_int64)]),
    ("vgs_grid_advance_bbox", C.c_int, [C.POINTER(VgsGridState), C.c_double, _P, C.POINTER(C.c_int32)]),
    ("vgs_set_grid", C.c_int, [_P, C.POINTER(VgsGridState)]),
    ("vgs_set_grid_covering", C.c_int, [_P, C.POINTER(VgsGridState)]),
    ("vgs_set_owned_region", C.c_int, [_P, _P, _P]),
    ("vgs_set_own_point_range", C.c_int, [_P, C.c_int64, C.c_int64]),
    ("vgs_get_boundary", C.c_int, [_P, _P, _P, _P]),
    ("vgs_get_owned_roots", C.c_int, [_P, _P, _P, _P]),
    ("vgs_apply_root_labels", C.c_int, [_P, _P, _P, C.c_int64]),
    ("vgs_get_boundary_roots", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("vgs_apply_tile_labels", C.c_int, [_P, C.c_int32, _P, _P, C.c_int64]),
]

_LIB = None


def build(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "vgs.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB
