"""vgs-svgs-segmentation_amd -- MI355X-native VGS / SVGS point-cloud segmentation.

Hand-written HIP kernels (gfx950) behind the C-ABI of include/vgs.h; this package is the thin host
layer: ctypes binding (_lib), the reference's class surface (api), deterministic scenes (scenes) and
the one-process-per-GPU tiling driver (dist).  The directory name contains '-', so import it through
the repo-root shim `vgs_svgs_segmentation_amd`.
"""
from . import _lib, pcd, scenes  # noqa: F401
from ._lib import VgsError, VgsParams, build  # noqa: F401
from .api import (Engine, SuperVoxelBasedSegmentation, VoxelBasedSegmentation, default_params, parse_task_file, pinned_empty,  # noqa: F401
                  segmentation_vgs)
