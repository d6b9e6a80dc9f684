"""CPU tests of the native tiled driver (include/vgs_tiles.h, libvgs_tiles.so): the library loads without a GPU and exports
every symbol the header declares, and its boundary merge (C++) gives the tables of the Python twin
(vgs-svgs-segmentation_amd/dist.py: merge_boundary_compact) on random per-rank records."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "vgs-svgs-segmentation_amd", "libvgs_tiles.so")


@pytest.fixture(scope="module")
def tiles_lib(vgs):
    vgs._lib.lib()                       # libvgs_hip.so first (libvgs_tiles.so links against it)
    return C.CDLL(LIB)


def test_library_exports_every_declared_symbol(tiles_lib):
    hdr = open(os.path.join(ROOT, "include", "vgs_tiles.h")).read()
    names = set(re.findall(r"\b(vgs_tiles_\w+)\s*\(", hdr))
    assert len(names) >= 10
    for n in names:
        assert hasattr(tiles_lib, n), n


def _random_records(rng, world, n_codes, voxels_min):
    """per-rank boundary records as vgs_get_boundary_roots gives them: unique codes per rank, several voxels per root"""
    recs, kept_local = [], []
    for r in range(world):
        m = int(rng.integers(0, n_codes))
        code = rng.choice(n_codes, size=m, replace=False).astype(np.uint64)
        roots = rng.integers(0, max(m // 3, 1), size=m).astype(np.int32) * 7 + r
        cnt_of_root = {int(x): int(rng.integers(1, 6)) for x in np.unique(roots)}
        cnt = np.array([cnt_of_root[int(x)] for x in roots], dtype=np.int32)
        recs.append((code, roots, cnt))
        kept_local.append(int(rng.integers(0, 50)))
    return recs, kept_local


@pytest.mark.parametrize("world,seed", [(2, 1), (4, 2), (8, 3), (8, 4), (3, 5)])
def test_boundary_merge_equals_the_python_twin(tiles_lib, vgs, world, seed):
    from vgs_svgs_segmentation_amd.dist import merge_boundary_compact
    rng = np.random.default_rng(seed)
    voxels_min = 3
    recs, kept_local = _random_records(rng, world, 400, voxels_min)
    base_py, lab_py, kept_py = merge_boundary_compact(recs, kept_local, voxels_min)
    off = np.zeros(world + 1, dtype=np.int64)
    off[1:] = np.cumsum([r[0].size for r in recs])
    code = np.concatenate([r[0] for r in recs]) if off[-1] else np.zeros(1, np.uint64)
    root = np.concatenate([r[1] for r in recs]) if off[-1] else np.zeros(1, np.int32)
    cnt = np.concatenate([r[2] for r in recs]) if off[-1] else np.zeros(1, np.int32)
    kl = np.asarray(kept_local, dtype=np.int64)
    base = np.zeros(world, dtype=np.int64)
    uoff = np.zeros(world + 1, dtype=np.int64)
    uroot = np.zeros(max(int(off[-1]), 1), dtype=np.int32)
    ulabel = np.zeros(max(int(off[-1]), 1), dtype=np.int32)
    kept = C.c_int64(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    st = tiles_lib.vgs_tiles_merge_boundary(world, p(off), p(code), p(root), p(cnt), p(kl), voxels_min, p(base), p(uoff), p(uroot), p(ulabel), C.byref(kept))
    assert st == 0
    np.testing.assert_array_equal(base, np.asarray(base_py))
    assert kept.value == kept_py
    for r in range(world):
        np.testing.assert_array_equal(uroot[uoff[r]:uoff[r + 1]], lab_py[r][0])
        np.testing.assert_array_equal(ulabel[uoff[r]:uoff[r + 1]], lab_py[r][1])
