"""The C++ host mirror (include/vgs_segmentation.hpp) with the reference's own driver call order
(examples/segmentation_vgs.cpp = segmentationVGS, reference `test`:9-86)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vgs-svgs-segmentation_amd", "csrc")
EXE = os.path.join(ROOT, "examples", "segmentation_vgs")


def test_cpp_driver_compiles_against_header():
    subprocess.check_call(["make", "-C", CSRC, "-s", "example"])
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_driver_matches_python_engine(gpu, tmp_path):
    subprocess.check_call(["make", "-C", CSRC, "-s", "example"])
    xyz = gpu.scenes.town_scene(60_000)
    f = tmp_path / "pts.f32"
    xyz.tofile(f)
    out = subprocess.check_output([EXE, str(f)], text=True).split()
    n, voxels, clusters, kept, labelled = (int(x) for x in out)
    eng = gpu.Engine(gpu.default_params(2))
    eng.set_points(xyz)
    eng.run()
    c = eng.counts()
    assert (n, voxels, clusters, kept) == (c["points"], c["voxels"], c["clusters"], c["kept"])
    assert labelled == int((eng.point_labels() >= 0).sum())
