"""csrc/regsort.hpp on its own: the register bitonic network (one, two, four keys per lane, two halves of 256, eight keys per lane)
against std::sort on 20 000 lists of 0 .. 448 keys (and, glued together, of up to 2048 keys for the workgroup form) -- zeros (dropped entries) and equal weights included -- and against the LDS
network it replaced.  The program is tools/regsort_test.hip, built by __graft_entry__.build() (make regsort_test)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "regsort_test")


@pytest.mark.parametrize("mean", [30, 100, 224])
def test_register_network_sorts_like_std_sort(mean):
    if not os.path.exists(EXE):
        pytest.skip("tools/regsort_test is not built (make -C vgs-svgs-segmentation_amd/csrc regsort_test)")
    out = subprocess.run([EXE, "20000", str(mean)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("(wrong keys 0)") == 5, out.stdout   # block form, one-word keys (round 6: ties, dropped entries, out-of-window lists), LDS network, eight keys per lane, two halves
