"""Register and scratch budget of the hot kernels, read from the compiler (hipcc cross-compiles gfx950 without a GPU): the one-wavefront
classes of the local cut must keep 80 registers (six wavefronts per SIMD) WITHOUT scratch -- a change that costs them a spilled register
costs the bulk launch 0.03 ms and 70 MB of scratch writes per step, silently (round 4 found one that way) -- and the supervoxel round
kernels must not spill either."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vgs-svgs-segmentation_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _usage(src, tmp_path):
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-ffp-contract=off", "-fno-fast-math",
                          "-I", os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, src), "-o", str(tmp_path / "dev.o"),
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=CSRC)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]+\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_one_wavefront_local_cut_keeps_its_registers_without_scratch(tmp_path):
    k = _usage("localcut.hip", tmp_path)
    hot = {n: v for n, v in k.items() if "k_localcut_waveILi96ELi448ELi1E" in n or "k_localcut_waveILi128ELi312ELi1E" in n}
    # <96,448,1,false> (bulk), <128,312,1,false> (class B) and <128,312,1,true> (round 5: the one-in-sixteen sample that books how the
    # lazy schedule fares on the scene -- its bookkeeping must not cost the others a register, which is why it is an instantiation of its own)
    # ... each twice since round 6: sorting one-word keys (the fifth template argument true; launched when 1 - cut >= 0.5) and with the 64-bit network
    assert len(hot) == 6, sorted(k)
    for name, u in hot.items():
        sampled = "ELi1ELb1ELb" in name   # the sample pays for its counters with three spilled registers (12 B per lane): one voxel in sixteen, off the bulk's stream
        wide_sort = name.split("PKj")[0].endswith("ELb0EEv")        # the 64-bit network: launched for cut > 0.5 only (and as the A/B twin, VGS_NO_SORT32)
        assert u["ScratchSize"] <= (16 if (sampled or wide_sort) else 0), (name, u)
        assert u["VGPRs Spill"] <= (6 if sampled else (4 if wide_sort else 0)), (name, u)
        assert u["VGPRs"] <= 80, (name, u)
        assert u["Occupancy"] >= 6, (name, u)
    # class C0: six workgroups of 26 KB per CU need six wavefronts per SIMD as well
    c0 = [v for n, v in k.items() if "k_localcut_waveILi320ELi2032ELi4E" in n]
    assert len(c0) == 1 and c0[0]["VGPRs"] <= 80 and c0[0]["VGPRs Spill"] == 0 and c0[0]["Occupancy"] >= 6, c0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_supervoxel_round_kernels_do_not_spill(tmp_path):
    k = _usage("vccs.hip", tmp_path)
    # ... and the two-ring normals over the tiles (round 6: k_pclt_normals<false / true>, six calls of each per step)
    hot = {n: v for n, v in k.items() if "k_vccs_expand_tiles" in n or "k_pclt_sweep" in n or "k_pclt_normals" in n}
    assert len(hot) == 5, sorted(k)
    for name, u in hot.items():
        assert u["ScratchSize"] == 0, (name, u)
