"""PCD reader / writer, coloured-cluster export and task-file reader of include/point_clouds_io.hpp (SURVEY.md 8f rows 1-2;
reference point_clouds_IO.h:64-108, point_clouds_IO.cpp:23-76, 148-169), checked against the independent numpy
implementation in vgs-svgs-segmentation_amd/pcd.py.  Host only: no GPU, no libvgs_hip.so."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vgs-svgs-segmentation_amd", "csrc")
TOOL = os.path.join(ROOT, "examples", "pcd_tool")


@pytest.fixture(scope="module")
def tool():
    subprocess.check_call(["make", "-C", CSRC, "-s", "../../examples/pcd_tool"])
    return TOOL


@pytest.fixture(scope="module")
def pcd():
    import vgs_svgs_segmentation_amd as v
    return v.pcd


def _cloud(n, seed):
    rng = np.random.default_rng(seed)
    xyz = (rng.standard_normal((n, 3)) * np.array([40.0, 25.0, 3.0])).astype(np.float32)
    xyz[::7, 2] = np.float32(1.5)          # repeated values: the LZF stream gets back references
    xyz[n // 2:n // 2 + 50] = xyz[:50]     # a repeated block
    return xyz


def test_lzf_roundtrip(pcd):
    rng = np.random.default_rng(3)
    for data in (b"", b"a", b"abcabcabcabcabcabcabcabcabc" * 40, bytes(rng.integers(0, 4, 5000, dtype=np.uint8)), bytes(rng.integers(0, 256, 3000, dtype=np.uint8)),
                 b"\x00" * 1000):
        comp = pcd.lzf_compress(data)
        assert pcd.lzf_decompress(comp, len(data)) == data
    assert len(pcd.lzf_compress(b"\x00" * 1000)) < 40   # back references are really produced


@pytest.mark.parametrize("mode", ["ascii", "binary", "binary_compressed"])
@pytest.mark.parametrize("out_mode", ["ascii", "binary"])
def test_cpp_reader_and_writer_against_numpy(tool, pcd, tmp_path, mode, out_mode):
    xyz = _cloud(2000, 11)
    src, dst = tmp_path / "in.pcd", tmp_path / "out.pcd"
    # extra fields and a shuffled field order: the reader picks x, y, z by name
    pcd.write_pcd(src, xyz, mode=mode, extra={"intensity": np.arange(2000, dtype=np.float32)}, field_order=["intensity", "z", "x", "y"])
    subprocess.check_call([tool, "convert", str(src), str(dst), out_mode])
    f, hdr = pcd.read_pcd(dst)
    assert hdr["FIELDS"] == ["x", "y", "z"] and hdr["DATA"] == [out_mode] and int(hdr["POINTS"][0]) == 2000
    got = np.stack([f["x"], f["y"], f["z"]], axis=1)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), xyz.view(np.uint32))   # bit exact, %.9g included


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_ply_reader(tool, pcd, tmp_path, fmt):
    """PLY vertices (point_clouds_IO.h:81-95): mixed property types and order, a face element after the vertices."""
    xyz = _cloud(300, 21)
    n = xyz.shape[0]
    inten = np.arange(n, dtype=np.uint8)
    hdr = (f"ply\nformat {fmt} 1.0\ncomment test\nelement vertex {n}\nproperty uchar intensity\nproperty double z\nproperty float x\n"
           "property float y\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n")
    src = tmp_path / "in.ply"
    with open(src, "wb") as f:
        f.write(hdr.encode())
        if fmt == "ascii":
            for k in range(n):
                f.write(f"{int(inten[k])} {float(xyz[k, 2]):.17g} {float(xyz[k, 0]):.9g} {float(xyz[k, 1]):.9g}\n".encode())
            f.write(b"3 0 1 2\n")
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            rec = np.zeros(n, dtype=[("i", "u1"), ("z", e + "f8"), ("x", e + "f4"), ("y", e + "f4")])
            rec["i"], rec["z"], rec["x"], rec["y"] = inten, xyz[:, 2].astype(np.float64), xyz[:, 0], xyz[:, 1]
            f.write(rec.tobytes())
            f.write(bytes([3]) + np.array([0, 1, 2], dtype=e + "i4").tobytes())
    dst = tmp_path / "out.pcd"
    subprocess.check_call([tool, "convert", str(src), str(dst), "binary"])
    got, _ = pcd.read_pcd(dst)
    out = np.stack([got["x"], got["y"], got["z"]], axis=1)
    assert np.array_equal(out.view(np.uint32), xyz.view(np.uint32))


def test_reader_rejects_bad_files(tool, pcd, tmp_path):
    bad = tmp_path / "bad.pcd"
    bad.write_text("FIELDS x y\nSIZE 4 4\nTYPE F F\nCOUNT 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2\n")
    assert subprocess.call([tool, "convert", str(bad), str(tmp_path / "o.pcd")], stderr=subprocess.DEVNULL) == 1   # no z field
    assert subprocess.call([tool, "convert", str(tmp_path / "missing.pcd"), str(tmp_path / "o.pcd")], stderr=subprocess.DEVNULL) == 1
    xyz = _cloud(100, 2)
    trunc = tmp_path / "trunc.pcd"
    pcd.write_pcd(trunc, xyz, mode="binary")
    blob = trunc.read_bytes()
    trunc.write_bytes(blob[:-40])
    assert subprocess.call([tool, "convert", str(trunc), str(tmp_path / "o.pcd")], stderr=subprocess.DEVNULL) == 1
    comp = tmp_path / "comp.pcd"
    pcd.write_pcd(comp, xyz, mode="binary_compressed")
    blob = bytearray(comp.read_bytes())
    blob[-20] ^= 0xE0   # corrupt a control byte region
    comp.write_bytes(bytes(blob))
    assert subprocess.call([tool, "convert", str(comp), str(tmp_path / "o.pcd")], stderr=subprocess.DEVNULL) in (0, 1)   # must not crash


@pytest.mark.parametrize("data", ["binary", "binary_compressed", "ascii"])
@pytest.mark.parametrize("count", ["-1 1 1 1", "0 1 1 1", "1000000000 1 1 1", "70000 1 1 1"])
def test_reader_rejects_hostile_count(tool, tmp_path, data, count):
    """COUNT < 1 (negative field offsets) or huge COUNT (overflowed point size) must be a clean error, never a crash."""
    import struct
    bad = tmp_path / "count.pcd"
    head = f"FIELDS pad x y z\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT {count}\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA {data}\n".encode()
    body = struct.pack("<8f", *range(8)) if data != "ascii" else b"0 1 2 3\n4 5 6 7\n"
    if data == "binary_compressed":
        body = struct.pack("<II", 32, 32) + body
    bad.write_bytes(head + body)
    assert subprocess.call([tool, "convert", str(bad), str(tmp_path / "o.pcd")], stderr=subprocess.DEVNULL) == 1


def test_reader_rejects_oversized_compressed_size_word(tool, pcd, tmp_path):
    """binary_compressed: a compressed-size word beyond the end of the file must not become a 4 GB allocation."""
    import struct
    xyz = _cloud(100, 3)
    comp = tmp_path / "comp.pcd"
    pcd.write_pcd(comp, xyz, mode="binary_compressed")
    blob = bytearray(comp.read_bytes())
    at = blob.index(b"DATA binary_compressed\n") + len(b"DATA binary_compressed\n")
    blob[at:at + 4] = struct.pack("<I", 0xFFFFFFF0)
    comp.write_bytes(bytes(blob))
    assert subprocess.call([tool, "convert", str(comp), str(tmp_path / "o.pcd")], stderr=subprocess.DEVNULL) == 1
    # a POINTS value far beyond the file
    big = tmp_path / "big.pcd"
    big.write_bytes(b"FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4000000000\nHEIGHT 1\nPOINTS 4000000000\nDATA binary_compressed\n"
                    + struct.pack("<II", 8, 0) + b"\0" * 8)
    assert subprocess.call([tool, "convert", str(big), str(tmp_path / "o.pcd")], stderr=subprocess.DEVNULL) == 1


def test_coloured_cluster_export(tool, pcd, tmp_path):
    xyz = _cloud(500, 5)
    src = tmp_path / "in.pcd"
    pcd.write_pcd(src, xyz, mode="binary")
    clusters = [[5, 3, 9, 100], [7], [499, 0, 250, 251, 252]]
    (tmp_path / "c.txt").write_text("\n".join(" ".join(str(i) for i in c) for c in clusters) + "\n")
    outs = []
    for name, seed in (("a.pcd", 7), ("b.pcd", 7), ("c.pcd", 8)):
        subprocess.check_call([tool, "colour", str(src), str(tmp_path / "c.txt"), str(tmp_path / name), str(seed)])
        outs.append(pcd.read_pcd(tmp_path / name)[0])
    a, b, c = outs
    order = [i for cl in clusters for i in cl]                 # cluster by cluster, member by member (point_clouds_IO.cpp:39-61)
    got = np.stack([a["x"], a["y"], a["z"]], axis=1)
    assert np.array_equal(got.view(np.uint32), xyz[order].view(np.uint32))
    rgb = a["rgb"].view(np.uint32)
    assert (rgb >> 24).max() == 0                              # 0x00RRGGBB
    bounds = np.cumsum([0] + [len(cl) for cl in clusters])
    per_cluster = [set(rgb[s:e].tolist()) for s, e in zip(bounds[:-1], bounds[1:])]
    assert all(len(s) == 1 for s in per_cluster)               # one colour per cluster
    assert len(set.union(*per_cluster)) == 3                   # three clusters, three colours
    assert np.array_equal(rgb, b["rgb"].view(np.uint32))       # seeded: reproducible (the reference uses srand(time(0)))
    assert not np.array_equal(rgb, c["rgb"].view(np.uint32))   # another seed, another palette


def test_task_file_reader_strips_cr(tool, tmp_path):
    lines = [f"line {k}" for k in range(61)]
    lines[24], lines[15], lines[21] = "3", "Town_Test.pcd", "Town_Test_SVGS.pcd"
    f = tmp_path / "task.txt"
    f.write_bytes("\r\n".join(lines).encode())                 # the shipped task files have CRLF line ends
    out = subprocess.check_output([tool, "task", str(f)], text=True).split()
    assert out == ["61", "3", "Town_Test.pcd", "Town_Test_SVGS.pcd"]


@pytest.mark.parametrize("name,n,res", [("TOWN", 60_000, 0.15), ("PC1M", 80_000, 0.05), ("URB10M", 120_000, 0.1)])
def test_cpp_scene_generator(tool, pcd, tmp_path, name, n, res):
    """include/vgs_scenes.hpp (the scenes as self-contained C++, so that the C++ examples need no Python): deterministic,
    the requested size, the viewpoint (0, 0, 1.5) inside it, the density per voxel face of the numpy generator the bench uses
    (same geometry and layout rules, its own random stream), and shuffled (the octree's origin depends on the first point)."""
    import vgs_svgs_segmentation_amd as v
    a, b = tmp_path / "a.pcd", tmp_path / "b.pcd"
    subprocess.check_call([tool, "scene", name, str(n), str(a)])
    subprocess.check_call([tool, "scene", name, str(n), str(b)])
    assert a.read_bytes() == b.read_bytes()
    fields, _ = pcd.read_pcd(a)
    xyz = np.stack([fields["x"], fields["y"], fields["z"]], axis=1)
    assert xyz.shape == (n, 3) and xyz.dtype == np.float32 and np.isfinite(xyz).all()
    lo, hi = xyz.min(axis=0), xyz.max(axis=0)
    assert (lo[:2] < 0).all() and (hi[:2] > 0).all() and lo[2] < 0.1   # (the viewpoint (0, 0, 1.5) is above the ground of every scene; scaled-down scenes are lower than it)
    ref = {"TOWN": v.scenes.town_scene, "PC1M": v.scenes.pc_scene, "URB10M": v.scenes.urban_scene}[name](n)
    # same extents (the layouts draw their building heights from different streams: x / y only), same occupied-voxel count within 15 %
    tol = max(0.05, 0.08 * float((hi - lo)[:2].max()))   # (tree crowns and poles are placed by the layout stream: they may stick out of the ground)
    np.testing.assert_allclose(lo[:2], ref.min(axis=0)[:2], atol=tol)
    np.testing.assert_allclose(hi[:2], ref.max(axis=0)[:2], atol=tol)
    vox = lambda p: len(np.unique(np.floor(p / res).astype(np.int64), axis=0))
    assert abs(vox(xyz) - vox(ref)) < 0.15 * vox(ref), (vox(xyz), vox(ref))
    # shuffled: consecutive points are not neighbours
    assert np.median(np.linalg.norm(np.diff(xyz[:2000], axis=0), axis=1)) > 5 * res
