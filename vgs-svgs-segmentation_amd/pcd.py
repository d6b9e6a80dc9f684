"""PCD files in numpy: the on-disk format either side of the segmentation path.

Independent of include/point_clouds_io.hpp (the C++ reader/writer used by examples/vgs_run); the tests run each
against the other.  Supports DATA ascii | binary | binary_compressed, any field list; `read_pcd` returns the fields
as a dict of arrays, `write_pcd` writes float32 x y z (+ optional packed rgb, + optional extra float fields).
"""
import numpy as np

_NP = {("F", 4): np.float32, ("F", 8): np.float64, ("I", 1): np.int8, ("I", 2): np.int16, ("I", 4): np.int32, ("I", 8): np.int64,
       ("U", 1): np.uint8, ("U", 2): np.uint16, ("U", 4): np.uint32, ("U", 8): np.uint64}


def lzf_compress(data: bytes) -> bytes:
    """Greedy LZF encoder (format of liblzf): literal runs of <= 32 bytes, back references of 3..264 bytes within 8191."""
    n = len(data)
    out = bytearray()
    lit = bytearray()
    table = {}
    i = 0

    def flush():
        for k in range(0, len(lit), 32):
            chunk = lit[k:k + 32]
            out.append(len(chunk) - 1)
            out.extend(chunk)
        lit.clear()

    while i < n:
        ref = -1
        if i + 2 < n:
            key = data[i:i + 3]
            ref = table.get(key, -1)
            table[key] = i
        if ref >= 0 and 0 < i - ref <= 8192:
            length = 3
            while i + length < n and length < 264 and data[ref + length] == data[i + length]:
                length += 1
            flush()
            dist = i - ref - 1
            ln = length - 2
            if ln < 7:
                out.append((ln << 5) | (dist >> 8))
            else:
                out.append((7 << 5) | (dist >> 8))
                out.append(ln - 7)
            out.append(dist & 0xFF)
            i += length
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def lzf_decompress(data: bytes, out_len: int) -> bytes:
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        ctrl = data[i]; i += 1
        if ctrl < 32:
            out.extend(data[i:i + ctrl + 1]); i += ctrl + 1
        else:
            ln = ctrl >> 5
            if ln == 7:
                ln += data[i]; i += 1
            dist = (((ctrl & 31) << 8) | data[i]) + 1; i += 1
            for _ in range(ln + 2):
                out.append(out[-dist])
    if len(out) != out_len:
        raise ValueError("LZF stream does not decode to the announced size")
    return bytes(out)


def write_pcd(path, xyz, mode="binary", rgb=None, extra=None, field_order=None):
    """xyz (N,3) float32; rgb (N,) uint32 0x00RRGGBB written as PCL's packed float field; extra: dict name -> (N,) float32."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    cols = {"x": xyz[:, 0], "y": xyz[:, 1], "z": xyz[:, 2]}
    if rgb is not None:
        cols["rgb"] = np.asarray(rgb, dtype=np.uint32).view(np.float32)
    for k, v in (extra or {}).items():
        cols[k] = np.asarray(v, dtype=np.float32)
    names = field_order or list(cols)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS " + " ".join(names) + "\nSIZE " + " ".join("4" for _ in names) +
           "\nTYPE " + " ".join("F" for _ in names) + "\nCOUNT " + " ".join("1" for _ in names) +
           f"\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {mode}\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if mode == "ascii":
            arr = np.stack([cols[k] for k in names], axis=1)
            for row in arr:
                f.write((" ".join(f"{float(v):.9g}" for v in row) + "\n").encode())
        elif mode == "binary":
            f.write(np.stack([cols[k] for k in names], axis=1).astype(np.float32).tobytes())
        elif mode == "binary_compressed":
            raw = b"".join(np.ascontiguousarray(cols[k], dtype=np.float32).tobytes() for k in names)  # field by field
            comp = lzf_compress(raw)
            f.write(np.array([len(comp), len(raw)], dtype=np.uint32).tobytes())
            f.write(comp)
        else:
            raise ValueError(mode)


def read_pcd(path):
    """Returns (fields: dict name -> array, header: dict)."""
    with open(path, "rb") as f:
        blob = f.read()
    pos = 0
    hdr = {}
    while True:
        end = blob.index(b"\n", pos)
        line = blob[pos:end].decode().strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, *vals = line.split()
        hdr[key] = vals
        if key == "DATA":
            break
    names = hdr["FIELDS"]
    sizes = [int(s) for s in hdr["SIZE"]]
    types = hdr["TYPE"]
    counts = [int(c) for c in hdr.get("COUNT", ["1"] * len(names))]
    n = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr["HEIGHT"][0])
    mode = hdr["DATA"][0]
    dt = np.dtype([(nm, _NP[(t, s)], (c,)) if c > 1 else (nm, _NP[(t, s)]) for nm, s, t, c in zip(names, sizes, types, counts)])
    if mode == "ascii":
        rows = [ln.split() for ln in blob[pos:].decode().splitlines() if ln.strip()]
        if len(rows) != n:
            raise ValueError("ascii body length")
        out = {}
        col = 0
        for nm, s, t, c in zip(names, sizes, types, counts):
            a = np.array([[float(r[col + k]) for k in range(c)] for r in rows], dtype=np.float64)
            out[nm] = a[:, 0].astype(_NP[(t, s)]) if c == 1 else a.astype(_NP[(t, s)])
            col += c
        return out, hdr
    if mode == "binary":
        rec = np.frombuffer(blob, dtype=dt, count=n, offset=pos)
        return {nm: rec[nm].copy() for nm in names}, hdr
    if mode == "binary_compressed":
        csize, usize = np.frombuffer(blob, dtype=np.uint32, count=2, offset=pos)
        raw = lzf_decompress(blob[pos + 8:pos + 8 + int(csize)], int(usize))
        out, off = {}, 0
        for nm, s, t, c in zip(names, sizes, types, counts):
            a = np.frombuffer(raw, dtype=_NP[(t, s)], count=n * c, offset=off)
            out[nm] = a.copy() if c == 1 else a.reshape(n, c).copy()
            off += n * c * s
        return out, hdr
    raise ValueError(mode)
