"""Deterministic synthetic scenes (SURVEY.md E / BASELINE.md section 2).

The reference ships no data (`Town_Test.pcd` is only named in README.md:20), so every workload is
generated here: surfaces sampled uniformly by area, Gaussian off-surface noise sigma = 3 mm, one
global shuffle (insertion order decides the octree origin), float32 output.
numpy's Philox bit generator keyed by the seed makes the clouds bit-identical on every box that runs
this image.  Scenes scale with the requested point count so that the point density per voxel face
stays at the nominal value (about 21-25 points per voxel-sized cell).
"""
from __future__ import annotations

import math

import numpy as np

NOISE_SIGMA = 0.003

SEEDS = {"TOWN": 20260101, "PC1M": 20260102, "URB10M": 20260103, "URB80M": 20260110}


class _Prims:
    def __init__(self):
        self.items = []  # (kind, area, params)

    def rect(self, origin, e1, e2):
        o, a, b = (np.asarray(v, dtype=np.float64) for v in (origin, e1, e2))
        n = np.cross(a, b)
        area = float(np.linalg.norm(n))
        self.items.append(("rect", area, (o, a, b, n / area)))

    def cylinder(self, cx, cy, r, z0, z1):
        self.items.append(("cyl", 2.0 * math.pi * r * (z1 - z0), (cx, cy, r, z0, z1)))

    def ball(self, c, r, budget_area=None):
        area = 4.0 * math.pi * r * r if budget_area is None else budget_area
        self.items.append(("ball", area, (np.asarray(c, dtype=np.float64), r)))

    def box(self, x0, y0, x1, y1, h, roof="flat"):
        # four facades + roof; no floor
        self.rect((x0, y0, 0), (x1 - x0, 0, 0), (0, 0, h))
        self.rect((x0, y1, 0), (x1 - x0, 0, 0), (0, 0, h))
        self.rect((x0, y0, 0), (0, y1 - y0, 0), (0, 0, h))
        self.rect((x1, y0, 0), (0, y1 - y0, 0), (0, 0, h))
        if roof == "flat":
            self.rect((x0, y0, h), (x1 - x0, 0, 0), (0, y1 - y0, 0))
        else:  # pitched: two slopes meeting at the ridge along x
            ym = 0.5 * (y0 + y1)
            rise = 0.35 * (y1 - y0)
            self.rect((x0, y0, h), (x1 - x0, 0, 0), (0, ym - y0, rise))
            self.rect((x0, y1, h), (x1 - x0, 0, 0), (0, ym - y1, rise))

    def total_area(self):
        return sum(a for _, a, _ in self.items)

    def sample(self, n, rng):
        areas = np.array([a for _, a, _ in self.items], dtype=np.float64)
        # deterministic proportional allocation (largest remainder)
        raw = areas / areas.sum() * n
        cnt = np.floor(raw).astype(np.int64)
        rem = n - int(cnt.sum())
        order = np.argsort(-(raw - cnt), kind="stable")
        cnt[order[:rem]] += 1
        out = np.empty((n, 3), dtype=np.float64)
        pos = 0
        for (kind, _, prm), m in zip(self.items, cnt):
            m = int(m)
            if m == 0:
                continue
            if kind == "rect":
                o, a, b, nrm = prm
                u = rng.random(m)
                v = rng.random(m)
                d = rng.standard_normal(m) * NOISE_SIGMA
                out[pos:pos + m] = o + u[:, None] * a + v[:, None] * b + d[:, None] * nrm
            elif kind == "cyl":
                cx, cy, r, z0, z1 = prm
                th = rng.random(m) * (2.0 * math.pi)
                z = z0 + rng.random(m) * (z1 - z0)
                rr = r + rng.standard_normal(m) * NOISE_SIGMA
                out[pos:pos + m, 0] = cx + rr * np.cos(th)
                out[pos:pos + m, 1] = cy + rr * np.sin(th)
                out[pos:pos + m, 2] = z
            else:  # ball volume
                c, r = prm
                v = rng.standard_normal((m, 3))
                v /= np.linalg.norm(v, axis=1)[:, None]
                rad = r * np.cbrt(rng.random(m))
                out[pos:pos + m] = c + v * rad[:, None]
            pos += m
        assert pos == n
        perm = rng.permutation(n)
        return out[perm].astype(np.float32)


def _rng(seed):
    return np.random.Generator(np.random.Philox(seed))


def pc_scene(n=1_000_000, seed=SEEDS["PC1M"]):
    """BASELINE config 2 "PC1M": ground + wall + vertical cylinder.  Nominal n = 1e6 gives about
    25.7 points per (0.05 m)^2; for other n the scene is scaled to keep that density."""
    s = math.sqrt(n / 1_000_000)
    P = _Prims()
    P.rect((-4 * s, -4 * s, 0), (8 * s, 0, 0), (0, 8 * s, 0))
    P.rect((-4 * s, 4 * s, 0), (8 * s, 0, 0), (0, 0, 3 * s))
    P.cylinder(2 * s, -2 * s, 0.5 * s, 0.0, 3 * s)
    return P.sample(n, _rng(seed))


def urban_scene(n=10_000_000, seed=SEEDS["URB10M"], center=(0.0, 0.0), nominal=10_000_000):
    """BASELINE config 3/4 "URB10M": ground, box buildings (facades + flat roofs), poles, trees.
    All horizontal extents scale with sqrt(n/nominal) so that the density stays at about 21 points per
    (0.1 m)^2 face; the layout (which lot holds what) is drawn from the seed."""
    s = math.sqrt(n / nominal)
    rng = _rng(seed)
    cx, cy = center
    L = 50.0 * s
    P = _Prims()
    lots = 4
    pitch = L / lots
    # buildings on a checkerboard of lots, ground everywhere (a real scan has no points under buildings,
    # but a full ground keeps (0,0,1.5) inside the scene for every n)
    P.rect((cx - L / 2, cy - L / 2, 0), (L, 0, 0), (0, L, 0))
    lay = _rng(seed + 1)
    for i in range(lots):
        for j in range(lots):
            x0 = cx - L / 2 + i * pitch
            y0 = cy - L / 2 + j * pitch
            if (i + j) % 2 == 0:
                w = pitch * (0.55 + 0.15 * lay.random())
                d = pitch * (0.40 + 0.15 * lay.random())
                h = (4.0 + 5.0 * lay.random()) * max(s, 0.35)
                bx = x0 + 0.5 * (pitch - w)
                by = y0 + 0.5 * (pitch - d)
                P.box(bx, by, bx + w, by + d, h)
            else:
                for _ in range(5):
                    px = x0 + pitch * (0.1 + 0.8 * lay.random())
                    py = y0 + pitch * (0.1 + 0.8 * lay.random())
                    P.cylinder(px, py, 0.15 * max(s, 0.5), 0.0, 5.0 * max(s, 0.4))
                for _ in range(2):
                    tr = 1.5 * max(s, 0.4)
                    tx = x0 + pitch * (0.2 + 0.6 * lay.random())
                    ty = y0 + pitch * (0.2 + 0.6 * lay.random())
                    P.ball((tx, ty, 2.0 * tr + 0.5), tr)
    return P.sample(n, rng)


def town_scene(n=500_000, seed=SEEDS["TOWN"]):
    """BASELINE config 1 stand-in "TOWN" (the real Town_Test.pcd is not available): ground, four
    buildings with pitched roofs, ten poles; about 25 points per (0.15 m)^2 at the nominal n."""
    s = math.sqrt(n / 500_000)
    P = _Prims()
    L = 14.0 * s
    P.rect((-L / 2, -L / 2, 0), (L, 0, 0), (0, L, 0))
    lay = _rng(seed + 1)
    for (qx, qy) in ((-1, -1), (1, -1), (-1, 1), (1, 1)):
        w, d = 3.2 * s, 2.4 * s
        h = (2.0 + 1.5 * lay.random()) * max(s, 0.4)
        bx = qx * L / 4 - w / 2
        by = qy * L / 4 - d / 2
        P.box(bx, by, bx + w, by + d, h, roof="pitched")
    for _ in range(10):
        px = (lay.random() - 0.5) * 0.9 * L
        py = (lay.random() - 0.5) * 0.9 * L
        P.cylinder(px, py, 0.1 * max(s, 0.5), 0.0, 4.0 * max(s, 0.4))
    return P.sample(n, _rng(seed))


def tiled_urban_scene(n_total=80_000_000, tiles=(4, 2), seed0=SEEDS["URB80M"], tile_index=None):
    """BASELINE config 5 "URB80M": a tiles[0] x tiles[1] grid of URB10M-like tiles (seeds seed0+k),
    centred on the origin.  With tile_index = k only that tile's points are generated (what one rank
    of the spatially sharded run loads)."""
    tx, ty = tiles
    nt = tx * ty
    per = n_total // nt
    s = math.sqrt(per / 10_000_000)
    pitch = 50.0 * s
    out = []
    for k in range(nt):
        if tile_index is not None and k != tile_index:
            continue
        i, j = k % tx, k // tx
        cx = (i - (tx - 1) / 2.0) * pitch
        cy = (j - (ty - 1) / 2.0) * pitch
        out.append(urban_scene(per, seed0 + k, center=(cx, cy)))
    return out[0] if tile_index is not None else np.concatenate(out, axis=0)


def noisy_surface_scene(n=5_000_000, seed=1, sigma=0.03):
    """"c3n": an undulating surface under centimetres of range noise (6000 points per m^2, sigma = 3 cm by default) -- the
    regime real terrestrial scans live in and the synthetic BASELINE scenes (sigma = 3 mm) do not reach: no segment of a
    neighbourhood forms early and freezes, so almost every voxel needs all of its heavy pairs.  (numpy's default PCG64
    stream: the scene DESIGN.md quoted since round 2, tools/fuzzy_bench.py.)"""
    rng = np.random.default_rng(seed)
    side = np.sqrt(n / 6000.0)
    x, y = rng.random(n) * side, rng.random(n) * side
    z = 0.3 * np.sin(2.0 * x) * np.cos(1.5 * y) + rng.normal(0, sigma, n) + 2.0
    return np.stack([x + 0.011, y + 0.017, z], axis=1).astype(np.float32)


def solid_block_scene(n=500_000, side=1.30, seed=12):
    """Points uniform in a solid cube (tools/xl_time.py): at voxel 0.05 m / graph 0.5 m its neighbourhoods are whole search
    balls of up to 4159 used voxels -- the regime above every one-wavefront and dense class of the local cut."""
    rng = np.random.default_rng(seed)
    return (rng.uniform(0, 1, (n, 3)) * side + np.array([1.0, -2.0, 0.2])).astype(np.float32)


def make_scene(name, n=None):
    name = name.upper()
    if name == "PC1M":
        return pc_scene(n or 1_000_000)
    if name == "URB10M":
        return urban_scene(n or 10_000_000)
    if name == "TOWN":
        return town_scene(n or 500_000)
    if name == "URB80M":
        return tiled_urban_scene(n or 80_000_000)
    if name == "C3N":
        return noisy_surface_scene(n or 5_000_000)
    if name == "BLOCK":
        return solid_block_scene(n or 500_000)
    raise ValueError(f"unknown scene {name}")
