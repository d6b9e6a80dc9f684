// vgs_scenes.hpp -- the synthetic scenes of SURVEY.md section E as self-contained C++ (no Python, no dependency): the C++ examples can
// produce a BASELINE workload by themselves (examples/pcd_tool scene URB10M 10000000 out.pcd; examples/vgs_run reads the PCD).
//
// The reference ships no data (Town_Test.pcd is only named in its README.md:20).  The scenes: surfaces sampled uniformly by area, the
// point budget split proportionally to area (largest remainder), Gaussian noise of 3 mm along the surface normal, ONE shuffle of the
// cloud (the insertion order decides the octree's origin: SURVEY B.1), the viewpoint (0, 0, 1.5) inside the scene, extents scaled with
// sqrt(n / nominal) so that the density stays at about 25 points per voxel face.  Geometry and layout rules are those of
// vgs-svgs-segmentation_amd/scenes.py (what bench.py and the tests use: numpy's Philox stream); the random STREAM here is this file's
// own -- a counter-based splitmix64, Box-Muller for the noise, Fisher-Yates for the shuffle -- so the two generators make the same
// kind of scene, not the same points.  Same seed, same n: bit-identical float32 output on every machine.
#ifndef VGS_SCENES_HPP_
#define VGS_SCENES_HPP_

#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

namespace vgs_scenes {

struct Rng {   // splitmix64 over a counter: value k of stream `seed` is a pure function of (seed, k)
  uint64_t seed, k = 0;
  explicit Rng(uint64_t s) : seed(s) {}
  uint64_t next() {
    uint64_t z = seed + (++k) * 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
  double normal() {                                                                  // Box-Muller, one value per call
    const double u1 = 1.0 - uniform(), u2 = uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
};

static const double kNoiseSigma = 0.003;

struct Prim { int kind; double area; double p[13]; };   // 0 rect (origin 3, e1 3, e2 3, normal 3), 1 cylinder (cx, cy, r, z0, z1), 2 ball (c 3, r)

struct Scene {
  std::vector<Prim> items;
  void rect(const double* o, const double* a, const double* b) {
    Prim q{};
    q.kind = 0;
    const double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    q.area = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int k = 0; k < 3; ++k) { q.p[k] = o[k]; q.p[3 + k] = a[k]; q.p[6 + k] = b[k]; q.p[9 + k] = n[k] / q.area; }
    items.push_back(q);
  }
  void rect(double ox, double oy, double oz, double ax, double ay, double az, double bx, double by, double bz) {
    const double o[3] = {ox, oy, oz}, a[3] = {ax, ay, az}, b[3] = {bx, by, bz};
    rect(o, a, b);
  }
  void cylinder(double cx, double cy, double r, double z0, double z1) {
    Prim q{};
    q.kind = 1; q.area = 6.283185307179586 * r * (z1 - z0);
    q.p[0] = cx; q.p[1] = cy; q.p[2] = r; q.p[3] = z0; q.p[4] = z1;
    items.push_back(q);
  }
  void ball(double cx, double cy, double cz, double r) {
    Prim q{};
    q.kind = 2; q.area = 12.566370614359172 * r * r;
    q.p[0] = cx; q.p[1] = cy; q.p[2] = cz; q.p[3] = r;
    items.push_back(q);
  }
  void box(double x0, double y0, double x1, double y1, double h, bool pitched) {   // four facades + roof; no floor
    rect(x0, y0, 0, x1 - x0, 0, 0, 0, 0, h);
    rect(x0, y1, 0, x1 - x0, 0, 0, 0, 0, h);
    rect(x0, y0, 0, 0, y1 - y0, 0, 0, 0, h);
    rect(x1, y0, 0, 0, y1 - y0, 0, 0, 0, h);
    if (!pitched) { rect(x0, y0, h, x1 - x0, 0, 0, 0, y1 - y0, 0); return; }
    const double ym = 0.5 * (y0 + y1), rise = 0.35 * (y1 - y0);
    rect(x0, y0, h, x1 - x0, 0, 0, 0, ym - y0, rise);
    rect(x0, y1, h, x1 - x0, 0, 0, 0, ym - y1, rise);
  }
  // n points as packed float32 xyz
  std::vector<float> sample(int64_t n, Rng& rng) const {
    double total = 0;
    for (const Prim& q : items) total += q.area;
    std::vector<int64_t> cnt(items.size());
    std::vector<std::pair<double, size_t>> frac(items.size());
    int64_t given = 0;
    for (size_t i = 0; i < items.size(); ++i) {
      const double raw = items[i].area / total * (double)n;
      cnt[i] = (int64_t)std::floor(raw);
      frac[i] = {-(raw - (double)cnt[i]), i};
      given += cnt[i];
    }
    std::stable_sort(frac.begin(), frac.end());   // largest remainder first, ties by index
    for (int64_t k = 0; k < n - given; ++k) cnt[frac[(size_t)k].second]++;
    std::vector<float> out((size_t)n * 3);
    int64_t pos = 0;
    for (size_t i = 0; i < items.size(); ++i) {
      const Prim& q = items[i];
      for (int64_t k = 0; k < cnt[i]; ++k, ++pos) {
        double x, y, z;
        if (q.kind == 0) {
          const double u = rng.uniform(), v = rng.uniform(), d = rng.normal() * kNoiseSigma;
          x = q.p[0] + u * q.p[3] + v * q.p[6] + d * q.p[9];
          y = q.p[1] + u * q.p[4] + v * q.p[7] + d * q.p[10];
          z = q.p[2] + u * q.p[5] + v * q.p[8] + d * q.p[11];
        } else if (q.kind == 1) {
          const double th = rng.uniform() * 6.283185307179586, zz = q.p[3] + rng.uniform() * (q.p[4] - q.p[3]), rr = q.p[2] + rng.normal() * kNoiseSigma;
          x = q.p[0] + rr * std::cos(th); y = q.p[1] + rr * std::sin(th); z = zz;
        } else {   // points uniform in the ball's volume (vegetation-like clutter)
          double vx, vy, vz, nn;
          do { vx = rng.normal(); vy = rng.normal(); vz = rng.normal(); nn = std::sqrt(vx * vx + vy * vy + vz * vz); } while (nn == 0.0);
          const double rad = q.p[3] * std::cbrt(rng.uniform());
          x = q.p[0] + vx / nn * rad; y = q.p[1] + vy / nn * rad; z = q.p[2] + vz / nn * rad;
        }
        out[(size_t)pos * 3] = (float)x; out[(size_t)pos * 3 + 1] = (float)y; out[(size_t)pos * 3 + 2] = (float)z;
      }
    }
    for (int64_t i = n - 1; i > 0; --i) {   // Fisher-Yates
      const int64_t j = (int64_t)(rng.uniform() * (double)(i + 1));
      for (int a = 0; a < 3; ++a) std::swap(out[(size_t)i * 3 + a], out[(size_t)j * 3 + a]);
    }
    return out;
  }
};

// BASELINE config 2 "PC1M": ground + wall + vertical cylinder; nominal n = 1e6 (about 25.7 points per (0.05 m)^2)
inline std::vector<float> pc_scene(int64_t n = 1000000, uint64_t seed = 20260102) {
  const double s = std::sqrt((double)n / 1.0e6);
  Scene P;
  P.rect(-4 * s, -4 * s, 0, 8 * s, 0, 0, 0, 8 * s, 0);
  P.rect(-4 * s, 4 * s, 0, 8 * s, 0, 0, 0, 0, 3 * s);
  P.cylinder(2 * s, -2 * s, 0.5 * s, 0.0, 3 * s);
  Rng rng(seed);
  return P.sample(n, rng);
}

// BASELINE config 3 / 4 "URB10M": ground, box buildings (facades + flat roofs), poles, trees; nominal n = 1e7
inline std::vector<float> urban_scene(int64_t n = 10000000, uint64_t seed = 20260103, double cx = 0.0, double cy = 0.0, double nominal = 1.0e7) {
  const double s = std::sqrt((double)n / nominal), L = 50.0 * s;
  const int lots = 4;
  const double pitch = L / lots;
  Scene P;
  P.rect(cx - L / 2, cy - L / 2, 0, L, 0, 0, 0, L, 0);
  Rng lay(seed + 1);
  for (int i = 0; i < lots; ++i)
    for (int j = 0; j < lots; ++j) {
      const double x0 = cx - L / 2 + i * pitch, y0 = cy - L / 2 + j * pitch;
      if ((i + j) % 2 == 0) {
        const double w = pitch * (0.55 + 0.15 * lay.uniform()), d = pitch * (0.40 + 0.15 * lay.uniform());
        const double h = (4.0 + 5.0 * lay.uniform()) * std::max(s, 0.35);
        const double bx = x0 + 0.5 * (pitch - w), by = y0 + 0.5 * (pitch - d);
        P.box(bx, by, bx + w, by + d, h, false);
      } else {
        for (int k = 0; k < 5; ++k) {
          const double px = x0 + pitch * (0.1 + 0.8 * lay.uniform()), py = y0 + pitch * (0.1 + 0.8 * lay.uniform());
          P.cylinder(px, py, 0.15 * std::max(s, 0.5), 0.0, 5.0 * std::max(s, 0.4));
        }
        for (int k = 0; k < 2; ++k) {
          const double tr = 1.5 * std::max(s, 0.4);
          const double tx = x0 + pitch * (0.2 + 0.6 * lay.uniform()), ty = y0 + pitch * (0.2 + 0.6 * lay.uniform());
          P.ball(tx, ty, 2.0 * tr + 0.5, tr);
        }
      }
    }
  Rng rng(seed);
  return P.sample(n, rng);
}

// BASELINE config 1 stand-in "TOWN": ground, four buildings with pitched roofs, ten poles; nominal n = 5e5
inline std::vector<float> town_scene(int64_t n = 500000, uint64_t seed = 20260101) {
  const double s = std::sqrt((double)n / 5.0e5), L = 14.0 * s;
  Scene P;
  P.rect(-L / 2, -L / 2, 0, L, 0, 0, 0, L, 0);
  Rng lay(seed + 1);
  const int q[4][2] = {{-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
  for (int k = 0; k < 4; ++k) {
    const double w = 3.2 * s, d = 2.4 * s, h = (2.0 + 1.5 * lay.uniform()) * std::max(s, 0.4);
    const double bx = q[k][0] * L / 4 - w / 2, by = q[k][1] * L / 4 - d / 2;
    P.box(bx, by, bx + w, by + d, h, true);
  }
  for (int k = 0; k < 10; ++k) {
    const double px = (lay.uniform() - 0.5) * 0.9 * L, py = (lay.uniform() - 0.5) * 0.9 * L;
    P.cylinder(px, py, 0.1 * std::max(s, 0.5), 0.0, 4.0 * std::max(s, 0.4));
  }
  Rng rng(seed);
  return P.sample(n, rng);
}

// BASELINE config 5 "URB80M": a tx x ty grid of URB10M-like tiles (seeds seed0 + k), centred on the origin; tile >= 0: that tile alone
inline std::vector<float> tiled_urban_scene(int64_t n_total = 80000000, int tx = 4, int ty = 2, uint64_t seed0 = 20260110, int tile = -1) {
  const int nt = tx * ty;
  const int64_t per = n_total / nt;
  const double pitch = 50.0 * std::sqrt((double)per / 1.0e7);
  std::vector<float> out;
  for (int k = 0; k < nt; ++k) {
    if (tile >= 0 && k != tile) continue;
    const int i = k % tx, j = k / tx;
    const std::vector<float> t = urban_scene(per, seed0 + (uint64_t)k, (i - (tx - 1) / 2.0) * pitch, (j - (ty - 1) / 2.0) * pitch);
    out.insert(out.end(), t.begin(), t.end());
  }
  return out;
}

// by name: TOWN, PC1M, URB10M, URB80M (n = 0: the nominal size)
inline bool make_scene(const std::string& name, int64_t n, std::vector<float>& xyz) {
  if (name == "TOWN") { xyz = town_scene(n > 0 ? n : 500000); return true; }
  if (name == "PC1M") { xyz = pc_scene(n > 0 ? n : 1000000); return true; }
  if (name == "URB10M") { xyz = urban_scene(n > 0 ? n : 10000000); return true; }
  if (name == "URB80M") { xyz = tiled_urban_scene(n > 0 ? n : 80000000); return true; }
  return false;
}

}  // namespace vgs_scenes

#endif
