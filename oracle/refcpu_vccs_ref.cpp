// oracle/refcpu_vccs_ref.cpp -- TEST INFRASTRUCTURE: the INDEPENDENT arithmetic leg of the supervoxel stage.
//
// refcpu_vccs.cpp restates the product's synchronous VCCS variant with the product's own arithmetic (csrc/vccs_common.h, float,
// libm-free, fixed-point sums) so that GPU and CPU agree bit for bit -- which also means a mistake in that arithmetic would show
// on neither side.  This file runs the SAME published steps (pcl::SupervoxelClustering as used by createSupervoxels, reference
// supervoxel_segmentation.h:265-284; Papon et al. 2013: 26-adjacency, one seed per occupied seed-resolution cell snapped to the
// nearest voxel, int(1.8 seed / res) expansion rounds on  D = w_s |dx| / seed + w_n (1 - |n1 . n2|)  with the colour term zero,
// centroid / normal updates after every round, five refinement passes) in "RefMath" style and shares NO code with the device:
// double precision throughout, libm sqrt / fabs, a cyclic Jacobi eigen-solver written here, plain floating-point sums, std::map
// lookups.  It includes neither vccs_common.h nor vgs_math.h.  Only the voxel table (the octree restatement, bit-exact against
// the GPU by its own tests) is taken from refcpu.cpp.
//
// Labels from the two legs differ where a float decision sits on an edge (a voxel equidistant from two supervoxels), so the
// comparison is a tolerance (tests/test_oracle_kat.py, tests/test_gpu_vccs.py): same supervoxel count, >= 97 % of the voxels in
// matching supervoxels after best-match relabelling.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

#include "refcpu.hpp"

namespace refcpu {
namespace {

// eigenvector of the smallest eigenvalue of a symmetric 3x3 matrix: cyclic Jacobi rotations in double
void smallest_eigenvector(const double C[3][3], double n[3]) {
  double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = C[i][j];
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (std::fabs(a[p][q]) < 1e-300) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
        for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
        for (int k = 0; k < 3; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
      }
  }
  int m = 0;
  for (int k = 1; k < 3; ++k) if (a[k][k] < a[m][m]) m = k;
  for (int k = 0; k < 3; ++k) n[k] = v[k][m];
}

}  // namespace

void vccs_supervoxels_refmath(const float* xyz, int64_t n, int stride, const Params& P, std::vector<int>& label, int& max_label) {
  VoxelTable T;
  build_voxel_table(xyz, n, stride, P.voxel_size, T);
  const int V = T.V();
  label.assign((size_t)n, 0);
  max_label = 0;
  if (V == 0) return;
  // voxel centroids
  std::vector<double> cen((size_t)V * 3), nrm((size_t)V * 3, 0.0);
  for (int v = 0; v < V; ++v) {
    double s[3] = {0, 0, 0};
    for (int k = T.start[v]; k < T.start[v + 1]; ++k) {
      const float* p = xyz + (int64_t)T.point_idx[k] * stride;
      for (int a = 0; a < 3; ++a) s[a] += (double)p[a];
    }
    const double cnt = (double)(T.start[v + 1] - T.start[v]);
    for (int a = 0; a < 3; ++a) cen[3 * (size_t)v + a] = s[a] / cnt;
  }
  // 26-adjacency through a map on the lattice key; the order of a voxel's neighbours is dz outermost, then dy, then dx -- the
  // order in which a tie between two labels at the same distance is met does not matter (ties go to the smaller label)
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, int> at;
  for (int v = 0; v < V; ++v) at[std::make_tuple(T.key[3 * v], T.key[3 * v + 1], T.key[3 * v + 2])] = v;
  std::vector<std::vector<int>> nbr((size_t)V);
  for (int v = 0; v < V; ++v)
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
      if (!dx && !dy && !dz) continue;
      const long long x = (long long)T.key[3 * v] + dx, y = (long long)T.key[3 * v + 1] + dy, z = (long long)T.key[3 * v + 2] + dz;
      if (x < 0 || y < 0 || z < 0) continue;
      auto it = at.find(std::make_tuple((uint32_t)x, (uint32_t)y, (uint32_t)z));
      if (it != at.end()) nbr[(size_t)v].push_back(it->second);
    }
  // normals: smallest-eigenvalue direction of the covariance of the voxel's and its neighbours' centroids, towards (0, 0, 0)
  for (int v = 0; v < V; ++v) {
    const int np = 1 + (int)nbr[(size_t)v].size();
    if (np < 3) continue;
    double mean[3] = {cen[3 * (size_t)v], cen[3 * (size_t)v + 1], cen[3 * (size_t)v + 2]};
    for (int t : nbr[(size_t)v]) for (int a = 0; a < 3; ++a) mean[a] += cen[3 * (size_t)t + a];
    for (int a = 0; a < 3; ++a) mean[a] /= (double)np;
    double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    auto acc = [&](int u) {
      double d[3];
      for (int a = 0; a < 3; ++a) d[a] = cen[3 * (size_t)u + a] - mean[a];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[i][j] += d[i] * d[j];
    };
    acc(v);
    for (int t : nbr[(size_t)v]) acc(t);
    double nn[3];
    smallest_eigenvector(C, nn);
    const double dot = -(nn[0] * cen[3 * (size_t)v] + nn[1] * cen[3 * (size_t)v + 1] + nn[2] * cen[3 * (size_t)v + 2]);
    const double sgn = dot < 0 ? -1.0 : 1.0;
    for (int a = 0; a < 3; ++a) nrm[3 * (size_t)v + a] = sgn * nn[a];
  }
  // seeds: one per occupied seed-size cell of the grid that hangs on the octree's minimum corner (as the product's), the voxel
  // nearest to the cell's centre; supervoxel k = k-th occupied cell in ascending (x, y, z) cell order
  const double seed = (double)P.seed_size;
  const double mn[3] = {(double)(float)T.min[0], (double)(float)T.min[1], (double)(float)T.min[2]};
  std::map<std::tuple<long long, long long, long long>, std::pair<double, int>> best;   // cell -> (d2, voxel)
  for (int v = 0; v < V; ++v) {
    long long c[3];
    double d2 = 0;
    for (int a = 0; a < 3; ++a) {
      c[a] = (long long)((cen[3 * (size_t)v + a] - mn[a]) / seed);
      if (c[a] < 0) c[a] = 0;
      const double d = cen[3 * (size_t)v + a] - (mn[a] + ((double)c[a] + 0.5) * seed);
      d2 += d * d;
    }
    auto key = std::make_tuple(c[0], c[1], c[2]);
    auto it = best.find(key);
    if (it == best.end() || d2 < it->second.first || (d2 == it->second.first && v < it->second.second)) best[key] = std::make_pair(d2, v);
  }
  const int K = (int)best.size();
  std::vector<int> seed_voxel;
  seed_voxel.reserve((size_t)K);
  for (auto& kv : best) seed_voxel.push_back(kv.second.second);
  // expansion: six passes of T rounds, every round synchronous
  const int Tn = (int)(1.8f * P.seed_size / P.voxel_size);
  const double w_s = (double)P.spatial_impt / seed, w_n = (double)P.normal_impt;
  std::vector<double> sc((size_t)K * 3, 0.0), sn((size_t)K * 3, 0.0);
  std::vector<int> lab((size_t)V, -1), lab2((size_t)V, -1);
  std::vector<double> dist((size_t)V, 1e300), dist2((size_t)V, 1e300);
  auto D = [&](int v, int l) {
    double ds = 0, dot = 0;
    for (int a = 0; a < 3; ++a) { const double d = cen[3 * (size_t)v + a] - sc[3 * (size_t)l + a]; ds += d * d; dot += nrm[3 * (size_t)v + a] * sn[3 * (size_t)l + a]; }
    return std::sqrt(ds) * w_s + w_n * (1.0 - std::fabs(dot));
  };
  for (int pass = 0; pass < 6; ++pass) {
    if (pass > 0) {   // refinement: re-seed at the member voxel nearest to the supervoxel's centroid
      std::vector<double> bd((size_t)K, 1e300);
      std::vector<int> bv((size_t)K, -1);
      for (int v = 0; v < V; ++v) {
        const int l = lab[(size_t)v];
        if (l < 0) continue;
        double d2 = 0;
        for (int a = 0; a < 3; ++a) { const double d = cen[3 * (size_t)v + a] - sc[3 * (size_t)l + a]; d2 += d * d; }
        if (d2 < bd[(size_t)l]) { bd[(size_t)l] = d2; bv[(size_t)l] = v; }
      }
      seed_voxel = bv;
    }
    std::fill(lab.begin(), lab.end(), -1);
    std::fill(dist.begin(), dist.end(), 1e300);
    for (int k = 0; k < K; ++k) {
      const int v = seed_voxel[(size_t)k];
      if (v < 0) { for (int a = 0; a < 3; ++a) { sc[3 * (size_t)k + a] = 0; sn[3 * (size_t)k + a] = 0; } continue; }
      lab[(size_t)v] = k; dist[(size_t)v] = 0.0;
      for (int a = 0; a < 3; ++a) { sc[3 * (size_t)k + a] = cen[3 * (size_t)v + a]; sn[3 * (size_t)k + a] = nrm[3 * (size_t)v + a]; }
    }
    for (int it = 0; it < Tn; ++it) {
      for (int v = 0; v < V; ++v) {
        int bl = lab[(size_t)v];
        double bdist = dist[(size_t)v];
        for (int t : nbr[(size_t)v]) {
          const int l = lab[(size_t)t];
          if (l < 0 || l == lab[(size_t)v]) continue;
          const double d = D(v, l);
          if (d < bdist || (d == bdist && l < bl)) { bdist = d; bl = l; }
        }
        lab2[(size_t)v] = bl; dist2[(size_t)v] = bdist;
      }
      lab.swap(lab2); dist.swap(dist2);
      std::vector<double> sums((size_t)K * 6, 0.0);
      std::vector<int> count((size_t)K, 0);
      for (int v = 0; v < V; ++v) {
        const int l = lab[(size_t)v];
        if (l < 0) continue;
        for (int a = 0; a < 3; ++a) { sums[6 * (size_t)l + a] += cen[3 * (size_t)v + a]; sums[6 * (size_t)l + 3 + a] += nrm[3 * (size_t)v + a]; }
        ++count[(size_t)l];
      }
      for (int k = 0; k < K; ++k) {
        if (!count[(size_t)k]) continue;   // keeps its previous state
        double m[3], len = 0;
        for (int a = 0; a < 3; ++a) { sc[3 * (size_t)k + a] = sums[6 * (size_t)k + a] / count[(size_t)k]; m[a] = sums[6 * (size_t)k + 3 + a] / count[(size_t)k]; len += m[a] * m[a]; }
        len = std::sqrt(len);
        for (int a = 0; a < 3; ++a) sn[3 * (size_t)k + a] = len > 0 ? m[a] / len : 0.0;
      }
    }
  }
  for (int64_t i = 0; i < n; ++i) {
    const int v = T.point_voxel[(size_t)i];
    label[(size_t)i] = (v < 0 || lab[(size_t)v] < 0) ? 0 : lab[(size_t)v] + 1;
  }
  max_label = K;
}


// ==================================================================================================================
// The independent leg of the PCL-ORDER stage (vccs_mode 1, the engine's default since round 6): the steps of
// refcpu_vccs.cpp::vccs_pcl_supervoxels -- pcl::SupervoxelClustering 1.8.1 as recalled there (reference supervoxel_segmentation.h:265-284:
// extract + refineSupervoxels(5)) -- written again in double precision with libm, the Jacobi solver above, two-pass covariances, plain
// floating-point means, std::map lookups and a brute-force nearest-voxel search; no vccs_common.h, no vgs_math.h.  Shared with the other
// leg: only the voxel table on the adjacency octree's own lattice (build_voxel_table_bbox, integer binning) and the parameters.  The
// number of expansion rounds is the float expression of the product, (int)(1.8f * seed / res): parameter arithmetic, not data arithmetic.
void vccs_pcl_supervoxels_refmath(const float* xyz, int64_t n, int stride, const Params& P, std::vector<int>& label, int& max_label) {
  VoxelTable T;
  build_voxel_table_bbox(xyz, n, stride, P.voxel_size, T);
  const int V = T.V();
  label.assign((size_t)n, 0);
  max_label = 0;
  if (V == 0) return;
  std::vector<double> cen((size_t)V * 3), nrm((size_t)V * 3, 0.0);
  for (int v = 0; v < V; ++v) {
    double s[3] = {0, 0, 0};
    for (int k = T.start[v]; k < T.start[v + 1]; ++k) {
      const float* p = xyz + (int64_t)T.point_idx[k] * stride;
      for (int a = 0; a < 3; ++a) s[a] += (double)p[a];
    }
    const double cnt = (double)(T.start[v + 1] - T.start[v]);
    for (int a = 0; a < 3; ++a) cen[3 * (size_t)v + a] = s[a] / cnt;
  }
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, int> at;
  for (int v = 0; v < V; ++v) at[std::make_tuple(T.key[3 * v], T.key[3 * v + 1], T.key[3 * v + 2])] = v;
  auto find = [&](long long x, long long y, long long z) -> int {
    if (x < 0 || y < 0 || z < 0) return -1;
    auto it = at.find(std::make_tuple((uint32_t)x, (uint32_t)y, (uint32_t)z));
    return it == at.end() ? -1 : it->second;
  };
  // the 27 cells around a leaf, the leaf itself included (computeNeighbors); their order does not matter here: sums and minima only
  std::vector<std::vector<int>> n27((size_t)V);
  for (int v = 0; v < V; ++v)
    for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) for (int dz = -1; dz <= 1; ++dz) {
      const int u = find((long long)T.key[3 * v] + dx, (long long)T.key[3 * v + 1] + dy, (long long)T.key[3 * v + 2] + dz);
      if (u >= 0) n27[(size_t)v].push_back(u);
    }
  // normal of leaf v from the multiset  [v] + for every t of its 27 cells (owned by `want` when want >= 0): [t] + the cells u of t (likewise)
  std::vector<int> owner((size_t)V, -1);
  auto leaf_normal = [&](int v, int want, double* out) {
    std::vector<int> idx;
    if (want < 0) idx.push_back(v);
    for (int t : n27[(size_t)v]) {
      if (want >= 0 && owner[(size_t)t] != want) continue;
      idx.push_back(t);
      for (int u : n27[(size_t)t]) if (want < 0 || owner[(size_t)u] == want) idx.push_back(u);
    }
    out[0] = out[1] = out[2] = 0.0;
    if (idx.size() < 3) return;
    double m[3] = {0, 0, 0};
    for (int u : idx) for (int a = 0; a < 3; ++a) m[a] += cen[3 * (size_t)u + a];
    for (int a = 0; a < 3; ++a) m[a] /= (double)idx.size();
    double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int u : idx) {
      const double d[3] = {cen[3 * (size_t)u] - m[0], cen[3 * (size_t)u + 1] - m[1], cen[3 * (size_t)u + 2] - m[2]};
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[a][b] += d[a] * d[b];
    }
    double e[3];
    smallest_eigenvector(C, e);
    const double* p0 = &cen[3 * (size_t)v];
    if (e[0] * (0.0 - p0[0]) + e[1] * (0.0 - p0[1]) + e[2] * (0.0 - p0[2]) < 0.0) { e[0] = -e[0]; e[1] = -e[1]; e[2] = -e[2]; }
    const double len = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    if (len > 0.0) for (int a = 0; a < 3; ++a) out[a] = e[a] / len;
  };
  for (int v = 0; v < V; ++v) leaf_normal(v, -1, &nrm[3 * (size_t)v]);
  // seeds: the voxel nearest to the centre of every occupied seed cell (cells anchored at the lattice's corner), cells in ascending
  // x-major Morton order of their indices; kept when more than 0.05 * (seed / 2)^2 * pi / res^2 voxels lie within seed / 2
  const double seed = (double)P.seed_size, res = (double)P.voxel_size;
  auto spread = [](uint64_t x) { uint64_t r = 0; for (int b = 0; b < 21; ++b) r |= ((x >> b) & 1ull) << (3 * b); return r; };
  std::map<uint64_t, std::pair<double, int>> cells;   // Morton of the cell -> (squared distance to its centre, voxel)
  for (int v = 0; v < V; ++v) {
    long long ci[3];
    double d2 = 0.0;
    for (int a = 0; a < 3; ++a) {
      ci[a] = (long long)((cen[3 * (size_t)v + a] - (double)(float)T.min[a]) / seed);
      if (ci[a] < 0) ci[a] = 0;
      const double cc = (double)(float)T.min[a] + ((double)ci[a] + 0.5) * seed;
      d2 += (cen[3 * (size_t)v + a] - cc) * (cen[3 * (size_t)v + a] - cc);
    }
    const uint64_t code = (spread((uint64_t)ci[0]) << 2) | (spread((uint64_t)ci[1]) << 1) | spread((uint64_t)ci[2]);
    auto it = cells.find(code);
    if (it == cells.end() || d2 < it->second.first) cells[code] = std::make_pair(d2, v);
  }
  const double rad = 0.5 * seed, min_points = 0.05 * rad * rad * 3.14159265358979323846 / (res * res);
  const int R = (int)(rad / res) + 1;
  std::vector<int> seeds;
  for (const auto& kv : cells) {
    const int s0 = kv.second.second;
    int num = 0;
    for (int dz = -R; dz <= R; ++dz) for (int dy = -R; dy <= R; ++dy) for (int dx = -R; dx <= R; ++dx) {
      const int u = find((long long)T.key[3 * s0] + dx, (long long)T.key[3 * s0 + 1] + dy, (long long)T.key[3 * s0 + 2] + dz);
      if (u < 0) continue;
      double e2 = 0.0;
      for (int a = 0; a < 3; ++a) e2 += (cen[3 * (size_t)u + a] - cen[3 * (size_t)s0 + a]) * (cen[3 * (size_t)u + a] - cen[3 * (size_t)s0 + a]);
      if (e2 < rad * rad) ++num;
    }
    if ((double)num > min_points) seeds.push_back(s0);
  }
  const int K = (int)seeds.size();
  if (K == 0) return;
  const int depth = (int)(1.8f * P.seed_size / P.voxel_size);
  const double w_s_over_seed = (double)P.spatial_impt / seed, w_n = (double)P.normal_impt;
  std::vector<double> dist((size_t)V, 1.0e300), sc((size_t)K * 3), sn((size_t)K * 3);
  std::vector<char> alive((size_t)K, 1);
  auto distance = [&](int v, int k) {
    double d2 = 0.0, dot = 0.0;
    for (int a = 0; a < 3; ++a) { const double e = cen[3 * (size_t)v + a] - sc[3 * (size_t)k + a]; d2 += e * e; dot += nrm[3 * (size_t)v + a] * sn[3 * (size_t)k + a]; }
    return std::sqrt(d2) * w_s_over_seed + w_n * (1.0 - std::fabs(dot));
  };
  auto update_centroids = [&]() {
    std::vector<double> sum((size_t)K * 6, 0.0);
    std::vector<int> cnt((size_t)K, 0);
    for (int v = 0; v < V; ++v) {
      const int l = owner[(size_t)v];
      if (l < 0) continue;
      for (int a = 0; a < 3; ++a) { sum[6 * (size_t)l + a] += cen[3 * (size_t)v + a]; sum[6 * (size_t)l + 3 + a] += nrm[3 * (size_t)v + a]; }
      cnt[(size_t)l]++;
    }
    for (int k = 0; k < K; ++k) {
      if (!alive[(size_t)k]) continue;
      if (cnt[(size_t)k] == 0) { alive[(size_t)k] = 0; continue; }
      double m[3];
      for (int a = 0; a < 3; ++a) { sc[3 * (size_t)k + a] = sum[6 * (size_t)k + a] / cnt[(size_t)k]; m[a] = sum[6 * (size_t)k + 3 + a] / cnt[(size_t)k]; }
      const double len = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
      for (int a = 0; a < 3; ++a) sn[3 * (size_t)k + a] = len > 0.0 ? m[a] / len : 0.0;
    }
  };
  auto expand_all = [&]() {
    std::vector<std::vector<int>> leaves((size_t)K);
    std::vector<char> taken((size_t)V);
    for (int i = 1; i < depth; ++i) {
      for (int k = 0; k < K; ++k) leaves[(size_t)k].clear();
      for (int v = 0; v < V; ++v) if (owner[(size_t)v] >= 0) leaves[(size_t)owner[(size_t)v]].push_back(v);
      std::fill(taken.begin(), taken.end(), 0);
      for (int k = 0; k < K; ++k) {   // the supervoxels take their turns one after the other
        if (!alive[(size_t)k]) continue;
        for (int leaf : leaves[(size_t)k]) {
          if (taken[(size_t)leaf]) continue;   // lost to an earlier supervoxel of this round
          for (int nb : n27[(size_t)leaf]) {
            if (owner[(size_t)nb] == k) continue;
            const double d = distance(nb, k);
            if (d < dist[(size_t)nb]) { dist[(size_t)nb] = d; owner[(size_t)nb] = k; taken[(size_t)nb] = 1; }
          }
        }
      }
      update_centroids();
    }
  };
  for (int k = 0; k < K; ++k) {
    owner[(size_t)seeds[(size_t)k]] = k;
    for (int a = 0; a < 3; ++a) { sc[3 * (size_t)k + a] = cen[3 * (size_t)seeds[(size_t)k] + a]; sn[3 * (size_t)k + a] = nrm[3 * (size_t)seeds[(size_t)k] + a]; }
  }
  expand_all();
  for (int pass = 0; pass < 5; ++pass) {
    // refineNormals: every owned leaf's normal again, from the leaves of its own supervoxel in its two rings
    {
      std::vector<double> fresh(nrm);
      for (int v = 0; v < V; ++v) if (owner[(size_t)v] >= 0) leaf_normal(v, owner[(size_t)v], &fresh[3 * (size_t)v]);
      nrm.swap(fresh);
    }
    // reseedSupervoxels: the voxel nearest to the supervoxel's centroid among ALL voxels, ties to the smaller voxel id; a search over
    // growing boxes of lattice cells around the centroid's cell until the best distance is inside the searched box
    std::vector<int> reseed((size_t)K, -1);
    for (int k = 0; k < K; ++k) {
      if (!alive[(size_t)k]) continue;
      long long c0[3];
      for (int a = 0; a < 3; ++a) c0[a] = (long long)std::floor((sc[3 * (size_t)k + a] - T.min[a]) / T.resolution);
      double best = 1.0e300;
      int bv = -1;
      for (int r = 1; r < 64; ++r) {
        for (long long z = c0[2] - r; z <= c0[2] + r; ++z) for (long long y = c0[1] - r; y <= c0[1] + r; ++y) for (long long x = c0[0] - r; x <= c0[0] + r; ++x) {
          const int v = find(x, y, z);
          if (v < 0) continue;
          double e2 = 0.0;
          for (int a = 0; a < 3; ++a) e2 += (cen[3 * (size_t)v + a] - sc[3 * (size_t)k + a]) * (cen[3 * (size_t)v + a] - sc[3 * (size_t)k + a]);
          if (e2 < best || (e2 == best && v < bv)) { best = e2; bv = v; }
        }
        if (bv >= 0 && std::sqrt(best) < (double)(r - 1) * res) break;   // nothing outside the box can be nearer
      }
      reseed[(size_t)k] = bv;
    }
    std::fill(owner.begin(), owner.end(), -1);
    std::fill(dist.begin(), dist.end(), 1.0e300);
    for (int k = 0; k < K; ++k) if (alive[(size_t)k] && reseed[(size_t)k] >= 0) owner[(size_t)reseed[(size_t)k]] = k;   // a contested voxel goes to the later one
    expand_all();
  }
  for (int64_t i = 0; i < n; ++i) {
    const int v = T.point_voxel[(size_t)i];
    label[(size_t)i] = (v < 0 || owner[(size_t)v] < 0) ? 0 : owner[(size_t)v] + 1;
  }
  for (int k = 0; k < K; ++k) if (alive[(size_t)k]) max_label = k + 1;
}

}  // namespace refcpu
