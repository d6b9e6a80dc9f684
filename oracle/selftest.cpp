// oracle/selftest.cpp -- TEST INFRASTRUCTURE.  The oracle under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU
// (`make -C oracle sanitize`): a small synthetic cloud through every path of refcpu.cpp -- both arithmetic modes, both data-flow
// flavours, the SVGS driver from a grid labelling and from the VCCS restatement -- plus the known-answer cuts of SURVEY.md C.
// Non-zero exit (or a sanitizer report, which aborts) fails tests/test_oracle_kat.py::test_oracle_under_sanitizers.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "refcpu.hpp"
#include "vccs_common.h"

static uint64_t splitmix(uint64_t& s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static float uni(uint64_t& s) { return (float)((splitmix(s) >> 40) * (1.0 / 16777216.0)); }

int main() {
  using namespace refcpu;
  // ground plane + wall + a sparse blob (unused voxels, NaN-free), one non-finite point, one repeated point
  uint64_t seed = 20260103;
  std::vector<float> xyz;
  auto add = [&](float x, float y, float z) { xyz.push_back(x); xyz.push_back(y); xyz.push_back(z); };
  for (int i = 0; i < 9000; ++i) add(uni(seed) * 3.0f - 1.5f, uni(seed) * 3.0f - 1.5f, 0.003f * (uni(seed) - 0.5f));
  for (int i = 0; i < 4000; ++i) add(uni(seed) * 3.0f - 1.5f, 1.5f + 0.003f * (uni(seed) - 0.5f), uni(seed) * 1.5f);
  for (int i = 0; i < 600; ++i) add(uni(seed) * 0.8f, uni(seed) * 0.8f - 1.0f, 0.5f + uni(seed) * 0.8f);
  add(NAN, 0.0f, 0.0f);
  add(xyz[0], xyz[1], xyz[2]);
  const int64_t n = (int64_t)(xyz.size() / 3);
  int fails = 0;
  std::vector<int> first_labels;
  for (int math = 0; math < 2; ++math)
    for (int flavour = 0; flavour < 2; ++flavour) {
      Params P;
      P.voxel_size = 0.1f; P.math = math; P.flavour = flavour;
      VgsResult R;
      run_vgs(xyz.data(), n, 3, P, R);
      if (R.T.V() <= 0 || (int64_t)R.point_label.size() != n || R.clusters_num <= 0) { std::printf("run_vgs(math %d, flavour %d): empty result\n", math, flavour); ++fails; }
      if (flavour == 0) first_labels = R.point_label;
      else {   // the two data flows of one arithmetic mode agree on almost every point
        int64_t same = 0;
        for (int64_t i = 0; i < n; ++i) same += (R.point_label[(size_t)i] >= 0) == (first_labels[(size_t)i] >= 0);
        if (same < n * 99 / 100) { std::printf("flavours disagree (math %d): %ld of %ld\n", math, (long)same, (long)n); ++fails; }
      }
    }
  {
    Params P;
    P.voxel_size = 0.05f; P.seed_size = 0.25f; P.sig_w = 1.0f; P.cut_thred = 0.5f; P.math = 1; P.flavour = 1;
    std::vector<int> lab; int mx = 0;
    vccs_supervoxels(xyz.data(), n, 3, P, lab, mx);
    if (mx <= 0 || (int64_t)lab.size() != n) { std::printf("vccs_supervoxels: no labels\n"); ++fails; }
    {   // the independent (double, libm) leg of the same steps: same supervoxel count
      std::vector<int> lab2; int mx2 = 0;
      vccs_supervoxels_refmath(xyz.data(), n, 3, P, lab2, mx2);
      if (mx2 != mx || (int64_t)lab2.size() != n) { std::printf("vccs_supervoxels_refmath: %d supervoxels, expected %d\n", mx2, mx); ++fails; }
    }
    for (int math = 0; math < 2; ++math) {
      P.math = math;
      SvgsResult S;
      run_svgs_from_labels(xyz.data(), n, 3, lab.data(), mx, P, S);
      if (S.nodes.empty() || (int64_t)S.point_label.size() != n) { std::printf("run_svgs_from_labels(math %d): empty result\n", math); ++fails; }
    }
  }
  {   // vccs_mode 1 (PCL order): the adjacency octree's own lattice, refineNormals, the re-seed by the nearest of all voxels
    Params P;
    P.voxel_size = 0.05f; P.seed_size = 0.25f; P.math = 1; P.flavour = 1;
    std::vector<int> lab; int mx = 0;
    vccs_pcl_supervoxels(xyz.data(), n, 3, P, lab, mx);
    if (mx <= 0 || (int64_t)lab.size() != n) { std::printf("vccs_pcl_supervoxels: no labels\n"); ++fails; }
    // defineBoundingBox + getKeyBitSize: the box is the cloud's bounding box padded symmetrically to 2^depth voxels (less epsilon)
    VoxelTable T, T0;
    build_voxel_table_bbox(xyz.data(), n, 3, P.voxel_size, T);
    build_voxel_table(xyz.data(), n, 3, P.voxel_size, T0);
    float mn[3] = {1e30f, 1e30f, 1e30f}, mxb[3] = {-1e30f, -1e30f, -1e30f};
    for (int64_t i = 0; i < n; ++i) { if (!std::isfinite(xyz[3 * i])) continue; for (int a = 0; a < 3; ++a) { mn[a] = std::fmin(mn[a], xyz[3 * i + a]); mxb[a] = std::fmax(mxb[a], xyz[3 * i + a]); } }
    for (int a = 0; a < 3; ++a) {
      const double side = (double)(1u << T.depth) * T.resolution, mid_box = T.min[a] + 0.5 * side, mid_cloud = 0.5 * ((double)mn[a] + (double)mxb[a]);
      if (std::fabs(mid_box - mid_cloud) > 1e-6 || T.min[a] > (double)mn[a] || T.min[a] + side < (double)mxb[a]) { std::printf("bbox lattice: axis %d box [%g, %g) cloud [%g, %g]\n", a, T.min[a], T.min[a] + side, mn[a], mxb[a]); ++fails; }
    }
    if (T.V() <= 0 || std::abs(T.V() - T0.V()) > T0.V() / 4) { std::printf("bbox lattice: %d voxels against %d on the grown lattice\n", T.V(), T0.V()); ++fails; }
    {   // every finite point is binned, all points of a voxel fall into one cell of the lattice
      int64_t binned = 0;
      for (int64_t i = 0; i < n; ++i) binned += T.point_voxel[(size_t)i] >= 0;
      if (binned != n - 1) { std::printf("bbox lattice: %ld of %ld points binned\n", (long)binned, (long)(n - 1)); ++fails; }
    }
    // vccs_nearest_voxel against a search over all voxels, from query points in and around the cloud
    std::vector<float> cen((size_t)T.V() * 3);
    for (int v = 0; v < T.V(); ++v) {
      float sx = 0, sy = 0, sz = 0;
      for (int k = T.start[v]; k < T.start[v + 1]; ++k) { const float* q = &xyz[3 * (size_t)T.point_idx[k]]; sx += q[0]; sy += q[1]; sz += q[2]; }
      const int cnt = T.start[v + 1] - T.start[v];
      cen[3 * v] = sx / cnt; cen[3 * v + 1] = sy / cnt; cen[3 * v + 2] = sz / cnt;
    }
    const uint32_t lim = 1u << T.depth;
    auto find = [&](uint32_t x, uint32_t y, uint32_t z) -> int {
      for (int v = 0; v < T.V(); ++v) if (T.key[3 * v] == x && T.key[3 * v + 1] == y && T.key[3 * v + 2] == z) return v;
      return -1;
    };
    uint64_t s2 = 99;
    for (int trial = 0; trial < 40; ++trial) {
      const float c[3] = {mn[0] + uni(s2) * (mxb[0] - mn[0]), mn[1] + uni(s2) * (mxb[1] - mn[1]), mn[2] + uni(s2) * (mxb[2] - mn[2])};
      const uint32_t kx = vm_axis_key(c[0], T.min[0], T.resolution), ky = vm_axis_key(c[1], T.min[1], T.resolution), kz = vm_axis_key(c[2], T.min[2], T.resolution);
      const unsigned long long got = vccs_nearest_voxel(c, kx, ky, kz, lim, P.voxel_size, find, [&](int v) { return (const float*)&cen[3 * (size_t)v]; });
      unsigned long long want = ~0ull;
      for (int v = 0; v < T.V(); ++v) {
        const float ex = cen[3 * v] - c[0], ey = cen[3 * v + 1] - c[1], ez = cen[3 * v + 2] - c[2];
        const unsigned long long key = ((unsigned long long)vm_bits((ex * ex + ey * ey) + ez * ez) << 32) | (unsigned long long)(uint32_t)v;
        want = key < want ? key : want;
      }
      if (got != want) { std::printf("vccs_nearest_voxel: trial %d voxel %u (d2 bits %u), all-voxel search %u (%u)\n", trial, (unsigned)got, (unsigned)(got >> 32), (unsigned)want, (unsigned)(want >> 32)); ++fails; }
    }
  }
  {   // KAT-C2 and KAT-C5
    const std::vector<float> W2 = {1.f, .9f, .2f, .9f, 1.f, .78f, .2f, .78f, 1.f};
    if (cut_graph_faithful(0.3f, W2, 3).size() != 3) { std::printf("KAT-C2 failed\n"); ++fails; }
    const std::vector<float> W5 = {1.f, .5f, .5f, .5f, 1.f, .95f, .5f, .95f, 1.f};
    if (cut_graph_faithful(0.3f, W5, 3).size() != 1) { std::printf("KAT-C5 failed\n"); ++fails; }
  }
  std::printf("oracle selftest: %d failure(s)\n", fails);
  return fails ? 1 : 0;
}
