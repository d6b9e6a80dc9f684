// oracle/selftest.cpp -- TEST INFRASTRUCTURE.  The oracle under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU
// (`make -C oracle sanitize`): a small synthetic cloud through every path of refcpu.cpp -- both arithmetic modes, both data-flow
// flavours, the SVGS driver from a grid labelling and from the VCCS restatement -- plus the known-answer cuts of SURVEY.md C.
// Non-zero exit (or a sanitizer report, which aborts) fails tests/test_oracle_kat.py::test_oracle_under_sanitizers.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "refcpu.hpp"

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
  {   // KAT-C2 and KAT-C5
    const std::vector<float> W2 = {1.f, .9f, .2f, .9f, 1.f, .78f, .2f, .78f, 1.f};
    if (cut_graph_faithful(0.3f, W2, 3).size() != 3) { std::printf("KAT-C2 failed\n"); ++fails; }
    const std::vector<float> W5 = {1.f, .5f, .5f, .5f, 1.f, .95f, .5f, .95f, 1.f};
    if (cut_graph_faithful(0.3f, W5, 3).size() != 1) { std::printf("KAT-C5 failed\n"); ++fails; }
  }
  std::printf("oracle selftest: %d failure(s)\n", fails);
  return fails ? 1 : 0;
}
