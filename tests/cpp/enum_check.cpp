// Host check of the label enumeration the supervoxel kernels use (csrc/vccs_common.h: vccs_enum_*): for random and adversarial
// neighbourhoods the successive minima must produce exactly the set of distinct labels other than the voxel's own, each once,
// and stop.  Built and run by tests/test_enum_arith.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include "vccs_common.h"

static int check(const std::vector<int>& nl, int own, bool ascending_labels) {
  std::set<int> want;
  for (int l : nl) if (l >= 0 && l != own) want.insert(l);
  std::vector<uint32_t> key(nl.size());
  uint32_t ref = 0;
  if (ascending_labels) { for (size_t o = 0; o < nl.size(); ++o) key[o] = (nl[o] >= 0 && nl[o] != own) ? (uint32_t)nl[o] : 0xffffffffu; }   // k_pclt_sweep: the keys are the labels
  else { ref = vccs_enum_ref(own); for (size_t o = 0; o < nl.size(); ++o) key[o] = vccs_enum_key(nl[o], ref); }
  std::vector<int> got;
  uint32_t off = 0;
  for (int guard = 0; guard < 64; ++guard) {
    const uint32_t kq = vccs_enum_next(key.data(), (int)key.size(), off);
    if (!vccs_enum_valid(kq)) break;
    got.push_back(ascending_labels ? (int)kq : vccs_enum_label(kq, ref));
    off = kq + 1u;
  }
  if (got.size() != want.size()) return 1;
  std::set<int> gs(got.begin(), got.end());
  if (gs != want) return 1;
  if (ascending_labels) for (size_t i = 1; i < got.size(); ++i) if (got[i] <= got[i - 1]) return 1;
  return 0;
}

int main() {
  srand(7);
  long bad = 0, runs = 0;
  const int big = 0x7ffffffd;   // the largest label the scheme admits
  for (int it = 0; it < 400000; ++it) {
    const int n = (it & 1) ? 26 : 27;
    int own = (rand() % 4 == 0) ? -1 : rand() % 50;
    if (rand() % 50 == 0) own = big - rand() % 3;
    std::vector<int> nl(n);
    const int mode = it % 4;   // 0: mixed, 1: every neighbour present and foreign, 2: all distinct, 3: sparse
    for (int o = 0; o < n; ++o) {
      const int r = rand() % 10;
      if (mode == 1) nl[o] = (rand() % 6) + (own == 0 ? 1 : 0) * 7;
      else if (mode == 2) nl[o] = 1000 + o * (1 + rand() % 1000);
      else if (mode == 3) nl[o] = r < 8 ? -1 : (r == 8 ? own : rand() % 3);
      else nl[o] = r < 3 ? -1 : (r < 6 ? own : (rand() % 8 == 0 ? big - rand() % 5 : rand() % 6));
    }
    bad += check(nl, own, false); ++runs;
    bad += check(nl, own, true); ++runs;
  }
  printf("runs=%ld bad=%ld\n", runs, bad);
  return bad != 0;
}
