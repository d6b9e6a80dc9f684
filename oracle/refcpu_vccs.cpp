// oracle/refcpu_vccs.cpp -- TEST INFRASTRUCTURE: CPU restatement of the VCCS-STYLE supervoxel stage of the product
// (csrc/vccs.hip, arithmetic in csrc/vccs_common.h).  PARITY UNPINNED with respect to pcl::SupervoxelClustering
// (reference supervoxel_segmentation.h:265-284): PCL is not available, its owner iteration is sequential and order
// dependent; this file restates the synchronous variant documented in vccs.hip so that the GPU stage can be checked
// label for label, and tests/test_gpu_vccs.py checks the published algorithm's invariants on top.
#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "refcpu.hpp"
#include "vccs_common.h"

namespace refcpu {

void vccs_supervoxels(const float* xyz, int64_t n, int stride, const Params& P, std::vector<int>& label, int& max_label) {
  VoxelTable T;
  build_voxel_table(xyz, n, stride, P.voxel_size, T);
  const int V = T.V();
  label.assign((size_t)n, 0);
  max_label = 0;
  if (V == 0) return;
  // centroids: sequential float sums in ascending point index order
  std::vector<float> cen((size_t)V * 3), nrm((size_t)V * 3);
  for (int v = 0; v < V; ++v) {
    float sx = 0, sy = 0, sz = 0;
    for (int k = T.start[v]; k < T.start[v + 1]; ++k) {
      const float* p = xyz + (int64_t)T.point_idx[k] * stride;
      sx = sx + p[0]; sy = sy + p[1]; sz = sz + p[2];
    }
    const int cnt = T.start[v + 1] - T.start[v];
    cen[3 * v] = sx / cnt; cen[3 * v + 1] = sy / cnt; cen[3 * v + 2] = sz / cnt;
  }
  // 26-neighbour table
  std::unordered_map<uint64_t, int> by_code;
  by_code.reserve((size_t)V * 2);
  for (int v = 0; v < V; ++v) by_code[vm_morton(T.key[3 * v], T.key[3 * v + 1], T.key[3 * v + 2])] = v;
  const uint32_t lim = 1u << T.depth;
  std::vector<int> nbr((size_t)V * 26, -1);
  for (int v = 0; v < V; ++v) {
    float pts[27 * 3];
    int np = 1;
    pts[0] = cen[3 * v]; pts[1] = cen[3 * v + 1]; pts[2] = cen[3 * v + 2];
    for (int o = 0; o < 26; ++o) {
      int dx, dy, dz;
      vccs_offset(o, &dx, &dy, &dz);
      const uint32_t nx = T.key[3 * v] + (uint32_t)dx, ny = T.key[3 * v + 1] + (uint32_t)dy, nz = T.key[3 * v + 2] + (uint32_t)dz;
      int t = -1;
      if (nx < lim && ny < lim && nz < lim) {
        auto it = by_code.find(vm_morton(nx, ny, nz));
        if (it != by_code.end()) t = it->second;
      }
      nbr[(size_t)26 * v + o] = t;
      if (t >= 0) { pts[3 * np] = cen[3 * t]; pts[3 * np + 1] = cen[3 * t + 1]; pts[3 * np + 2] = cen[3 * t + 2]; ++np; }
    }
    vccs_normal_from_points(pts, np, &nrm[3 * v]);
  }
  // seeds
  const float seed = P.seed_size;
  const float mn[3] = {(float)T.min[0], (float)T.min[1], (float)T.min[2]};
  std::vector<uint64_t> cell((size_t)V);
  for (int v = 0; v < V; ++v) cell[v] = vccs_seed_cell(cen[3 * v], cen[3 * v + 1], cen[3 * v + 2], mn[0], mn[1], mn[2], seed);
  std::vector<uint64_t> uniq(cell);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  const int K = (int)uniq.size();
  std::vector<uint64_t> seed_key((size_t)K, ~0ull);
  for (int v = 0; v < V; ++v) {
    const int k = (int)(std::lower_bound(uniq.begin(), uniq.end(), cell[v]) - uniq.begin());
    const float d2 = vccs_cell_center_d2(cell[v], cen[3 * v], cen[3 * v + 1], cen[3 * v + 2], mn[0], mn[1], mn[2], seed);
    const uint64_t key = ((uint64_t)vm_bits(d2) << 32) | (uint64_t)(uint32_t)v;
    if (key < seed_key[k]) seed_key[k] = key;
  }
  // expansion passes
  const int Tn = (int)(1.8f * P.seed_size / P.voxel_size);
  const float w_s_over_seed = P.spatial_impt / P.seed_size;
  const float w_n = P.normal_impt;
  std::vector<float> sc((size_t)K * 3, 0.f), sn((size_t)K * 3, 0.f);
  std::vector<int> lab((size_t)V, -1), lab2((size_t)V, -1);
  std::vector<float> dist((size_t)V, 3.0e38f), dist2((size_t)V, 3.0e38f);
  std::vector<long long> sums((size_t)K * 6);
  std::vector<unsigned> count((size_t)K);
  for (int pass = 0; pass < 6; ++pass) {
    if (pass > 0) {
      std::fill(seed_key.begin(), seed_key.end(), ~0ull);
      for (int v = 0; v < V; ++v) {
        const int l = lab[v];
        if (l < 0) continue;
        const float dx = cen[3 * v] - sc[3 * l], dy = cen[3 * v + 1] - sc[3 * l + 1], dz = cen[3 * v + 2] - sc[3 * l + 2];
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        const uint64_t key = ((uint64_t)vm_bits(d2) << 32) | (uint64_t)(uint32_t)v;
        if (key < seed_key[l]) seed_key[l] = key;
      }
    }
    std::fill(lab.begin(), lab.end(), -1);
    std::fill(dist.begin(), dist.end(), 3.0e38f);
    for (int k = 0; k < K; ++k) {
      if (seed_key[k] == ~0ull) { for (int a = 0; a < 3; ++a) { sc[3 * k + a] = 0.f; sn[3 * k + a] = 0.f; } continue; }
      const uint32_t v = (uint32_t)seed_key[k];
      lab[v] = k; dist[v] = 0.0f;
      for (int a = 0; a < 3; ++a) { sc[3 * k + a] = cen[3 * v + a]; sn[3 * k + a] = nrm[3 * v + a]; }
    }
    for (int it = 0; it < Tn; ++it) {
      for (int v = 0; v < V; ++v) {
        int best_l = lab[v];
        float best_d = dist[v];
        for (int o = 0; o < 26; ++o) {
          const int t = nbr[(size_t)26 * v + o];
          if (t < 0) continue;
          const int l = lab[t];
          if (l < 0 || l == lab[v]) continue;
          const float d = vccs_distance(&cen[3 * v], &nrm[3 * v], &sc[3 * l], &sn[3 * l], w_s_over_seed, w_n);
          if (d < best_d || (d == best_d && l < best_l)) { best_d = d; best_l = l; }
        }
        lab2[v] = best_l; dist2[v] = best_d;
      }
      lab.swap(lab2); dist.swap(dist2);
      std::fill(sums.begin(), sums.end(), 0);
      std::fill(count.begin(), count.end(), 0u);
      for (int v = 0; v < V; ++v) {
        const int l = lab[v];
        if (l < 0) continue;
        for (int a = 0; a < 3; ++a) { sums[(size_t)6 * l + a] += vccs_fix_pos(cen[3 * v + a]); sums[(size_t)6 * l + 3 + a] += vccs_fix_nrm(nrm[3 * v + a]); }
        count[l]++;
      }
      for (int k = 0; k < K; ++k)
        if (count[k]) vccs_state_from_sums(&sums[(size_t)6 * k], count[k], &sc[3 * k], &sn[3 * k]);
    }
  }
  for (int64_t i = 0; i < n; ++i) {
    const int v = T.point_voxel[(size_t)i];
    label[(size_t)i] = (v < 0 || lab[v] < 0) ? 0 : lab[v] + 1;
  }
  max_label = K;
}


// ==================================================================================================================
// vccs_mode 1 ("PCL order"): pcl::SupervoxelClustering 1.8.1 restated step by step (recalled from upstream, NOT verifiable
// here: parity with PCL stays unpinned).  What follows PCL: normals from the 2-ring of voxel centroids through the single-pass
// covariance of computePointNormal, flipped towards (0,0,0); seeds = the voxel nearest to the centre of every occupied
// seed_res cell, cells in ascending Morton order (getOccupiedVoxelCenters), a seed kept only if more than
// 0.05 * (seed/2)^2 * pi / res^2 voxels lie within seed/2 of it; expansion for (int)(1.8 seed / res) - 1 rounds in which the
// supervoxels take their turns ONE AFTER THE OTHER in label order, each offering its current centroid to the 27-neighbourhood
// of the leaves it still owns at its turn (a voxel goes to the offer strictly below its recorded distance; recorded distances
// persist), centroids updated after every round, supervoxels left without voxels removed; refineSupervoxels(5): re-seed, reset
// every voxel, expand again.  Known divergences (csrc/vccs.hip lists them): the lattice is the class's own octree's (PCL's
// adjacency octree anchors at the cloud's minimum), the seed grid is anchored at that lattice's corner, re-seeding takes the
// nearest of the supervoxel's OWN voxels (PCL asks a kd-tree for the nearest of all), refineNormals is skipped, centroid sums
// are integer fixed point (order-free on the GPU).
void vccs_pcl_supervoxels(const float* xyz, int64_t n, int stride, const Params& P, std::vector<int>& label, int& max_label) {
  VoxelTable T;
  build_voxel_table_bbox(xyz, n, stride, P.voxel_size, T);   // the adjacency octree's own lattice (round 5): box from the cloud's bounding box
  const int V = T.V();
  label.assign((size_t)n, 0);
  max_label = 0;
  if (V == 0) return;
  std::vector<float> cen((size_t)V * 3), nrm((size_t)V * 3);
  for (int v = 0; v < V; ++v) {
    float sx = 0, sy = 0, sz = 0;
    for (int k = T.start[v]; k < T.start[v + 1]; ++k) {
      const float* p = xyz + (int64_t)T.point_idx[k] * stride;
      sx = sx + p[0]; sy = sy + p[1]; sz = sz + p[2];
    }
    const int cnt = T.start[v + 1] - T.start[v];
    cen[3 * v] = sx / cnt; cen[3 * v + 1] = sy / cnt; cen[3 * v + 2] = sz / cnt;
  }
  std::unordered_map<uint64_t, int> by_code;
  by_code.reserve((size_t)V * 2);
  for (int v = 0; v < V; ++v) by_code[vm_morton(T.key[3 * v], T.key[3 * v + 1], T.key[3 * v + 2])] = v;
  const uint32_t lim = 1u << T.depth;
  auto find = [&](uint32_t x, uint32_t y, uint32_t z) -> int {
    if (!(x < lim && y < lim && z < lim)) return -1;
    auto it = by_code.find(vm_morton(x, y, z));
    return it == by_code.end() ? -1 : it->second;
  };
  // 27-neighbour table in PCL's computeNeighbors order (the leaf itself included)
  std::vector<int> n27((size_t)V * 27, -1);
  for (int v = 0; v < V; ++v)
    for (int o = 0; o < 27; ++o) {
      int dx, dy, dz;
      vccs_offset27(o, &dx, &dy, &dz);
      n27[(size_t)27 * v + o] = find(T.key[3 * v] + (uint32_t)dx, T.key[3 * v + 1] + (uint32_t)dy, T.key[3 * v + 2] + (uint32_t)dz);
    }
  // normals: indices = [v] + for every neighbour t: [t] + the neighbours of t (computeVoxelData)
  {
    std::vector<VccsAccu> A1((size_t)V);
    for (int t = 0; t < V; ++t) {
      vccs_accu_zero(&A1[t]);
      for (int o = 0; o < 27; ++o) { const int u = n27[(size_t)27 * t + o]; if (u >= 0) vccs_accu_point(&A1[t], &cen[3 * u]); }
    }
    for (int v = 0; v < V; ++v) {
      VccsAccu A;
      vccs_accu_zero(&A);
      vccs_accu_point(&A, &cen[3 * v]);
      for (int o = 0; o < 27; ++o) {
        const int t = n27[(size_t)27 * v + o];
        if (t < 0) continue;
        vccs_accu_point(&A, &cen[3 * t]);
        vccs_accu_add(&A, &A1[t]);
      }
      vccs_accu_normal(&A, &cen[3 * v], &nrm[3 * v]);
    }
  }
  // seeds
  const float seed = P.seed_size, res = P.voxel_size;
  const float mn[3] = {(float)T.min[0], (float)T.min[1], (float)T.min[2]};
  std::vector<uint64_t> cell((size_t)V), mcell((size_t)V);
  for (int v = 0; v < V; ++v) {
    cell[v] = vccs_seed_cell(cen[3 * v], cen[3 * v + 1], cen[3 * v + 2], mn[0], mn[1], mn[2], seed);
    mcell[v] = vm_morton((uint32_t)((cell[v] >> 42) & 0x1fffff), (uint32_t)((cell[v] >> 21) & 0x1fffff), (uint32_t)(cell[v] & 0x1fffff));
  }
  std::vector<uint64_t> uniq(mcell);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  const int K0 = (int)uniq.size();
  std::vector<uint64_t> seed_key((size_t)K0, ~0ull);
  for (int v = 0; v < V; ++v) {
    const int k = (int)(std::lower_bound(uniq.begin(), uniq.end(), mcell[v]) - uniq.begin());
    const float d2 = vccs_cell_center_d2(cell[v], cen[3 * v], cen[3 * v + 1], cen[3 * v + 2], mn[0], mn[1], mn[2], seed);
    const uint64_t key = ((uint64_t)vm_bits(d2) << 32) | (uint64_t)(uint32_t)v;
    if (key < seed_key[k]) seed_key[k] = key;
  }
  // seed rejection (selectInitialSupervoxelSeeds): count the voxels within seed / 2 of the seed voxel's centroid
  const float rad = 0.5f * seed, rad2 = rad * rad, min_points = vccs_seed_min_points(seed, res);
  const int R = (int)(rad / res) + 1;
  std::vector<int> seeds;
  for (int k = 0; k < K0; ++k) {
    const int s = (int)(uint32_t)seed_key[k];
    int num = 0;
    for (int dz = -R; dz <= R; ++dz)
      for (int dy = -R; dy <= R; ++dy)
        for (int dx = -R; dx <= R; ++dx) {
          const int u = find(T.key[3 * s] + (uint32_t)dx, T.key[3 * s + 1] + (uint32_t)dy, T.key[3 * s + 2] + (uint32_t)dz);
          if (u < 0) continue;
          const float ex = cen[3 * u] - cen[3 * s], ey = cen[3 * u + 1] - cen[3 * s + 1], ez = cen[3 * u + 2] - cen[3 * s + 2];
          if ((ex * ex + ey * ey) + ez * ez < rad2) ++num;
        }
    if ((float)num > min_points) seeds.push_back(s);
  }
  const int K = (int)seeds.size();
  max_label = 0;
  if (K == 0) return;
  // expansion
  const int depth = (int)(1.8f * seed / res);
  const float w_s_over_seed = P.spatial_impt / seed, w_n = P.normal_impt;
  const float FMAX = 3.4028235e38f;
  std::vector<int> owner((size_t)V, -1);
  std::vector<float> dist((size_t)V, FMAX);
  std::vector<float> sc((size_t)K * 3), sn((size_t)K * 3);
  std::vector<char> alive((size_t)K, 1);
  std::vector<long long> sums((size_t)K * 6);
  std::vector<unsigned> count((size_t)K);
  auto update_centroids = [&]() {
    std::fill(sums.begin(), sums.end(), 0);
    std::fill(count.begin(), count.end(), 0u);
    for (int v = 0; v < V; ++v) {
      const int l = owner[v];
      if (l < 0) continue;
      for (int a = 0; a < 3; ++a) { sums[(size_t)6 * l + a] += vccs_fix_pos(cen[3 * v + a]); sums[(size_t)6 * l + 3 + a] += vccs_fix_nrm(nrm[3 * v + a]); }
      count[l]++;
    }
    for (int k = 0; k < K; ++k) {
      if (!alive[k]) continue;
      if (count[k] == 0) { alive[k] = 0; continue; }   // a supervoxel without voxels is removed for good
      vccs_state_from_sums(&sums[(size_t)6 * k], count[k], &sc[3 * k], &sn[3 * k]);
    }
  };
  auto expand_all = [&]() {
    std::vector<std::vector<int>> leaves((size_t)K);
    std::vector<char> taken((size_t)V);
    for (int i = 1; i < depth; ++i) {
      for (int k = 0; k < K; ++k) leaves[k].clear();
      for (int v = 0; v < V; ++v) if (owner[v] >= 0) leaves[owner[v]].push_back(v);   // ascending voxel id = idx_ order
      std::fill(taken.begin(), taken.end(), 0);
      for (int k = 0; k < K; ++k) {
        if (!alive[k]) continue;
        for (int leaf : leaves[k]) {
          // taken by an earlier supervoxel of this round: erased from leaves_; a voxel this supervoxel wins (back) joins
          // leaves_ only after its turn (new_owned), so it is not expanded from in this round either
          if (taken[leaf]) continue;
          for (int o = 0; o < 27; ++o) {
            const int nb = n27[(size_t)27 * leaf + o];
            if (nb < 0 || owner[nb] == k) continue;
            const float d = vccs_distance(&cen[3 * nb], &nrm[3 * nb], &sc[3 * k], &sn[3 * k], w_s_over_seed, w_n);
            if (d < dist[nb]) { dist[nb] = d; owner[nb] = k; taken[nb] = 1; }
          }
        }
      }
      update_centroids();
    }
  };
  for (int k = 0; k < K; ++k) {
    owner[seeds[k]] = k;   // addLeaf: the recorded distance stays at its initial maximum
    for (int a = 0; a < 3; ++a) { sc[3 * k + a] = cen[3 * seeds[k] + a]; sn[3 * k + a] = nrm[3 * seeds[k] + a]; }
  }
  expand_all();
  std::vector<VccsAccu> A1((size_t)V);
  for (int pass = 0; pass < 5; ++pass) {   // refineSupervoxels(5): refineNormals of every supervoxel, reseedSupervoxels, expandSupervoxels
    // SupervoxelHelper::refineNormals: the normal of every leaf again, from the leaves of ITS supervoxel only -- indices = for every
    // neighbour t of the leaf (itself included) owned by the supervoxel: [t] + the neighbours of t owned by it (round 5)
    for (int t = 0; t < V; ++t) {
      vccs_accu_zero(&A1[t]);
      const int k = owner[t];
      if (k < 0) continue;
      for (int o = 0; o < 27; ++o) { const int u = n27[(size_t)27 * t + o]; if (u >= 0 && owner[u] == k) vccs_accu_point(&A1[t], &cen[3 * u]); }
    }
    for (int v = 0; v < V; ++v) {
      const int k = owner[v];
      if (k < 0) continue;
      VccsAccu A;
      vccs_accu_zero(&A);
      for (int o = 0; o < 27; ++o) {
        const int t = n27[(size_t)27 * v + o];
        if (t < 0 || owner[t] != k) continue;
        vccs_accu_point(&A, &cen[3 * t]);
        vccs_accu_add(&A, &A1[t]);
      }
      vccs_accu_normal(&A, &cen[3 * v], &nrm[3 * v]);
    }
    // reseedSupervoxels: the voxel nearest to the supervoxel's centroid among ALL voxels (vccs_nearest_voxel); supervoxels take their
    // new seed in label order, so a voxel two of them name goes to the later one (addLeaf overwrites owner_) and the earlier one is left
    // without a voxel -- removed at the next centroid update.  (PCL keeps the leaf in the earlier helper's own set as well and would let
    // it expand from a voxel it does not own: not followed.)
    std::vector<uint64_t> rk((size_t)K, ~0ull);
    for (int k = 0; k < K; ++k) {
      if (!alive[k]) continue;
      const uint32_t kx = vm_axis_key(sc[3 * k], T.min[0], T.resolution), ky = vm_axis_key(sc[3 * k + 1], T.min[1], T.resolution),
                     kz = vm_axis_key(sc[3 * k + 2], T.min[2], T.resolution);
      rk[k] = vccs_nearest_voxel(&sc[3 * k], kx, ky, kz, lim, res, find, [&](int v) { return (const float*)&cen[3 * (size_t)v]; });
    }
    std::fill(owner.begin(), owner.end(), -1);
    std::fill(dist.begin(), dist.end(), FMAX);
    for (int k = 0; k < K; ++k) if (alive[k] && rk[k] != ~0ull) owner[(uint32_t)rk[k]] = k;   // centroids stay what they were
    expand_all();
  }
  for (int64_t i = 0; i < n; ++i) {
    const int v = T.point_voxel[(size_t)i];
    label[(size_t)i] = (v < 0 || owner[v] < 0) ? 0 : owner[v] + 1;
  }
  for (int k = 0; k < K; ++k) if (alive[k]) max_label = k + 1;   // getMaxLabel: the largest label still in use
}

}  // namespace refcpu
