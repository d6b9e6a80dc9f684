// vccs_common.h -- arithmetic of the VCCS-style supervoxel stage (vccs.hip), shared with the oracle's restatement
// (oracle/refcpu_vccs.cpp) the same way vgs_math.h is: float/double IEEE primitives only, no libm, so host and
// device agree bit for bit.  See vccs.hip for what the stage replaces (pcl::SupervoxelClustering, SS:265-284)
// and why its parity with PCL is unpinned.
#ifndef VCCS_COMMON_H_
#define VCCS_COMMON_H_

#include "vgs_math.h"

// 26-neighbourhood, dz outermost, the centre skipped
VGS_HD void vccs_offset(int o, int* dx, int* dy, int* dz) {
  const int k = o < 13 ? o : o + 1;
  *dx = k % 3 - 1; *dy = (k / 3) % 3 - 1; *dz = k / 9 - 1;
}

// normal of a voxel: smallest-eigenvalue direction of the covariance of the centroids of the voxel (pts[0..2]) and
// its occupied neighbours, flipped towards the viewpoint (0,0,0); zero when fewer than 3 centroids are available
// (the tail of vccs_normal_from_points: C holds the six upper sums, p0 is the voxel's own centroid; vccs.hip's tile kernel gathers the
// centroids twice in the same order instead of keeping them in an array of 81 floats, which lives in scratch memory on the GPU)
VGS_HD void vccs_normal_finish(float* C, const float* p0, float* n) {
  C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
  float evecs[9], evals[3];
  vm_eigen33(C, evecs, evals);
  float nx = evecs[0], ny = evecs[3], nz = evecs[6];
  if ((nx * (0.f - p0[0]) + ny * (0.f - p0[1]) + nz * (0.f - p0[2])) < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
  n[0] = nx; n[1] = ny; n[2] = nz;
}
VGS_HD void vccs_normal_from_points(const float* pts, int np, float* n) {
  n[0] = 0.f; n[1] = 0.f; n[2] = 0.f;
  if (np < 3) return;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int k = 0; k < np; ++k) { sx = sx + pts[3 * k]; sy = sy + pts[3 * k + 1]; sz = sz + pts[3 * k + 2]; }
  const float mx = sx / np, my = sy / np, mz = sz / np;
  float C[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < np; ++k) {
    const float d0 = pts[3 * k] - mx, d1 = pts[3 * k + 1] - my, d2 = pts[3 * k + 2] - mz;
    C[0] = C[0] + d0 * d0; C[1] = C[1] + d0 * d1; C[2] = C[2] + d0 * d2;
    C[4] = C[4] + d1 * d1; C[5] = C[5] + d1 * d2; C[8] = C[8] + d2 * d2;
  }
  vccs_normal_finish(C, pts, n);
}

VGS_HD uint64_t vccs_seed_cell(float cx, float cy, float cz, float min_x, float min_y, float min_z, float seed) {
  const double s = (double)seed;
  long long ix = (long long)(((double)cx - (double)min_x) / s);
  long long iy = (long long)(((double)cy - (double)min_y) / s);
  long long iz = (long long)(((double)cz - (double)min_z) / s);
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (iz < 0) iz = 0;
  return ((uint64_t)(ix & 0x1fffff) << 42) | ((uint64_t)(iy & 0x1fffff) << 21) | (uint64_t)(iz & 0x1fffff);
}

VGS_HD float vccs_cell_center_d2(uint64_t cell, float cx, float cy, float cz, float min_x, float min_y, float min_z, float seed) {
  const double s = (double)seed;
  const float ccx = (float)((double)min_x + ((double)((cell >> 42) & 0x1fffff) + 0.5) * s);
  const float ccy = (float)((double)min_y + ((double)((cell >> 21) & 0x1fffff) + 0.5) * s);
  const float ccz = (float)((double)min_z + ((double)(cell & 0x1fffff) + 0.5) * s);
  const float dx = cx - ccx, dy = cy - ccy, dz = cz - ccz;
  return (dx * dx + dy * dy) + dz * dz;
}

// D = w_s * |dx| / seed_res + w_n * (1 - |n1 . n2|)   (colour term is identically zero here)
VGS_HD float vccs_distance(const float* c, const float* n, const float* sc, const float* sn, float w_s_over_seed, float w_n) {
  const float dx = c[0] - sc[0], dy = c[1] - sc[1], dz = c[2] - sc[2];
  const float ds = vm_sqrt((dx * dx + dy * dy) + dz * dz);
  const float dn = 1.0f - vm_abs((n[0] * sn[0] + n[1] * sn[1]) + n[2] * sn[2]);
  return ds * w_s_over_seed + w_n * dn;
}

// ------------------------------------------------------------------------------------------------------------------
// Distinct labels of a neighbourhood as successive minima (vccs.hip: vccs_best_offer, k_pclt_sweep; checked on the host by
// tests/test_enum_arith.py).  key = (label ^ ref) - 1 with ref = the voxel's own label (an unowned voxel: 2^31 - 2) is
// 0xffffffff for the own label, >= 2^31 for "no neighbour" (-1) and < 2^31 for every other label (labels stay below 2^31 - 2).
// The next key above the ones already taken is  off + min over the keys of (key - off)  in unsigned arithmetic with
// off = previous + 1: a key already taken wraps to the top, and the voxel's own key is always part of the minimum, so that a
// neighbourhood whose labels have all been taken ends at 0xffffffff instead of wrapping round to one of them.
VGS_HD uint32_t vccs_enum_ref(int own) { return own >= 0 ? (uint32_t)own : 0x7ffffffeu; }
VGS_HD uint32_t vccs_enum_key(int label, uint32_t ref) { return ((uint32_t)label ^ ref) - 1u; }
VGS_HD int vccs_enum_label(uint32_t key, uint32_t ref) { return (int)((key + 1u) ^ ref); }
VGS_HD bool vccs_enum_valid(uint32_t key) { return key < 0x80000000u; }
// smallest key >= off (returned as a key, not as a difference); keys: n values
VGS_HD uint32_t vccs_enum_next(const uint32_t* keys, int n, uint32_t off) {
  uint32_t m = 0xffffffffu - off;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int o = 0; o < n; ++o) { const uint32_t d = keys[o] - off; m = d < m ? d : m; }
  return m + off;
}

// fixed point so that the per-supervoxel sums do not depend on the order of the atomic adds
VGS_HD long long vccs_fix_pos(float x) { return (long long)__builtin_rint((double)x * 65536.0); }
VGS_HD long long vccs_fix_nrm(float x) { return (long long)__builtin_rint((double)x * 1048576.0); }

VGS_HD void vccs_state_from_sums(const long long* sums, unsigned int count, float* c, float* n) {
  const double inv = 1.0 / (double)count;
  for (int a = 0; a < 3; ++a) c[a] = (float)(((double)sums[a] * inv) / 65536.0);
  float m[3];
  for (int a = 0; a < 3; ++a) m[a] = (float)(((double)sums[3 + a] * inv) / 1048576.0);
  const float len = vm_sqrt((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]);
  if (len > 0.f) { n[0] = m[0] / len; n[1] = m[1] / len; n[2] = m[2] / len; }
  else { n[0] = 0.f; n[1] = 0.f; n[2] = 0.f; }
}

// ------------------------------------------------------------------------------------------------------------------
// vccs_mode 1 ("PCL order"): pcl::SupervoxelClustering's own steps restated (PCL 1.8.1, recalled from upstream; unpinned --
// see vccs.hip).  Arithmetic shared with the oracle (refcpu_vccs.cpp: vccs_pcl_supervoxels) so that both agree bit for bit.
// ------------------------------------------------------------------------------------------------------------------
// computeMeanAndCovarianceMatrix's single-pass accumulator (float, as PCL's Scalar): sums of x*x, x*y, x*z, y*y, y*z, z*z,
// x, y, z over a multiset of voxel centroids
struct VccsAccu { float a[9]; float n; };
VGS_HD void vccs_accu_zero(VccsAccu* A) { for (int k = 0; k < 9; ++k) A->a[k] = 0.f; A->n = 0.f; }
VGS_HD void vccs_accu_point(VccsAccu* A, const float* p) {
  A->a[0] = A->a[0] + p[0] * p[0]; A->a[1] = A->a[1] + p[0] * p[1]; A->a[2] = A->a[2] + p[0] * p[2];
  A->a[3] = A->a[3] + p[1] * p[1]; A->a[4] = A->a[4] + p[1] * p[2]; A->a[5] = A->a[5] + p[2] * p[2];
  A->a[6] = A->a[6] + p[0]; A->a[7] = A->a[7] + p[1]; A->a[8] = A->a[8] + p[2];
  A->n = A->n + 1.0f;
}
VGS_HD void vccs_accu_add(VccsAccu* A, const VccsAccu* B) { for (int k = 0; k < 9; ++k) A->a[k] = A->a[k] + B->a[k]; A->n = A->n + B->n; }
// computePointNormal + flipNormalTowardsViewpoint(0, 0, 0) + normalize: cov = E[pp^T] - mean mean^T, smallest-eigenvalue
// direction; fewer than three indices give no normal (PCL sets NaN; zero here: a zero normal gives the largest normal distance)
VGS_HD void vccs_accu_normal(const VccsAccu* A, const float* self, float* n) {
  n[0] = 0.f; n[1] = 0.f; n[2] = 0.f;
  if (A->n < 3.0f) return;
  float m[9];
  for (int k = 0; k < 9; ++k) m[k] = A->a[k] / A->n;
  float C[9];
  C[0] = m[0] - m[6] * m[6]; C[1] = m[1] - m[6] * m[7]; C[2] = m[2] - m[6] * m[8];
  C[4] = m[3] - m[7] * m[7]; C[5] = m[4] - m[7] * m[8]; C[8] = m[5] - m[8] * m[8];
  C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
  float evecs[9], evals[3];
  vm_eigen33(C, evecs, evals);
  float nx = evecs[0], ny = evecs[3], nz = evecs[6];
  if ((nx * (0.f - self[0]) + ny * (0.f - self[1])) + nz * (0.f - self[2]) < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
  const float len = vm_sqrt((nx * nx + ny * ny) + nz * nz);
  if (len > 0.f) { n[0] = nx / len; n[1] = ny / len; n[2] = nz / len; }
}
// the 27 cells of a leaf's neighbour list in PCL's computeNeighbors order (dx outermost, the leaf itself included)
VGS_HD void vccs_offset27(int o, int* dx, int* dy, int* dz) { *dx = o / 9 - 1; *dy = (o / 3) % 3 - 1; *dz = o % 3 - 1; }
// index of offset (dx, dy, dz) != 0 in the 26-neighbour table of vccs_offset (dz outermost, centre skipped)
VGS_HD int vccs_index26(int dx, int dy, int dz) { const int k = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1); return k < 13 ? k : k - 1; }
// selectInitialSupervoxelSeeds: a seed needs more than this many voxels within half a seed size
VGS_HD float vccs_seed_min_points(float seed, float res) {
  const float r = 0.5f * seed;
  return 0.05f * (r * r) * 3.1415926536f / (res * res);
}

// reseedSupervoxels (SupervoxelClustering::reseedSupervoxels, PCL 1.8.1 recalled): the new seed of a supervoxel is the voxel whose
// centroid is nearest to the supervoxel's centroid among ALL voxels (PCL asks voxel_kdtree_ for one neighbour; FLANN's tie order is
// replaced by (distance, voxel id)).  Here: the lattice cells around the centroid's own cell, shell by shell (Chebyshev radius r).  A
// voxel's centroid lies in its cell, the supervoxel's centroid in its own, so everything beyond shell r is at least r * res away: the
// search ends once the best squared distance is below that bound (less a thousandth of a voxel for the rounding of the means).  Oracle
// and device run this same procedure, so they agree whatever the coordinates; it IS the nearest of all wherever float means stay within
// a thousandth of a voxel of their cells.  find(x, y, z) -> voxel id or -1; cen(v) -> pointer to the voxel's centroid.
VGS_HD bool vccs_shell_settles(float best_d2, int r, float res) {
  const float bound = (float)r * res - 1.0e-3f * res;
  return bound > 0.f && best_d2 < bound * bound;
}
template <class Find, class Cen>
VGS_HD unsigned long long vccs_nearest_voxel(const float* c, uint32_t kx, uint32_t ky, uint32_t kz, uint32_t lim, float res, Find find, Cen cen) {
  unsigned long long best = ~0ull;
  for (int r = 0; ; ++r) {
    for (int dz = -r; dz <= r; ++dz)
      for (int dy = -r; dy <= r; ++dy) {
        const bool face = (dz == -r || dz == r || dy == -r || dy == r);
        for (int dx = -r; dx <= r; dx += (face || r == 0) ? 1 : 2 * r) {   // inside a slab row only the two end cells belong to the shell
          const uint32_t x = kx + (uint32_t)dx, y = ky + (uint32_t)dy, z = kz + (uint32_t)dz;   // (below zero wraps above lim)
          if (!(x < lim && y < lim && z < lim)) continue;
          const int v = find(x, y, z);
          if (v < 0) continue;
          const float* p = cen(v);
          const float ex = p[0] - c[0], ey = p[1] - c[1], ez = p[2] - c[2];
          const unsigned long long key = ((unsigned long long)vm_bits((ex * ex + ey * ey) + ez * ez) << 32) | (unsigned long long)(uint32_t)v;
          best = key < best ? key : best;
        }
      }
    if (best != ~0ull && vccs_shell_settles(vm_from_bits((uint32_t)(best >> 32)), r, res)) break;
    if ((uint32_t)r >= lim) break;
  }
  return best;
}

#endif
