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
  C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
  float evecs[9], evals[3];
  vm_eigen33(C, evecs, evals);
  float nx = evecs[0], ny = evecs[3], nz = evecs[6];
  if ((nx * (0.f - pts[0]) + ny * (0.f - pts[1]) + nz * (0.f - pts[2])) < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
  n[0] = nx; n[1] = ny; n[2] = nz;
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

#endif
