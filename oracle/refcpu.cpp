// oracle/refcpu.cpp -- TEST INFRASTRUCTURE (see refcpu.hpp header note; parity unpinned).
// CPU restatement of the reference hot path.  Citations: VS: = voxel_segmentation.h,
// SS: = supervoxel_segmentation.h, T: = test (all under /root/reference), SURVEY = SURVEY.md.
#include "refcpu.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <unordered_map>

#include "vgs_math.h"  // DevMath: the product's arithmetic specification (oracle -> product, never the reverse)

namespace refcpu {

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// =============================================================================================
// B.1  pcl::octree::OctreePointCloud used without defineBoundingBox (T:51-54)
// =============================================================================================
namespace {
struct Octree {
  double min[3], max[3];
  double res;
  int depth = 0;
  bool defined = false;
  uint64_t shift[3] = {0, 0, 0};  // voxels by which existing keys moved up on each axis (root re-rooting)
  const double eps = (double)std::numeric_limits<float>::epsilon();

  // OctreePointCloud::getKeyBitSize for the empty tree
  void key_bit_size_first() {
    unsigned mk[3];
    for (int a = 0; a < 3; ++a) mk[a] = (unsigned)((max[a] - min[a]) / res);
    unsigned max_voxels = std::max(std::max(std::max(mk[0], mk[1]), mk[2]), 2u);
    depth = std::max(std::min(32u, (unsigned)std::ceil(std::log2((double)max_voxels) - eps)), 0u);
    double side = (double)(1u << depth) * res - eps;
    for (int a = 0; a < 3; ++a) {
      double over = (side - (max[a] - min[a])) / 2.0;
      min[a] -= over;
      max[a] += over;
    }
  }
  // OctreePointCloud::adoptBoundingBoxToPoint
  void adopt(const float* p) {
    while (true) {
      bool lo[3], hi[3];
      bool any = false;
      for (int a = 0; a < 3; ++a) {
        lo[a] = defined && ((double)p[a] < min[a]);
        hi[a] = defined && ((double)p[a] >= max[a]);
        any = any || lo[a] || hi[a];
      }
      if (!any && defined) break;
      if (defined) {
        double side = (double)(1u << depth) * res;
        for (int a = 0; a < 3; ++a)
          if (!hi[a]) { min[a] -= side; shift[a] += (1ull << depth); }  // old root becomes the upper child
        depth++;
        side = (double)(1u << depth) * res - eps;
        for (int a = 0; a < 3; ++a) max[a] = min[a] + side;
      } else {
        for (int a = 0; a < 3; ++a) {
          min[a] = (double)p[a] - res / 2;
          max[a] = (double)p[a] + res / 2;
        }
        key_bit_size_first();
        defined = true;
      }
    }
  }
};
}  // namespace

static void build_voxel_table_impl(const float* xyz, int64_t n, int stride, float voxel_size, VoxelTable& T, bool bbox_first);
void build_voxel_table(const float* xyz, int64_t n, int stride, float voxel_size, VoxelTable& T) { build_voxel_table_impl(xyz, n, stride, voxel_size, T, false); }
// pcl::octree::OctreePointCloudAdjacency::addPointsFromInputCloud (the octree pcl::SupervoxelClustering builds for itself, SS:265-284;
// PCL 1.8.1, recalled -- unpinned like B.1): the bounding box of the finite points is computed first (float min / max) and handed to
// defineBoundingBox, whose getKeyBitSize on the still empty tree pads it symmetrically to the cube of 2^depth voxels (the same code
// that pads the box around the first point above); then the points are inserted as in B.1, growth included should one not fit.
void build_voxel_table_bbox(const float* xyz, int64_t n, int stride, float voxel_size, VoxelTable& T) { build_voxel_table_impl(xyz, n, stride, voxel_size, T, true); }
static void build_voxel_table_impl(const float* xyz, int64_t n, int stride, float voxel_size, VoxelTable& T, bool bbox_first) {
  Octree oc;
  oc.res = (double)voxel_size;  // ctor takes double(voxel_size) (T:51, VS:84)
  if (bbox_first) {
    float mn[3] = {3.4028235e38f, 3.4028235e38f, 3.4028235e38f}, mx[3] = {-3.4028235e38f, -3.4028235e38f, -3.4028235e38f};
    bool any = false;
    for (int64_t i = 0; i < n; ++i) {
      const float* p = xyz + i * stride;
      if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;
      any = true;
      for (int a = 0; a < 3; ++a) { mn[a] = p[a] < mn[a] ? p[a] : mn[a]; mx[a] = p[a] > mx[a] ? p[a] : mx[a]; }
    }
    if (any) {
      for (int a = 0; a < 3; ++a) { oc.min[a] = (double)mn[a]; oc.max[a] = (double)mx[a]; }
      oc.key_bit_size_first();
      oc.defined = true;
    }
  }
  // addPointsFromInputCloud: points in index order; non-finite points skipped.  Each point's key is
  // generated with the bounding box AS IT IS WHEN THE POINT IS INSERTED (genOctreeKeyforPoint); every
  // later growth step that lowers min on an axis re-roots the tree with the old root in the upper
  // half, i.e. adds 2^depth to all existing keys on that axis.  Restated with per-point key +
  // (final shift - shift at insertion).
  std::vector<uint32_t> pk((size_t)n * 3, 0);
  std::vector<uint64_t> pshift;  // 3 per epoch
  std::vector<int> pepoch((size_t)n, -1);
  if (oc.defined) { pshift.push_back(oc.shift[0]); pshift.push_back(oc.shift[1]); pshift.push_back(oc.shift[2]); }   // (bbox_first: the box exists before the first point)
  for (int64_t i = 0; i < n; ++i) {
    const float* p = xyz + i * stride;
    if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;
    int old_depth = oc.depth;
    bool was_defined = oc.defined;
    oc.adopt(p);
    if (!was_defined || oc.depth != old_depth) {
      pshift.push_back(oc.shift[0]); pshift.push_back(oc.shift[1]); pshift.push_back(oc.shift[2]);
    }
    pepoch[(size_t)i] = (int)pshift.size() / 3 - 1;
    for (int a = 0; a < 3; ++a) pk[(size_t)i * 3 + a] = vm_axis_key(p[a], oc.min[a], oc.res);
  }
  for (int a = 0; a < 3; ++a) { T.min[a] = oc.min[a]; T.max[a] = oc.max[a]; }
  T.resolution = oc.res;
  T.depth = oc.depth;
  T.point_voxel.assign((size_t)n, -1);
  std::vector<std::pair<uint64_t, int>> code_idx;
  code_idx.reserve((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    int e = pepoch[(size_t)i];
    if (e < 0) continue;
    uint32_t k[3];
    for (int a = 0; a < 3; ++a) k[a] = (uint32_t)(pk[(size_t)i * 3 + a] + (oc.shift[a] - pshift[(size_t)e * 3 + a]));
    code_idx.emplace_back(vm_morton(k[0], k[1], k[2]), (int)i);
  }
  // LeafNodeIterator (PCL 1.8.1): depth first, children popped 7 -> 0 => descending Morton code;
  // inside a leaf: insertion (ascending index) order.
  std::sort(code_idx.begin(), code_idx.end(), [](const std::pair<uint64_t, int>& a, const std::pair<uint64_t, int>& b) {
    if (a.first != b.first) return a.first > b.first;
    return a.second < b.second;
  });
  T.key.clear(); T.start.clear(); T.point_idx.clear(); T.center.clear();
  const float res_f = voxel_size;  // voxel_resolution_ is a float member (VS:1121)
  const float min_f[3] = {(float)oc.min[0], (float)oc.min[1], (float)oc.min[2]};  // setBoundingBox stores floats (VS:1123)
  uint64_t prev = ~0ull;
  bool first = true;
  for (size_t j = 0; j < code_idx.size(); ++j) {
    uint64_t c = code_idx[j].first;
    if (first || c != prev) {
      uint32_t kx = vm_compact21(c >> 2), ky = vm_compact21(c >> 1), kz = vm_compact21(c);
      T.key.push_back(kx); T.key.push_back(ky); T.key.push_back(kz);
      T.start.push_back((int)T.point_idx.size());
      T.center.push_back(vm_voxel_center(kx, res_f, min_f[0]));
      T.center.push_back(vm_voxel_center(ky, res_f, min_f[1]));
      T.center.push_back(vm_voxel_center(kz, res_f, min_f[2]));
      prev = c;
      first = false;
    }
    T.point_voxel[(size_t)code_idx[j].second] = (int)T.start.size() - 1;
    T.point_idx.push_back(code_idx[j].second);
  }
  T.start.push_back((int)T.point_idx.size());
}

// =============================================================================================
// B.3  pcl::eigen33 in float with libm (RefMath).  DevMath uses vm_eigen33.
// =============================================================================================
namespace {
void roots2_ref(float b, float c, float* roots) {
  roots[0] = 0.0f;
  float d = (float)((double)(b * b) - 4.0 * (double)c);  // Scalar(b*b - 4.0*c): float product, double difference
  if (d < 0.0) d = 0.0f;
  float sd = ::sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}
void roots3_ref(const float* m, float* roots) {
  float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
  float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  float c2 = m[0] + m[4] + m[8];
  if (std::fabs(c0) < std::numeric_limits<float>::epsilon()) { roots2_ref(c2, c1, roots); return; }
  const float s_inv3 = (float)(1.0 / 3.0);
  const float s_sqrt3 = std::sqrt(3.0f);
  float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0f) a_over_3 = 0.0f;
  float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0f) q = 0.0f;
  float rho = std::sqrt(-a_over_3);
  float theta = std::atan2(std::sqrt(-q), half_b) * s_inv3;
  float cos_theta = std::cos(theta);
  float sin_theta = std::sin(theta);
  roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
  if (roots[1] >= roots[2]) {
    std::swap(roots[1], roots[2]);
    if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
  }
  if (roots[0] <= 0) roots2_ref(c2, c1, roots);
}
void cross3(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
float sq3(const float* a) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }
float best_cross_ref(const float* sm, float lambda, float* out) {
  float t[9];
  std::memcpy(t, sm, sizeof(t));
  t[0] -= lambda; t[4] -= lambda; t[8] -= lambda;
  float v1[3], v2[3], v3[3];
  cross3(&t[0], &t[3], v1); cross3(&t[0], &t[6], v2); cross3(&t[3], &t[6], v3);
  float l1 = sq3(v1), l2 = sq3(v2), l3 = sq3(v3);
  const float* v; float l;
  if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
  else { v = v3; l = l3; }
  float s = std::sqrt(l);
  out[0] = v[0] / s; out[1] = v[1] / s; out[2] = v[2] / s;
  return l;
}
void unit_orthogonal_ref(const float* v, float* o) {
  const float prec = 1e-5f;
  bool xs = std::fabs(v[0]) <= std::fabs(v[2]) * prec, ys = std::fabs(v[1]) <= std::fabs(v[2]) * prec;
  if (!xs || !ys) {
    float inv = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1]);
    o[0] = -v[1] * inv; o[1] = v[0] * inv; o[2] = 0;
  } else {
    float inv = 1.0f / std::sqrt(v[1] * v[1] + v[2] * v[2]);
    o[0] = 0; o[1] = -v[2] * inv; o[2] = v[1] * inv;
  }
}
void normalize_ref(float* v) {
  float n2 = sq3(v);
  if (n2 > 0) { float nn = std::sqrt(n2); v[0] /= nn; v[1] /= nn; v[2] /= nn; }
}
void eigen33_ref(const float* mat, float* evecs, float* evals) {
  float scale = 0;
  for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(mat[i]));
  if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
  float sm[9];
  for (int i = 0; i < 9; ++i) sm[i] = mat[i] / scale;
  roots3_ref(sm, evals);
  const float eps = std::numeric_limits<float>::epsilon();
  float c0[3], c1[3], c2[3];
  if ((evals[2] - evals[0]) <= eps) {
    c0[0] = 1; c0[1] = 0; c0[2] = 0; c1[0] = 0; c1[1] = 1; c1[2] = 0; c2[0] = 0; c2[1] = 0; c2[2] = 1;
  } else if ((evals[1] - evals[0]) <= eps) {
    best_cross_ref(sm, evals[2], c2);
    unit_orthogonal_ref(c2, c1);
    cross3(c1, c2, c0);
  } else if ((evals[2] - evals[1]) <= eps) {
    best_cross_ref(sm, evals[0], c0);
    unit_orthogonal_ref(c0, c1);
    cross3(c0, c1, c2);
  } else {
    float mmax[3];
    unsigned min_el = 2, max_el = 2;
    mmax[2] = best_cross_ref(sm, evals[2], c2);
    mmax[1] = best_cross_ref(sm, evals[1], c1);
    min_el = mmax[1] <= mmax[min_el] ? 1u : min_el;
    max_el = mmax[1] > mmax[max_el] ? 1u : max_el;
    mmax[0] = best_cross_ref(sm, evals[0], c0);
    min_el = mmax[0] <= mmax[min_el] ? 0u : min_el;
    max_el = mmax[0] > mmax[max_el] ? 0u : max_el;
    unsigned mid_el = 3 - min_el - max_el;
    float* col[3] = {c0, c1, c2};
    float tmp[3];
    cross3(col[(min_el + 1) % 3], col[(min_el + 2) % 3], tmp);
    normalize_ref(tmp);
    std::memcpy(col[min_el], tmp, sizeof(tmp));
    cross3(col[(mid_el + 1) % 3], col[(mid_el + 2) % 3], tmp);
    normalize_ref(tmp);
    std::memcpy(col[mid_el], tmp, sizeof(tmp));
  }
  for (int r = 0; r < 3; ++r) { evecs[r * 3] = c0[r]; evecs[r * 3 + 1] = c1[r]; evecs[r * 3 + 2] = c2[r]; }
  evals[0] *= scale; evals[1] *= scale; evals[2] *= scale;
}

// VS:1169-1219 / SS:796-837 with the promotions of the source expressions
void eigen_features_ref(const float* ev, bool svgs, float* F) {
  if (ev[0] == 0 && ev[1] == 0 && ev[2] == 0) { for (int i = 0; i < 8; ++i) F[i] = 0; return; }
  double s = std::sqrt(std::pow((double)ev[0], 2) + std::pow((double)ev[1], 2) + std::pow((double)ev[2], 2));
  float e3 = (float)((double)ev[0] / s), e2 = (float)((double)ev[1] / s), e1 = (float)((double)ev[2] / s);
  float lin, pla, sca, ani;
  if (e1 == 0) { lin = 0; pla = 1; sca = 0; }
  else { lin = (e1 - e2) / e1; pla = (e2 - e3) / e1; sca = e3 / e1; }
  float cur = e3 / (e1 + e2 + e3);
  if (svgs) ani = (e1 == 0) ? 0.0f : (e1 - e3) / e1;
  else ani = (e2 == 0) ? 0.0f : (e1 - e3) / e1;
  float ent, omn;
  if (e1 * e2 * e3 == 0) ent = 0;
  else ent = -1 * (e1 * std::log(e1) + e2 * std::log(e2) + e3 * std::log(e3));
  omn = std::pow((float)(e1 * e2 * e3), (float)(1.0 / 3));
  F[0] = lin; F[1] = pla; F[2] = sca;
  if (svgs) { F[3] = ani; F[4] = cur; } else { F[3] = cur; F[4] = ani; }
  F[5] = ent; F[6] = e1 + e2 + e3; F[7] = omn;
}
}  // namespace

void eigen33(const float* m9, int math, float* evecs9, float* evals3) {
  if (math == 1) vm_eigen33(m9, evecs9, evals3); else eigen33_ref(m9, evecs9, evals3);
}
void eigen_features(const float* ev3, bool svgs, int math, float* F8) {
  if (math == 1) vm_eigen_features(ev3, svgs ? 1 : 0, F8); else eigen_features_ref(ev3, svgs, F8);
}

// =============================================================================================
// A.2  per-node attributes (VS:1358-1429, 1147-1228, 1533-1594; SS:988-1040, 745-847, 1372-1435)
// =============================================================================================
void compute_node(const float* xyz, int stride, const int* idx, int count, int math, bool svgs, Node& out) {
  // centroid: float running sums in list order, divided by the int count (VS:1364-1375)
  float sx = 0, sy = 0, sz = 0;
  for (int k = 0; k < count; ++k) {
    const float* p = xyz + (int64_t)idx[k] * stride;
    sx = sx + p[0]; sy = sy + p[1]; sz = sz + p[2];
  }
  out.c[0] = sx / count; out.c[1] = sy / count; out.c[2] = sz / count;
  // covariance: mean recomputed, sum of outer products; not divided by n in VGS (VS:1592), /n in SVGS (SS:1425)
  float C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (count > 3) {
    float mx = sx / count, my = sy / count, mz = sz / count;
    for (int k = 0; k < count; ++k) {
      const float* p = xyz + (int64_t)idx[k] * stride;
      float d[3] = {p[0] - mx, p[1] - my, p[2] - mz};
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = C[r * 3 + c] + d[r] * d[c];
    }
    if (svgs)
      for (int i = 0; i < 9; ++i) C[i] = C[i] / count;
  }
  float evecs[9], evals[3];
  if (math == 1) vm_eigen33(C, evecs, evals); else eigen33_ref(C, evecs, evals);
  // normal = eigenvector of the smallest eigenvalue, flipped towards (0,0,1.5) seen from the first point (VS:1394-1421)
  const float* p0 = xyz + (int64_t)idx[0] * stride;
  float vx = 0 - p0[0], vy = 0 - p0[1], vz = (float)(1.5 - (double)p0[2]);
  float nx = evecs[0], ny = evecs[3], nz = evecs[6];
  if ((nx * vx + ny * vy + nz * vz) < 0) { nx = nx * -1; ny = ny * -1; nz = nz * -1; }
  out.n[0] = nx; out.n[1] = ny; out.n[2] = nz;
  if (math == 1) vm_eigen_features(evals, svgs ? 1 : 0, out.f); else eigen_features_ref(evals, svgs, out.f);
  out.nf = 8;
  out.used = true;
}

// =============================================================================================
// A.3 / A.6  measuringDistance + distanceWeight
// =============================================================================================
static VgsNode to_dev_node(const Node& a) {
  VgsNode v;
  for (int i = 0; i < 3; ++i) { v.c[i] = a.c[i]; v.n[i] = a.n[i]; }
  for (int i = 0; i < 8; ++i) v.f[i] = a.f[i];
  v.flags = 0;
  if (a.c[0] != 0 && a.c[1] != 0 && a.c[2] != 0) v.flags |= VGS_F_POS;   // VS:1829
  if (a.n[0] != 0 && a.n[1] != 0 && a.n[2] != 0) v.flags |= VGS_F_NRM;   // VS:1840
  if (a.nf > 1) v.flags |= VGS_F_EIG;
  return v;
}

static VgsWeightParams to_dev_params(const Params& P, bool svgs) {
  VgsWeightParams W;
  W.inv_sig_p = 1.0f / P.sig_p; W.inv_sig_n = 1.0f / P.sig_n; W.inv_sig_o = 1.0f / P.sig_o;
  W.inv_sig_e = 1.0f / P.sig_e; W.inv_sig_c = 1.0f / P.sig_c;
  W.inv_sig_w2 = 1.0f / (P.sig_w * P.sig_w);
  W.svgs = svgs ? 1 : 0;
  return W;
}

// RefMath restatement of VS:1597-1720 (svgs: SS:1756-1878).  Variable names follow the source.
static void pair_distances_ref(const Node& A, const Node& B, bool svgs, float out[5]) {
  const bool pv1 = (A.c[0] != 0 && A.c[1] != 0 && A.c[2] != 0), pv2 = (B.c[0] != 0 && B.c[1] != 0 && B.c[2] != 0);
  const bool nv1 = (A.n[0] != 0 && A.n[1] != 0 && A.n[2] != 0), nv2 = (B.n[0] != 0 && B.n[1] != 0 && B.n[2] != 0);
  float dist_space = 100, dist_angle = 100, dist_stair = 100, dist_eigen = 100, dist_convx = 100;
  float cos_v1_dist = 0, cos_v2_dist = 0, dist_v1_v2 = 0, cos_v1_v2 = 0, cos_d_s = 0;
  float thred_singular = 0;
  float dist_v1 = 0, dist_v2 = 0, dist_o1 = 0, dist_o2 = 0;
  double a_1 = 0, a_2 = 0, a_1_2 = 0, a_d_s = 0, a_d_s1 = 0, a_d_s2 = 0, PI = 3.1415926;
  float norm_v1_v2[3] = {0, 0, 0}, product_v1_v2[3] = {0, 0, 0};  // empty vectors in the source (UB if read) -> zeros
  const float* c1 = A.c; const float* c2 = B.c; const float* n1 = A.n; const float* n2 = B.n;
  if (pv1 && pv2) {
    dist_v1_v2 = (float)std::sqrt(std::pow((double)(c1[0] - c2[0]), 2) + std::pow((double)(c1[1] - c2[1]), 2) +
                                  std::pow((double)(c1[2] - c2[2]), 2));
    dist_space = dist_v1_v2;
    if (dist_v1_v2 != 0) {
      for (int i = 0; i < 3; ++i) norm_v1_v2[i] = (c1[i] - c2[i]) / dist_v1_v2;
      product_v1_v2[0] = c1[1] * c2[2] - c1[2] * c2[1];
      product_v1_v2[1] = c1[2] * c2[0] - c1[0] * c2[2];
      product_v1_v2[2] = c1[0] * c2[1] - c1[1] * c2[0];
    }
  }
  if (nv1 && nv2) {
    const bool guard = svgs ? (dist_v1_v2 != 0) : (dist_space != 0);
    if (guard) {
      cos_v1_v2 = (n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2]);
      cos_v1_dist = (n1[0] * norm_v1_v2[0] + n1[1] * norm_v1_v2[1] + n1[2] * norm_v1_v2[2]);
      cos_v2_dist = (n2[0] * norm_v1_v2[0] + n2[1] * norm_v1_v2[1] + n2[2] * norm_v1_v2[2]);
      cos_d_s = (product_v1_v2[0] * norm_v1_v2[0] + product_v1_v2[1] * norm_v1_v2[1] + product_v1_v2[2] * norm_v1_v2[2]);
      a_1 = std::acos(cos_v1_dist);   // float overloads (the source has `using namespace std`, IOH:46)
      a_2 = std::acos(cos_v2_dist);
      a_1_2 = std::acos(cos_v1_v2);
      a_d_s1 = std::acos(cos_d_s);
      a_d_s2 = PI - a_d_s1;
      dist_angle = std::acos(cos_v1_v2);
      dist_v1 = n1[0] * c1[0] + n1[1] * c1[1] + n1[2] * c1[2];
      dist_v2 = n2[0] * c2[0] + n2[1] * c2[1] + n2[2] * c2[2];
      dist_o1 = n1[0] * c2[0] + n1[1] * c2[1] + n1[2] * c2[2];
      dist_o2 = n2[0] * c1[0] + n2[1] * c1[1] + n2[2] * c1[2];
      if (dist_v1_v2 != 0)
        dist_stair = (float)std::sqrt(std::pow((double)(dist_o1 - dist_v1), 2) + std::pow((double)(dist_o2 - dist_v2), 2));
      else
        dist_stair = 0;
    } else if (svgs) {
      dist_stair = 0;
    }
    double temp_a = 0.5, temp_off = PI / 6, max_singular = PI / 2;
    thred_singular = (float)((double)(float)max_singular / (1 + std::exp(-1 * temp_a * (a_1_2 - temp_off))));
    a_d_s = a_d_s1;
    if (a_d_s1 > a_d_s2) a_d_s = a_d_s2;
    if (a_d_s > thred_singular) dist_convx = (float)std::fabs(a_1 - a_2);
    else dist_convx = (float)PI;
  }
  if (A.nf > 1 && B.nf > 1) {
    float eigen_cos = 0, eigen_abs1 = 0, eigen_abs2 = 0;
    for (int i = svgs ? 0 : 4; i < A.nf; ++i) {
      eigen_cos = eigen_cos + A.f[i] * B.f[i];
      eigen_abs1 = eigen_abs1 + A.f[i] * A.f[i];
      eigen_abs2 = eigen_abs2 + B.f[i] * B.f[i];
    }
    if (eigen_abs1 != 0 && eigen_abs2 != 0) dist_eigen = 1.0f - eigen_cos / (std::sqrt(eigen_abs1) * std::sqrt(eigen_abs2));
  }
  out[0] = dist_space; out[1] = dist_angle; out[2] = dist_stair; out[3] = dist_eigen; out[4] = dist_convx;
}

void pair_distances(const Node& a, const Node& b, bool svgs, int math, float out[5]) {
  if (math == 1) { vm_pair_distances(to_dev_node(a), to_dev_node(b), svgs ? 1 : 0, out); return; }
  pair_distances_ref(a, b, svgs, out);
}

static float distance_weight_ref(const float d[5], const Params& P, bool svgs) {
  float similarity_dist, similarity_weight;
  if (!svgs) {  // VS:1736-1737
    similarity_dist = (float)std::sqrt(std::pow((double)(d[0] / P.sig_p), 2) + std::pow((double)(d[1] / P.sig_n), 2) +
                                       std::pow((double)(d[2] / P.sig_o), 2) + std::pow((double)(d[4] / P.sig_c), 2) +
                                       std::pow((double)(d[3] / P.sig_e), 2));
  } else {      // SS:1900
    similarity_dist = (float)std::sqrt(std::pow((double)d[0], 2) / P.sig_p + std::pow((double)d[1], 2) / P.sig_n +
                                       std::pow((double)d[3], 2) / P.sig_e + std::pow((double)d[2], 2) / P.sig_o);
  }
  similarity_weight = (float)std::exp(-0.5 * similarity_dist / std::pow((double)P.sig_w, 2));
  return similarity_weight;
}

float distance_weight(const float d[5], const Params& P, bool svgs) {
  if (P.math == 1) return vm_distance_weight(d, to_dev_params(P, svgs));
  return distance_weight_ref(d, P, svgs);
}

float pair_weight(const Node& a, const Node& b, const Params& P, bool svgs) {
  float d[5];
  pair_distances(a, b, svgs, P.math, d);
  return distance_weight(d, P, svgs);
}

// =============================================================================================
// A.4  cutGraphSegmentation (VS:1913-2029, SS:1908-2054)
// =============================================================================================
namespace {
struct WeightIndex { float Weight; int Index; };
bool godown(const WeightIndex& a, const WeightIndex& b) { return a.Weight > b.Weight; }

// shared merge loop over an ordered edge list (v1 = column, v2 = row of the flattened matrix)
struct Merger {
  int n; float cut;
  std::vector<float> seg_int;
  std::vector<std::vector<int>> seg_ver;
  std::vector<int> seg_size, ver_seg;
  Merger(int n_, float cut_) : n(n_), cut(cut_), seg_int(n_, 1.0f), seg_ver(n_), seg_size(n_, 1), ver_seg(n_) {
    for (int i = 0; i < n; ++i) { seg_ver[i].push_back(i); ver_seg[i] = i; }
  }
  void edge(int v1, int v2, float w) {
    if (ver_seg[v1] == ver_seg[v2]) return;
    int s1 = ver_seg[v1], s2 = ver_seg[v2];
    float v1_mint = seg_int[s1] - cut / seg_size[s1];
    float v2_mint = seg_int[s2] - cut / seg_size[s2];
    int keep, gone; float thr;
    if (v1_mint >= v2_mint) { keep = s1; gone = s2; thr = v1_mint; }
    else { keep = s2; gone = s1; thr = v2_mint; }
    if (w > thr) {
      seg_int[keep] = w;
      for (int j = 0; j < seg_size[gone]; ++j) {
        seg_ver[keep].push_back(seg_ver[gone][j]);
        ver_seg[seg_ver[gone][j]] = keep;
      }
      seg_size[keep] += seg_size[gone];
      seg_size[gone] = 0;
      seg_ver[gone].clear();
    }
  }
  std::vector<int> component_of_zero() const {
    int seg_con = 0;
    for (int i = 0; i < n; ++i)
      for (size_t j = 0; j < seg_ver[i].size(); ++j)
        if (seg_ver[i][j] == 0) { seg_con = i; break; }
    return seg_ver[seg_con];
  }
};
}  // namespace

std::vector<int> cut_graph_faithful(float cut, const std::vector<float>& W, int n) {
  const int weight_size = n * n;
  // Eigen column-major storage viewed as 1 x n^2: entry k = M(row = k % n, col = k / n)
  std::vector<WeightIndex> arr;
  arr.reserve(weight_size);
  for (int k = 0; k < weight_size; ++k) {
    int col = k / n, row = k - col * n;
    arr.push_back({W[(size_t)row * n + col], k});
  }
  // Q3: NaN weights make the source's comparator inconsistent (UB).  Resolution: NaN edges sort last
  // and never merge (w > thr is false for NaN either way).
  auto mid = std::stable_partition(arr.begin(), arr.end(), [](const WeightIndex& a) { return !(a.Weight != a.Weight); });
  std::sort(arr.begin(), mid, godown);
  Merger M(n, cut);
  for (int i = 0; i < weight_size; ++i) {
    int v1 = arr[i].Index / n;
    int v2 = arr[i].Index - v1 * n;
    M.edge(v1, v2, arr[i].Weight);
  }
  return M.component_of_zero();
}

std::vector<int> cut_graph_lean(float cut, std::vector<LeanEdge>& edges, int n) {
  // deterministic order: weight descending, then k = a*n + b ascending (a < b); NaN edges dropped by the caller
  std::sort(edges.begin(), edges.end(), [n](const LeanEdge& x, const LeanEdge& y) {
    if (x.w != y.w) return x.w > y.w;
    return (int64_t)x.a * n + x.b < (int64_t)y.a * n + y.b;
  });
  Merger M(n, cut);
  for (const LeanEdge& e : edges) M.edge(e.a, e.b, e.w);
  return M.component_of_zero();
}

// =============================================================================================
// A.4-A.5  segmentVoxelCloudWithGraphModel / segmentSupervoxelCloudWithGraphModel graph part
// =============================================================================================
namespace {
// buildAdjacencyGraph + measuringDistance with the source's by-value vector traffic (faithful timing flavour)
std::vector<float> measuring_distance_byvalue(std::vector<float> v1_center, std::vector<float> v2_center, std::vector<float> v1_norm,
                                              std::vector<float> v2_norm, std::vector<float> v1_eigen, std::vector<float> v2_eigen,
                                              bool svgs, int math) {
  Node A, B;
  if (v1_center.size() > 1) for (int i = 0; i < 3; ++i) A.c[i] = v1_center[i];
  if (v2_center.size() > 1) for (int i = 0; i < 3; ++i) B.c[i] = v2_center[i];
  if (v1_norm.size() > 1) for (int i = 0; i < 3; ++i) A.n[i] = v1_norm[i];
  if (v2_norm.size() > 1) for (int i = 0; i < 3; ++i) B.n[i] = v2_norm[i];
  A.nf = (int)v1_eigen.size(); B.nf = (int)v2_eigen.size();
  for (int i = 0; i < A.nf && i < 8; ++i) A.f[i] = v1_eigen[i];
  for (int i = 0; i < B.nf && i < 8; ++i) B.f[i] = v2_eigen[i];
  float d[5];
  pair_distances(A, B, svgs, math, d);
  std::vector<float> similarity_dist;
  for (int i = 0; i < 5; ++i) similarity_dist.push_back(d[i]);
  return similarity_dist;
}
float distance_weight_byvalue(std::vector<float> dist_all, const Params& P, bool svgs) {
  float d[5] = {dist_all[0], dist_all[1], dist_all[2], dist_all[3], dist_all[4]};
  return distance_weight(d, P, svgs);
}
void node_vectors(const Node& nd, std::vector<float>& pos, std::vector<float>& nrm, std::vector<float>& eig) {
  pos.clear(); nrm.clear(); eig.clear();
  if (nd.c[0] != 0 && nd.c[1] != 0 && nd.c[2] != 0) { pos.push_back(nd.c[0]); pos.push_back(nd.c[1]); pos.push_back(nd.c[2]); }
  else pos.push_back(0);
  if (nd.n[0] != 0 && nd.n[1] != 0 && nd.n[2] != 0) { nrm.push_back(nd.n[0]); nrm.push_back(nd.n[1]); nrm.push_back(nd.n[2]); }
  else nrm.push_back(0);
  for (int i = 0; i < nd.nf; ++i) eig.push_back(nd.f[i]);
}
}  // namespace

void segment_graph(const std::vector<Node>& nodes, const std::vector<std::vector<int>>& adjacency, const Params& P, bool svgs,
                   GraphResult& R) {
  const int V = (int)nodes.size();
  R.adjacency = adjacency;
  R.connect_cut.assign(V, {});
  R.pair_evals = 0;
  // weight of any edge touching an unused node (all five distances stay 100)
  float d100[5] = {100, 100, 100, 100, 100};
  const float w_dead = distance_weight(d100, P, svgs);
  const bool prune_unused = !(w_dead > 1.0f - P.cut_thred);
  // ---- per-node local graph + cut (VS:376-412) ----
  // The reference is single threaded (P.threads = 1).  The local cuts are independent per node, so the "CPU-lean, all cores"
  // context row of BASELINE.md 2 runs this loop with OpenMP over nodes (same results: every node writes its own list).
  int64_t pair_evals = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : pair_evals) num_threads(P.threads > 1 ? P.threads : 1)
  for (int i = 0; i < V; ++i) {
    if (!nodes[i].used) continue;
    const std::vector<int>& adj = adjacency[i];
    const int n = (int)adj.size();
    std::vector<int> local;
    if (P.flavour == 0) {
      std::vector<float> W((size_t)n * n, 0.0f);
      std::vector<float> p1, n1, e1, p2, n2, e2;
      for (int a = 0; a < n; ++a) {
        node_vectors(nodes[adj[a]], p1, n1, e1);
        for (int b = 0; b < n; ++b) {
          node_vectors(nodes[adj[b]], p2, n2, e2);
          if (a != b) {
            std::vector<float> dist_all = measuring_distance_byvalue(p1, p2, n1, n2, e1, e2, svgs, P.math);
            W[(size_t)a * n + b] = distance_weight_byvalue(dist_all, P, svgs);
            pair_evals++;
          } else {
            W[(size_t)a * n + b] = 1;
          }
        }
      }
      std::vector<float> Wcopy = W;  // matrix passed by value (VS:1914)
      local = cut_graph_faithful(P.cut_thred, Wcopy, n);
    } else {
      std::vector<LeanEdge> edges;
      for (int a = 0; a < n; ++a) {
        if (prune_unused && !nodes[adj[a]].used) continue;
        for (int b = a + 1; b < n; ++b) {
          if (prune_unused && !nodes[adj[b]].used) continue;
          float w = pair_weight(nodes[adj[a]], nodes[adj[b]], P, svgs);
          pair_evals++;
          if (w != w) continue;  // NaN never merges (Q3)
          edges.push_back({w, a, b});
        }
      }
      local = cut_graph_lean(P.cut_thred, edges, n);
    }
    std::vector<int>& out = R.connect_cut[i];
    for (int v : local) out.push_back(adj[v]);
  }
  R.pair_evals = pair_evals;
  // ---- crossValidation (VS:2111-2179): sequential, in place ----
  R.connect_cross = R.connect_cut;
  {
    auto& L = R.connect_cross;
    for (int i = 0; i < V; ++i) {
      int inthis = (int)L[i].size();
      if (inthis > 1) {
        std::vector<int> keep;
        for (int j = 0; j < inthis; ++j) {
          int s = L[i][j];
          bool found = false;
          for (int v : L[s]) if (v == i) found = true;
          if (found) keep.push_back(s);
        }
        L[i] = keep;
      }
    }
  }
  // ---- closestCheck (VS:2181-2303): sequential, mutating ----
  R.connect_final = R.connect_cross;
  R.q7_out_of_range = 0;
  {
    auto& L = R.connect_final;
    for (int i = 0; i < V; ++i) {
      int inthis = (int)L[i].size();
      if (!(inthis > 0 && inthis < 2)) continue;
      // voxels_adjacency_idx_[i] = [count, idx...]: size = n + 1
      const std::vector<int>& adj = adjacency[i];
      if (!((int)adj.size() + 1 > P.adjacency_min)) continue;
      float min_dis = 0;
      int min_idx = -1;
      for (int j = 0; j < (int)adj.size() + 1; ++j) {
        int t;
        if (j == 0) {
          if (!P.q7_count_as_index) continue;
          t = (int)adj.size();                         // Q7: the leading count read as a voxel id
          if (t >= V) { R.q7_out_of_range++; continue; }  // out of range is UB in the source; skipped here
        } else {
          t = adj[j - 1];
        }
        if ((int)L[t].size() > 1) {
          float temp_dis = pair_weight(nodes[i], nodes[t], P, svgs);  // distanceProbability == distanceWeight (Q8)
          if (temp_dis >= min_dis) { min_dis = temp_dis; min_idx = t; }
        }
      }
      if (min_idx != -1) { L[i].push_back(min_idx); L[min_idx].push_back(i); }
    }
  }
  // ---- clusteringVoxels + recursionSearch (VS:2032-2099): pre-order DFS, seed appended last ----
  R.clusters.clear();
  R.node_cluster.assign(V, -1);
  {
    const auto& L = R.connect_final;
    std::vector<char> clustered(V, 0);
    for (int i = 0; i < V; ++i) {
      if (clustered[i]) continue;
      std::vector<int> cl;
      clustered[i] = 1;
      std::vector<std::pair<int, size_t>> stack;  // (node whose list is being scanned, position)
      stack.emplace_back(i, 0);
      while (!stack.empty()) {
        auto& top = stack.back();
        const std::vector<int>& lst = L[top.first];
        if (top.second >= lst.size()) { stack.pop_back(); continue; }
        int v = lst[top.second++];
        if (!clustered[v]) {
          cl.push_back(v);
          clustered[v] = 1;
          stack.emplace_back(v, 0);
        }
      }
      cl.push_back(i);
      for (int v : cl) R.node_cluster[v] = (int)R.clusters.size();
      R.clusters.push_back(cl);
    }
  }
}

// =============================================================================================
// B.2  KdTreeFLANN::radiusSearch restated as an exact grid search
// =============================================================================================
namespace {
struct Key3 { int x, y, z; bool operator==(const Key3& o) const { return x == o.x && y == o.y && z == o.z; } };
struct Key3Hash { size_t operator()(const Key3& k) const { return ((size_t)(uint32_t)k.x * 73856093u) ^ ((size_t)(uint32_t)k.y * 19349663u) ^ ((size_t)(uint32_t)k.z * 83492791u); } };

// points: 3*M floats; for each point all points with d2 < float(r*r), sorted by (d2, index)
void radius_search_all(const std::vector<float>& pts, double radius, std::vector<std::vector<int>>& out) {
  const int M = (int)pts.size() / 3;
  out.assign(M, {});
  const float r2 = (float)(radius * radius);
  const double cell = radius > 0 ? radius : 1.0;
  std::unordered_map<Key3, std::vector<int>, Key3Hash> grid;
  auto cell_of = [&](const float* p) { return Key3{(int)std::floor(p[0] / cell), (int)std::floor(p[1] / cell), (int)std::floor(p[2] / cell)}; };
  for (int i = 0; i < M; ++i) grid[cell_of(&pts[3 * i])].push_back(i);
  std::vector<std::pair<float, int>> found;
  for (int i = 0; i < M; ++i) {
    const float* q = &pts[3 * i];
    Key3 c = cell_of(q);
    found.clear();
    for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) for (int dz = -1; dz <= 1; ++dz) {
      auto it = grid.find(Key3{c.x + dx, c.y + dy, c.z + dz});
      if (it == grid.end()) continue;
      for (int j : it->second) {
        const float* p = &pts[3 * j];
        // flann::L2_Simple<float>: result += diff*diff over x,y,z
        float d2 = 0;
        float t = q[0] - p[0]; d2 += t * t;
        t = q[1] - p[1]; d2 += t * t;
        t = q[2] - p[2]; d2 += t * t;
        if (d2 < r2) found.emplace_back(d2, j);
      }
    }
    std::sort(found.begin(), found.end());
    out[i].reserve(found.size());
    for (auto& f : found) out[i].push_back(f.second);
  }
}
}  // namespace

// =============================================================================================
// VGS driver (T:51-76)
// =============================================================================================
void run_vgs(const float* xyz, int64_t n, int stride, const Params& P, VgsResult& R) {
  double t0 = now_s();
  build_voxel_table(xyz, n, stride, P.voxel_size, R.T);
  double t1 = now_s();
  const int V = R.T.V();
  R.nodes.assign(V, Node());
  for (int v = 0; v < V; ++v) {
    int cnt = R.T.start[v + 1] - R.T.start[v];
    if (cnt > P.points_min) compute_node(xyz, stride, &R.T.point_idx[R.T.start[v]], cnt, P.math, false, R.nodes[v]);
  }
  double t2 = now_s();
  std::vector<std::vector<int>> adjacency;
  radius_search_all(R.T.center, (double)P.graph_size, adjacency);
  double t3 = now_s();
  segment_graph(R.nodes, adjacency, P, false, R.G);
  double t4 = now_s();
  // drawColorMapofPointsinClusters (VS:963-1009): clusters with size > voxels_min; points per member voxel
  R.clusters_num = (int)R.G.clusters.size();
  R.clusters_points.clear();
  R.point_label.assign((size_t)n, -1);
  for (const auto& cl : R.G.clusters) {
    if (!((int)cl.size() > P.voxels_min)) continue;
    std::vector<int> pts;
    for (int v : cl)
      for (int k = R.T.start[v]; k < R.T.start[v + 1]; ++k) pts.push_back(R.T.point_idx[k]);
    for (int p : pts) R.point_label[p] = (int)R.clusters_points.size();
    R.clusters_points.push_back(pts);
  }
  double t5 = now_s();
  R.t.voxelize = t1 - t0; R.t.features = t2 - t1; R.t.adjacency = t3 - t2; R.t.graph = t4 - t3; R.t.labels = t5 - t4;
  R.t.total = t5 - t0;
}

// =============================================================================================
// SVGS driver from a per-point supervoxel labelling (SS:279-331, 1238-1303, 1477-1521, 362-421, 2079-2130)
// =============================================================================================
void run_svgs_from_labels(const float* xyz, int64_t n, int stride, const int* sv_label, int max_label, const Params& P,
                          SvgsResult& R) {
  double t0 = now_s();
  R.sv_label.assign(sv_label, sv_label + n);
  R.max_label = max_label;
  std::vector<std::vector<int>> map((size_t)max_label + 1);
  for (int64_t j = 0; j < n; ++j) {
    int l = sv_label[j];
    if (l > 0 && l <= max_label) map[l].push_back((int)j);
  }
  R.sv_points.clear();
  for (int k = 0; k < max_label; ++k)  // Q12: label == max_label is never visited (SS:313)
    if (!map[k].empty()) R.sv_points.push_back(map[k]);
  const int S = (int)R.sv_points.size();
  R.nodes.assign(S, Node());
  std::vector<float> centroids((size_t)S * 3);
  for (int s = 0; s < S; ++s) {
    compute_node(xyz, stride, R.sv_points[s].data(), (int)R.sv_points[s].size(), P.math, true, R.nodes[s]);
    for (int a = 0; a < 3; ++a) centroids[3 * s + a] = R.nodes[s].c[a];
  }
  double t1 = now_s();
  std::vector<std::vector<int>> adjacency;
  radius_search_all(centroids, (double)P.graph_size, adjacency);
  double t2 = now_s();
  segment_graph(R.nodes, adjacency, P, true, R.G);
  double t3 = now_s();
  R.clusters_num = (int)R.G.clusters.size();
  R.clusters_points.clear();
  R.point_label.assign((size_t)n, -1);
  for (const auto& cl : R.G.clusters) {
    if (cl.empty()) continue;
    std::vector<int> pts;
    for (int s : cl) for (int p : R.sv_points[s]) pts.push_back(p);
    for (int p : pts) R.point_label[p] = (int)R.clusters_points.size();
    R.clusters_points.push_back(pts);
  }
  double t4 = now_s();
  R.t.features = t1 - t0; R.t.adjacency = t2 - t1; R.t.graph = t3 - t2; R.t.labels = t4 - t3; R.t.total = t4 - t0;
}

}  // namespace refcpu
