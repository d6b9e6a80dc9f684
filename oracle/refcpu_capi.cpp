// oracle/refcpu_capi.cpp -- TEST INFRASTRUCTURE: flat C entry points over refcpu for ctypes (tests/,
// __graft_entry__.smoke(), bench.py cpu_baseline).  Never linked into the product library.
#include <cstring>

#include "refcpu.hpp"
#include "vgs_math.h"

using namespace refcpu;

extern "C" {

struct RefParamsC {
  float voxel_size, graph_size, sig_p, sig_n, sig_o, sig_e, sig_c, sig_w, cut_thred;
  int points_min, adjacency_min, voxels_min;
  float seed_size, color_impt, spatial_impt, normal_impt;
  int math, flavour, q7_count_as_index, threads;
};

static Params to_params(const RefParamsC* c) {
  Params P;
  P.voxel_size = c->voxel_size; P.graph_size = c->graph_size;
  P.sig_p = c->sig_p; P.sig_n = c->sig_n; P.sig_o = c->sig_o; P.sig_e = c->sig_e; P.sig_c = c->sig_c; P.sig_w = c->sig_w;
  P.cut_thred = c->cut_thred;
  P.points_min = c->points_min; P.adjacency_min = c->adjacency_min; P.voxels_min = c->voxels_min;
  P.seed_size = c->seed_size; P.color_impt = c->color_impt; P.spatial_impt = c->spatial_impt; P.normal_impt = c->normal_impt;
  P.math = c->math; P.flavour = c->flavour; P.q7_count_as_index = c->q7_count_as_index;
  P.threads = c->threads > 1 ? c->threads : 1;
  return P;
}

struct RefHandle {
  int kind;  // 0 vgs, 1 svgs
  int64_t n;
  VgsResult vgs;
  SvgsResult svgs;
  const GraphResult& G() const { return kind == 0 ? vgs.G : svgs.G; }
  const std::vector<Node>& nodes() const { return kind == 0 ? vgs.nodes : svgs.nodes; }
};

void* ref_vgs_run(const float* xyz, int64_t n, int stride_floats, const RefParamsC* p) {
  RefHandle* h = new RefHandle();
  h->kind = 0; h->n = n;
  run_vgs(xyz, n, stride_floats, to_params(p), h->vgs);
  return h;
}

void* ref_svgs_run_from_labels(const float* xyz, int64_t n, int stride_floats, const int* labels, int max_label, const RefParamsC* p) {
  RefHandle* h = new RefHandle();
  h->kind = 1; h->n = n;
  run_svgs_from_labels(xyz, n, stride_floats, labels, max_label, to_params(p), h->svgs);
  return h;
}

// voxel table only (cheap; for binning parity at large N)
void* ref_voxelize(const float* xyz, int64_t n, int stride_floats, float voxel_size) {
  RefHandle* h = new RefHandle();
  h->kind = 0; h->n = n;
  build_voxel_table(xyz, n, stride_floats, voxel_size, h->vgs.T);
  return h;
}

// the adjacency octree of pcl::SupervoxelClustering: box defined from the cloud's bounding box (vccs_mode 1)
void* ref_voxelize_bbox(const float* xyz, int64_t n, int stride_floats, float voxel_size) {
  RefHandle* h = new RefHandle();
  h->kind = 0; h->n = n;
  build_voxel_table_bbox(xyz, n, stride_floats, voxel_size, h->vgs.T);
  return h;
}

// VCCS-style supervoxel stage: labels (N ints, 0 = unassigned); returns max_label
int ref_vccs(const float* xyz, int64_t n, int stride_floats, const RefParamsC* p, int* labels) {
  std::vector<int> lab;
  int max_label = 0;
  vccs_supervoxels(xyz, n, stride_floats, to_params(p), lab, max_label);
  std::memcpy(labels, lab.data(), lab.size() * sizeof(int));
  return max_label;
}

int ref_vccs_refmath(const float* xyz, int64_t n, int stride_floats, const RefParamsC* p, int* labels) {
  std::vector<int> lab;
  int max_label = 0;
  vccs_supervoxels_refmath(xyz, n, stride_floats, to_params(p), lab, max_label);
  std::memcpy(labels, lab.data(), lab.size() * sizeof(int));
  return max_label;
}

int ref_vccs_pcl(const float* xyz, int64_t n, int stride_floats, const RefParamsC* p, int* labels) {
  std::vector<int> lab;
  int max_label = 0;
  vccs_pcl_supervoxels(xyz, n, stride_floats, to_params(p), lab, max_label);
  std::memcpy(labels, lab.data(), lab.size() * sizeof(int));
  return max_label;
}
int ref_vccs_pcl_refmath(const float* xyz, int64_t n, int stride_floats, const RefParamsC* p, int* labels) {
  std::vector<int> lab;
  int max_label = 0;
  vccs_pcl_supervoxels_refmath(xyz, n, stride_floats, to_params(p), lab, max_label);
  std::memcpy(labels, lab.data(), lab.size() * sizeof(int));
  return max_label;
}
void ref_free(void* hv) { delete (RefHandle*)hv; }

// out[0]=nodes V, [1]=sum adjacency, [2]=clusters_num, [3]=kept clusters, [4]=pair_evals, [5]=octree depth,
// [6]=q7 out of range, [7]=N' (finite points), [8]=used nodes
void ref_counts(void* hv, int64_t* out) {
  RefHandle* h = (RefHandle*)hv;
  const GraphResult& G = h->G();
  int64_t V = h->kind == 0 ? h->vgs.T.V() : (int64_t)h->svgs.nodes.size();
  out[0] = V;
  int64_t e = 0;
  for (auto& a : G.adjacency) e += (int64_t)a.size();
  out[1] = e;
  out[2] = h->kind == 0 ? h->vgs.clusters_num : h->svgs.clusters_num;
  out[3] = h->kind == 0 ? (int64_t)h->vgs.clusters_points.size() : (int64_t)h->svgs.clusters_points.size();
  out[4] = G.pair_evals;
  out[5] = h->kind == 0 ? h->vgs.T.depth : 0;
  out[6] = G.q7_out_of_range;
  out[7] = h->kind == 0 ? (int64_t)h->vgs.T.point_idx.size() : h->n;
  int64_t u = 0;
  for (auto& nd : h->nodes()) u += nd.used ? 1 : 0;
  out[8] = u;
}

void ref_vgs_bbox(void* hv, double* out6) {
  RefHandle* h = (RefHandle*)hv;
  for (int a = 0; a < 3; ++a) { out6[a] = h->vgs.T.min[a]; out6[3 + a] = h->vgs.T.max[a]; }
}

void ref_vgs_voxel_table(void* hv, uint32_t* key, int* start, int* point_idx, int* point_voxel, float* center) {
  RefHandle* h = (RefHandle*)hv;
  const VoxelTable& T = h->vgs.T;
  if (key) std::memcpy(key, T.key.data(), T.key.size() * sizeof(uint32_t));
  if (start) std::memcpy(start, T.start.data(), T.start.size() * sizeof(int));
  if (point_idx) std::memcpy(point_idx, T.point_idx.data(), T.point_idx.size() * sizeof(int));
  if (point_voxel) std::memcpy(point_voxel, T.point_voxel.data(), T.point_voxel.size() * sizeof(int));
  if (center) std::memcpy(center, T.center.data(), T.center.size() * sizeof(float));
}

void ref_nodes(void* hv, float* centroid, float* normal, float* eig8, uint8_t* used) {
  RefHandle* h = (RefHandle*)hv;
  const std::vector<Node>& nodes = h->nodes();
  for (size_t v = 0; v < nodes.size(); ++v) {
    for (int a = 0; a < 3; ++a) { centroid[3 * v + a] = nodes[v].c[a]; normal[3 * v + a] = nodes[v].n[a]; }
    for (int a = 0; a < 8; ++a) eig8[8 * v + a] = nodes[v].f[a];
    used[v] = nodes[v].used ? 1 : 0;
  }
}

// which: 0 adjacency, 1 connect after cut, 2 after crossValidation, 3 after closestCheck, 4 clusters,
//        5 cluster point lists (getClusterIdx), 6 supervoxel point lists
static const std::vector<std::vector<int>>& pick(RefHandle* h, int which) {
  const GraphResult& G = h->G();
  switch (which) {
    case 0: return G.adjacency;
    case 1: return G.connect_cut;
    case 2: return G.connect_cross;
    case 3: return G.connect_final;
    case 4: return G.clusters;
    case 5: return h->kind == 0 ? h->vgs.clusters_points : h->svgs.clusters_points;
    default: return h->svgs.sv_points;
  }
}
int64_t ref_lists_size(void* hv, int which, int64_t* n_lists) {
  const auto& L = pick((RefHandle*)hv, which);
  *n_lists = (int64_t)L.size();
  int64_t tot = 0;
  for (auto& l : L) tot += (int64_t)l.size();
  return tot;
}
void ref_lists(void* hv, int which, int64_t* offsets, int* idx) {
  const auto& L = pick((RefHandle*)hv, which);
  int64_t o = 0;
  for (size_t i = 0; i < L.size(); ++i) {
    offsets[i] = o;
    std::memcpy(idx + o, L[i].data(), L[i].size() * sizeof(int));
    o += (int64_t)L[i].size();
  }
  offsets[L.size()] = o;
}

void ref_labels(void* hv, int* point_label, int* node_cluster) {
  RefHandle* h = (RefHandle*)hv;
  const std::vector<int>& pl = h->kind == 0 ? h->vgs.point_label : h->svgs.point_label;
  if (point_label) std::memcpy(point_label, pl.data(), pl.size() * sizeof(int));
  if (node_cluster) std::memcpy(node_cluster, h->G().node_cluster.data(), h->G().node_cluster.size() * sizeof(int));
}

void ref_times(void* hv, double* out7) {
  RefHandle* h = (RefHandle*)hv;
  const StageTimes& t = h->kind == 0 ? h->vgs.t : h->svgs.t;
  out7[0] = t.voxelize; out7[1] = t.features; out7[2] = t.adjacency; out7[3] = t.graph; out7[4] = t.merge; out7[5] = t.labels; out7[6] = t.total;
}

// ------------------------------------------------------------------ known-answer entry points
// node layout: c[3], n[3], f[8], nf (as float), used (as float) = 16 floats
static Node node_from(const float* a) {
  Node nd;
  for (int i = 0; i < 3; ++i) { nd.c[i] = a[i]; nd.n[i] = a[3 + i]; }
  for (int i = 0; i < 8; ++i) nd.f[i] = a[6 + i];
  nd.nf = (int)a[14];
  nd.used = a[15] != 0;
  return nd;
}
void ref_pair_distances(const float* a16, const float* b16, int svgs, int math, float* out5) {
  pair_distances(node_from(a16), node_from(b16), svgs != 0, math, out5);
}
float ref_distance_weight(const float* d5, const RefParamsC* p, int svgs) { return distance_weight(d5, to_params(p), svgs != 0); }
float ref_pair_weight(const float* a16, const float* b16, const RefParamsC* p, int svgs) {
  return pair_weight(node_from(a16), node_from(b16), to_params(p), svgs != 0);
}
int ref_cut_graph(float cut, const float* W, int n, int flavour, int* out) {
  std::vector<int> r;
  if (flavour == 0) {
    std::vector<float> Wv(W, W + (size_t)n * n);
    r = cut_graph_faithful(cut, Wv, n);
  } else {
    std::vector<LeanEdge> e;
    for (int a = 0; a < n; ++a)
      for (int b = a + 1; b < n; ++b) {
        float w = W[(size_t)a * n + b];
        if (w != w) continue;
        e.push_back({w, a, b});
      }
    r = cut_graph_lean(cut, e, n);
  }
  for (size_t i = 0; i < r.size(); ++i) out[i] = r[i];
  return (int)r.size();
}
// schedule bounds of the lazy local cut (csrc/vgs_math.h): exported so that CPU tests can check bound >= weight
static VgsNode dev_node(const float* a) {
  VgsNode v;
  for (int i = 0; i < 3; ++i) { v.c[i] = a[i]; v.n[i] = a[3 + i]; }
  for (int i = 0; i < 8; ++i) v.f[i] = a[6 + i];
  v.flags = 0; v.pad = 0;
  if (a[0] != 0 && a[1] != 0 && a[2] != 0) v.flags |= VGS_F_POS;
  if (a[3] != 0 && a[4] != 0 && a[5] != 0) v.flags |= VGS_F_NRM;
  if ((int)a[14] > 1) v.flags |= VGS_F_EIG;
  return v;
}
static VgsWeightParams dev_params(const RefParamsC* p, int svgs) {
  VgsWeightParams W;
  W.inv_sig_p = 1.0f / p->sig_p; W.inv_sig_n = 1.0f / p->sig_n; W.inv_sig_o = 1.0f / p->sig_o;
  W.inv_sig_e = 1.0f / p->sig_e; W.inv_sig_c = 1.0f / p->sig_c;
  W.inv_sig_w2 = 1.0f / (p->sig_w * p->sig_w);
  W.svgs = svgs;
  return W;
}
void ref_weight_and_bounds(const float* a16, const float* b16, const RefParamsC* p, int svgs, float* out3) {
  VgsNode A = dev_node(a16), B = dev_node(b16);
  VgsWeightParams W = dev_params(p, svgs);
  out3[0] = vm_pair_weight(A, B, W);
  out3[1] = vm_weight_bound_da(A, B, W);
  float d2 = 1.0e4f;
  if ((A.flags & VGS_F_POS) && (B.flags & VGS_F_POS)) {
    float dx = A.c[0] - B.c[0], dy = A.c[1] - B.c[1], dz = A.c[2] - B.c[2];
    d2 = (dx * dx + dy * dy) + dz * dz;
  }
  out3[2] = vm_weight_bound_d(d2, W);
}
// vm_pair_weight_both against two plain evaluations, over n pairs of records (a16[k], b16[k]): returns the number of pairs where either
// orientation differs in any bit (tests/test_oracle_kat.py: must be 0)
int ref_weight_both_mismatches(const float* a16, const float* b16, int n, const RefParamsC* p, int svgs) {
  const VgsWeightParams W = dev_params(p, svgs);
  int bad = 0;
  for (int k = 0; k < n; ++k) {
    const VgsNode A = dev_node(a16 + 16 * (size_t)k), B = dev_node(b16 + 16 * (size_t)k);
    float w12, w21;
    vm_pair_weight_both(A, B, W, &w12, &w21);
    const float r12 = vm_pair_weight(A, B, W), r21 = vm_pair_weight(B, A, W);
    if (vm_bits(w12) != vm_bits(r12) || vm_bits(w21) != vm_bits(r21)) ++bad;
  }
  return bad;
}
void ref_compute_node(const float* xyz, int stride_floats, const int* idx, int count, int math, int svgs, float* out16) {
  Node nd;
  compute_node(xyz, stride_floats, idx, count, math, svgs != 0, nd);
  for (int i = 0; i < 3; ++i) { out16[i] = nd.c[i]; out16[3 + i] = nd.n[i]; }
  for (int i = 0; i < 8; ++i) out16[6 + i] = nd.f[i];
  out16[14] = (float)nd.nf; out16[15] = nd.used ? 1.0f : 0.0f;
}
void ref_eigen_features(const float* ev3, int svgs, int math, float* out8) { eigen_features(ev3, svgs != 0, math, out8); }
void ref_eigen33(const float* m9, int math, float* evecs9, float* evals3) { eigen33(m9, math, evecs9, evals3); }
// DevMath scalar functions for ulp tests: 0 acos, 1 exp, 2 log, 3 atan2(y,x), 4 sin_small, 5 cos_small
void ref_devmath(int fn, const float* x, const float* y, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    float s, c;
    switch (fn) {
      case 0: out[i] = vm_acos(x[i]); break;
      case 1: out[i] = vm_exp(x[i]); break;
      case 2: out[i] = vm_log(x[i]); break;
      case 3: out[i] = vm_atan2_ypos(y[i], x[i]); break;
      case 4: vm_sincos_small(x[i], &s, &c); out[i] = s; break;
      default: vm_sincos_small(x[i], &s, &c); out[i] = c; break;
    }
  }
}
}  // extern "C"
