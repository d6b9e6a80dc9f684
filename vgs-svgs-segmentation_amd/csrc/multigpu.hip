// multigpu.hip -- tile support: shared grid chaining, ownership, boundary records, final labels.
// The reference is single-process; scenes shard by spatial tile, one context per GPU.  Everything per voxel is
// local to a ball of radius graph_size, only the connected components are global: each rank segments its tile
// plus a halo, trusts the mutual connections that have an owned endpoint and the re-attachments of its owned voxels,
// and publishes one (voxel code, local root) record per boundary voxel (k_boundary).  One all-gather of these records
// (RCCL, done by the host driver) lets every rank run the same small union-find over (rank, root) pairs.
#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "vgs_context.hpp"

// per voxel: does it hold points this rank loaded itself (mix[v]) / points that came with another rank's strip (mix[V + v])?
__global__ void k_point_sources(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ pt_vox, int64_t nf, int64_t own_first, int64_t n_own,
                                int64_t V, uint8_t* __restrict__ mix) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nf) return;
  const uint32_t v = pt_vox[j];
  const int64_t i = (int64_t)perm[j];
  if (v != 0xffffffffu) mix[((i >= own_first && i < own_first + n_own) ? 0 : V) + v] = 1;   // every writer stores the same value
}

// straddle[v]: bit 0 = the voxel's cube reaches over a border (or holds points of both sides); bit 1 (round 5) = its centre lies within
// `near` of a border line -- only such a voxel can have a neighbour of the other ownership in its search ball, so only its row is read
// when the boundary records are made (k_boundary read every row: 0.15 ms for a handful of records)
__global__ void k_owned(const uint64_t* __restrict__ vox_code, int64_t V, float res_f, float min_x, float min_y, double lo_x, double lo_y,
                        double hi_x, double hi_y, uint8_t* __restrict__ owned, uint8_t* __restrict__ straddle, const uint8_t* __restrict__ mix, double near) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const uint64_t code = vox_code[v];
  // the centre is a function of the global code and the shared grid only: every rank decides ownership alike
  const double cx = (double)vm_voxel_center(vm_compact21(code >> 2), res_f, min_x);
  const double cy = (double)vm_voxel_center(vm_compact21(code >> 1), res_f, min_y);
  owned[v] = (cx >= lo_x && cx < hi_x && cy >= lo_y && cy < hi_y) ? 1 : 0;
  // the voxel's cube reaches over a border of the region: points on both sides (loaded by different ranks) fall into it
  const double h = 0.5001 * (double)res_f;
  bool st = fabs(cx - lo_x) < h || fabs(cx - hi_x) < h || fabs(cy - lo_y) < h || fabs(cy - hi_y) < h;
  // ... or the ranks were handed points beyond their regions (objects that reach over a tile's edge): an owned voxel with points
  // another rank loaded, a halo voxel with points this rank loaded.  Both ranks see that (the holder sends such points in its
  // strip) and both publish the voxel, so the holder learns the label of its points from the owner's record.
  if (mix) st = st || (owned[v] ? mix[V + v] != 0 : mix[v] != 0);
  const bool nr = fabs(cx - lo_x) < near || fabs(cx - hi_x) < near || fabs(cy - lo_y) < near || fabs(cy - hi_y) < near;
  straddle[v] = (uint8_t)((st ? 1 : 0) | (nr ? 2 : 0));
}

vgs_status vgs_compute_owned(vgs_ctx* c) {
  VGS_HIP_TRY(c, c->owned.ensure(c->V > 0 ? c->V : 1));
  VGS_HIP_TRY(c, c->straddle.ensure(c->V > 0 ? c->V : 1));
  if (c->V == 0) return VGS_OK;
  const uint8_t* mix = nullptr;
  if (c->n_own >= 0 && c->Nf > 0) {
    VGS_HIP_TRY(c, c->mixsrc.ensure(2 * (size_t)c->V));
    VGS_HIP_TRY(c, hipMemsetAsync(c->mixsrc.p, 0, 2 * (size_t)c->V, c->stream));
    hipLaunchKernelGGL(k_point_sources, dim3((unsigned)((c->Nf + 255) / 256)), dim3(256), 0, c->stream, c->perm_b.p, c->pt_vox.p, c->Nf, c->own_first, c->n_own, c->V, c->mixsrc.p);
    mix = c->mixsrc.p;
  }
  hipLaunchKernelGGL(k_owned, dim3((unsigned)((c->V + 255) / 256)), dim3(256), 0, c->stream, c->vox_code.p, c->V, c->P.voxel_size,
                     (float)c->box.min[0], (float)c->box.min[1], c->own_lo[0], c->own_lo[1], c->own_hi[0], c->own_hi[1], c->owned.p, c->straddle.p, mix,
                     (double)c->P.graph_size + 2.0 * (double)c->P.voxel_size);   // neighbours' centres are closer than graph_size; two voxels of margin
  VGS_HIP_TRY(c, hipGetLastError());
  return VGS_OK;
}

// Boundary voxels of a tile: every voxel through which a local component can continue on another rank.
//   * a voxel (owned or halo) with a mutual connection to a voxel of the other ownership: both ranks see that
//     connection exactly (halo = 2*graph_size + voxel_size), both publish both endpoints, each from its own row;
//   * an owned voxel with a halo voxel in its neighbourhood: it may be the closestCheck target of an isolated voxel
//     of the neighbouring rank (VS:2293).  Only the owner of the isolated voxel decides that re-attachment, so the
//     owner of the target cannot know about it and must publish the target unconditionally -- else the other
//     rank's (target code, root) record finds no partner and the re-attached voxel becomes a one-voxel segment;
//   * both ends of a re-attachment of an owned voxel to a halo voxel;
//   * every voxel, used or not, whose cube reaches over the border (k_boundary_straddle): the ranks on both sides hold
//     points of it, and the rank that does not own it learns its label through the record of the rank that does.
// At most two records per voxel (duplicates are removed by the sort that follows).
// (round 5) a wavefront takes 64 used voxels: every lane checks ITS voxel's re-attachment and whether it lies near a border line; the
// rows of the near ones (a few per cent of a tile) are then read by the whole wavefront, one after the other
__global__ __launch_bounds__(64) void k_boundary(const uint32_t* __restrict__ used_ids, int64_t U, const uint64_t* __restrict__ adj_key,
                                                 const uint32_t* __restrict__ adj_cnt, int adj_stride, const uint8_t* __restrict__ mutual,
                                                 const int32_t* __restrict__ attach, const uint8_t* __restrict__ owned,
                                                 const uint32_t* __restrict__ parent, const uint64_t* __restrict__ vox_code,
                                                 unsigned long long cap, unsigned long long* __restrict__ n_out,
                                                 uint64_t* __restrict__ out_code, int32_t* __restrict__ out_root, const uint8_t* __restrict__ straddle) {
  const int lane = threadIdx.x;
  const int64_t u_mine = (int64_t)blockIdx.x * 64 + lane;
  uint32_t i_mine = 0;
  bool near_mine = false, oi_mine = false;
  if (u_mine < U) { i_mine = used_ids[u_mine]; near_mine = (straddle[i_mine] & 2) != 0; oi_mine = owned[i_mine] != 0; }
  // far from every border line all of the row's voxels share the voxel's ownership: nothing to publish but a re-attachment
  unsigned long long pub_mask = 0ull;
  unsigned long long todo = __ballot(near_mine);
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    const int64_t u = (int64_t)blockIdx.x * 64 + src;
    const bool oi = __shfl((int)oi_mine, src, 64) != 0;
    const int n = (int)adj_cnt[u];
    const uint64_t* row = adj_key + u * adj_stride;
    const uint8_t* mrow = mutual + u * adj_stride;
    bool pub = false;
    for (int k = lane; k < n; k += 64) {
      const uint32_t t = (uint32_t)row[k];
      const bool ot = owned[t] != 0;
      pub = pub || (oi ? !ot : (ot && mrow[k] != 0));
    }
    if (__ballot(pub) != 0ull) pub_mask |= 1ull << src;
  }
  if (u_mine >= U) return;
  const bool pub = ((pub_mask >> lane) & 1ull) != 0ull;
  const int32_t a = attach[i_mine];
  const bool att = a >= 0 && oi_mine && !owned[a];   // the count-as-index target (Q7) need not be a neighbour
  const unsigned int cnt = (pub || att ? 1u : 0u) + (att ? 1u : 0u);
  if (cnt == 0) return;
  const unsigned long long p = atomicAdd(n_out, (unsigned long long)cnt);
  if (p + cnt > cap) return;
  out_code[p] = vox_code[i_mine]; out_root[p] = (int32_t)parent[i_mine];
  if (att) { out_code[p + 1] = vox_code[a]; out_root[p + 1] = (int32_t)parent[a]; }
}

__global__ void k_boundary_straddle(const uint8_t* __restrict__ straddle, int64_t V, const uint32_t* __restrict__ parent,
                                    const uint64_t* __restrict__ vox_code, unsigned long long cap, unsigned long long* __restrict__ n_out,
                                    uint64_t* __restrict__ out_code, int32_t* __restrict__ out_root) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V || !(straddle[v] & 1)) return;
  const unsigned long long p = atomicAdd(n_out, 1ull);
  if (p < cap) { out_code[p] = vox_code[v]; out_root[p] = (int32_t)parent[v]; }
}

// halo voxels take the label of their local component too: a voxel that reaches over the border holds points of this
// rank although another rank owns it (its root is then a boundary root and carries the owner's label)
__global__ void k_apply_root_labels(const uint32_t* __restrict__ parent, const int32_t* __restrict__ root_label, int64_t V,
                                    int32_t* __restrict__ vox_label) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  vox_label[v] = root_label[parent[v]];
}

__global__ void k_point_labels2(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ pt_vox, const int32_t* __restrict__ vox_label,
                                int64_t N, int32_t* __restrict__ label) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const uint32_t v = pt_vox[j];
  label[perm[j]] = (v == 0xffffffffu) ? -1 : vox_label[v];
}

// ---- compact tile protocol ----------------------------------------------------------------------------------
__global__ void k_bnd_heads(const uint64_t* __restrict__ code_sorted, int64_t n, uint32_t* __restrict__ head) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) head[k] = (k == 0 || code_sorted[k] != code_sorted[k - 1]) ? 1u : 0u;
}

// one record per boundary voxel; marks the roots they name
__global__ void k_bnd_compact(const uint64_t* __restrict__ code_sorted, const int32_t* __restrict__ root_sorted, const uint32_t* __restrict__ head,
                              const uint32_t* __restrict__ scan_incl, int64_t n, const uint32_t* __restrict__ csz,
                              uint64_t* __restrict__ out_code, int32_t* __restrict__ out_root, int32_t* __restrict__ out_cnt,
                              uint8_t* __restrict__ broot) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n || !head[k]) return;
  const uint32_t pos = scan_incl[k] - 1u;
  const int32_t r = root_sorted[k];
  out_code[pos] = code_sorted[k];
  out_root[pos] = r;
  out_cnt[pos] = (int32_t)csz[r];
  broot[r] = 1;
}

// kept components that no boundary record names: flag per voxel id (their rank = exclusive scan of the flags)
__global__ void k_local_kept_flags(const uint32_t* __restrict__ parent, const uint32_t* __restrict__ csz, const uint8_t* __restrict__ broot,
                                   int64_t V, int voxels_min, uint32_t* __restrict__ flag) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  flag[v] = (parent[v] == (uint32_t)v && !broot[v] && (int)csz[v] > voxels_min) ? 1u : 0u;
}

__global__ void k_tile_root_labels(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ rank_excl, int64_t V, int32_t local_base,
                                   int32_t* __restrict__ root_label) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  root_label[v] = flag[v] ? local_base + (int32_t)rank_excl[v] : -1;
}

__global__ void k_scatter_labels(const int32_t* __restrict__ root, const int32_t* __restrict__ label, int64_t n, int64_t V,
                                 int32_t* __restrict__ root_label, unsigned int* __restrict__ bad) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int32_t r = root[k];
  if (r < 0 || r >= V) { atomicOr(bad, 1u); return; }
  root_label[r] = label[k];
}

// per-workgroup bounding box of the finite points: out[6 * block + (min x, y, z, max x, y, z)]
__global__ __launch_bounds__(256) void k_points_bbox(const float* __restrict__ xyz, int stride_f, int64_t n, float* __restrict__ out,
                                                     unsigned long long* __restrict__ n_finite) {
  __shared__ float s_v[6][4];
  __shared__ unsigned long long s_n[4];
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  unsigned long long cnt = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = xyz + i * stride_f;
    const float x = p[0], y = p[1], z = p[2];
    const bool fin = (vm_bits(x) & 0x7f800000u) != 0x7f800000u && (vm_bits(y) & 0x7f800000u) != 0x7f800000u && (vm_bits(z) & 0x7f800000u) != 0x7f800000u;
    if (fin) {
      lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
      hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
      ++cnt;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
    cnt += __shfl_xor(cnt, o, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { for (int a = 0; a < 3; ++a) { s_v[a][wave] = lo[a]; s_v[3 + a][wave] = hi[a]; } s_n[wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int a = 0; a < 3; ++a) {
      out[6 * blockIdx.x + a] = fminf(fminf(s_v[a][0], s_v[a][1]), fminf(s_v[a][2], s_v[a][3]));
      out[6 * blockIdx.x + 3 + a] = fmaxf(fmaxf(s_v[3 + a][0], s_v[3 + a][1]), fmaxf(s_v[3 + a][2], s_v[3 + a][3]));
    }
    atomicAdd(n_finite, s_n[0] + s_n[1] + s_n[2] + s_n[3]);
  }
}

extern "C" {

vgs_status vgs_points_bbox(vgs_ctx* c, float* bbox6, int64_t* n_finite) {
  if (!c || !bbox6 || !n_finite) return VGS_E_ARG;
  if (c->stage < ST_POINTS) { c->err = "vgs_points_bbox: no input cloud"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  *n_finite = 0;
  for (int a = 0; a < 3; ++a) { bbox6[a] = 0.f; bbox6[3 + a] = 0.f; }
  if (c->N == 0) return VGS_OK;
  const int NBLK = 512;
  VGS_HIP_TRY(c, c->sort_tmp.ensure((size_t)NBLK * 6 * sizeof(float) + 16));
  VGS_HIP_TRY(c, c->counters.ensure(64));
  float* d_out = (float*)c->sort_tmp.p;
  unsigned long long* d_n = (unsigned long long*)(c->counters.p + 34);
  VGS_HIP_TRY(c, hipMemsetAsync(d_n, 0, 8, c->stream));
  hipLaunchKernelGGL(k_points_bbox, dim3(NBLK), dim3(256), 0, c->stream, c->xyz, c->stride_f, c->N, d_out, d_n);
  std::vector<float> h((size_t)NBLK * 6);
  unsigned long long nf = 0;
  VGS_HIP_TRY(c, hipMemcpyAsync(h.data(), d_out, h.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipMemcpyAsync(&nf, d_n, 8, hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  VGS_HIP_TRY(c, hipGetLastError());
  *n_finite = (int64_t)nf;
  if (nf == 0) return VGS_OK;
  for (int a = 0; a < 3; ++a) { bbox6[a] = 3.0e38f; bbox6[3 + a] = -3.0e38f; }
  for (int b = 0; b < NBLK; ++b)
    for (int a = 0; a < 3; ++a) { bbox6[a] = std::min(bbox6[a], h[6 * b + a]); bbox6[3 + a] = std::max(bbox6[3 + a], h[6 * b + 3 + a]); }
  return VGS_OK;
}

// Host replay of the octree growth over a cloud of which only the bounding box is known.  PCL grows the box for the first
// point outside it, per axis upwards if the point lies at or above the box's max and downwards otherwise (also on the axes
// where the point is inside).  When every point outside the box gives the same answer on every axis -- points stick out
// above the max on exactly one axis and nowhere below, or only below mins -- the step does not depend on WHICH point is
// first, so it can be taken from the bounding box alone; the test is repeated for the grown box.  Otherwise *need_scan = 1
// and g is left at the last state reached: vgs_grid_advance continues from there with the points themselves (every point
// in front of the next violator is inside the current box, so scanning from the first point finds the same violator).
vgs_status vgs_grid_advance_bbox(vgs_grid_state* g, double voxel_size, const float* bbox6, int32_t* need_scan) {
  if (!g || !bbox6 || !need_scan || !(voxel_size > 0.0)) return VGS_E_ARG;
  *need_scan = 0;
  if (!g->defined) { *need_scan = 1; return VGS_OK; }   // the first box is placed around the first point itself
  const double eps = 1.1920928955078125e-07;
  for (int guard = 0; guard < 64; ++guard) {
    const double side = (double)(1u << g->depth) * voxel_size;
    int n_hi = 0, n_lo = 0, a_hi = -1;
    for (int a = 0; a < 3; ++a) {
      const double mx = g->min[a] + side - eps;
      if ((double)bbox6[3 + a] >= mx) { ++n_hi; a_hi = a; }
      if ((double)bbox6[a] < g->min[a]) ++n_lo;
    }
    if (n_hi == 0 && n_lo == 0) return VGS_OK;   // every point is inside
    const bool det = (n_lo == 0 && n_hi == 1) || (n_hi == 0);
    if (!det || g->depth >= 30) { *need_scan = 1; return VGS_OK; }
    for (int a = 0; a < 3; ++a)
      if (a != a_hi || n_hi == 0) { g->min[a] -= side; g->shift[a] += (1ull << g->depth); }   // old root becomes the upper child
    g->depth++;
  }
  *need_scan = 1;
  return VGS_OK;
}

vgs_status vgs_grid_state_init(vgs_grid_state* g) {
  if (!g) return VGS_E_ARG;
  std::memset(g, 0, sizeof(*g));
  return VGS_OK;
}

static void box_from_state(const vgs_grid_state* g, double res, OctreeBox& box) {
  box = OctreeBox();
  box.res = res;
  box.defined = g->defined != 0;
  box.depth = g->depth;
  const double eps = 1.1920928955078125e-07;
  for (int a = 0; a < 3; ++a) {
    box.min[a] = g->min[a];
    box.shift[a] = g->shift[a];
    box.max[a] = g->min[a] + (double)(1u << g->depth) * res - eps;
  }
}

vgs_status vgs_grid_advance(vgs_ctx* c, vgs_grid_state* g) {
  if (!c || !g) return VGS_E_ARG;
  if (c->stage < ST_POINTS) { c->err = "vgs_grid_advance: no input cloud"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  OctreeBox box;
  box_from_state(g, (double)c->P.voxel_size, box);
  const bool was_pinned = c->grid_pinned;
  c->grid_pinned = false;
  vgs_status st = vgs_grow_box_from(c, box, false);
  c->grid_pinned = was_pinned;
  if (st != VGS_OK) return st;
  g->defined = box.defined ? 1 : 0;
  g->depth = box.depth;
  for (int a = 0; a < 3; ++a) { g->min[a] = box.min[a]; g->shift[a] = box.shift[a]; }
  return VGS_OK;
}

vgs_status vgs_set_grid(vgs_ctx* c, const vgs_grid_state* g) {
  if (!c || !g) return VGS_E_ARG;
  if (!g->defined) { c->err = "vgs_set_grid: undefined grid state"; return VGS_E_ARG; }
  box_from_state(g, (double)c->P.voxel_size, c->box);
  c->grid_pinned = true;
  c->grid_covers = false;
  if (c->stage > ST_POINTS) c->stage = ST_POINTS;
  return VGS_OK;
}

vgs_status vgs_set_grid_covering(vgs_ctx* c, const vgs_grid_state* g) {
  vgs_status s = vgs_set_grid(c, g);
  if (s == VGS_OK) c->grid_covers = true;
  return s;
}

vgs_status vgs_set_owned_region(vgs_ctx* c, const double* lo, const double* hi) {
  if (!c || !lo || !hi) return VGS_E_ARG;
  c->own_lo[0] = lo[0]; c->own_lo[1] = lo[1]; c->own_hi[0] = hi[0]; c->own_hi[1] = hi[1];
  c->have_region = true;
  if (c->stage > ST_ADJACENCY) c->stage = ST_ADJACENCY;
  return VGS_OK;
}

vgs_status vgs_set_own_point_range(vgs_ctx* c, int64_t first, int64_t n_own) {
  if (!c || n_own < -1 || first < 0) return VGS_E_ARG;
  c->own_first = first;
  c->n_own = n_own;
  if (c->stage > ST_ADJACENCY) c->stage = ST_ADJACENCY;
  return VGS_OK;
}

vgs_status vgs_get_boundary(vgs_ctx* c, int64_t* n_records, uint64_t* code, int32_t* root) {
  if (!c || !n_records) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED || !c->have_region) { c->err = "vgs_get_boundary: segment a context with an owned region first"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  if (c->V == 0) { *n_records = 0; return VGS_OK; }
  unsigned long long cap = c->bnd_code.cap;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (cap < 1024) cap = 1u << 20;
    VGS_HIP_TRY(c, c->bnd_code.ensure(cap)); VGS_HIP_TRY(c, c->bnd_root.ensure(cap));
    VGS_HIP_TRY(c, c->counters.ensure(64));
    unsigned long long* d_n = (unsigned long long*)c->counters.p + 32;
    VGS_HIP_TRY(c, hipMemsetAsync(d_n, 0, 8, c->stream));
    const uint8_t* mutual = c->conn.p + (size_t)c->U * c->adj_stride;
    if (c->U > 0)
      hipLaunchKernelGGL(k_boundary, dim3((unsigned)((c->U + 63) / 64)), dim3(64), 0, c->stream, c->used_ids.p, c->U, c->adj_key.p, c->adj_cnt.p,
                       c->adj_stride, mutual, c->attach.p, c->owned.p, c->parent.p, c->vox_code.p, cap, d_n, c->bnd_code.p, c->bnd_root.p, c->straddle.p);
    hipLaunchKernelGGL(k_boundary_straddle, dim3((unsigned)((c->V + 255) / 256)), dim3(256), 0, c->stream, c->straddle.p, c->V, c->parent.p,
                       c->vox_code.p, cap, d_n, c->bnd_code.p, c->bnd_root.p);
    unsigned long long n = 0;
    VGS_HIP_TRY(c, hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, c->stream));
    VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n <= cap) {
      *n_records = (int64_t)n;
      if (code && n) VGS_HIP_TRY(c, hipMemcpy(code, c->bnd_code.p, n * 8, hipMemcpyDeviceToHost));
      if (root && n) VGS_HIP_TRY(c, hipMemcpy(root, c->bnd_root.p, n * 4, hipMemcpyDeviceToHost));
      return VGS_OK;
    }
    cap = n + 1024;
  }
  c->err = "vgs_get_boundary: record buffer overflow";
  return VGS_E_NOMEM;
}

vgs_status vgs_get_boundary_roots(vgs_ctx* c, int64_t* n_records, uint64_t* code, int32_t* root, int32_t* owned_voxels, int64_t* n_kept_local) {
  if (!c || !n_records) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED || !c->have_region) { c->err = "vgs_get_boundary_roots: segment a context with an owned region first"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  const int64_t V = c->V;
  const int TB = 256;
  if (!code || c->bnd_unique < 0) {
    // raw records (both endpoints of every crossing connection), then one per voxel
    int64_t n_raw = 0;
    vgs_status st = vgs_get_boundary(c, &n_raw, nullptr, nullptr);   // leaves the records in bnd_code / bnd_root on the device
    if (st != VGS_OK) return st;
    VGS_HIP_TRY(c, c->broot.ensure(V > 0 ? V : 1));
    VGS_HIP_TRY(c, hipMemsetAsync(c->broot.p, 0, V > 0 ? V : 1, c->stream));
    int64_t n_unique = 0;
    if (n_raw > 0) {
      const size_t n = (size_t)n_raw;
      VGS_HIP_TRY(c, c->bnd_code2.ensure(2 * n)); VGS_HIP_TRY(c, c->bnd_root2.ensure(2 * n)); VGS_HIP_TRY(c, c->bnd_cnt.ensure(n));
      VGS_HIP_TRY(c, c->head_flag.ensure(n + 1)); VGS_HIP_TRY(c, c->kept_rank.ensure(n + 1));
      uint64_t* code_sorted = c->bnd_code2.p + n;   // second halves: sort outputs
      int32_t* root_sorted = c->bnd_root2.p + n;
      size_t sort_bytes = 0, scan_bytes = 0;
      VGS_HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, sort_bytes, c->bnd_code.p, code_sorted, c->bnd_root.p, root_sorted, n, 0, 64, c->stream));
      VGS_HIP_TRY(c, rocprim::inclusive_scan(nullptr, scan_bytes, c->head_flag.p, c->kept_rank.p, n, rocprim::plus<uint32_t>(), c->stream));
      VGS_HIP_TRY(c, c->sort_tmp.ensure(std::max(sort_bytes, scan_bytes)));
      VGS_HIP_TRY(c, rocprim::radix_sort_pairs(c->sort_tmp.p, sort_bytes, c->bnd_code.p, code_sorted, c->bnd_root.p, root_sorted, n, 0, 64, c->stream));
      hipLaunchKernelGGL(k_bnd_heads, dim3((unsigned)((n + TB - 1) / TB)), dim3(TB), 0, c->stream, code_sorted, (int64_t)n, c->head_flag.p);
      VGS_HIP_TRY(c, rocprim::inclusive_scan(c->sort_tmp.p, scan_bytes, c->head_flag.p, c->kept_rank.p, n, rocprim::plus<uint32_t>(), c->stream));
      hipLaunchKernelGGL(k_bnd_compact, dim3((unsigned)((n + TB - 1) / TB)), dim3(TB), 0, c->stream, code_sorted, root_sorted, c->head_flag.p,
                         c->kept_rank.p, (int64_t)n, c->csz.p, c->bnd_code2.p, c->bnd_root2.p, c->bnd_cnt.p, c->broot.p);
      uint32_t last = 0;
      VGS_HIP_TRY(c, hipMemcpyAsync(&last, c->kept_rank.p + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
      VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
      n_unique = (int64_t)last;
    }
    // kept components without boundary records: flags in head_flag[0, V), exclusive ranks in kept_rank[0, V]
    int64_t kept_local = 0;
    if (V > 0) {
      VGS_HIP_TRY(c, c->head_flag.ensure(V + 1)); VGS_HIP_TRY(c, c->kept_rank.ensure(V + 1));
      hipLaunchKernelGGL(k_local_kept_flags, dim3((unsigned)((V + TB - 1) / TB)), dim3(TB), 0, c->stream, c->parent.p, c->csz.p, c->broot.p, V,
                         c->P.voxels_min, c->head_flag.p);
      size_t scan_bytes = 0;
      VGS_HIP_TRY(c, rocprim::exclusive_scan(nullptr, scan_bytes, c->head_flag.p, c->kept_rank.p, 0u, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
      VGS_HIP_TRY(c, c->sort_tmp.ensure(scan_bytes));
      VGS_HIP_TRY(c, rocprim::exclusive_scan(c->sort_tmp.p, scan_bytes, c->head_flag.p, c->kept_rank.p, 0u, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
      uint32_t last_rank = 0, last_flag = 0;
      VGS_HIP_TRY(c, hipMemcpyAsync(&last_rank, c->kept_rank.p + (V - 1), 4, hipMemcpyDeviceToHost, c->stream));
      VGS_HIP_TRY(c, hipMemcpyAsync(&last_flag, c->head_flag.p + (V - 1), 4, hipMemcpyDeviceToHost, c->stream));
      VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
      kept_local = (int64_t)last_rank + last_flag;
    }
    VGS_HIP_TRY(c, hipGetLastError());
    c->bnd_unique = n_unique;
    c->bnd_kept_local = kept_local;
  }
  *n_records = c->bnd_unique;
  if (n_kept_local) *n_kept_local = c->bnd_kept_local;
  const size_t n = (size_t)c->bnd_unique;
  if (code && n) VGS_HIP_TRY(c, hipMemcpy(code, c->bnd_code2.p, n * 8, hipMemcpyDeviceToHost));
  if (root && n) VGS_HIP_TRY(c, hipMemcpy(root, c->bnd_root2.p, n * 4, hipMemcpyDeviceToHost));
  if (owned_voxels && n) VGS_HIP_TRY(c, hipMemcpy(owned_voxels, c->bnd_cnt.p, n * 4, hipMemcpyDeviceToHost));
  return VGS_OK;
}

vgs_status vgs_apply_tile_labels(vgs_ctx* c, int32_t local_base, const int32_t* root, const int32_t* label, int64_t n_roots) {
  if (c) c->cl_valid = false;
  if (!c || (n_roots > 0 && (!root || !label))) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED || c->bnd_unique < 0) { c->err = "vgs_apply_tile_labels: vgs_get_boundary_roots first"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  // these kernels rewrite pt_label in place: a download that was opened on it must be through first
  if (c->d2h_open && c->d2h_src == c->pt_label.p) { VGS_HIP_TRY(c, hipEventSynchronize(c->ev_d2h)); c->d2h_open = false; }
  const int64_t V = c->V, N = c->N;
  if (V == 0) return VGS_OK;
  const int TB = 256;
  VGS_HIP_TRY(c, c->root_label.ensure(V));
  hipLaunchKernelGGL(k_tile_root_labels, dim3((unsigned)((V + TB - 1) / TB)), dim3(TB), 0, c->stream, c->head_flag.p, c->kept_rank.p, V, local_base,
                     c->root_label.p);
  if (n_roots > 0) {
    // labels of the boundary roots: small upload, scattered on the device
    VGS_HIP_TRY(c, c->bnd_root2.ensure(2 * (size_t)n_roots + 2)); VGS_HIP_TRY(c, c->counters.ensure(64));
    int32_t* d_root = c->bnd_root2.p;
    int32_t* d_label = c->bnd_root2.p + n_roots;
    unsigned int* d_bad = (unsigned int*)(c->counters.p + 33);
    VGS_HIP_TRY(c, hipMemsetAsync(d_bad, 0, 4, c->stream));
    VGS_HIP_TRY(c, hipMemcpyAsync(d_root, root, (size_t)n_roots * 4, hipMemcpyHostToDevice, c->stream));
    VGS_HIP_TRY(c, hipMemcpyAsync(d_label, label, (size_t)n_roots * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_scatter_labels, dim3((unsigned)((n_roots + TB - 1) / TB)), dim3(TB), 0, c->stream, d_root, d_label, n_roots, V,
                       c->root_label.p, d_bad);
    unsigned int bad = 0;
    VGS_HIP_TRY(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (bad) { c->err = "vgs_apply_tile_labels: root out of range"; return VGS_E_ARG; }
  }
  hipLaunchKernelGGL(k_apply_root_labels, dim3((unsigned)((V + TB - 1) / TB)), dim3(TB), 0, c->stream, c->parent.p, c->root_label.p, V,
                     c->vox_label.p);
  hipLaunchKernelGGL(k_point_labels2, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, c->stream, c->perm_b.p, c->pt_vox.p, c->vox_label.p, N,
                     c->pt_label.p);
  c->pt_labels_pending = false;   // every point's label has just been written
  VGS_HIP_TRY(c, hipEventRecord(c->ev[15], c->stream));   // label downloads order themselves behind this (vgs_get_point_labels_async)
  c->labels_event_valid = true;
  VGS_HIP_TRY(c, hipGetLastError());
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->bnd_unique = -1;  // bnd_root2 was reused
  return VGS_OK;
}

vgs_status vgs_get_owned_roots(vgs_ctx* c, int64_t* n_roots, int32_t* root, int32_t* owned_voxels) {
  if (!c || !n_roots) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED) { c->err = "vgs_get_owned_roots: segment first"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  std::vector<uint32_t> par((size_t)c->V), csz((size_t)c->V);
  if (c->V > 0) {
    VGS_HIP_TRY(c, hipMemcpy(par.data(), c->parent.p, par.size() * 4, hipMemcpyDeviceToHost));
    VGS_HIP_TRY(c, hipMemcpy(csz.data(), c->csz.p, csz.size() * 4, hipMemcpyDeviceToHost));
  }
  int64_t n = 0;
  for (int64_t v = 0; v < c->V; ++v)
    if (par[v] == (uint32_t)v && csz[v] > 0) {
      if (root) root[n] = (int32_t)v;
      if (owned_voxels) owned_voxels[n] = (int32_t)csz[v];
      ++n;
    }
  *n_roots = n;
  return VGS_OK;
}

vgs_status vgs_apply_root_labels(vgs_ctx* c, const int32_t* root, const int32_t* label, int64_t n_roots) {
  if (c) c->cl_valid = false;
  if (!c || (n_roots > 0 && (!root || !label))) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED) { c->err = "vgs_apply_root_labels: segment first"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  // these kernels rewrite pt_label in place: a download that was opened on it must be through first
  if (c->d2h_open && c->d2h_src == c->pt_label.p) { VGS_HIP_TRY(c, hipEventSynchronize(c->ev_d2h)); c->d2h_open = false; }
  const int64_t V = c->V, N = c->N;
  if (V == 0) return VGS_OK;
  std::vector<int32_t> map((size_t)V, -1);
  for (int64_t k = 0; k < n_roots; ++k) {
    if (root[k] < 0 || root[k] >= V) { c->err = "vgs_apply_root_labels: root out of range"; return VGS_E_ARG; }
    map[root[k]] = label[k];
  }
  VGS_HIP_TRY(c, c->root_label.ensure(V));
  VGS_HIP_TRY(c, hipMemcpy(c->root_label.p, map.data(), (size_t)V * 4, hipMemcpyHostToDevice));
  const int TB = 256;
  hipLaunchKernelGGL(k_apply_root_labels, dim3((unsigned)((V + TB - 1) / TB)), dim3(TB), 0, c->stream, c->parent.p, c->root_label.p, V,
                     c->vox_label.p);
  hipLaunchKernelGGL(k_point_labels2, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, c->stream, c->perm_b.p, c->pt_vox.p, c->vox_label.p, N,
                     c->pt_label.p);
  c->pt_labels_pending = false;   // every point's label has just been written
  VGS_HIP_TRY(c, hipEventRecord(c->ev[15], c->stream));   // label downloads order themselves behind this (vgs_get_point_labels_async)
  c->labels_event_valid = true;
  VGS_HIP_TRY(c, hipGetLastError());
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VGS_OK;
}

}  // extern "C"
