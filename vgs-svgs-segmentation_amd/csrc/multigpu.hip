// multigpu.hip -- tile support: shared grid chaining, ownership, boundary records, final labels.
// The reference is single-process; scenes shard by spatial tile, one context per GPU.  Everything per voxel is
// local to a ball of radius graph_size, only the connected components are global: each rank segments its tile
// plus a halo, trusts the connections that have an owned endpoint, and publishes one (voxel code, local root)
// record per endpoint of every connection that crosses the ownership border.  One all-gather of these records
// (RCCL, done by the host driver) lets every rank run the same small union-find over (rank, root) pairs.
#include <algorithm>
#include <cstring>
#include <vector>

#include "vgs_context.hpp"

__global__ void k_owned(const uint64_t* __restrict__ vox_code, int64_t V, float res_f, float min_x, float min_y, double lo_x, double lo_y,
                        double hi_x, double hi_y, uint8_t* __restrict__ owned) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const uint64_t code = vox_code[v];
  // the centre is a function of the global code and the shared grid only: every rank decides ownership alike
  const double cx = (double)vm_voxel_center(vm_compact21(code >> 2), res_f, min_x);
  const double cy = (double)vm_voxel_center(vm_compact21(code >> 1), res_f, min_y);
  owned[v] = (cx >= lo_x && cx < hi_x && cy >= lo_y && cy < hi_y) ? 1 : 0;
}

vgs_status vgs_compute_owned(vgs_ctx* c) {
  VGS_HIP_TRY(c, c->owned.ensure(c->V > 0 ? c->V : 1));
  if (c->V == 0) return VGS_OK;
  hipLaunchKernelGGL(k_owned, dim3((unsigned)((c->V + 255) / 256)), dim3(256), 0, c->stream, c->vox_code.p, c->V, c->P.voxel_size,
                     (float)c->box.min[0], (float)c->box.min[1], c->own_lo[0], c->own_lo[1], c->own_hi[0], c->own_hi[1], c->owned.p);
  VGS_HIP_TRY(c, hipGetLastError());
  return VGS_OK;
}

// both endpoints of every final connection (mutual or re-attachment) that crosses the ownership border
__global__ __launch_bounds__(64) void k_boundary(const uint32_t* __restrict__ used_ids, int64_t U, const uint64_t* __restrict__ adj_key,
                                                 const uint32_t* __restrict__ adj_cnt, int adj_stride, const uint8_t* __restrict__ mutual,
                                                 const int32_t* __restrict__ attach, const uint8_t* __restrict__ owned,
                                                 const uint32_t* __restrict__ parent, const uint64_t* __restrict__ vox_code,
                                                 unsigned long long cap, unsigned long long* __restrict__ n_out,
                                                 uint64_t* __restrict__ out_code, int32_t* __restrict__ out_root) {
  const int64_t u = blockIdx.x;
  if (u >= U) return;
  const int lane = threadIdx.x;
  const uint32_t i = used_ids[u];
  const int n = (int)adj_cnt[u];
  const uint64_t* row = adj_key + u * adj_stride;
  const uint8_t* mrow = mutual + u * adj_stride;
  const bool oi = owned[i] != 0;
  for (int base = 0; base < n + 1; base += 64) {
    const int k = base + lane;
    bool cross = false;
    uint32_t t = 0;
    if (k < n) {
      t = (uint32_t)row[k];
      cross = mrow[k] && t > i && ((owned[t] != 0) != oi);
    } else if (k == n) {
      const int32_t a = attach[i];
      if (a >= 0) { t = (uint32_t)a; cross = oi && !owned[t]; }
    }
    const unsigned long long mk = __ballot(cross);
    if (mk == 0ull) continue;
    unsigned long long basepos = 0;
    const int l0 = __ffsll((long long)mk) - 1;
    if (lane == l0) basepos = atomicAdd(n_out, 2ull * (unsigned long long)__popcll(mk));
    basepos = __shfl((long long)basepos, l0, 64);
    if (cross) {
      const unsigned long long p = basepos + 2ull * (unsigned long long)__popcll(mk & ((1ull << lane) - 1ull));
      if (p + 1 < cap) {
        out_code[p] = vox_code[i]; out_root[p] = (int32_t)parent[i];
        out_code[p + 1] = vox_code[t]; out_root[p + 1] = (int32_t)parent[t];
      }
    }
  }
}

__global__ void k_apply_root_labels(const uint32_t* __restrict__ parent, const uint8_t* __restrict__ owned, const int32_t* __restrict__ root_label,
                                    int64_t V, int32_t* __restrict__ vox_label) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  vox_label[v] = (!owned || owned[v]) ? root_label[parent[v]] : -1;
}

__global__ void k_point_labels2(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ pt_vox, const int32_t* __restrict__ vox_label,
                                int64_t N, int32_t* __restrict__ label) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const uint32_t v = pt_vox[j];
  label[perm[j]] = (v == 0xffffffffu) ? -1 : vox_label[v];
}

extern "C" {

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
  if (c->stage > ST_POINTS) c->stage = ST_POINTS;
  return VGS_OK;
}

vgs_status vgs_set_owned_region(vgs_ctx* c, const double* lo, const double* hi) {
  if (!c || !lo || !hi) return VGS_E_ARG;
  c->own_lo[0] = lo[0]; c->own_lo[1] = lo[1]; c->own_hi[0] = hi[0]; c->own_hi[1] = hi[1];
  c->have_region = true;
  if (c->stage > ST_ADJACENCY) c->stage = ST_ADJACENCY;
  return VGS_OK;
}

vgs_status vgs_get_boundary(vgs_ctx* c, int64_t* n_records, uint64_t* code, int32_t* root) {
  if (!c || !n_records) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED || !c->have_region) { c->err = "vgs_get_boundary: segment a context with an owned region first"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  if (c->U == 0) { *n_records = 0; return VGS_OK; }
  unsigned long long cap = c->bnd_code.cap;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (cap < 1024) cap = 1u << 20;
    VGS_HIP_TRY(c, c->bnd_code.ensure(cap)); VGS_HIP_TRY(c, c->bnd_root.ensure(cap));
    VGS_HIP_TRY(c, c->counters.ensure(64));
    unsigned long long* d_n = (unsigned long long*)c->counters.p + 32;
    VGS_HIP_TRY(c, hipMemsetAsync(d_n, 0, 8, c->stream));
    const uint8_t* mutual = c->conn.p + (size_t)c->U * c->adj_stride;
    hipLaunchKernelGGL(k_boundary, dim3((unsigned)c->U), dim3(64), 0, c->stream, c->used_ids.p, c->U, c->adj_key.p, c->adj_cnt.p,
                       c->adj_stride, mutual, c->attach.p, c->owned.p, c->parent.p, c->vox_code.p, cap, d_n, c->bnd_code.p, c->bnd_root.p);
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
  if (!c || (n_roots > 0 && (!root || !label))) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED) { c->err = "vgs_apply_root_labels: segment first"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
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
  hipLaunchKernelGGL(k_apply_root_labels, dim3((unsigned)((V + TB - 1) / TB)), dim3(TB), 0, c->stream, c->parent.p,
                     c->have_region ? c->owned.p : nullptr, c->root_label.p, V, c->vox_label.p);
  hipLaunchKernelGGL(k_point_labels2, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, c->stream, c->perm_b.p, c->pt_vox.p, c->vox_label.p, N,
                     c->pt_label.p);
  VGS_HIP_TRY(c, hipGetLastError());
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VGS_OK;
}

}  // extern "C"
