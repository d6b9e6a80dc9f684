// features.hip -- stage a3: per-voxel centroid, covariance, 3x3 eigen solve, normal, 8 eigen features.
// Replaces calcualteVoxelCloudAttributes and its helpers (voxel_segmentation.h:290-369, 1358-1429,
// 1147-1228, 1533-1594).  The arithmetic is vgs_math.h (DevMath); sums run in ascending point index
// order exactly like the reference's loops, so the records are bit-identical to the oracle's.
//
// Data flow: points were gathered into leaf order by the voxelize stage, so a voxel's points are one
// contiguous run of xs/ys/zs.  A workgroup owns 256 consecutive voxels = one contiguous point range;
// it streams that range through LDS in coalesced tiles and every thread folds the part of the tile
// that belongs to its voxel (two sweeps: sums, then centred second moments).
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "vgs_context.hpp"
#include "brick_table.hpp"

#define FEAT_TB 256
#define FEAT_TILE 2048

// (TB voxels per workgroup: 256 for VGS voxels of ten points; 64 for supervoxels of a hundred -- a workgroup streams TB runs through LDS and
// only the threads whose run lies in the current tile work, so long runs want small workgroups: config 4 11.65 -> 11.48 ms)
template <int TB>
__global__ __launch_bounds__(TB) void k_features(const float* __restrict__ xs, const float* __restrict__ ys,
                                                      const float* __restrict__ zs, const uint32_t* __restrict__ vox_start,
                                                      int64_t V, int points_min, int svgs, const uint64_t* __restrict__ vox_code,
                                                      NodeRec* __restrict__ node, uint32_t* __restrict__ used_flag) {
  __shared__ float lx[FEAT_TILE], ly[FEAT_TILE], lz[FEAT_TILE];
  const int64_t v0 = (int64_t)blockIdx.x * TB;
  const int64_t v = v0 + threadIdx.x;
  const int64_t vend = (v0 + TB < V) ? v0 + TB : V;
  const uint32_t p_begin = vox_start[v0];
  const uint32_t p_end = vox_start[vend];
  uint32_t my_s = 0, my_e = 0;
  if (v < V) { my_s = vox_start[v]; my_e = vox_start[v + 1]; }
  const int cnt = (int)(my_e - my_s);
  const bool used = (v < V) && (cnt > points_min);

  // sweep 1: running sums in ascending index order (VS:1364-1375)
  float sx = 0.f, sy = 0.f, sz = 0.f;
  float fx = 0.f, fy = 0.f, fz = 0.f;  // first point of the voxel (normal flip, VS:1394-1396)
  for (uint32_t t0 = p_begin; t0 < p_end; t0 += FEAT_TILE) {
    const uint32_t tn = (p_end - t0 < FEAT_TILE) ? (p_end - t0) : FEAT_TILE;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < tn; k += TB) { lx[k] = xs[t0 + k]; ly[k] = ys[t0 + k]; lz[k] = zs[t0 + k]; }
    __syncthreads();
    if (used) {
      uint32_t a = my_s > t0 ? my_s : t0;
      uint32_t b = my_e < t0 + tn ? my_e : t0 + tn;
      for (uint32_t j = a; j < b; ++j) {
        const uint32_t k = j - t0;
        if (j == my_s) { fx = lx[k]; fy = ly[k]; fz = lz[k]; }
        sx = sx + lx[k]; sy = sy + ly[k]; sz = sz + lz[k];
      }
    }
  }
  const float mx = sx / cnt, my = sy / cnt, mz = sz / cnt;

  // sweep 2: sum of outer products of (p - mean), sequential, no contraction (VS:1556-1583)
  float C[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_cov = used && cnt > 3;  // VS:1554: a voxel with <= 3 points keeps the zero matrix
  for (uint32_t t0 = p_begin; t0 < p_end; t0 += FEAT_TILE) {
    const uint32_t tn = (p_end - t0 < FEAT_TILE) ? (p_end - t0) : FEAT_TILE;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < tn; k += TB) { lx[k] = xs[t0 + k]; ly[k] = ys[t0 + k]; lz[k] = zs[t0 + k]; }
    __syncthreads();
    if (do_cov) {
      uint32_t a = my_s > t0 ? my_s : t0;
      uint32_t b = my_e < t0 + tn ? my_e : t0 + tn;
      for (uint32_t j = a; j < b; ++j) {
        const uint32_t k = j - t0;
        const float d0 = lx[k] - mx, d1 = ly[k] - my, d2 = lz[k] - mz;
        C[0] = C[0] + d0 * d0; C[1] = C[1] + d0 * d1; C[2] = C[2] + d0 * d2;
        C[4] = C[4] + d1 * d1; C[5] = C[5] + d1 * d2; C[8] = C[8] + d2 * d2;
      }
    }
  }
  if (v >= V) return;
  NodeRec r;
  for (int i = 0; i < 3; ++i) { r.c[i] = 0.f; r.n[i] = 0.f; }
  for (int i = 0; i < 8; ++i) r.f[i] = 0.f;
  r.flags = 0;
  // VGS: the low 10 bits of the voxel's lattice coordinates ride along in the record's spare word, so that whoever gathers
  // two records also knows their lattice offset (near-pair lists of the local cut); supervoxels are not on a lattice
  r.pad = 0;
  if (!svgs) {
    const uint64_t code = vox_code[v];
    r.pad = (vm_compact21(code >> 2) & 1023u) | ((vm_compact21(code >> 1) & 1023u) << 10) | ((vm_compact21(code) & 1023u) << 20);
  }
  if (used) {
    C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
    if (svgs) for (int i = 0; i < 9; ++i) C[i] = C[i] / cnt;  // SS:1425
    float evecs[9], evals[3];
    vm_eigen33(C, evecs, evals);
    float nx = evecs[0], ny = evecs[3], nz = evecs[6];
    const float vx = 0.f - fx, vy = 0.f - fy, vz = 1.5f - fz;
    if ((nx * vx + ny * vy + nz * vz) < 0.f) { nx = nx * -1.f; ny = ny * -1.f; nz = nz * -1.f; }
    r.c[0] = mx; r.c[1] = my; r.c[2] = mz;
    r.n[0] = nx; r.n[1] = ny; r.n[2] = nz;
    vm_eigen_features(evals, svgs, r.f);
    uint32_t fl = VGS_F_EIG;
    if (mx != 0.f && my != 0.f && mz != 0.f) fl |= VGS_F_POS;
    if (nx != 0.f && ny != 0.f && nz != 0.f) fl |= VGS_F_NRM;
    r.flags = fl;
  }
  node[v] = r;
  used_flag[v] = used ? 1u : 0u;
}

__global__ void k_compact_used(const uint32_t* __restrict__ used_flag, const uint32_t* __restrict__ excl, int64_t V,
                               uint32_t* __restrict__ used_ids, uint32_t* __restrict__ used_rank, unsigned long long* __restrict__ n_used) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  if (v == V - 1) *n_used = (unsigned long long)excl[v] + used_flag[v];
  if (used_flag[v]) { used_ids[excl[v]] = (uint32_t)v; used_rank[v] = excl[v]; }
  else used_rank[v] = 0xffffffffu;
}

vgs_status vgs_stage_features(vgs_ctx* c) {
  const int64_t V = c->V;
  c->U = 0;
  if (V == 0) { c->counts[VGS_N_USED] = 0; return VGS_OK; }
  VGS_HIP_TRY(c, c->node.ensure(V));
  VGS_HIP_TRY(c, c->used_ids.ensure(V)); VGS_HIP_TRY(c, c->used_rank.ensure(V));
  VGS_HIP_TRY(c, c->head_flag.ensure(V + 1)); VGS_HIP_TRY(c, c->perm_a.ensure(V + 1));
  uint32_t* used_flag = c->head_flag.p;  // free after voxelize
  uint32_t* excl = c->perm_a.p;
  const unsigned nb = (unsigned)((V + FEAT_TB - 1) / FEAT_TB);
  if (c->P.method == 3)
    hipLaunchKernelGGL(k_features<64>, dim3((unsigned)((V + 63) / 64)), dim3(64), 0, c->stream, c->xs.p, c->ys.p, c->zs.p, c->vox_start.p, V,
                       -1 /* every supervoxel is used (SS:1288) */, 1, c->vox_code.p, c->node.p, used_flag);
  else
    hipLaunchKernelGGL(k_features<FEAT_TB>, dim3(nb), dim3(FEAT_TB), 0, c->stream, c->xs.p, c->ys.p, c->zs.p, c->vox_start.p, V,
                       c->P.points_min, 0, c->vox_code.p, c->node.p, used_flag);
  size_t bytes = 0;
  VGS_HIP_TRY(c, rocprim::exclusive_scan(nullptr, bytes, used_flag, excl, 0u, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
  VGS_HIP_TRY(c, c->sort_tmp.ensure(bytes));
  VGS_HIP_TRY(c, rocprim::exclusive_scan(c->sort_tmp.p, bytes, used_flag, excl, 0u, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
  VGS_HIP_TRY(c, c->counters.ensure(64));
  unsigned long long* d_nu = (unsigned long long*)(c->counters.p + 36);
  hipLaunchKernelGGL(k_compact_used, dim3(nb), dim3(FEAT_TB), 0, c->stream, used_flag, excl, V, c->used_ids.p, c->used_rank.p, d_nu);
  unsigned long long nu = 0;
  c->bricks_ready = false;
  if (c->P.method == 2 && vgs_can_split_readback(c)) {
    // the brick table of the adjacency stage needs the voxel codes and the used flags, not the count: it is built while the host
    // fetches the count (vgs_stage_adjacency finds bricks_ready and skips its own build)
    { vgs_status sb = vgs_readback_begin(c, d_nu, 8); if (sb != VGS_OK) return sb; }
    { vgs_status bs = vgs_build_bricks(c, c->node.p); if (bs != VGS_OK) return bs; }
    c->bricks_ready = true;
    { vgs_status se = vgs_readback_end(c, &nu, 8); if (se != VGS_OK) return se; }
  } else {
    VGS_READBACK(c, &nu, d_nu, 8);
  }
  VGS_HIP_TRY(c, hipGetLastError());
  c->U = (int64_t)nu;
  c->counts[VGS_N_USED] = c->U;
  return VGS_OK;
}
