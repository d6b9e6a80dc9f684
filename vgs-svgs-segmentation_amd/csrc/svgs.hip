// svgs.hip -- SVGS-specific stages: supervoxels as graph nodes.
// Replaces the bookkeeping half of createSupervoxels (supervoxel_segmentation.h:279-331: per-point labels ->
// per-supervoxel point lists, label 0 = unassigned, label == max_label never visited, SS:313) and
// findAllSupervoxelNeighbors / getOneSupervoxelNeighbor (SS:1477-1521, 1544-1563: radius search of
// graph_resolution_ over supervoxel CENTROIDS, FLANN semantics as in adjacency.hip).  Attributes, local cuts,
// crossValidation, closestCheck and clustering reuse the VGS kernels with the SS formula switches.
//
// Centroids are not on a lattice, so the neighbour search uses a uniform grid with cell = graph_size: supervoxels
// are sorted by cell code, each wavefront probes the 27 cells around its supervoxel through a hash of the occupied
// cells, keeps what the float predicate accepts and sorts by (d2, id) in LDS.
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

#include "vgs_context.hpp"

#define SV_ROW 512  // row stride of the neighbour table (entries per supervoxel); overflow is reported

// ---------------------------------------------------------------- grouping points by supervoxel
__global__ void k_sv_keys(const int32_t* __restrict__ label, int64_t n, int32_t max_label, const float* __restrict__ xyz, int stride_f,
                          uint32_t* __restrict__ key, uint32_t* __restrict__ perm) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t l = label[i];
  // supervoxels_point_idx_ is built for k in [0, max_label): label 0 collects nothing, label max_label is skipped (SS:301-322)
  key[i] = (l > 0 && l < max_label) ? (uint32_t)l : 0xffffffffu;
  perm[i] = (uint32_t)i;
  (void)xyz; (void)stride_f;
}

__global__ void k_sv_heads(const uint32_t* __restrict__ key, int64_t n, uint32_t* __restrict__ head) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t k = key[j];
  head[j] = (k != 0xffffffffu && (j == 0 || key[j - 1] != k)) ? 1u : 0u;
}

__global__ void k_sv_count_valid(const uint32_t* __restrict__ key, int64_t n, unsigned long long* __restrict__ n_valid) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (key[mid] != 0xffffffffu) lo = mid + 1; else hi = mid; }
  *n_valid = (unsigned long long)lo;
}

__global__ void k_sv_table(const uint32_t* __restrict__ key, const uint32_t* __restrict__ head, const uint32_t* __restrict__ scan, int64_t n,
                           uint32_t* __restrict__ pt_vox, uint32_t* __restrict__ vox_start, uint64_t* __restrict__ vox_code) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t k = key[j];
  if (k == 0xffffffffu) { pt_vox[j] = 0xffffffffu; return; }
  const uint32_t s = scan[j] - 1u;
  pt_vox[j] = s;
  if (head[j]) { vox_start[s] = (uint32_t)j; vox_code[s] = (uint64_t)k; }  // vox_code keeps the VCCS label of the supervoxel
}

__global__ void k_sv_gather(const float* __restrict__ xyz, int stride_f, const uint32_t* __restrict__ perm, int64_t nf,
                            float* __restrict__ xs, float* __restrict__ ys, float* __restrict__ zs) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nf) return;
  const float* p = xyz + (int64_t)perm[j] * stride_f;
  xs[j] = p[0]; ys[j] = p[1]; zs[j] = p[2];
}

__global__ void k_sv_set(uint32_t* p, uint32_t v) { *p = v; }

vgs_status vgs_stage_svgs_group(vgs_ctx* c) {
  const int64_t N = c->N;
  c->V = 0; c->Nf = 0; c->U = 0;
  if (N == 0) return VGS_OK;
  VGS_HIP_TRY(c, c->sv_key_a.ensure(N)); VGS_HIP_TRY(c, c->sv_key_b.ensure(N));
  VGS_HIP_TRY(c, c->perm_a.ensure(N)); VGS_HIP_TRY(c, c->perm_b.ensure(N));
  VGS_HIP_TRY(c, c->head_flag.ensure(N)); VGS_HIP_TRY(c, c->pt_vox.ensure(N));
  VGS_HIP_TRY(c, c->counters.ensure(64));
  const int TB = 256;
  const unsigned nb = (unsigned)((N + TB - 1) / TB);
  hipLaunchKernelGGL(k_sv_keys, dim3(nb), dim3(TB), 0, c->stream, c->sv_label.p, N, c->sv_max_label, c->xyz, c->stride_f, c->sv_key_a.p,
                     c->perm_a.p);
  size_t sort_bytes = 0, scan_bytes = 0;
  VGS_HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, sort_bytes, c->sv_key_a.p, c->sv_key_b.p, c->perm_a.p, c->perm_b.p, (size_t)N, 0, 32, c->stream));
  VGS_HIP_TRY(c, rocprim::inclusive_scan(nullptr, scan_bytes, c->head_flag.p, c->perm_a.p, (size_t)N, rocprim::plus<uint32_t>(), c->stream));
  VGS_HIP_TRY(c, c->sort_tmp.ensure(std::max(sort_bytes, scan_bytes)));
  // stable sort by label: ascending point index inside a supervoxel, like points_label_map (SS:296-306)
  VGS_HIP_TRY(c, rocprim::radix_sort_pairs(c->sort_tmp.p, sort_bytes, c->sv_key_a.p, c->sv_key_b.p, c->perm_a.p, c->perm_b.p, (size_t)N, 0, 32, c->stream));
  unsigned long long* d_cnt = (unsigned long long*)c->counters.p;
  hipLaunchKernelGGL(k_sv_heads, dim3(nb), dim3(TB), 0, c->stream, c->sv_key_b.p, N, c->head_flag.p);
  hipLaunchKernelGGL(k_sv_count_valid, dim3(1), dim3(1), 0, c->stream, c->sv_key_b.p, N, d_cnt);
  uint32_t* scan = c->perm_a.p;
  VGS_HIP_TRY(c, rocprim::inclusive_scan(c->sort_tmp.p, scan_bytes, c->head_flag.p, scan, (size_t)N, rocprim::plus<uint32_t>(), c->stream));
  unsigned long long nf = 0;
  uint32_t s_total = 0;
  VGS_HIP_TRY(c, hipMemcpyAsync(&nf, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipMemcpyAsync(&s_total, scan + (N - 1), 4, hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->Nf = (int64_t)nf;
  c->V = (int64_t)s_total;
  VGS_HIP_TRY(c, c->vox_code.ensure(c->V + 1)); VGS_HIP_TRY(c, c->vox_start.ensure(c->V + 1));
  hipLaunchKernelGGL(k_sv_table, dim3(nb), dim3(TB), 0, c->stream, c->sv_key_b.p, c->head_flag.p, scan, N, c->pt_vox.p, c->vox_start.p,
                     c->vox_code.p);
  hipLaunchKernelGGL(k_sv_set, dim3(1), dim3(1), 0, c->stream, c->vox_start.p + c->V, (uint32_t)c->Nf);
  VGS_HIP_TRY(c, c->xs.ensure(c->Nf + 1)); VGS_HIP_TRY(c, c->ys.ensure(c->Nf + 1)); VGS_HIP_TRY(c, c->zs.ensure(c->Nf + 1));
  if (c->Nf > 0)
    hipLaunchKernelGGL(k_sv_gather, dim3((unsigned)((c->Nf + TB - 1) / TB)), dim3(TB), 0, c->stream, c->xyz, c->stride_f, c->perm_b.p, c->Nf,
                       c->xs.p, c->ys.p, c->zs.p);
  VGS_HIP_TRY(c, hipGetLastError());
  c->counts[VGS_N_FINITE] = c->Nf;
  c->counts[VGS_N_VOXELS] = c->V;
  c->counts[VGS_N_SUPERVOXELS] = c->V;
  return VGS_OK;
}

// ---------------------------------------------------------------- radius search over centroids
#define SV_OFF (1 << 20)
__device__ __forceinline__ uint64_t sv_cell_code(int cx, int cy, int cz) {
  return ((uint64_t)(uint32_t)(cx + SV_OFF) << 42) | ((uint64_t)(uint32_t)(cy + SV_OFF) << 21) | (uint64_t)(uint32_t)(cz + SV_OFF);
}
__device__ __forceinline__ uint32_t sv_hash_slot(uint64_t code, uint32_t hbits) {
  return (uint32_t)((code * 0x9E3779B97F4A7C15ull) >> (64 - hbits));
}

__global__ void k_sv_cells(const NodeRec* __restrict__ node, int64_t S, double cell, uint64_t* __restrict__ code, uint32_t* __restrict__ id) {
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const int cx = (int)floor((double)node[s].c[0] / cell), cy = (int)floor((double)node[s].c[1] / cell), cz = (int)floor((double)node[s].c[2] / cell);
  code[s] = sv_cell_code(cx, cy, cz);
  id[s] = (uint32_t)s;
}

__global__ void k_sv_cell_heads(const uint64_t* __restrict__ code, int64_t S, uint32_t* __restrict__ head) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= S) return;
  head[j] = (j == 0 || code[j - 1] != code[j]) ? 1u : 0u;
}

__global__ void k_sv_cell_table(const uint64_t* __restrict__ code, const uint32_t* __restrict__ head, const uint32_t* __restrict__ scan, int64_t S,
                                uint32_t* __restrict__ cell_start, unsigned long long* __restrict__ hkey, uint32_t* __restrict__ hval,
                                uint32_t hbits) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= S || !head[j]) return;
  const uint32_t cidx = scan[j] - 1u;
  cell_start[cidx] = (uint32_t)j;
  const unsigned long long key = code[j] + 1ull;
  const uint32_t mask = (1u << hbits) - 1u;
  uint32_t s = sv_hash_slot(key, hbits);
  while (true) {
    unsigned long long prev = atomicCAS(&hkey[s], 0ull, key);
    if (prev == 0ull) { hval[s] = cidx; return; }
    s = (s + 1) & mask;
  }
}

__global__ __launch_bounds__(64) void k_sv_neighbours(const NodeRec* __restrict__ node, int64_t S, double cell, float r2,
                                                      const uint64_t* __restrict__ hkey, const uint32_t* __restrict__ hval, uint32_t hbits,
                                                      const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ sorted_id,
                                                      uint64_t* __restrict__ adj_key, uint32_t* __restrict__ adj_cnt,
                                                      uint32_t* __restrict__ adj_nall, unsigned long long* __restrict__ overflow) {
  __shared__ uint64_t lst[SV_ROW];
  const int lane = threadIdx.x;
  const int64_t s = blockIdx.x;
  if (s >= S) return;
  const float qx = node[s].c[0], qy = node[s].c[1], qz = node[s].c[2];
  const int cx = (int)floor((double)qx / cell), cy = (int)floor((double)qy / cell), cz = (int)floor((double)qz / cell);
  int cnt = 0;
  for (int o = 0; o < 27; ++o) {
    const int dx = o % 3 - 1, dy = (o / 3) % 3 - 1, dz = o / 9 - 1;
    const uint64_t key = sv_cell_code(cx + dx, cy + dy, cz + dz) + 1ull;
    const uint32_t mask = (1u << hbits) - 1u;
    uint32_t slot = sv_hash_slot(key, hbits);
    int cidx = -1;
    while (true) {
      const uint64_t k = hkey[slot];
      if (k == key) { cidx = (int)hval[slot]; break; }
      if (k == 0ull) break;
      slot = (slot + 1) & mask;
    }
    if (cidx < 0) continue;
    const uint32_t b = cell_start[cidx], e = cell_start[cidx + 1];
    for (uint32_t base = b; base < e; base += 64) {
      const uint32_t j = base + lane;
      bool keep = false;
      uint64_t k64 = 0;
      if (j < e) {
        const uint32_t t = sorted_id[j];
        // flann::L2_Simple<float>: result += diff * diff over x, y, z
        const float tx = qx - node[t].c[0], ty = qy - node[t].c[1], tz = qz - node[t].c[2];
        const float d2 = (tx * tx + ty * ty) + tz * tz;
        if (d2 < r2) { keep = true; k64 = ((uint64_t)vm_bits(d2) << 32) | (uint64_t)t; }
      }
      const unsigned long long mk = __ballot(keep);
      if (keep) { const int pos = cnt + __popcll(mk & ((1ull << lane) - 1ull)); if (pos < SV_ROW) lst[pos] = k64; }
      cnt += __popcll(mk);
    }
  }
  if (cnt > SV_ROW) { if (lane == 0) atomicAdd(overflow, 1ull); cnt = SV_ROW; }
  int np = 64;
  while (np < cnt) np <<= 1;
  for (int k = cnt + lane; k < np; k += 64) lst[k] = ~0ull;
  __syncthreads();
  for (int size = 2; size <= np; size <<= 1) {
    for (int strd = size >> 1; strd > 0; strd >>= 1) {
      for (int t = lane; t < (np >> 1); t += 64) {
        const int lo = ((t / strd) * (strd << 1)) + (t & (strd - 1));
        const int hi = lo + strd;
        const bool up = ((lo & size) == 0);
        const uint64_t a = lst[lo], bb = lst[hi];
        if ((a > bb) == up) { lst[lo] = bb; lst[hi] = a; }
      }
      __syncthreads();
    }
  }
  uint64_t* row = adj_key + s * SV_ROW;
  for (int k = lane; k < cnt; k += 64) row[k] = lst[k];
  if (lane == 0) { adj_cnt[s] = (uint32_t)cnt; adj_nall[s] = (uint32_t)cnt; }
}

vgs_status vgs_stage_svgs_neighbours(vgs_ctx* c) {
  const int64_t S = c->V;
  if (S == 0) return VGS_OK;
  const double cell = (double)c->P.graph_size;
  const float r2 = (float)(cell * cell);
  const int TB = 256;
  const unsigned nb = (unsigned)((S + TB - 1) / TB);
  VGS_HIP_TRY(c, c->cell_code_a.ensure(S)); VGS_HIP_TRY(c, c->cell_code_b.ensure(S));
  VGS_HIP_TRY(c, c->cell_id_a.ensure(S)); VGS_HIP_TRY(c, c->cell_id_b.ensure(S)); VGS_HIP_TRY(c, c->cell_start.ensure(S + 1));
  VGS_HIP_TRY(c, c->head_flag.ensure(S + 1)); VGS_HIP_TRY(c, c->perm_a.ensure(S + 1));
  hipLaunchKernelGGL(k_sv_cells, dim3(nb), dim3(TB), 0, c->stream, c->node.p, S, cell, c->cell_code_a.p, c->cell_id_a.p);
  size_t sort_bytes = 0, scan_bytes = 0;
  VGS_HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, sort_bytes, c->cell_code_a.p, c->cell_code_b.p, c->cell_id_a.p, c->cell_id_b.p, (size_t)S, 0, 63, c->stream));
  VGS_HIP_TRY(c, rocprim::inclusive_scan(nullptr, scan_bytes, c->head_flag.p, c->perm_a.p, (size_t)S, rocprim::plus<uint32_t>(), c->stream));
  VGS_HIP_TRY(c, c->sort_tmp.ensure(std::max(sort_bytes, scan_bytes)));
  VGS_HIP_TRY(c, rocprim::radix_sort_pairs(c->sort_tmp.p, sort_bytes, c->cell_code_a.p, c->cell_code_b.p, c->cell_id_a.p, c->cell_id_b.p, (size_t)S, 0, 63, c->stream));
  hipLaunchKernelGGL(k_sv_cell_heads, dim3(nb), dim3(TB), 0, c->stream, c->cell_code_b.p, S, c->head_flag.p);
  uint32_t* scan = c->perm_a.p;
  VGS_HIP_TRY(c, rocprim::inclusive_scan(c->sort_tmp.p, scan_bytes, c->head_flag.p, scan, (size_t)S, rocprim::plus<uint32_t>(), c->stream));
  uint32_t n_cells = 0;
  VGS_HIP_TRY(c, hipMemcpyAsync(&n_cells, scan + (S - 1), 4, hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  uint32_t hbits = 4;
  while ((1ull << hbits) < (uint64_t)(2 * (uint64_t)n_cells + 2)) ++hbits;
  const size_t H = (size_t)1 << hbits;
  VGS_HIP_TRY(c, c->hkey.ensure(H)); VGS_HIP_TRY(c, c->hval.ensure(H));
  VGS_HIP_TRY(c, hipMemsetAsync(c->hkey.p, 0, H * 8, c->stream));
  hipLaunchKernelGGL(k_sv_cell_table, dim3(nb), dim3(TB), 0, c->stream, c->cell_code_b.p, c->head_flag.p, scan, S, c->cell_start.p,
                     (unsigned long long*)c->hkey.p, c->hval.p, hbits);
  hipLaunchKernelGGL(k_sv_set, dim3(1), dim3(1), 0, c->stream, c->cell_start.p + n_cells, (uint32_t)S);
  c->adj_stride = SV_ROW;
  c->adj_pruned = vgs_unused_are_inert(c->P);  // every supervoxel is used: the rows are complete either way
  VGS_HIP_TRY(c, c->adj_key.ensure((size_t)S * SV_ROW));
  VGS_HIP_TRY(c, c->adj_cnt.ensure(S)); VGS_HIP_TRY(c, c->adj_mused.ensure(S));
  VGS_HIP_TRY(c, c->counters.ensure(64));
  unsigned long long* d_ovf = (unsigned long long*)c->counters.p + 40;
  VGS_HIP_TRY(c, hipMemsetAsync(d_ovf, 0, 8, c->stream));
  hipLaunchKernelGGL(k_sv_neighbours, dim3((unsigned)S), dim3(64), 0, c->stream, c->node.p, S, cell, r2, c->hkey.p, c->hval.p, hbits,
                     c->cell_start.p, c->cell_id_b.p, c->adj_key.p, c->adj_cnt.p, c->adj_mused.p, d_ovf);
  unsigned long long ovf = 0;
  VGS_HIP_TRY(c, hipMemcpyAsync(&ovf, d_ovf, 8, hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  VGS_HIP_TRY(c, hipGetLastError());
  if (ovf) { c->err = "a supervoxel has more than 512 neighbours within graph_size"; return VGS_E_UNSUPPORTED; }
  c->adj_r2 = r2;
  return VGS_OK;
}
