// vccs.hip -- stage a12: supervoxel creation on the GPU (svgs_supervoxels).
// Replaces the call to pcl::SupervoxelClustering<PointXYZRGBA>(voxel_res, seed_res).extract + refineSupervoxels(5)
// and getLabeledCloud / getMaxLabel in createSupervoxels (reference supervoxel_segmentation.h:265-284).
//
// PARITY UNPINNED: pcl::SupervoxelClustering (VCCS, Papon et al. 2013) is a third-party dependency that is not
// under /root/reference and cannot be built here.  This is a VCCS-STYLE restatement of the published algorithm
// (SURVEY.md B.4): voxel adjacency (26-connectivity), seeds on a seed_res grid snapped to the nearest voxel,
// breadth-first expansion for int(1.8 * seed_res / voxel_res) rounds where a voxel goes to the adjacent
// supervoxel minimising  D = w_s * |dx| / seed_res + w_n * (1 - |n1.n2|)  (colour term is zero: the reference
// copies XYZ into XYZRGBA, SS:258), centroids updated every round, then five refinement passes (re-seed at the
// centroid, expand again).  PCL's owner iteration is sequential and order dependent; here every round is one
// synchronous sweep (all voxels decide from the state at the start of the round), which is deterministic and
// data-parallel.  It is validated by invariants (tests/test_gpu_vccs.py) and by exact agreement with the oracle's
// CPU restatement of THIS algorithm (oracle/refcpu_vccs.cpp); sums are integer fixed point so that atomics commute.
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

#include "vgs_context.hpp"
#include "vccs_common.h"
#include "brick_table.hpp"

// ---------------------------------------------------------------- voxel attributes
__global__ void k_vccs_centroid(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
                                const uint32_t* __restrict__ vox_start, int64_t V, float* __restrict__ cen) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const uint32_t s = vox_start[v], e = vox_start[v + 1];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (uint32_t j = s; j < e; ++j) { sx = sx + xs[j]; sy = sy + ys[j]; sz = sz + zs[j]; }
  const int cnt = (int)(e - s);
  cen[3 * v + 0] = sx / cnt; cen[3 * v + 1] = sy / cnt; cen[3 * v + 2] = sz / cnt;
}

// 26-neighbour table (offset order of vccs_common.h) and the voxel normal from the neighbourhood's centroids.
// Neighbours are looked up in the brick table (a few MB, L2 resident) instead of a per-voxel hash.
__global__ void k_vccs_neighbours(const uint64_t* __restrict__ vox_code, int64_t V, int depth, const Brick* __restrict__ bricks, uint32_t hbits,
                                  const float* __restrict__ cen, int32_t* __restrict__ nbr, float* __restrict__ nrm, int4* __restrict__ nbr4) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  int quad[28];
  for (int k = 26; k < 28; ++k) quad[k] = -1;
  const uint64_t code = vox_code[v];
  const uint32_t kx = vm_compact21(code >> 2), ky = vm_compact21(code >> 1), kz = vm_compact21(code);
  const uint32_t lim = 1u << depth;
  float pts[27 * 3];
  int np = 0;
  pts[0] = cen[3 * v]; pts[1] = cen[3 * v + 1]; pts[2] = cen[3 * v + 2];
  np = 1;
  for (int o = 0; o < 26; ++o) {
    int dx, dy, dz;
    vccs_offset(o, &dx, &dy, &dz);
    const uint32_t nx = kx + (uint32_t)dx, ny = ky + (uint32_t)dy, nz = kz + (uint32_t)dz;
    int t = -1;
    if (nx < lim && ny < lim && nz < lim) { bool unused_flag; t = brick_find(bricks, hbits, nx, ny, nz, &unused_flag); }
    nbr[(int64_t)o * V + v] = t;   // [26][V]: a wavefront reads one offset of 64 consecutive voxels
    quad[o] = t;
    if (nrm && t >= 0) { pts[3 * np] = cen[3 * t]; pts[3 * np + 1] = cen[3 * t + 1]; pts[3 * np + 2] = cen[3 * t + 2]; ++np; }
  }
  if (nrm) {   // (vccs_mode 1 has its own two-ring normals and passes null)
    float n[3];
    vccs_normal_from_points(pts, np, n);
    nrm[3 * v] = n[0]; nrm[3 * v + 1] = n[1]; nrm[3 * v + 2] = n[2];
  }
  // the same table four entries to a load ([7][V] of int4: the expansion rounds read it 54 times and are bound by their number of
  // vector-memory instructions, not by bytes)
  if (nbr4)
    for (int k = 0; k < 7; ++k) nbr4[(int64_t)k * V + v] = make_int4(quad[4 * k], quad[4 * k + 1], quad[4 * k + 2], quad[4 * k + 3]);
}

// ---------------------------------------------------------------- seeding
__global__ void k_vccs_cell_codes(const float* __restrict__ cen, int64_t V, float min_x, float min_y, float min_z, float seed,
                                  uint64_t* __restrict__ code, uint32_t* __restrict__ id) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  code[v] = vccs_seed_cell(cen[3 * v], cen[3 * v + 1], cen[3 * v + 2], min_x, min_y, min_z, seed);
  id[v] = (uint32_t)v;
}

__global__ void k_vccs_heads(const uint64_t* __restrict__ code, int64_t V, uint32_t* __restrict__ head) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V) return;
  head[j] = (j == 0 || code[j - 1] != code[j]) ? 1u : 0u;
}

// sorted by cell code: voxel at sorted position j belongs to cell rank scan[j]-1; pick the voxel nearest to the cell centre
__global__ void k_vccs_pick_seeds(const uint64_t* __restrict__ code, const uint32_t* __restrict__ sorted_id, const uint32_t* __restrict__ scan,
                                  int64_t V, const float* __restrict__ cen, float min_x, float min_y, float min_z, float seed,
                                  unsigned long long* __restrict__ seed_key) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  // the voxels of a cell are neighbours in the sorted order: the minimum is taken over each run of equal cells among the lanes first
  // (segmented scan, six shuffle steps) and only the last lane of a run goes to memory -- a cell holds some thirty voxels
  uint32_t cell = 0xffffffffu;
  unsigned long long key = ~0ull;
  if (j < V) {
    const uint32_t v = sorted_id[j];
    cell = scan[j] - 1u;
    const float d2 = vccs_cell_center_d2(code[j], cen[3 * v], cen[3 * v + 1], cen[3 * v + 2], min_x, min_y, min_z, seed);
    key = ((unsigned long long)vm_bits(d2) << 32) | (unsigned long long)v;
  }
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t co = (uint32_t)__shfl_up((int)cell, o, 64);
    const unsigned long long ko = ((unsigned long long)(uint32_t)__shfl_up((int)(key >> 32), o, 64) << 32) | (unsigned long long)(uint32_t)__shfl_up((int)key, o, 64);
    if (lane >= o && co == cell && ko < key) key = ko;   // runs are contiguous: equal cells o lanes apart span one run
  }
  const uint32_t cnext = (uint32_t)__shfl_down((int)cell, 1, 64);
  if (cell == 0xffffffffu || (lane < 63 && cnext == cell)) return;   // not the last lane of its run
  atomicMin(&seed_key[cell], key);
}

__global__ void k_vccs_fill_u64(unsigned long long* p, int64_t n, unsigned long long v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---------------------------------------------------------------- expansion
struct VccsState {  // per supervoxel
  float c[3];
  float n[3];
};

__global__ void k_vccs_reset(int64_t V, int32_t* __restrict__ label, float* __restrict__ dist) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  label[v] = -1;
  dist[v] = 3.0e38f;
}

__global__ void k_vccs_plant(const unsigned long long* __restrict__ seed_key, int K, const float* __restrict__ cen, const float* __restrict__ nrm,
                             int32_t* __restrict__ label, float* __restrict__ dist, VccsState* __restrict__ st,
                             long long* __restrict__ sums, unsigned int* __restrict__ count) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const unsigned long long key = seed_key[k];
  for (int a = 0; a < 6; ++a) sums[6 * k + a] = 0;
  count[k] = 0;
  if (key == ~0ull) { for (int a = 0; a < 3; ++a) { st[k].c[a] = 0.f; st[k].n[a] = 0.f; } return; }  // supervoxel without voxels
  const uint32_t v = (uint32_t)key;
  label[v] = k;
  dist[v] = 0.0f;
  for (int a = 0; a < 3; ++a) {
    st[k].c[a] = cen[3 * v + a]; st[k].n[a] = nrm[3 * v + a];
    sums[6 * k + a] = vccs_fix_pos(cen[3 * v + a]); sums[6 * k + 3 + a] = vccs_fix_nrm(nrm[3 * v + a]);
  }
  count[k] = 1;
}

// one synchronous round: every voxel looks at the labels its 26 neighbours (and itself) had at the start of the round.
// The per-supervoxel sums move with the voxels that change owner.  A workgroup's 256 voxels are neighbours in space (Morton
// order) and belong to a handful of supervoxels, so their contributions are first added up in a small LDS table keyed by
// label and only the table goes to memory: one global atomic per (workgroup, supervoxel, component) instead of fourteen
// per voxel that changes -- the contended 64-bit atomics were half of the stage's time.  Integer sums: any order, same result.
#define VX_SLOTS 128
// The best offer among the labels of the 26 neighbours.  A voxel on a border sees two to four other supervoxels, each through
// several neighbours, and the evaluation of one label is a gather of the supervoxel's state and a wait for it: the DISTINCT labels
// are enumerated first and the states of the first four requested together.  (Round 4: the rounds turned out to be bound by
// instruction issue, not by memory -- 26 x (five compares, an if-chain over four registers) of half-rate VALU per voxel.)  The
// enumeration is arithmetic: key = (label ^ ref) - 1 with ref = the voxel's own label is 0xffffffff for the own label, >= 2^31 for
// "no neighbour" (-1) and < 2^31 for every other label (labels stay below 2^31 - 2); the smallest key is a minimum over 26 values (v_min3), and the next one
// is the minimum of key - (previous + 1) in unsigned arithmetic, where everything already taken wraps to the top: one subtraction
// and a third of a min3 per neighbour and level, all full rate, and a level is only run while some lane of the wavefront still
// has a label to find.  Minimum by (distance, label): the order of evaluation does not matter.
__device__ __forceinline__ void vccs_best_offer(const int (&nl)[26], int own, const float (&c)[3], const float (&n)[3], const VccsState* __restrict__ st,
                                                float w_s_over_seed, float w_n, int& best_l, float& best_d) {
  const uint32_t ref = vccs_enum_ref(own);
  uint32_t key[26];
#pragma unroll
  for (int o = 0; o < 26; ++o) key[o] = vccs_enum_key(nl[o], ref);
  int sl[4] = {-1, -1, -1, -1};
  bool more = true;
  uint32_t off = 0u;
  bool live[4] = {false, false, false, false};   // (wave-uniform) some lane has a label at this level
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q > 0 && !live[q - 1]) break;
    const uint32_t kq = vccs_enum_next(key, 26, off);
    more = more && vccs_enum_valid(kq);
    live[q] = __ballot(more) != 0ull;
    if (!live[q]) break;
    sl[q] = more ? vccs_enum_label(kq, ref) : -1;
    off = kq + 1u;
  }
  VccsState A[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (live[q]) A[q] = st[sl[q] >= 0 ? sl[q] : 0];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (!live[q]) continue;
    const int l = sl[q];
    const float d = vccs_distance(c, n, A[q].c, A[q].n, w_s_over_seed, w_n);
    if (l >= 0 && (d < best_d || (d == best_d && l < best_l))) { best_d = d; best_l = l; }
  }
  // a fifth label and beyond (corners where many supervoxels meet): one at a time
  while (live[3] && __ballot(more) != 0ull) {
    const uint32_t kq = vccs_enum_next(key, 26, off);
    more = more && vccs_enum_valid(kq);
    if (__ballot(more) == 0ull) break;
    off = kq + 1u;
    const int l = vccs_enum_label(kq, ref);
    if (more && l >= 0) {
      const float d = vccs_distance(c, n, st[l].c, st[l].n, w_s_over_seed, w_n);
      if (d < best_d || (d == best_d && l < best_l)) { best_d = d; best_l = l; }
    }
  }
}
__global__ __launch_bounds__(256) void k_vccs_expand(int64_t V, const int4* __restrict__ nbr4, const float* __restrict__ cen,
                              const float* __restrict__ nrm, const int32_t* __restrict__ label_in, const float* __restrict__ dist_in,
                              const VccsState* __restrict__ st, float w_s_over_seed, float w_n, int32_t* __restrict__ label_out,
                              float* __restrict__ dist_out, long long* __restrict__ sums, unsigned int* __restrict__ count) {
  __shared__ int s_key[VX_SLOTS];
  __shared__ unsigned long long s_sum[VX_SLOTS][6];
  __shared__ int s_cnt[VX_SLOTS];
  for (int k = threadIdx.x; k < VX_SLOTS; k += blockDim.x) {
    s_key[k] = -1; s_cnt[k] = 0;
    for (int a = 0; a < 6; ++a) s_sum[k][a] = 0ull;
  }
  __syncthreads();
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < V) {
    int best_l = label_in[v];
    float best_d = dist_in[v];
    const float c[3] = {cen[3 * v], cen[3 * v + 1], cen[3 * v + 2]};
    const float n[3] = {nrm[3 * v], nrm[3 * v + 1], nrm[3 * v + 2]};
    const int own = best_l;
    // all 26 neighbour ids, then all 26 labels: independent loads in flight together (the serial walk was latency bound)
    int nl[26];
    // (Round 4 measured this table as 16-bit rank differences -- voxels are in Morton order, a spatial neighbour is almost always
    // within +-32 k ranks; 52 instead of 104 bytes per voxel and round: the stage went from 12.6 to 16.9 ms.  The kernel is bound by
    // its 52 vector-memory instructions per thread, not by HBM bytes; the escape path for far neighbours adds 26 masked loads.)
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int4 q = nbr4[(int64_t)k * V + v];
      nl[4 * k] = q.x; nl[4 * k + 1] = q.y;
      if (k < 6) { nl[4 * k + 2] = q.z; nl[4 * k + 3] = q.w; }
    }
#pragma unroll
    for (int o = 0; o < 26; ++o) nl[o] = nl[o] >= 0 ? label_in[nl[o]] : -1;
    vccs_best_offer(nl, own, c, n, st, w_s_over_seed, w_n, best_l, best_d);
    label_out[v] = best_l;
    dist_out[v] = best_d;
    // moving this voxel's contribution from its old owner to the new one gives exactly the sums a full re-accumulation
    // would (and after the first rounds only the frontier moves)
    if (best_l != own) {
      long long f[6];
      for (int a = 0; a < 3; ++a) { f[a] = vccs_fix_pos(c[a]); f[3 + a] = vccs_fix_nrm(n[a]); }
      for (int side = 0; side < 2; ++side) {
        const int l = side ? best_l : own;
        if (l < 0) continue;
        const long long sgn = side ? 1 : -1;
        // slot of label l in the workgroup's table (open addressing); a full table sends the contribution straight to memory
        int slot = -1;
        unsigned int h = ((unsigned int)l * 2654435761u) >> 25;   // 7 bits
        for (int probe = 0; probe < VX_SLOTS; ++probe) {
          const int prev = atomicCAS(&s_key[h], -1, l);
          if (prev == -1 || prev == l) { slot = (int)h; break; }
          h = (h + 1) & (VX_SLOTS - 1);
        }
        if (slot >= 0) {
          for (int a = 0; a < 6; ++a) atomicAdd(&s_sum[slot][a], (unsigned long long)(sgn * f[a]));
          atomicAdd(&s_cnt[slot], (int)sgn);
        } else {
          for (int a = 0; a < 6; ++a) atomicAdd((unsigned long long*)&sums[6 * l + a], (unsigned long long)(sgn * f[a]));
          if (side) atomicAdd(&count[l], 1u); else atomicSub(&count[l], 1u);
        }
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < VX_SLOTS; k += blockDim.x) {
    const int l = s_key[k];
    if (l < 0) continue;
    for (int a = 0; a < 6; ++a) { const unsigned long long x = s_sum[k][a]; if (x) atomicAdd((unsigned long long*)&sums[6 * l + a], x); }
    const int dc = s_cnt[k];
    if (dc > 0) atomicAdd(&count[l], (unsigned int)dc); else if (dc < 0) atomicSub(&count[l], (unsigned int)(-dc));
  }
}

// ---------------------------------------------------------------- expansion over tiles (round 4)
// The round above is bound by its vector-memory instructions: per voxel seven loads of neighbour ids and 26 divergent gathers of
// their labels, 216 B through the memory pipeline.  Voxels are sorted by Morton code, so an aligned 8 x 8 x 8 block of lattice
// cells -- a TILE -- is one contiguous run of the voxel array (about 60 voxels on a scanned surface).  One wavefront takes one tile:
// it writes the labels of the tile's own voxels (a coalesced load) and of the voxels in the one-cell shell around it (a list built
// once per run: at most seven tiles see a voxel in their shell) into a 10 x 10 x 10 array in LDS, and every voxel then reads its
// 26 neighbours from there (the array is first filled with "no voxel": four 16-byte LDS stores per lane).  Per voxel and round
// that is 4 B of label, 2 B of cell index and about 12 B of shell instead of the 216.  Same synchronous round, same result.
#define VT_CELLS 1000
#define VT_SHELL 488
__device__ __forceinline__ void vt_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
__global__ void k_vccs_tile_heads(const uint64_t* __restrict__ code, int64_t V, uint32_t* __restrict__ head) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  head[v] = (v == 0 || (code[v - 1] >> 9) != (code[v] >> 9)) ? 1u : 0u;
}
__global__ void k_vccs_tile_starts(const uint32_t* __restrict__ head, const uint32_t* __restrict__ scan, int64_t V, uint32_t* __restrict__ tile_start,
                                   uint32_t* __restrict__ tile_of) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  if (tile_of) tile_of[v] = scan[v] - 1u;
  if (head[v]) tile_start[scan[v] - 1u] = (uint32_t)v;
  if (v == V - 1) tile_start[scan[v]] = (uint32_t)V;
}
// Round 4, second step: the same arrangement BUILDS the tiles.  The wavefront writes the ids of its own voxels into the 10^3 array,
// looks the 488 cells of the shell up in the brick table (five times fewer probes than 26 per voxel), and the array then answers
// "who is my neighbour" for the normals: the [26][V] neighbour table of k_vccs_neighbours is neither written nor read.
__global__ __launch_bounds__(64) void k_vccs_tile_setup(const uint64_t* __restrict__ vox_code, int depth, const Brick* __restrict__ bricks, uint32_t hbits,
                                                        const float* __restrict__ cen, const uint32_t* __restrict__ tile_start, uint16_t* __restrict__ cell,
                                                        float* __restrict__ nrm, uint2* __restrict__ halo, unsigned long long pool_cap,
                                                        unsigned long long* __restrict__ pool_count, uint2* __restrict__ meta,
                                                        const uint32_t* __restrict__ tile_of, int32_t* __restrict__ nbr_tiles) {
  __shared__ __attribute__((aligned(16))) int I[VT_CELLS];
  __shared__ uint2 s_ent[VT_SHELL];
  __shared__ int s_n;
  __shared__ int s_nt[27];   // (vccs_mode 1) the tile on each of the 26 sides that holds a voxel of the shell, -1: none
  if (threadIdx.x < 27) s_nt[threadIdx.x] = -1;
  const int lane = threadIdx.x;
  const int t = (int)blockIdx.x;
  for (int i = lane; i < VT_CELLS / 4; i += 64) ((int4*)I)[i] = make_int4(-1, -1, -1, -1);
  if (lane == 0) s_n = 0;
  const uint32_t ts = tile_start[t], te = tile_start[t + 1];
  const uint64_t code0 = vox_code[ts];
  const uint32_t ox = vm_compact21(code0 >> 2) & ~7u, oy = vm_compact21(code0 >> 1) & ~7u, oz = vm_compact21(code0) & ~7u;
  const uint32_t lim = 1u << depth;
  vt_sync();
  for (uint32_t v = ts + (uint32_t)lane; v < te; v += 64u) {
    const uint32_t lo = (uint32_t)vox_code[v] & 511u;   // z0 y0 x0 z1 y1 x1 z2 y2 x2 from bit 0
    const int z = (int)((lo & 1u) | ((lo >> 2) & 2u) | ((lo >> 4) & 4u));
    const int y = (int)(((lo >> 1) & 1u) | ((lo >> 3) & 2u) | ((lo >> 5) & 4u));
    const int x = (int)(((lo >> 2) & 1u) | ((lo >> 4) & 2u) | ((lo >> 6) & 4u));
    const int ci = (x + 1) + 10 * (y + 1) + 100 * (z + 1);
    I[ci] = (int)v;
    cell[v] = (uint16_t)ci;
  }
  for (int ci = lane; ci < VT_CELLS; ci += 64) {
    const int cx = ci % 10, cy = (ci / 10) % 10, cz = ci / 100;
    if ((unsigned)(cx - 1) < 8u && (unsigned)(cy - 1) < 8u && (unsigned)(cz - 1) < 8u) continue;   // one of the tile's own cells
    const uint32_t gx = ox + (uint32_t)(cx - 1), gy = oy + (uint32_t)(cy - 1), gz = oz + (uint32_t)(cz - 1);   // (below zero wraps above lim)
    if (!(gx < lim && gy < lim && gz < lim)) continue;
    bool unused_flag;
    const int u = brick_find(bricks, hbits, gx, gy, gz, &unused_flag);
    if (u < 0) continue;
    I[ci] = u;
    s_ent[atomicAdd(&s_n, 1)] = make_uint2((uint32_t)u, (uint32_t)ci);
    if (nbr_tiles) s_nt[(cx == 0 ? 0 : (cx == 9 ? 2 : 1)) + 3 * (cy == 0 ? 0 : (cy == 9 ? 2 : 1)) + 9 * (cz == 0 ? 0 : (cz == 9 ? 2 : 1))] = (int)tile_of[u];   // (every voxel on that side says the same)
  }
  vt_sync();
  if (nbr_tiles && lane < 27) nbr_tiles[(int64_t)t * 27 + lane] = s_nt[lane];
  const int n = s_n;
  unsigned long long off = 0ull;
  if (lane == 0) off = atomicAdd(pool_count, (unsigned long long)n);
  off = ((unsigned long long)(uint32_t)__shfl((int)(off >> 32), 0, 64) << 32) | (unsigned long long)(uint32_t)__shfl((int)off, 0, 64);
  const bool fits = off + (unsigned long long)n <= pool_cap;   // always (a voxel lies in the shell of at most seven tiles)
  if (fits) for (int i = lane; i < n; i += 64) halo[off + (unsigned long long)i] = s_ent[i];
  if (lane == 0) meta[t] = fits ? make_uint2((uint32_t)off, (uint32_t)n) : make_uint2(0u, 0u);
  // the voxel normals from the neighbourhood's centroids, neighbours in the order of vccs_offset as k_vccs_neighbours takes them
  // (vccs_mode 1 has its own two-ring normals: nrm is null)
  if (nrm == nullptr) return;
  for (uint32_t v = ts + (uint32_t)lane; v < te; v += 64u) {
    const uint32_t lo = (uint32_t)vox_code[v] & 511u;
    const int z = (int)((lo & 1u) | ((lo >> 2) & 2u) | ((lo >> 4) & 4u));
    const int y = (int)(((lo >> 1) & 1u) | ((lo >> 3) & 2u) | ((lo >> 5) & 4u));
    const int x = (int)(((lo >> 2) & 1u) | ((lo >> 4) & 2u) | ((lo >> 6) & 4u));
    const int ci = (x + 1) + 10 * (y + 1) + 100 * (z + 1);
    // (vccs_normal_from_points without its array of 27 centroids -- 324 bytes of scratch per lane, 3 GB of scratch traffic per run:
    // the centroids are gathered twice in the same order, the second time out of the cache; same operations in the same order)
    const float p0[3] = {cen[3 * (int64_t)v], cen[3 * (int64_t)v + 1], cen[3 * (int64_t)v + 2]};
    float sx = 0.f, sy = 0.f, sz = 0.f;
    sx = sx + p0[0]; sy = sy + p0[1]; sz = sz + p0[2];
    int np = 1;
#pragma unroll
    for (int o = 0; o < 26; ++o) {
      const int k = o < 13 ? o : o + 1;   // vccs_offset(o)
      const int u = I[ci + (k % 3 - 1) + 10 * ((k / 3) % 3 - 1) + 100 * (k / 9 - 1)];
      if (u >= 0) { sx = sx + cen[3 * (int64_t)u]; sy = sy + cen[3 * (int64_t)u + 1]; sz = sz + cen[3 * (int64_t)u + 2]; ++np; }
    }
    float nn[3] = {0.f, 0.f, 0.f};
    if (np >= 3) {
      const float mx = sx / np, my = sy / np, mz = sz / np;
      float C[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      {
        const float d0 = p0[0] - mx, d1 = p0[1] - my, d2 = p0[2] - mz;
        C[0] = C[0] + d0 * d0; C[1] = C[1] + d0 * d1; C[2] = C[2] + d0 * d2;
        C[4] = C[4] + d1 * d1; C[5] = C[5] + d1 * d2; C[8] = C[8] + d2 * d2;
      }
#pragma unroll
      for (int o = 0; o < 26; ++o) {
        const int k = o < 13 ? o : o + 1;
        const int u = I[ci + (k % 3 - 1) + 10 * ((k / 3) % 3 - 1) + 100 * (k / 9 - 1)];
        if (u >= 0) {
          const float d0 = cen[3 * (int64_t)u] - mx, d1 = cen[3 * (int64_t)u + 1] - my, d2 = cen[3 * (int64_t)u + 2] - mz;
          C[0] = C[0] + d0 * d0; C[1] = C[1] + d0 * d1; C[2] = C[2] + d0 * d2;
          C[4] = C[4] + d1 * d1; C[5] = C[5] + d1 * d2; C[8] = C[8] + d2 * d2;
        }
      }
      vccs_normal_finish(C, p0, nn);
    }
    nrm[3 * (int64_t)v] = nn[0]; nrm[3 * (int64_t)v + 1] = nn[1]; nrm[3 * (int64_t)v + 2] = nn[2];
  }
}
// slots of a tile's table of per-supervoxel sums: a tile's window meets up to two dozen supervoxels at seed = 5 voxels; a contribution
// that finds the table full goes to memory with seven 64-bit atomics (all of them that way: the stage takes twice as long).  Config 4
// with 16 / 32 / 64 / 128 slots: 11.30 / 10.92 / 11.05 / 11.2 ms (64 slots cost the fifth LDS granule)
#ifndef VT_SLOTS
#define VT_SLOTS 32
#endif
__global__ __launch_bounds__(64) void k_vccs_expand_tiles(int T, const uint32_t* __restrict__ tile_start, const uint2* __restrict__ meta,
                              const uint2* __restrict__ halo, const uint16_t* __restrict__ cell,
                              const float* __restrict__ cen, const float* __restrict__ nrm, const int32_t* __restrict__ label_in,
                              const float* __restrict__ dist_in, const VccsState* __restrict__ st, float w_s_over_seed, float w_n,
                              int32_t* __restrict__ label_out, float* __restrict__ dist_out, long long* __restrict__ sums,
                              unsigned int* __restrict__ count) {
  // one wavefront = one workgroup = one tile: no workgroup barrier, a tile of 500 voxels holds nobody else up, 5 KB of LDS
  __shared__ __attribute__((aligned(16))) int L[VT_CELLS];
  __shared__ int s_key[VT_SLOTS];
  __shared__ int s_sum[VT_SLOTS][6];
  __shared__ int s_cnt[VT_SLOTS];
  const int lane = threadIdx.x;
  const int t = (int)blockIdx.x;
  for (int k = lane; k < VT_SLOTS; k += 64) {
    s_key[k] = -1; s_cnt[k] = 0;
    for (int a = 0; a < 6; ++a) s_sum[k][a] = 0;
  }
  for (int i = lane; i < VT_CELLS / 4; i += 64) ((int4*)L)[i] = make_int4(-1, -1, -1, -1);   // an empty cell reads as "no neighbour"
  const uint32_t ts = tile_start[t], te = tile_start[t + 1];
  const uint2 m = meta[t];
  long long base[3];
  for (int a = 0; a < 3; ++a) base[a] = vccs_fix_pos(cen[3 * (int64_t)ts + a]);   // (wave-uniform)
  // what the first 64 voxels need is requested before the shell is staged
  struct Vox { int ci; float d; float c[3]; float n[3]; };
  auto load_vox = [&](uint32_t v, Vox& x) {
    x.ci = (int)cell[v]; x.d = dist_in[v];
    for (int a = 0; a < 3; ++a) { x.c[a] = cen[3 * (int64_t)v + a]; x.n[a] = nrm[3 * (int64_t)v + a]; }
  };
  // (requesting the second 64 voxels' inputs up front as well costs 18 registers -- four instead of five wavefronts per SIMD -- and
  // was measured slower: 12.14 against 11.98 ms for config 4)
  Vox x0;
  int lab0 = -1;
  const uint32_t v0 = ts + (uint32_t)lane;
  if (v0 < te) { load_vox(v0, x0); lab0 = label_in[v0]; }
  uint2 e0 = make_uint2(0u, 0u), e1 = make_uint2(0u, 0u);
  if ((uint32_t)lane < m.y) e0 = halo[m.x + (uint32_t)lane];
  if ((uint32_t)lane + 64u < m.y) e1 = halo[m.x + (uint32_t)lane + 64u];
  int h0 = -1, h1 = -1;
  if ((uint32_t)lane < m.y) h0 = label_in[e0.x];
  if ((uint32_t)lane + 64u < m.y) h1 = label_in[e1.x];
  vt_sync();
  if ((uint32_t)lane < m.y) L[e0.y] = h0;
  if ((uint32_t)lane + 64u < m.y) L[e1.y] = h1;
  for (uint32_t i = (uint32_t)lane + 128u; i < m.y; i += 64u) { const uint2 e = halo[m.x + i]; L[e.y] = label_in[e.x]; }
  if (v0 < te) L[x0.ci] = lab0;
  for (uint32_t v = v0 + 64u; v < te; v += 64u) L[cell[v]] = label_in[v];
  vt_sync();
  bool touched = false;
  for (uint32_t v = v0; v < te; v += 64u) {
    Vox x;
    if (v == v0) x = x0; else load_vox(v, x);
    const int ci = x.ci;
    const int own = L[ci];
    int best_l = own;
    float best_d = x.d;
    int nl[26];
#pragma unroll
    for (int o = 0; o < 26; ++o) {
      const int k = o < 13 ? o : o + 1;   // vccs_offset(o)
      nl[o] = L[ci + (k % 3 - 1) + 10 * ((k / 3) % 3 - 1) + 100 * (k / 9 - 1)];
    }
    vccs_best_offer(nl, own, x.c, x.n, st, w_s_over_seed, w_n, best_l, best_d);
    label_out[v] = best_l;
    dist_out[v] = best_d;
    if (best_l != own) {
      touched = true;
      long long f[6];
      // (vccs_fix_pos / vccs_fix_nrm in single precision: the scaling by a power of two is exact in either format, so the rounding
      // to an integer sees the same real number and the results are equal; the double-precision forms cost eight f64 operations each)
      for (int a = 0; a < 3; ++a) { f[a] = (long long)__builtin_rintf(x.c[a] * 65536.0f); f[3 + a] = (long long)__builtin_rintf(x.n[a] * 1048576.0f); }
      // inside a tile the fixed-point positions differ by a few voxels: 32-bit sums relative to the tile's first voxel (a position
      // farther than 2^21 units away -- voxels of several metres -- goes straight to memory); normals are below 2^20 as they are
      int r[6];
      bool small = true;
      for (int a = 0; a < 3; ++a) {
        const long long d = f[a] - base[a];
        small = small && d > -(1ll << 21) && d < (1ll << 21);
        r[a] = (int)d; r[3 + a] = (int)f[3 + a];
      }
      for (int side = 0; side < 2; ++side) {
        const int l = side ? best_l : own;
        if (l < 0) continue;
        const int sgn = side ? 1 : -1;
        int slot = -1;
        if (small) {
          unsigned int h = ((unsigned int)l * 2654435761u) >> (VT_SLOTS == 128 ? 25 : (VT_SLOTS == 64 ? 26 : (VT_SLOTS == 32 ? 27 : 28)));
          for (int probe = 0; probe < VT_SLOTS; ++probe) {
            const int prev = atomicCAS(&s_key[h], -1, l);
            if (prev == -1 || prev == l) { slot = (int)h; break; }
            h = (h + 1) & (VT_SLOTS - 1);
          }
        }
        if (slot >= 0) {
          for (int a = 0; a < 6; ++a) atomicAdd(&s_sum[slot][a], sgn * r[a]);
          atomicAdd(&s_cnt[slot], sgn);
        } else {
          for (int a = 0; a < 6; ++a) atomicAdd((unsigned long long*)&sums[6 * l + a], (unsigned long long)((long long)sgn * f[a]));
          if (side) atomicAdd(&count[l], 1u); else atomicSub(&count[l], 1u);
        }
      }
    }
  }
  if (__ballot(touched) == 0ull) return;
  vt_sync();
  for (int k = lane; k < VT_SLOTS; k += 64) {
    const int l = s_key[k];
    if (l >= 0) {
      const int dc = s_cnt[k];
      for (int a = 0; a < 6; ++a) {
        const long long x = (long long)s_sum[k][a] + (a < 3 ? (long long)dc * base[a] : 0ll);
        if (x) atomicAdd((unsigned long long*)&sums[6 * l + a], (unsigned long long)x);
      }
      if (dc > 0) atomicAdd(&count[l], (unsigned int)dc); else if (dc < 0) atomicSub(&count[l], (unsigned int)(-dc));
    }
  }
}

__global__ void k_vccs_accumulate(int64_t V, const int32_t* __restrict__ label, const float* __restrict__ cen, const float* __restrict__ nrm,
                                  long long* __restrict__ sums /* 6 per supervoxel */, unsigned int* __restrict__ count) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const int l = label[v];
  if (l < 0) return;
  for (int a = 0; a < 3; ++a) {
    atomicAdd((unsigned long long*)&sums[6 * l + a], (unsigned long long)vccs_fix_pos(cen[3 * v + a]));
    atomicAdd((unsigned long long*)&sums[6 * l + 3 + a], (unsigned long long)vccs_fix_nrm(nrm[3 * v + a]));
  }
  atomicAdd(&count[l], 1u);
}

__global__ void k_vccs_update(int K, const long long* __restrict__ sums, const unsigned int* __restrict__ count, VccsState* __restrict__ st) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  if (count[k] == 0) return;  // keeps its previous state
  vccs_state_from_sums(&sums[6 * k], count[k], st[k].c, st[k].n);
}

// refinement: new seed = member voxel closest to the supervoxel centroid (smallest (distance^2, voxel id) key).  The minimum
// is first taken per workgroup in an LDS table keyed by label (see k_vccs_expand), then once per table entry in memory.
__global__ __launch_bounds__(256) void k_vccs_reseed(int64_t V, const int32_t* __restrict__ label, const float* __restrict__ cen,
                                                     const VccsState* __restrict__ st, unsigned long long* __restrict__ seed_key) {
  __shared__ int s_key[VX_SLOTS];
  __shared__ unsigned long long s_min[VX_SLOTS];
  for (int k = threadIdx.x; k < VX_SLOTS; k += blockDim.x) { s_key[k] = -1; s_min[k] = ~0ull; }
  __syncthreads();
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < V) {
    const int l = label[v];
    if (l >= 0) {
      const float dx = cen[3 * v] - st[l].c[0], dy = cen[3 * v + 1] - st[l].c[1], dz = cen[3 * v + 2] - st[l].c[2];
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      const unsigned long long key = ((unsigned long long)vm_bits(d2) << 32) | (unsigned long long)v;
      int slot = -1;
      unsigned int h = ((unsigned int)l * 2654435761u) >> 25;
      for (int probe = 0; probe < VX_SLOTS; ++probe) {
        const int prev = atomicCAS(&s_key[h], -1, l);
        if (prev == -1 || prev == l) { slot = (int)h; break; }
        h = (h + 1) & (VX_SLOTS - 1);
      }
      if (slot >= 0) atomicMin(&s_min[slot], key);
      else atomicMin(&seed_key[l], key);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < VX_SLOTS; k += blockDim.x)
    if (s_key[k] >= 0 && s_min[k] != ~0ull) atomicMin(&seed_key[s_key[k]], s_min[k]);
}

__global__ void k_vccs_point_labels(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ pt_vox, const int32_t* __restrict__ label,
                                    int64_t N, int32_t* __restrict__ out) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const uint32_t v = pt_vox[j];
  out[perm[j]] = (v == 0xffffffffu || label[v] < 0) ? 0 : label[v] + 1;  // getLabeledCloud: 0 = unassigned (SS:303)
}

// ================================================================ vccs_mode 1: PCL's own order
// pcl::SupervoxelClustering (1.8.1) step by step, restated from recollection (PCL is not available here: parity with it stays
// unpinned; oracle/refcpu_vccs.cpp: vccs_pcl_supervoxels is the sequential CPU statement these kernels equal label for label):
//   * normals from the 2-ring of voxel centroids (indices = [v] + for every neighbour t: [t] + the neighbours of t, the leaf
//     itself being its own neighbour) through computePointNormal's single-pass covariance, flipped towards (0,0,0);
//   * seeds: the voxel nearest to the centre of every occupied seed_res cell, cells in ascending Morton order
//     (getOccupiedVoxelCenters), rejected unless more than 0.05 (seed/2)^2 pi / res^2 voxels lie within seed/2;
//   * expansion, (int)(1.8 seed / res) - 1 rounds: the supervoxels take their turns ONE AFTER THE OTHER in label order, each
//     offering its current centroid to the 27-neighbourhood of the leaves it still owns at its turn; a voxel goes to an offer
//     strictly below its recorded distance, and recorded distances persist.  Sequential semantics, parallel execution: a leaf
//     is expanded from at its owner S's turn iff no supervoxel with a smaller label took it earlier in the round ("live"), and
//     whether T < S takes it depends only on the leaves of T that are live -- a recursion on smaller labels.  The live flags
//     are therefore iterated to their fixed point (a few sweeps), then every voxel folds the offers of the supervoxels that
//     touch it through live leaves in label order.  The result is exactly the sequential one (tests: equal to the oracle);
//   * centroids from all leaves after every round (integer fixed-point sums), supervoxels without voxels removed for good;
//   * refineSupervoxels(5): re-seed at the member voxel nearest to the centroid, reset every voxel, expand again.
// Divergences from PCL that remain: the lattice is the class's own octree's (PCL's adjacency octree anchors at the
// cloud's minimum corner) and the seed grid hangs on it; re-seeding searches the supervoxel's own voxels (PCL: kd-tree over all);
// refineNormals is skipped; FLANN's tie order is replaced by (distance, voxel id); centroid sums are fixed point.
__device__ __forceinline__ int vccs_nbr27(const int32_t* __restrict__ nbr, int64_t V, int64_t v, int o) {
  int dx, dy, dz;
  vccs_offset27(o, &dx, &dy, &dz);
  if (dx == 0 && dy == 0 && dz == 0) return (int)v;
  return nbr[(int64_t)vccs_index26(dx, dy, dz) * V + v];
}

// owner == null: computeVoxelData's two-ring normals of every voxel.  owner != null (round 5): SupervoxelHelper::refineNormals -- only the
// leaves of the voxel's own supervoxel count, in both rings, the voxel itself is not listed up front, an unowned voxel keeps its normal
__global__ void k_pcl_accu1(int64_t V, const int32_t* __restrict__ nbr, const float* __restrict__ cen, VccsAccu* __restrict__ A1,
                            const int32_t* __restrict__ owner) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V) return;
  const int k = owner ? owner[t] : 0;
  if (k < 0) return;
  VccsAccu A;
  vccs_accu_zero(&A);
  for (int o = 0; o < 27; ++o) { const int u = vccs_nbr27(nbr, V, t, o); if (u >= 0 && (!owner || owner[u] == k)) vccs_accu_point(&A, &cen[3 * (int64_t)u]); }
  A1[t] = A;
}

__global__ void k_pcl_normals(int64_t V, const int32_t* __restrict__ nbr, const float* __restrict__ cen, const VccsAccu* __restrict__ A1,
                              float* __restrict__ nrm, const int32_t* __restrict__ owner) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const int k = owner ? owner[v] : 0;
  if (k < 0) return;
  VccsAccu A;
  vccs_accu_zero(&A);
  if (!owner) vccs_accu_point(&A, &cen[3 * v]);
  for (int o = 0; o < 27; ++o) {
    const int t = vccs_nbr27(nbr, V, v, o);
    if (t < 0 || (owner && owner[t] != k)) continue;
    vccs_accu_point(&A, &cen[3 * (int64_t)t]);
    const VccsAccu B = A1[t];
    vccs_accu_add(&A, &B);
  }
  float n[3];
  vccs_accu_normal(&A, &cen[3 * v], n);
  nrm[3 * v] = n[0]; nrm[3 * v + 1] = n[1]; nrm[3 * v + 2] = n[2];
}

// The same two kernels over the tiles of k_vccs_tile_setup (round 6): the tile's 10^3 array of voxel ids -- its own voxels and the shell, from
// the lists the live sweeps use -- answers "who is my neighbour at offset o", so the [26][V] table (26 x 4 bytes per voxel and kernel: three
// quarters of either kernel's traffic) is neither read nor, over tiles, built.  Same neighbours in the same order (vccs_offset27: dx outermost),
// same operations: the accumulators and normals are bit for bit those of k_pcl_accu1 / k_pcl_normals.
// What the neighbours contribute to both kernels -- owner and centroid -- is staged in LDS once per tile as well (entry = the tile's voxels, then
// its shell); the second kernel reads the first one's 40-byte accumulators from L2, a kilobyte per voxel, and that is what it waits for with or
// without the table.  A tile with more entries than the arrays hold (a solid block: up to 1000) reads everything from memory as before.
#define VTN_TB 128
#ifndef VTN_CAP
#define VTN_CAP 448
#endif
template <bool NORMALS>
__global__ __launch_bounds__(VTN_TB) void k_pclt_normals(const uint32_t* __restrict__ tile_start, const uint2* __restrict__ meta, const uint2* __restrict__ halo,
                                                         const uint16_t* __restrict__ cell, const float* __restrict__ cen, VccsAccu* __restrict__ A1 /* NORMALS: read */,
                                                         float* __restrict__ nrm, const int32_t* __restrict__ owner) {
  __shared__ __attribute__((aligned(16))) int I[VT_CELLS];   // cell -> entry (staged tiles) or voxel id, -1: no voxel
  __shared__ float4 s_pc[VTN_CAP];                                   // centroid, owner (bits)
  __shared__ int s_id[NORMALS ? VTN_CAP : 1];                        // NORMALS: entry -> voxel, for the first kernel's accumulators (from L2: staging
                                                                     // them as well -- 31 KB a tile, five workgroups per CU -- was measured slower, 388 against 275 us)
  const int tid = threadIdx.x;
  const int t = (int)blockIdx.x;
  for (int i = tid; i < VT_CELLS / 4; i += VTN_TB) ((int4*)I)[i] = make_int4(-1, -1, -1, -1);
  const uint32_t ts = tile_start[t], te = tile_start[t + 1];
  const uint2 m = meta[t];
  const uint32_t n_own = te - ts, n_ent = n_own + m.y;
  const bool staged = n_ent <= (uint32_t)VTN_CAP;   // (uniform)
  __syncthreads();
  for (uint32_t e = (uint32_t)tid; e < n_ent; e += VTN_TB) {
    uint32_t id, ci;
    if (e < n_own) { id = ts + e; ci = cell[id]; } else { const uint2 h = halo[m.x + (e - n_own)]; id = h.x; ci = h.y; }
    I[ci] = staged ? (int)e : (int)id;
    if (staged) {
      s_pc[e] = make_float4(cen[3 * (int64_t)id], cen[3 * (int64_t)id + 1], cen[3 * (int64_t)id + 2], __int_as_float(owner ? owner[id] : 0));
      if (NORMALS) s_id[e] = (int)id;
    }
  }
  __syncthreads();
  for (uint32_t e = (uint32_t)tid; e < n_own; e += VTN_TB) {
    const uint32_t v = ts + e;
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (staged) me = s_pc[e];
    const int k = owner ? (staged ? __float_as_int(me.w) : owner[v]) : 0;
    if (k < 0) continue;
    const int ci = (int)cell[v];
    VccsAccu A;
    vccs_accu_zero(&A);
    float pv[3];
    if (staged) { pv[0] = me.x; pv[1] = me.y; pv[2] = me.z; }
    else { pv[0] = cen[3 * (int64_t)v]; pv[1] = cen[3 * (int64_t)v + 1]; pv[2] = cen[3 * (int64_t)v + 2]; }
    if (NORMALS && !owner) vccs_accu_point(&A, pv);
#pragma unroll
    for (int o = 0; o < 27; ++o) {
      const int u = I[ci + (o / 9 - 1) + 10 * ((o / 3) % 3 - 1) + 100 * (o % 3 - 1)];   // vccs_offset27(o); o == 13 is the voxel itself
      if (u < 0) continue;
      float pu[3];
      VccsAccu B;
      if (staged) {
        const float4 q = s_pc[u];
        if (owner && __float_as_int(q.w) != k) continue;
        pu[0] = q.x; pu[1] = q.y; pu[2] = q.z;
        if (NORMALS) B = A1[s_id[u]];
      } else {
        if (owner && owner[u] != k) continue;
        pu[0] = cen[3 * (int64_t)u]; pu[1] = cen[3 * (int64_t)u + 1]; pu[2] = cen[3 * (int64_t)u + 2];
        if (NORMALS) B = A1[u];
      }
      vccs_accu_point(&A, pu);
      if (NORMALS) vccs_accu_add(&A, &B);
    }
    if (NORMALS) {
      float n[3];
      vccs_accu_normal(&A, pv, n);
      nrm[3 * (int64_t)v] = n[0]; nrm[3 * (int64_t)v + 1] = n[1]; nrm[3 * (int64_t)v + 2] = n[2];
    } else {
      A1[v] = A;
    }
  }
}

// seed cells in Morton order: sort key = Morton code of the cell's integer coordinates
__global__ void k_pcl_cell_codes(const float* __restrict__ cen, int64_t V, float min_x, float min_y, float min_z, float seed,
                                 uint64_t* __restrict__ code, uint32_t* __restrict__ id) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const uint64_t cell = vccs_seed_cell(cen[3 * v], cen[3 * v + 1], cen[3 * v + 2], min_x, min_y, min_z, seed);
  code[v] = vm_morton((uint32_t)((cell >> 42) & 0x1fffff), (uint32_t)((cell >> 21) & 0x1fffff), (uint32_t)(cell & 0x1fffff));
  id[v] = (uint32_t)v;
}

__global__ void k_pcl_pick_seeds(const uint64_t* __restrict__ mcode, const uint32_t* __restrict__ sorted_id, const uint32_t* __restrict__ scan,
                                 int64_t V, const float* __restrict__ cen, float min_x, float min_y, float min_z, float seed,
                                 unsigned long long* __restrict__ seed_key) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V) return;
  const uint32_t v = sorted_id[j];
  const uint32_t cellrank = scan[j] - 1u;
  const uint64_t m = mcode[j];
  const uint64_t cell = ((uint64_t)vm_compact21(m >> 2) << 42) | ((uint64_t)vm_compact21(m >> 1) << 21) | (uint64_t)vm_compact21(m);
  const float d2 = vccs_cell_center_d2(cell, cen[3 * v], cen[3 * v + 1], cen[3 * v + 2], min_x, min_y, min_z, seed);
  atomicMin(&seed_key[cellrank], ((unsigned long long)vm_bits(d2) << 32) | (unsigned long long)v);
}

// selectInitialSupervoxelSeeds: a seed voxel stays if more than min_points voxel centroids lie within seed / 2 of its own
// (round 6: sixteen lanes per seed -- the 343 probes of a cube of seven cells were one thread's dependent chain, 0.66 ms for 115 k seeds)
__global__ __launch_bounds__(256) void k_pcl_seed_filter(const unsigned long long* __restrict__ seed_key, int K0, const uint64_t* __restrict__ vox_code, int depth,
                                  const Brick* __restrict__ bricks, uint32_t hbits, const float* __restrict__ cen, float rad2, int R,
                                  float min_points, uint32_t* __restrict__ keep) {
  const int k = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4), sub = (int)(threadIdx.x & 15u);
  int num = 0;
  if (k < K0) {
    const uint32_t s = (uint32_t)seed_key[k];
    const uint64_t code = vox_code[s];
    const uint32_t kx = vm_compact21(code >> 2), ky = vm_compact21(code >> 1), kz = vm_compact21(code);
    const uint32_t lim = 1u << depth;
    const float sx = cen[3 * (int64_t)s], sy = cen[3 * (int64_t)s + 1], sz = cen[3 * (int64_t)s + 2];
    const int D = 2 * R + 1, cells = D * D * D;
    for (int ci = sub; ci < cells; ci += 16) {
      const int dx = ci % D - R, dy = (ci / D) % D - R, dz = ci / (D * D) - R;
      const uint32_t nx = kx + (uint32_t)dx, ny = ky + (uint32_t)dy, nz = kz + (uint32_t)dz;
      if (!(nx < lim && ny < lim && nz < lim)) continue;
      bool unused_flag;
      const int u = brick_find(bricks, hbits, nx, ny, nz, &unused_flag);
      if (u < 0) continue;
      const float ex = cen[3 * (int64_t)u] - sx, ey = cen[3 * (int64_t)u + 1] - sy, ez = cen[3 * (int64_t)u + 2] - sz;
      if ((ex * ex + ey * ey) + ez * ez < rad2) ++num;
    }
  }
  for (int o = 8; o > 0; o >>= 1) num += __shfl_xor(num, o, 64);   // (a count: any order)
  if (k < K0 && sub == 0) keep[k] = ((float)num > min_points) ? 1u : 0u;
}

__global__ void k_pcl_compact_seeds(const unsigned long long* __restrict__ seed_key, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ keep_scan_excl,
                                    int K0, uint32_t* __restrict__ seeds) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K0 && keep[k]) seeds[keep_scan_excl[k]] = (uint32_t)seed_key[k];
}

__global__ void k_pcl_reset(int64_t V, int32_t* __restrict__ owner, float* __restrict__ dist) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < V) { owner[v] = -1; dist[v] = 3.4028235e38f; }
}

// addLeaf of the seeds: the recorded distance of the seed voxel stays at its initial maximum
__global__ void k_pcl_plant_first(const uint32_t* __restrict__ seeds, int K, const float* __restrict__ cen, const float* __restrict__ nrm,
                                  int32_t* __restrict__ owner, VccsState* __restrict__ st, uint8_t* __restrict__ alive) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const uint32_t s = seeds[k];
  owner[s] = k;
  alive[k] = 1;
  for (int a = 0; a < 3; ++a) { st[k].c[a] = cen[3 * (int64_t)s + a]; st[k].n[a] = nrm[3 * (int64_t)s + a]; }
}

// reseedSupervoxels (round 5): the voxel nearest to the supervoxel's centroid among ALL voxels -- PCL asks its kd-tree; here the lattice
// cells around the centroid's own cell, shell by shell, through the brick table (vccs_common.h: vccs_nearest_voxel).  One thread per
// supervoxel: three or four shells, a couple of hundred probes.
__global__ __launch_bounds__(256) void k_pcl_reseed_nearest(int K, const uint8_t* __restrict__ alive, const VccsState* __restrict__ st, const float* __restrict__ cen,
                                     const Brick* __restrict__ bricks, uint32_t hbits, int depth, double min_x, double min_y, double min_z, double res_d,
                                     float res, unsigned long long* __restrict__ seed_key) {
  // (round 6: sixteen lanes per supervoxel share the cells of every shell; the minimum of a shell's keys and the test behind it are those of
  // vccs_nearest_voxel -- the oracle's one-thread form -- whatever lane found them)
  const int k = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4), sub = (int)(threadIdx.x & 15u);
  const bool on = k < K && alive[k < K ? k : 0] != 0;
  float c[3] = {0.f, 0.f, 0.f};
  uint32_t kx = 0, ky = 0, kz = 0;
  if (on) {
    c[0] = st[k].c[0]; c[1] = st[k].c[1]; c[2] = st[k].c[2];
    kx = vm_axis_key(c[0], min_x, res_d); ky = vm_axis_key(c[1], min_y, res_d); kz = vm_axis_key(c[2], min_z, res_d);
  }
  const uint32_t lim = 1u << depth;
  unsigned long long best = ~0ull;
  bool done = !on;
  for (int r = 0; ; ++r) {
    if (!done) {
      const int D = 2 * r + 1, cells = D * D * D;
      for (int ci = sub; ci < cells; ci += 16) {
        const int dx = ci % D - r, dy = (ci / D) % D - r, dz = ci / (D * D) - r;
        if (dx != -r && dx != r && dy != -r && dy != r && dz != -r && dz != r) continue;   // the shell only
        const uint32_t x = kx + (uint32_t)dx, y = ky + (uint32_t)dy, z = kz + (uint32_t)dz;
        if (!(x < lim && y < lim && z < lim)) continue;
        bool unused_flag;
        const int v = brick_find(bricks, hbits, x, y, z, &unused_flag);
        if (v < 0) continue;
        const float* p = cen + 3 * (int64_t)v;
        const float ex = p[0] - c[0], ey = p[1] - c[1], ez = p[2] - c[2];
        const unsigned long long key = ((unsigned long long)vm_bits((ex * ex + ey * ey) + ez * ez) << 32) | (unsigned long long)(uint32_t)v;
        best = key < best ? key : best;
      }
    }
    for (int o = 8; o > 0; o >>= 1) {
      const unsigned long long other = ((unsigned long long)(uint32_t)__shfl_xor((int)(best >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)best, o, 64);
      best = other < best ? other : best;
    }
    if (!done && ((best != ~0ull && vccs_shell_settles(vm_from_bits((uint32_t)(best >> 32)), r, res)) || (uint32_t)r >= lim)) done = true;
    if (__ballot(!done) == 0ull) break;
  }
  if (k < K && sub == 0) seed_key[k] = on ? best : ~0ull;
}
// the supervoxel keeps its centroid; its new only leaf is that voxel.  Supervoxels take their seeds in label order: a voxel that two of them
// name goes to the later one (atomicMax = the sequential overwrite), the earlier one starts the pass without a voxel
__global__ void k_pcl_plant_again(const unsigned long long* __restrict__ seed_key, int K, const uint8_t* __restrict__ alive, int32_t* __restrict__ owner) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K || !alive[k] || seed_key[k] == ~0ull) return;
  atomicMax(&owner[(uint32_t)seed_key[k]], k);
}

// The offers a voxel gets in one round, folded in label order: the supervoxels < limit that touch it through a live leaf of
// its 27-neighbourhood (itself included).  Returns the owner after the fold; *d_out its recorded distance.
__device__ __forceinline__ int vccs_pcl_fold(int64_t V, int64_t v, const int32_t* __restrict__ nbr, const int32_t* __restrict__ owner0,
                                             const float* __restrict__ dist0, const uint8_t* __restrict__ live, const float* __restrict__ cen,
                                             const float* __restrict__ nrm, const VccsState* __restrict__ st, float w_s_over_seed, float w_n, int limit,
                                             float* d_out) {
  int lab[27];
  int nl = 0;
  const int own = owner0[v];
  for (int o = 0; o < 27; ++o) {
    const int l = vccs_nbr27(nbr, V, v, o);
    if (l < 0) continue;
    const int s = owner0[l];
    // an offer of the voxel's own owner changes nothing as long as nobody else offers: it is kept in the list only for the case
    // that another supervoxel takes the voxel first, so the live flag of such a leaf is not even loaded when it cannot matter
    if (s < 0 || s >= limit) continue;
    if (!live[l]) continue;
    int pos = nl;
    bool dup = false;
    for (int q = 0; q < nl; ++q) if (lab[q] == s) { dup = true; break; }
    if (dup) continue;
    while (pos > 0 && lab[pos - 1] > s) { lab[pos] = lab[pos - 1]; --pos; }   // insertion: ascending labels
    lab[pos] = s;
    ++nl;
  }
  int cur = own;
  float cd = dist0[v];
  if (nl == 0 || (nl == 1 && lab[0] == own)) { *d_out = cd; return cur; }   // interior voxel: nobody else reaches it
  const float c[3] = {cen[3 * v], cen[3 * v + 1], cen[3 * v + 2]};
  const float n[3] = {nrm[3 * v], nrm[3 * v + 1], nrm[3 * v + 2]};
  for (int q = 0; q < nl; ++q) {
    const int s = lab[q];
    if (s == cur) continue;
    const float d = vccs_distance(c, n, st[s].c, st[s].n, w_s_over_seed, w_n);
    if (d < cd) { cd = d; cur = s; }
  }
  *d_out = cd;
  return cur;
}

// one sweep of the live flags: a leaf of S is live iff the offers of the supervoxels before S leave it with S
__global__ void k_pcl_live(int64_t V, const int32_t* __restrict__ nbr, const int32_t* __restrict__ owner0, const float* __restrict__ dist0,
                           const uint8_t* __restrict__ live_in, const float* __restrict__ cen, const float* __restrict__ nrm,
                           const VccsState* __restrict__ st, float w_s_over_seed, float w_n, uint8_t* __restrict__ live_out,
                           unsigned int* __restrict__ changed) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const int s = owner0[v];
  uint8_t l = 0;
  if (s >= 0) {
    float d;
    l = vccs_pcl_fold(V, v, nbr, owner0, dist0, live_in, cen, nrm, st, w_s_over_seed, w_n, s, &d) == s ? 1 : 0;
  }
  live_out[v] = l;
  if (l != live_in[v]) atomicOr(changed, 1u);
}

__global__ void k_pcl_live_init(int64_t V, const int32_t* __restrict__ owner, uint8_t* __restrict__ live) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < V) live[v] = owner[v] >= 0 ? 1 : 0;
}

__global__ void k_pcl_claim(int64_t V, const int32_t* __restrict__ nbr, const int32_t* __restrict__ owner0, const float* __restrict__ dist0,
                            const uint8_t* __restrict__ live, const float* __restrict__ cen, const float* __restrict__ nrm,
                            const VccsState* __restrict__ st, float w_s_over_seed, float w_n, int32_t* __restrict__ owner1, float* __restrict__ dist1) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  float d;
  owner1[v] = vccs_pcl_fold(V, v, nbr, owner0, dist0, live, cen, nrm, st, w_s_over_seed, w_n, 0x7fffffff, &d);
  dist1[v] = d;
}

// per-supervoxel sums of all leaves: added up per workgroup in an LDS table keyed by label first (a workgroup's 256 voxels are
// neighbours in space and belong to a handful of supervoxels), one global atomic per (workgroup, supervoxel, component)
__global__ __launch_bounds__(256) void k_pcl_accumulate(int64_t V, const int32_t* __restrict__ label, const float* __restrict__ cen,
                                                        const float* __restrict__ nrm, long long* __restrict__ sums, unsigned int* __restrict__ count) {
  __shared__ int s_key[VX_SLOTS];
  __shared__ unsigned long long s_sum[VX_SLOTS][6];
  __shared__ int s_cnt[VX_SLOTS];
  for (int k = threadIdx.x; k < VX_SLOTS; k += blockDim.x) {
    s_key[k] = -1; s_cnt[k] = 0;
    for (int a = 0; a < 6; ++a) s_sum[k][a] = 0ull;
  }
  __syncthreads();
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < V) {
    const int l = label[v];
    if (l >= 0) {
      long long f[6];
      for (int a = 0; a < 3; ++a) { f[a] = vccs_fix_pos(cen[3 * v + a]); f[3 + a] = vccs_fix_nrm(nrm[3 * v + a]); }
      int slot = -1;
      unsigned int h = ((unsigned int)l * 2654435761u) >> 25;   // 7 bits
      for (int probe = 0; probe < VX_SLOTS; ++probe) {
        const int prev = atomicCAS(&s_key[h], -1, l);
        if (prev == -1 || prev == l) { slot = (int)h; break; }
        h = (h + 1) & (VX_SLOTS - 1);
      }
      if (slot >= 0) {
        for (int a = 0; a < 6; ++a) atomicAdd(&s_sum[slot][a], (unsigned long long)f[a]);
        atomicAdd(&s_cnt[slot], 1);
      } else {   // a full table sends the contribution straight to memory
        for (int a = 0; a < 6; ++a) atomicAdd((unsigned long long*)&sums[6 * l + a], (unsigned long long)f[a]);
        atomicAdd(&count[l], 1u);
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < VX_SLOTS; k += blockDim.x) {
    const int l = s_key[k];
    if (l < 0) continue;
    for (int a = 0; a < 6; ++a) { const unsigned long long x = s_sum[k][a]; if (x) atomicAdd((unsigned long long*)&sums[6 * l + a], x); }
    if (s_cnt[k] > 0) atomicAdd(&count[l], (unsigned int)s_cnt[k]);
  }
}

__global__ void k_pcl_update(int K, const long long* __restrict__ sums, const unsigned int* __restrict__ count, VccsState* __restrict__ st,
                             uint8_t* __restrict__ alive) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K || !alive[k]) return;
  if (count[k] == 0) { alive[k] = 0; return; }   // removed for good
  vccs_state_from_sums(&sums[6 * k], count[k], st[k].c, st[k].n);
}

__global__ void k_pcl_max_label(int K, const uint8_t* __restrict__ alive, unsigned int* __restrict__ out) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K && alive[k]) atomicMax(out, (unsigned int)(k + 1));
}

// ---- vccs_mode 1 over tiles (round 4).  The live sweeps and the final claim read, for each of a voxel's 27 cells, the owner and the
// live flag: both travel in one word, P = owner << 1 | live (-1: no owner), staged per tile in the 10^3 LDS array of
// k_vccs_expand_tiles.  The fold -- distinct owners below the limit that reach the voxel through a live leaf, in ascending label
// order, an offer strictly below the recorded distance taking the voxel -- enumerates them as successive minima (vccs_enum_next: the
// keys ARE the labels, so the order is the sequential one) instead of an insertion sort over 27 gathered values.
// (round 5: also marks every tile "changed" for the round's first sweep -- was a memset per round between two dependent kernels)
__global__ void k_pclt_init(int64_t V, const int32_t* __restrict__ owner, int32_t* __restrict__ P, uint32_t* __restrict__ tchg, int NT) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < V) { const int s = owner[v]; P[v] = s < 0 ? -1 : ((s << 1) | 1); }
  if (tchg && v < NT) tchg[v] = 0x01010101u;
}
template <bool CLAIM>
__global__ __launch_bounds__(64) void k_pclt_sweep(const uint32_t* __restrict__ tile_start, const uint2* __restrict__ meta, const uint2* __restrict__ halo,
                                                   const uint16_t* __restrict__ cell, const int32_t* P_in /* may be P_out: in-place sweeps */, const float* __restrict__ dist0,
                                                   const float* __restrict__ cen, const float* __restrict__ nrm, const VccsState* __restrict__ st,
                                                   float w_s_over_seed, float w_n, int32_t* P_out, unsigned int* __restrict__ changed,
                                                   int32_t* __restrict__ owner1, float* __restrict__ dist1, long long* __restrict__ sums,
                                                   unsigned int* __restrict__ count, const int32_t* __restrict__ nbr_tiles,
                                                   const uint32_t* __restrict__ tchg_in, uint32_t* __restrict__ tchg_out,
                                                   unsigned int* __restrict__ dbg_moved = nullptr) {
  // A sweep's output for a tile is a function of the flags in its window -- its own voxels and the shell, which lie in the tile and in
  // the tiles on its 26 sides.  If none of these changed a flag in the previous sweep, the window is what it was then and so is the
  // output: the tile copies its flags and is done.  From the second sweep of a round on most tiles are that quiet.
  if (!CLAIM && tchg_in) {
    const int t = (int)blockIdx.x, lane = threadIdx.x;
    uint32_t f = 0u;
    if (lane < 27) { const int nt = lane == 13 ? t : nbr_tiles[(int64_t)t * 27 + lane]; if (nt >= 0) f = tchg_in[nt]; }
    if (__ballot(f != 0u) == 0ull) {
      if (P_out != P_in) {   // (in place -- round 6 -- a quiet tile has nothing to copy)
        const uint32_t ts = tile_start[t], te = tile_start[t + 1];
        for (uint32_t v = ts + (uint32_t)lane; v < te; v += 64u) P_out[v] = P_in[v];
      }
      if (lane == 0) tchg_out[t] = 0u;
      return;
    }
  }
  __shared__ __attribute__((aligned(16))) int L[VT_CELLS];
  // CLAIM: the per-supervoxel sums move with the voxels that change owner, per tile in an LDS table keyed by label (k_vccs_expand_tiles)
  __shared__ int s_key[CLAIM ? VT_SLOTS : 1];
  __shared__ int s_sum[CLAIM ? VT_SLOTS : 1][6];
  __shared__ int s_cnt[CLAIM ? VT_SLOTS : 1];
  const int lane = threadIdx.x;
  const int t = (int)blockIdx.x;
  for (int i = lane; i < VT_CELLS / 4; i += 64) ((int4*)L)[i] = make_int4(-1, -1, -1, -1);
  if (CLAIM) for (int k = lane; k < VT_SLOTS; k += 64) {
    s_key[k] = -1; s_cnt[k] = 0;
    for (int a = 0; a < 6; ++a) s_sum[k][a] = 0;
  }
  const uint32_t ts = tile_start[t], te = tile_start[t + 1];
  const uint2 m = meta[t];
  long long base[3] = {0, 0, 0};
  if (CLAIM) for (int a = 0; a < 3; ++a) base[a] = vccs_fix_pos(cen[3 * (int64_t)ts + a]);   // (wave-uniform)
  bool touched = false;
  const uint32_t v0 = ts + (uint32_t)lane;
  int ci0 = 0, p0 = -1;
  if (v0 < te) { ci0 = (int)cell[v0]; p0 = P_in[v0]; }
  uint2 e0 = make_uint2(0u, 0u), e1 = make_uint2(0u, 0u);
  if ((uint32_t)lane < m.y) e0 = halo[m.x + (uint32_t)lane];
  if ((uint32_t)lane + 64u < m.y) e1 = halo[m.x + (uint32_t)lane + 64u];
  int h0 = -1, h1 = -1;
  if ((uint32_t)lane < m.y) h0 = P_in[e0.x];
  if ((uint32_t)lane + 64u < m.y) h1 = P_in[e1.x];
  vt_sync();
  if ((uint32_t)lane < m.y) L[e0.y] = h0;
  if ((uint32_t)lane + 64u < m.y) L[e1.y] = h1;
  for (uint32_t i = (uint32_t)lane + 128u; i < m.y; i += 64u) { const uint2 e = halo[m.x + i]; L[e.y] = P_in[e.x]; }
  if (v0 < te) L[ci0] = p0;
  for (uint32_t v = v0 + 64u; v < te; v += 64u) L[cell[v]] = P_in[v];
  vt_sync();
  bool any_change = false;
  for (uint32_t v = v0; v < te; v += 64u) {
    const int ci = v == v0 ? ci0 : (int)cell[v];
    const int p = L[ci];
    const int own = p < 0 ? -1 : (p >> 1);
    if (!CLAIM && own < 0) { P_out[v] = -1; continue; }   // no leaf, no live flag
    const uint32_t limit = CLAIM ? 0x7fffffffu : (uint32_t)own;
    uint32_t key[27];
#pragma unroll
    for (int o = 0; o < 27; ++o) {
      const int q = L[ci + (o % 3 - 1) + 10 * ((o / 3) % 3 - 1) + 100 * (o / 9 - 1)];
      const uint32_t s = (uint32_t)(q >> 1);
      key[o] = (q >= 0 && (q & 1) && s < limit) ? s : 0xffffffffu;
    }
    int cur = own;
    float cd = 0.f;
    bool have = false;
    float c[3] = {0.f, 0.f, 0.f}, n[3] = {0.f, 0.f, 0.f};
    uint32_t off = 0u;
    bool more = true;
    while (true) {
      const uint32_t kq = vccs_enum_next(key, 27, off);
      more = more && vccs_enum_valid(kq);
      if (__ballot(more) == 0ull) break;
      off = kq + 1u;
      if (more && (int)kq != cur) {
        if (!have) {
          have = true;
          cd = dist0[v];
          for (int a = 0; a < 3; ++a) { c[a] = cen[3 * (int64_t)v + a]; n[a] = nrm[3 * (int64_t)v + a]; }
        }
        const int s = (int)kq;
        const float d = vccs_distance(c, n, st[s].c, st[s].n, w_s_over_seed, w_n);
        if (d < cd) { cd = d; cur = s; }
      }
    }
    if (CLAIM) {
      owner1[v] = cur;
      dist1[v] = have ? cd : dist0[v];
      if (cur != own) {   // (an offer was evaluated: c and n are loaded)
        touched = true;
        long long f[6];
        for (int a = 0; a < 3; ++a) { f[a] = vccs_fix_pos(c[a]); f[3 + a] = vccs_fix_nrm(n[a]); }
        int r[6];
        bool small = true;
        for (int a = 0; a < 3; ++a) {
          const long long d = f[a] - base[a];
          small = small && d > -(1ll << 21) && d < (1ll << 21);
          r[a] = (int)d; r[3 + a] = (int)f[3 + a];
        }
        for (int side = 0; side < 2; ++side) {
          const int l = side ? cur : own;
          if (l < 0) continue;
          const int sgn = side ? 1 : -1;
          int slot = -1;
          if (small) {
            unsigned int h = ((unsigned int)l * 2654435761u) >> (VT_SLOTS == 128 ? 25 : (VT_SLOTS == 64 ? 26 : (VT_SLOTS == 32 ? 27 : 28)));
            for (int probe = 0; probe < VT_SLOTS; ++probe) {
              const int prev = atomicCAS(&s_key[h], -1, l);
              if (prev == -1 || prev == l) { slot = (int)h; break; }
              h = (h + 1) & (VT_SLOTS - 1);
            }
          }
          if (slot >= 0) {
            for (int a = 0; a < 6; ++a) atomicAdd(&s_sum[slot][a], sgn * r[a]);
            atomicAdd(&s_cnt[slot], sgn);
          } else {
            for (int a = 0; a < 6; ++a) atomicAdd((unsigned long long*)&sums[6 * l + a], (unsigned long long)((long long)sgn * f[a]));
            if (side) atomicAdd(&count[l], 1u); else atomicSub(&count[l], 1u);
          }
        }
      }
    } else {
      const int pn = (own << 1) | (cur == own ? 1 : 0);
      P_out[v] = pn;
      any_change = any_change || pn != p;
    }
  }
  if (!CLAIM) {
    const bool chg = __ballot(any_change) != 0ull;
    if (lane == 0) { if (chg) atomicOr(changed, 1u); if (tchg_out) tchg_out[t] = chg ? 1u : 0u; }
  }
  if (CLAIM && dbg_moved) { const int nm = __popcll(__ballot(touched)); if (lane == 0 && nm) atomicAdd(dbg_moved, (unsigned int)nm); }   // (lanes, not voxels: a lower bound)
  if (CLAIM && __ballot(touched) != 0ull) {
    vt_sync();
    for (int k = lane; k < VT_SLOTS; k += 64) {
      const int l = s_key[k];
      if (l >= 0) {
        const int dc = s_cnt[k];
        for (int a = 0; a < 6; ++a) {
          const long long x = (long long)s_sum[k][a] + (a < 3 ? (long long)dc * base[a] : 0ll);
          if (x) atomicAdd((unsigned long long*)&sums[6 * l + a], (unsigned long long)x);
        }
        if (dc > 0) atomicAdd(&count[l], (unsigned int)dc); else if (dc < 0) atomicSub(&count[l], (unsigned int)(-dc));
      }
    }
  }
}
// the sums at the start of a pass: every living supervoxel owns exactly its seed voxel
__global__ void k_pclt_seed_sums(int K, const int32_t* __restrict__ seed_of, const unsigned long long* __restrict__ seed_key, const uint8_t* __restrict__ alive,
                                 const float* __restrict__ cen, const float* __restrict__ nrm, long long* __restrict__ sums, unsigned int* __restrict__ count,
                                 const int32_t* __restrict__ owner) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  for (int a = 0; a < 6; ++a) sums[6 * k + a] = 0;
  count[k] = 0;
  int64_t v = -1;
  if (seed_of) v = (int64_t)seed_of[k];                                                     // the first pass: the seeds themselves
  else if (alive[k] && seed_key[k] != ~0ull) v = (int64_t)(uint32_t)seed_key[k];            // re-seeding (k_pcl_plant_again)
  if (v < 0 || (owner && owner[v] != k)) return;                                            // (... lost to a later supervoxel that named the same voxel)
  for (int a = 0; a < 3; ++a) { sums[6 * k + a] = vccs_fix_pos(cen[3 * v + a]); sums[6 * k + 3 + a] = vccs_fix_nrm(nrm[3 * v + a]); }
  count[k] = 1;
}

static vgs_status vgs_stage_vccs_pcl(vgs_ctx* c) {
  const int64_t V = c->V, N = c->N;
  const int TB = 256;
  const unsigned nbV = (unsigned)((V + TB - 1) / TB);
  DevBuf<float>& cen = c->vc_cen; DevBuf<float>& nrm = c->vc_nrm; DevBuf<float>& dist = c->vc_dist;
  VGS_HIP_TRY(c, cen.ensure(3 * V)); VGS_HIP_TRY(c, nrm.ensure(3 * V)); VGS_HIP_TRY(c, dist.ensure(2 * V));
  const bool tiles = !c->K.no_vccs_tiles;
  const bool tile_normals = tiles && !c->K.vccs_nbr_normals;   // (round 6) the two-ring normals over the tiles: no [26][V] neighbour table
  if (!tile_normals) VGS_HIP_TRY(c, c->vc_nbr.ensure(26 * V));
  VGS_HIP_TRY(c, c->vc_label.ensure(2 * V));
  static_assert(sizeof(VccsAccu) == 40, "VccsAccu");
  VGS_HIP_TRY(c, c->vc_accu.ensure(10 * (size_t)V)); VGS_HIP_TRY(c, c->vc_live.ensure(2 * (size_t)V));
  VGS_HIP_TRY(c, c->counters.ensure(64));
  hipLaunchKernelGGL(k_vccs_centroid, dim3(nbV), dim3(TB), 0, c->stream, c->xs.p, c->ys.p, c->zs.p, c->vox_start.p, V, cen.p);
  { vgs_status bs = vgs_build_bricks(c, nullptr); if (bs != VGS_OK) return bs; }
  // the 26-neighbour table (no 1-ring normals: the 2-ring ones follow)
  if (!tile_normals)
    hipLaunchKernelGGL(k_vccs_neighbours, dim3(nbV), dim3(TB), 0, c->stream, c->vox_code.p, V, c->box.depth, (const Brick*)c->hkey.p, c->hbits, cen.p,
                       c->vc_nbr.p, (float*)nullptr, (int4*)nullptr);
  // tiles for the live sweeps and the claim (k_pclt_sweep)
  int NT = 0;
  if (tiles) {
    VGS_HIP_TRY(c, c->head_flag.ensure(V + 1)); VGS_HIP_TRY(c, c->perm_a.ensure(V + 1));
    size_t scan_bytes_t = 0;
    VGS_HIP_TRY(c, rocprim::inclusive_scan(nullptr, scan_bytes_t, c->head_flag.p, c->perm_a.p, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
    VGS_HIP_TRY(c, c->sort_tmp.ensure(scan_bytes_t));
    VGS_HIP_TRY(c, c->vc_tile_start.ensure(V + 1));
    hipLaunchKernelGGL(k_vccs_tile_heads, dim3(nbV), dim3(TB), 0, c->stream, c->vox_code.p, V, c->head_flag.p);
    VGS_HIP_TRY(c, rocprim::inclusive_scan(c->sort_tmp.p, scan_bytes_t, c->head_flag.p, c->perm_a.p, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
    VGS_HIP_TRY(c, c->vc_tile_of.ensure(V));
    hipLaunchKernelGGL(k_vccs_tile_starts, dim3(nbV), dim3(TB), 0, c->stream, c->head_flag.p, c->perm_a.p, V, c->vc_tile_start.p, c->vc_tile_of.p);
    uint32_t T32 = 0;
    VGS_READBACK(c, &T32, c->perm_a.p + (V - 1), 4);
    NT = (int)T32;
    const unsigned long long pool_cap = 7ull * (unsigned long long)V;
    VGS_HIP_TRY(c, c->vc_cell.ensure(V)); VGS_HIP_TRY(c, c->vc_halo.ensure(pool_cap));
    VGS_HIP_TRY(c, c->vc_tile_meta.ensure(NT)); VGS_HIP_TRY(c, c->vc_pool.ensure(1)); VGS_HIP_TRY(c, c->vc_plive.ensure(2 * (size_t)V));
    VGS_HIP_TRY(c, c->vc_nbr_tiles.ensure(27 * (size_t)NT)); VGS_HIP_TRY(c, c->vc_tchg.ensure(2 * (size_t)NT));
    VGS_HIP_TRY(c, hipMemsetAsync(c->vc_pool.p, 0, 8, c->stream));
    hipLaunchKernelGGL(k_vccs_tile_setup, dim3((unsigned)NT), dim3(64), 0, c->stream, c->vox_code.p, c->box.depth, (const Brick*)c->hkey.p, c->hbits, cen.p,
                       c->vc_tile_start.p, c->vc_cell.p, (float*)nullptr, (uint2*)c->vc_halo.p, pool_cap, (unsigned long long*)c->vc_pool.p,
                       (uint2*)c->vc_tile_meta.p, (const uint32_t*)c->vc_tile_of.p, c->vc_nbr_tiles.p);
  }
  auto two_ring_normals = [&](const int32_t* owner) {   // computeVoxelData (owner == null) / SupervoxelHelper::refineNormals
    if (tile_normals) {
      hipLaunchKernelGGL(k_pclt_normals<false>, dim3((unsigned)NT), dim3(VTN_TB), 0, c->stream, c->vc_tile_start.p, (const uint2*)c->vc_tile_meta.p, (const uint2*)c->vc_halo.p,
                         c->vc_cell.p, cen.p, (VccsAccu*)c->vc_accu.p, (float*)nullptr, owner);
      hipLaunchKernelGGL(k_pclt_normals<true>, dim3((unsigned)NT), dim3(VTN_TB), 0, c->stream, c->vc_tile_start.p, (const uint2*)c->vc_tile_meta.p, (const uint2*)c->vc_halo.p,
                         c->vc_cell.p, cen.p, (VccsAccu*)c->vc_accu.p, nrm.p, owner);
    } else {
      hipLaunchKernelGGL(k_pcl_accu1, dim3(nbV), dim3(TB), 0, c->stream, V, c->vc_nbr.p, cen.p, (VccsAccu*)c->vc_accu.p, owner);
      hipLaunchKernelGGL(k_pcl_normals, dim3(nbV), dim3(TB), 0, c->stream, V, c->vc_nbr.p, cen.p, (const VccsAccu*)c->vc_accu.p, nrm.p, owner);
    }
  };
  two_ring_normals(nullptr);
  // ---- seeds ----
  const float seed = c->P.seed_size, res = c->P.voxel_size;
  const float mnx = (float)c->box.min[0], mny = (float)c->box.min[1], mnz = (float)c->box.min[2];
  VGS_HIP_TRY(c, c->cell_code_a.ensure(V)); VGS_HIP_TRY(c, c->cell_code_b.ensure(V));
  VGS_HIP_TRY(c, c->cell_id_a.ensure(V)); VGS_HIP_TRY(c, c->cell_id_b.ensure(V));
  VGS_HIP_TRY(c, c->head_flag.ensure(V + 1)); VGS_HIP_TRY(c, c->perm_a.ensure(V + 1));
  hipLaunchKernelGGL(k_pcl_cell_codes, dim3(nbV), dim3(TB), 0, c->stream, cen.p, V, mnx, mny, mnz, seed, c->cell_code_a.p, c->cell_id_a.p);
  size_t sort_bytes = 0, scan_bytes = 0, scan2_bytes = 0;
  VGS_HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, sort_bytes, c->cell_code_a.p, c->cell_code_b.p, c->cell_id_a.p, c->cell_id_b.p, (size_t)V, 0, 63, c->stream));
  VGS_HIP_TRY(c, rocprim::inclusive_scan(nullptr, scan_bytes, c->head_flag.p, c->perm_a.p, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
  VGS_HIP_TRY(c, rocprim::exclusive_scan(nullptr, scan2_bytes, c->head_flag.p, c->perm_a.p, 0u, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
  VGS_HIP_TRY(c, c->sort_tmp.ensure(std::max(sort_bytes, std::max(scan_bytes, scan2_bytes))));
  VGS_HIP_TRY(c, rocprim::radix_sort_pairs(c->sort_tmp.p, sort_bytes, c->cell_code_a.p, c->cell_code_b.p, c->cell_id_a.p, c->cell_id_b.p, (size_t)V, 0, 63, c->stream));
  hipLaunchKernelGGL(k_vccs_heads, dim3(nbV), dim3(TB), 0, c->stream, c->cell_code_b.p, V, c->head_flag.p);
  uint32_t* scan = c->perm_a.p;
  VGS_HIP_TRY(c, rocprim::inclusive_scan(c->sort_tmp.p, scan_bytes, c->head_flag.p, scan, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
  uint32_t K0u = 0;
  VGS_HIP_TRY(c, hipMemcpyAsync(&K0u, scan + (V - 1), 4, hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  const int K0 = (int)K0u;
  const unsigned nbK0 = (unsigned)((K0 + TB - 1) / TB);
  VGS_HIP_TRY(c, c->vc_seedkey.ensure(K0));
  unsigned long long* seed_key = (unsigned long long*)c->vc_seedkey.p;
  hipLaunchKernelGGL(k_vccs_fill_u64, dim3(nbK0), dim3(TB), 0, c->stream, seed_key, (int64_t)K0, ~0ull);
  hipLaunchKernelGGL(k_pcl_pick_seeds, dim3(nbV), dim3(TB), 0, c->stream, c->cell_code_b.p, c->cell_id_b.p, scan, V, cen.p, mnx, mny, mnz, seed, seed_key);
  // rejection + compaction (the scan buffers are free again: keep flags in head_flag, their exclusive scan in perm_a, seeds in cell_id_a)
  const float rad = 0.5f * seed;
  hipLaunchKernelGGL(k_pcl_seed_filter, dim3((unsigned)(((int64_t)K0 * 16 + TB - 1) / TB)), dim3(TB), 0, c->stream, seed_key, K0, c->vox_code.p, c->box.depth, (const Brick*)c->hkey.p, c->hbits, cen.p,
                     rad * rad, (int)(rad / res) + 1, vccs_seed_min_points(seed, res), c->head_flag.p);
  VGS_HIP_TRY(c, rocprim::exclusive_scan(c->sort_tmp.p, scan2_bytes, c->head_flag.p, c->perm_a.p, 0u, (size_t)K0, rocprim::plus<uint32_t>(), c->stream));
  uint32_t tail[2] = {0, 0};
  VGS_HIP_TRY(c, hipMemcpyAsync(&tail[0], c->perm_a.p + (K0 - 1), 4, hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipMemcpyAsync(&tail[1], c->head_flag.p + (K0 - 1), 4, hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  const int K = (int)(tail[0] + tail[1]);
  c->sv_max_label = 0;
  c->counts[VGS_N_SUPERVOXELS] = K;
  if (K == 0) {
    VGS_HIP_TRY(c, hipMemsetAsync(c->sv_label.p, 0, (size_t)N * 4, c->stream));
    VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->sv_have_labels = true;
    return VGS_OK;
  }
  uint32_t* seeds = c->cell_id_a.p;
  hipLaunchKernelGGL(k_pcl_compact_seeds, dim3(nbK0), dim3(TB), 0, c->stream, seed_key, c->head_flag.p, c->perm_a.p, K0, seeds);
  const unsigned nbK = (unsigned)((K + TB - 1) / TB);
  VGS_HIP_TRY(c, c->vc_sums.ensure(6 * (size_t)K)); VGS_HIP_TRY(c, c->vc_count.ensure(K)); VGS_HIP_TRY(c, c->vc_state.ensure(6 * (size_t)K));
  VGS_HIP_TRY(c, c->vc_alive.ensure(K)); VGS_HIP_TRY(c, c->vc_seedkey.ensure(std::max(K, K0)));
  seed_key = (unsigned long long*)c->vc_seedkey.p;
  VccsState* state = (VccsState*)c->vc_state.p;
  // ---- extract + refineSupervoxels(5) ----
  const int depth = (int)(1.8f * seed / res);
  const float w_s_over_seed = c->P.spatial_impt / seed, w_n = c->P.normal_impt;
  int32_t* own[2] = {c->vc_label.p, c->vc_label.p + V};
  float* dst[2] = {dist.p, dist.p + V};
  uint8_t* live[2] = {c->vc_live.p, c->vc_live.p + V};
  int32_t* plive[2] = {tiles ? c->vc_plive.p : nullptr, tiles ? c->vc_plive.p + V : nullptr};
  uint32_t* tchg[2] = {tiles ? c->vc_tchg.p : nullptr, tiles ? c->vc_tchg.p + NT : nullptr};   // per tile: a flag changed in that sweep
  // the sweeps' change counters: a ring of words zeroed once per pass (was a memset in front of every batch of sweeps: 80 per step)
  constexpr int RING = 4096;
  VGS_HIP_TRY(c, c->vc_ring.ensure(RING));
  unsigned int* const d_ring = c->vc_ring.p;
  int ring_at = RING;   // (forces the first fill)
  unsigned int* d_changed = (unsigned int*)(c->counters.p + 56);   // (the final read-back of the largest label in use)
  int cur = 0;
  unsigned int* dbg_moved = nullptr;   // VGS_DEBUG: lanes that moved a voxel, per pass and round
  if (c->K.debug) { VGS_HIP_TRY(c, c->vc_dbg.ensure(128)); VGS_HIP_TRY(c, hipMemsetAsync(c->vc_dbg.p, 0, 512, c->stream)); dbg_moved = c->vc_dbg.p; }
  hipLaunchKernelGGL(k_pcl_reset, dim3(nbV), dim3(TB), 0, c->stream, V, own[cur], dst[cur]);
  hipLaunchKernelGGL(k_pcl_plant_first, dim3(nbK), dim3(TB), 0, c->stream, seeds, K, cen.p, nrm.p, own[cur], state, c->vc_alive.p);
  if (tiles) hipLaunchKernelGGL(k_pclt_seed_sums, dim3(nbK), dim3(TB), 0, c->stream, K, (const int32_t*)seeds, (const unsigned long long*)nullptr, (const uint8_t*)nullptr,
                                cen.p, nrm.p, c->vc_sums.p, c->vc_count.p, (const int32_t*)nullptr);
  int dbg_settled[16] = {0};   // VGS_DEBUG: rounds by the index of their first sweep that changed nothing
  const bool inplace = tiles && !c->K.vccs_pingpong;   // (VGS_VCCS_PINGPONG: the sweeps alternate between two flag arrays, as until round 5)
  for (int pass = 0; pass < 6; ++pass) {
    if (pass > 0) {
      // refineSupervoxels: refineNormals of every supervoxel (from its own leaves), reseedSupervoxels (nearest of all voxels), expansion
      two_ring_normals((const int32_t*)own[cur]);
      hipLaunchKernelGGL(k_pcl_reseed_nearest, dim3((unsigned)(((int64_t)K * 16 + TB - 1) / TB)), dim3(TB), 0, c->stream, K, (const uint8_t*)c->vc_alive.p, (const VccsState*)state, cen.p,
                         (const Brick*)c->hkey.p, c->hbits, c->box.depth, c->box.min[0], c->box.min[1], c->box.min[2], c->box.res, res, seed_key);
      hipLaunchKernelGGL(k_pcl_reset, dim3(nbV), dim3(TB), 0, c->stream, V, own[cur], dst[cur]);
      hipLaunchKernelGGL(k_pcl_plant_again, dim3(nbK), dim3(TB), 0, c->stream, seed_key, K, c->vc_alive.p, own[cur]);
      if (tiles) hipLaunchKernelGGL(k_pclt_seed_sums, dim3(nbK), dim3(TB), 0, c->stream, K, (const int32_t*)nullptr, (const unsigned long long*)seed_key,
                                    (const uint8_t*)c->vc_alive.p, cen.p, nrm.p, c->vc_sums.p, c->vc_count.p, (const int32_t*)own[cur]);
    }
    for (int it = 1; it < depth; ++it) {
      int lc = 0;
      if (tiles) hipLaunchKernelGGL(k_pclt_init, dim3(nbV), dim3(TB), 0, c->stream, V, own[cur], plive[lc], tchg[lc], NT);   // (NT <= V) the first sweep of a round works every tile
      else hipLaunchKernelGGL(k_pcl_live_init, dim3(nbV), dim3(TB), 0, c->stream, V, own[cur], live[lc]);
      // fixed point of the live flags: the recursion is on smaller labels, so it ends -- a sweep that changes nothing is the proof.
      // Sweeps go out in pairs with one read-back per pair (a sweep at the fixed point changes nothing, so a spare one is harmless):
      // half the host round trips.  Leaving the loop without that proof would give labels that differ from the sequential
      // order silently: it is an error.
      // Over tiles a sweep at the fixed point costs a few microseconds (every tile is quiet), a host round trip costs thirty: the first
      // batch of a round is four sweeps (a round needs four or five; six were measured slower than pairs), later ones are pairs.
      bool settled = false;
      for (int sweep = 0, nb = tiles ? 4 : 2; sweep < 4096 && !settled; sweep += nb, nb = 2) {
        unsigned int ch[6] = {1u, 1u, 1u, 1u, 1u, 1u};
        if (ring_at + nb > RING) { VGS_HIP_TRY(c, hipMemsetAsync(d_ring, 0, (size_t)RING * 4, c->stream)); ring_at = 0; }
        unsigned int* const d_chg = d_ring + ring_at;
        ring_at += nb;
        for (int k = 0; k < nb; ++k) {
          if (tiles)
            // In place (round 6): a tile reads its window from the one flag array and writes its own voxels back into it.  Another tile may
            // read them before or after -- an asynchronous iteration, and the fixed point of the live flags is unique (the recursion is on
            // smaller labels), so it is reached whatever the order; the proof of arrival is unchanged (a sweep in which nobody wrote read only
            // final values), and so is the quiet-tile test (a tile whose window tiles wrote nothing in the last sweep read what is there now).
            // Quiet tiles then copy nothing, and news travels within a sweep.  The per-tile change flags still alternate.
            hipLaunchKernelGGL(k_pclt_sweep<false>, dim3((unsigned)NT), dim3(64), 0, c->stream, c->vc_tile_start.p, (const uint2*)c->vc_tile_meta.p,
                               (const uint2*)c->vc_halo.p, c->vc_cell.p, plive[inplace ? 0 : lc], dst[cur], cen.p, nrm.p, state, w_s_over_seed, w_n, plive[inplace ? 0 : (lc ^ 1)],
                               d_chg + k, (int32_t*)nullptr, (float*)nullptr, (long long*)nullptr, (unsigned int*)nullptr,
                               (const int32_t*)c->vc_nbr_tiles.p, (const uint32_t*)tchg[lc], tchg[lc ^ 1]);
          else
            hipLaunchKernelGGL(k_pcl_live, dim3(nbV), dim3(TB), 0, c->stream, V, c->vc_nbr.p, own[cur], dst[cur], live[lc], cen.p, nrm.p, state, w_s_over_seed, w_n,
                               live[lc ^ 1], d_chg + k);
          lc ^= 1;
        }
        VGS_READBACK(c, ch, d_chg, 4 * (size_t)nb);
        for (int k = 0; k < nb; ++k) { if (!settled && !ch[k] && c->K.debug) ++dbg_settled[std::min(sweep + k, 15)]; settled = settled || !ch[k]; }   // (after an unchanged sweep the two flag arrays are equal: either is the fixed point)
      }
      if (!settled) { c->err = "svgs_supervoxels (vccs_mode 1): the live flags did not reach their fixed point in 4096 sweeps"; return VGS_E_STATE; }
      if (tiles)
        hipLaunchKernelGGL(k_pclt_sweep<true>, dim3((unsigned)NT), dim3(64), 0, c->stream, c->vc_tile_start.p, (const uint2*)c->vc_tile_meta.p,
                           (const uint2*)c->vc_halo.p, c->vc_cell.p, plive[inplace ? 0 : lc], dst[cur], cen.p, nrm.p, state, w_s_over_seed, w_n, (int32_t*)nullptr,
                           (unsigned int*)nullptr, own[cur ^ 1], dst[cur ^ 1], c->vc_sums.p, c->vc_count.p, (const int32_t*)nullptr,
                           (const uint32_t*)nullptr, (uint32_t*)nullptr, dbg_moved ? dbg_moved + pass * 16 + it : (unsigned int*)nullptr);
      else
        hipLaunchKernelGGL(k_pcl_claim, dim3(nbV), dim3(TB), 0, c->stream, V, c->vc_nbr.p, own[cur], dst[cur], live[lc], cen.p, nrm.p, state, w_s_over_seed, w_n,
                           own[cur ^ 1], dst[cur ^ 1]);
      cur ^= 1;
      if (!tiles) {   // (over tiles the claim moved the sums with the voxels that changed owner: integers, the same totals)
        VGS_HIP_TRY(c, hipMemsetAsync(c->vc_sums.p, 0, 6 * (size_t)K * sizeof(long long), c->stream));
        VGS_HIP_TRY(c, hipMemsetAsync(c->vc_count.p, 0, (size_t)K * 4, c->stream));
        hipLaunchKernelGGL(k_pcl_accumulate, dim3(nbV), dim3(TB), 0, c->stream, V, own[cur], cen.p, nrm.p, c->vc_sums.p, c->vc_count.p);
      }
      hipLaunchKernelGGL(k_pcl_update, dim3(nbK), dim3(TB), 0, c->stream, K, c->vc_sums.p, c->vc_count.p, state, c->vc_alive.p);
    }
  }
  hipLaunchKernelGGL(k_vccs_point_labels, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, c->stream, c->perm_b.p, c->pt_vox.p, own[cur], N, c->sv_label.p);
  VGS_HIP_TRY(c, hipMemsetAsync(d_changed, 0, 4, c->stream));
  hipLaunchKernelGGL(k_pcl_max_label, dim3(nbK), dim3(TB), 0, c->stream, K, c->vc_alive.p, d_changed);
  unsigned int mx = 0;
  VGS_READBACK(c, &mx, d_changed, 4);
  VGS_HIP_TRY(c, hipGetLastError());
  if (c->K.debug) {
    fprintf(stderr, "[vgs] vccs_mode 1: rounds by the index of their first unchanged sweep:");
    for (int k = 0; k < 16; ++k) if (dbg_settled[k]) fprintf(stderr, " %d:%d", k, dbg_settled[k]);
    fprintf(stderr, "\n");
  }
  if (dbg_moved) {
    unsigned int h[128];
    VGS_HIP_TRY(c, hipMemcpy(h, dbg_moved, sizeof(h), hipMemcpyDeviceToHost));
    for (int pass = 0; pass < 6; ++pass) {
      fprintf(stderr, "[vgs] vccs_mode 1 pass %d: lanes that moved a voxel per round:", pass);
      for (int it = 1; it < depth && it < 16; ++it) fprintf(stderr, " %u", h[pass * 16 + it]);
      fprintf(stderr, "\n");
    }
  }
  c->sv_max_label = (int32_t)mx;   // getMaxLabel(): the largest label still in use
  c->sv_have_labels = true;
  return VGS_OK;
}

// ---------------------------------------------------------------- driver
vgs_status vgs_stage_vccs(vgs_ctx* c) {
  // the class's own octree at voxel_resolution_ (SS:85, test:138-142) provides the VCCS voxels
  // vccs_mode 1 (round 5): pcl::SupervoxelClustering bins into an octree of its OWN (OctreePointCloudAdjacency), whose box is defined from
  // the cloud's bounding box before the first point goes in -- defineBoundingBox + getKeyBitSize on the empty tree pad it symmetrically to
  // the cube of 2^depth voxels -- instead of growing from the first point as the class's octree does.  The box is put together on the host
  // from the device's bounding box, the growth scan runs over it once (a point on the padded box's upper face still grows it), and the
  // voxelize stage takes it as a grid that covers the cloud.  Whatever the context had pinned (nothing, outside tiles) comes back after.
  const bool own_lattice = c->P.vccs_mode == 1;
  struct Restore {   // (at every way out of the stage: its kernels read c->box until the end)
    vgs_ctx* c; bool on, pinned, covers; OctreeBox box;
    ~Restore() { if (on) { c->grid_pinned = pinned; c->grid_covers = covers; if (pinned) c->box = box; } }
  } restore{c, own_lattice, c->grid_pinned, c->grid_covers, c->box};
  if (own_lattice) {
    float bb[6]; int64_t nf = 0;
    vgs_status sb = vgs_points_bbox(c, bb, &nf);
    if (sb != VGS_OK) return sb;
    if (nf > 0) {
      const double eps = 1.1920928955078125e-07, res = (double)c->P.voxel_size;
      double mn[3], mx[3];
      for (int a = 0; a < 3; ++a) { mn[a] = (double)bb[a]; mx[a] = (double)bb[3 + a]; }
      unsigned mk[3];
      for (int a = 0; a < 3; ++a) mk[a] = (unsigned)((mx[a] - mn[a]) / res);
      const unsigned max_voxels = std::max(std::max(std::max(mk[0], mk[1]), mk[2]), 2u);
      vgs_grid_state g;
      vgs_grid_state_init(&g);
      g.depth = (int32_t)std::max(std::min(32u, (unsigned)std::ceil(std::log2((double)max_voxels) - eps)), 0u);
      const double side = (double)(1u << g.depth) * res - eps;
      for (int a = 0; a < 3; ++a) g.min[a] = mn[a] - (side - (mx[a] - mn[a])) / 2.0;
      g.defined = 1;
      if ((sb = vgs_grid_advance(c, &g)) != VGS_OK) return sb;
      if ((sb = vgs_set_grid_covering(c, &g)) != VGS_OK) return sb;
    }
  }
  vgs_status st0 = vgs_stage_voxelize(c);
  if (st0 != VGS_OK) return st0;
  const int64_t V = c->V, N = c->N;
  VGS_HIP_TRY(c, c->sv_label.ensure(N > 0 ? N : 1));
  c->sv_max_label = 0;
  if (V == 0) { if (N > 0) VGS_HIP_TRY(c, hipMemsetAsync(c->sv_label.p, 0, N * 4, c->stream)); c->sv_have_labels = true; return VGS_OK; }
  if (c->P.vccs_mode == 1) return vgs_stage_vccs_pcl(c);
  const int TB = 256;
  const unsigned nbV = (unsigned)((V + TB - 1) / TB);
  static_assert(sizeof(VccsState) == 24, "VccsState");
  DevBuf<float>& cen = c->vc_cen; DevBuf<float>& nrm = c->vc_nrm; DevBuf<float>& dist = c->vc_dist;
  VGS_HIP_TRY(c, cen.ensure(3 * V)); VGS_HIP_TRY(c, nrm.ensure(3 * V)); VGS_HIP_TRY(c, dist.ensure(2 * V));
  VGS_HIP_TRY(c, c->vc_label.ensure(2 * V));
  hipLaunchKernelGGL(k_vccs_centroid, dim3(nbV), dim3(TB), 0, c->stream, c->xs.p, c->ys.p, c->zs.p, c->vox_start.p, V, cen.p);
  { vgs_status bs = vgs_build_bricks(c, nullptr); if (bs != VGS_OK) return bs; }
  if (c->K.no_vccs_tiles) {
    VGS_HIP_TRY(c, c->vc_nbr.ensure(26 * V)); VGS_HIP_TRY(c, c->vc_nbr4.ensure(28 * (size_t)V));
    hipLaunchKernelGGL(k_vccs_neighbours, dim3(nbV), dim3(TB), 0, c->stream, c->vox_code.p, V, c->box.depth, (const Brick*)c->hkey.p, c->hbits, cen.p,
                       c->vc_nbr.p, nrm.p, (int4*)c->vc_nbr4.p);
  }
  // ---- seeds: one per occupied seed_res cell, snapped to the voxel nearest to the cell centre ----
  const float seed = c->P.seed_size;
  const float mnx = (float)c->box.min[0], mny = (float)c->box.min[1], mnz = (float)c->box.min[2];
  VGS_HIP_TRY(c, c->cell_code_a.ensure(V)); VGS_HIP_TRY(c, c->cell_code_b.ensure(V));
  VGS_HIP_TRY(c, c->cell_id_a.ensure(V)); VGS_HIP_TRY(c, c->cell_id_b.ensure(V));
  VGS_HIP_TRY(c, c->head_flag.ensure(V + 1)); VGS_HIP_TRY(c, c->perm_a.ensure(V + 1));
  // ---- tiles of the expansion rounds (k_vccs_expand_tiles): runs of equal code >> 9, numbered by a scan; their number comes back with
  // the number of seeds below
  const bool tiles = !c->K.no_vccs_tiles;
  uint32_t T32 = 0;
  if (tiles) {
    size_t scan_bytes_t = 0;
    VGS_HIP_TRY(c, rocprim::inclusive_scan(nullptr, scan_bytes_t, c->head_flag.p, c->perm_a.p, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
    VGS_HIP_TRY(c, c->sort_tmp.ensure(scan_bytes_t));
    VGS_HIP_TRY(c, c->vc_tile_start.ensure(V + 1));
    hipLaunchKernelGGL(k_vccs_tile_heads, dim3(nbV), dim3(TB), 0, c->stream, c->vox_code.p, V, c->head_flag.p);
    VGS_HIP_TRY(c, rocprim::inclusive_scan(c->sort_tmp.p, scan_bytes_t, c->head_flag.p, c->perm_a.p, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
    hipLaunchKernelGGL(k_vccs_tile_starts, dim3(nbV), dim3(TB), 0, c->stream, c->head_flag.p, c->perm_a.p, V, c->vc_tile_start.p, (uint32_t*)nullptr);
    VGS_HIP_TRY(c, hipMemcpyAsync(&T32, c->perm_a.p + (V - 1), 4, hipMemcpyDeviceToHost, c->stream));
  }
  hipLaunchKernelGGL(k_vccs_cell_codes, dim3(nbV), dim3(TB), 0, c->stream, cen.p, V, mnx, mny, mnz, seed, c->cell_code_a.p, c->cell_id_a.p);
  size_t sort_bytes = 0, scan_bytes = 0;
  VGS_HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, sort_bytes, c->cell_code_a.p, c->cell_code_b.p, c->cell_id_a.p, c->cell_id_b.p, (size_t)V, 0, 63, c->stream));
  VGS_HIP_TRY(c, rocprim::inclusive_scan(nullptr, scan_bytes, c->head_flag.p, c->perm_a.p, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
  VGS_HIP_TRY(c, c->sort_tmp.ensure(std::max(sort_bytes, scan_bytes)));
  VGS_HIP_TRY(c, rocprim::radix_sort_pairs(c->sort_tmp.p, sort_bytes, c->cell_code_a.p, c->cell_code_b.p, c->cell_id_a.p, c->cell_id_b.p, (size_t)V, 0, 63, c->stream));
  hipLaunchKernelGGL(k_vccs_heads, dim3(nbV), dim3(TB), 0, c->stream, c->cell_code_b.p, V, c->head_flag.p);
  uint32_t* scan = c->perm_a.p;
  VGS_HIP_TRY(c, rocprim::inclusive_scan(c->sort_tmp.p, scan_bytes, c->head_flag.p, scan, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
  uint32_t K32 = 0;
  VGS_HIP_TRY(c, hipMemcpyAsync(&K32, scan + (V - 1), 4, hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  const int K = (int)K32;
  const unsigned nbK = (unsigned)((K + TB - 1) / TB);
  const int NT = (int)T32;
  if (tiles) {
    const unsigned long long pool_cap = 7ull * (unsigned long long)V;   // a voxel lies in the shell of at most seven tiles
    VGS_HIP_TRY(c, c->vc_cell.ensure(V)); VGS_HIP_TRY(c, c->vc_halo.ensure(pool_cap));
    VGS_HIP_TRY(c, c->vc_tile_meta.ensure(NT)); VGS_HIP_TRY(c, c->vc_pool.ensure(1));
    VGS_HIP_TRY(c, hipMemsetAsync(c->vc_pool.p, 0, 8, c->stream));
    hipLaunchKernelGGL(k_vccs_tile_setup, dim3((unsigned)NT), dim3(64), 0, c->stream, c->vox_code.p, c->box.depth, (const Brick*)c->hkey.p, c->hbits, cen.p,
                       c->vc_tile_start.p, c->vc_cell.p, nrm.p, (uint2*)c->vc_halo.p, pool_cap, (unsigned long long*)c->vc_pool.p, (uint2*)c->vc_tile_meta.p,
                       (const uint32_t*)nullptr, (int32_t*)nullptr);
  }
  VGS_HIP_TRY(c, c->vc_seedkey.ensure(K)); VGS_HIP_TRY(c, c->vc_sums.ensure(6 * (size_t)K)); VGS_HIP_TRY(c, c->vc_count.ensure(K));
  VGS_HIP_TRY(c, c->vc_state.ensure(6 * (size_t)K));
  VccsState* state = (VccsState*)c->vc_state.p;
  unsigned long long* seed_key = (unsigned long long*)c->vc_seedkey.p;
  hipLaunchKernelGGL(k_vccs_fill_u64, dim3(nbK), dim3(TB), 0, c->stream, seed_key, (int64_t)K, ~0ull);
  hipLaunchKernelGGL(k_vccs_pick_seeds, dim3(nbV), dim3(TB), 0, c->stream, c->cell_code_b.p, c->cell_id_b.p, scan, V, cen.p, mnx, mny, mnz, seed,
                     seed_key);
  // ---- extract + refineSupervoxels(5): six passes of T expansion rounds ----
  const int T = (int)(1.8f * c->P.seed_size / c->P.voxel_size);
  const float w_s_over_seed = c->P.spatial_impt / c->P.seed_size;
  const float w_n = c->P.normal_impt;
  int32_t* lab[2] = {c->vc_label.p, c->vc_label.p + V};
  float* dst[2] = {dist.p, dist.p + V};
  int cur = 0;
  for (int pass = 0; pass < 6; ++pass) {
    if (pass > 0) {
      hipLaunchKernelGGL(k_vccs_fill_u64, dim3(nbK), dim3(TB), 0, c->stream, seed_key, (int64_t)K, ~0ull);
      hipLaunchKernelGGL(k_vccs_reseed, dim3(nbV), dim3(TB), 0, c->stream, V, lab[cur], cen.p, state, seed_key);
    }
    hipLaunchKernelGGL(k_vccs_reset, dim3(nbV), dim3(TB), 0, c->stream, V, lab[cur], dst[cur]);
    hipLaunchKernelGGL(k_vccs_plant, dim3(nbK), dim3(TB), 0, c->stream, seed_key, K, cen.p, nrm.p, lab[cur], dst[cur], state, c->vc_sums.p, c->vc_count.p);
    for (int it = 0; it < T; ++it) {
      if (tiles)
        hipLaunchKernelGGL(k_vccs_expand_tiles, dim3((unsigned)NT), dim3(64), 0, c->stream, NT, c->vc_tile_start.p, (const uint2*)c->vc_tile_meta.p,
                           (const uint2*)c->vc_halo.p, c->vc_cell.p, cen.p, nrm.p, lab[cur], dst[cur], state, w_s_over_seed, w_n,
                           lab[cur ^ 1], dst[cur ^ 1], c->vc_sums.p, c->vc_count.p);
      else
        hipLaunchKernelGGL(k_vccs_expand, dim3(nbV), dim3(TB), 0, c->stream, V, (const int4*)c->vc_nbr4.p, cen.p, nrm.p, lab[cur], dst[cur], state, w_s_over_seed, w_n,
                           lab[cur ^ 1], dst[cur ^ 1], c->vc_sums.p, c->vc_count.p);
      cur ^= 1;
      hipLaunchKernelGGL(k_vccs_update, dim3(nbK), dim3(TB), 0, c->stream, K, c->vc_sums.p, c->vc_count.p, state);
    }
  }
  hipLaunchKernelGGL(k_vccs_point_labels, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, c->stream, c->perm_b.p, c->pt_vox.p, lab[cur], N,
                     c->sv_label.p);
  VGS_HIP_TRY(c, hipGetLastError());
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->sv_max_label = K;  // labels 1..K: getMaxLabel() returns K (the supervoxel with that label is then skipped, SS:313)
  c->sv_have_labels = true;
  c->counts[VGS_N_SUPERVOXELS] = K;
  return VGS_OK;
}
