// adjacency.hip -- stage a4: radius graph over voxel centres.
// Replaces buildVoxelCentersKdtree + findAllVoxelAdjacency + getOneVoxelAdjacency
// (voxel_segmentation.h:223-287, 1742-1747); FLANN radius-search semantics per SURVEY.md B.2:
// float d2 = (dx*dx + dy*dy) + dz*dz between centres, keep d2 < float(r*r), order by (d2, id).
//
// Voxel centres lie on a lattice, so the kd-tree is replaced by a hash of the voxel codes and a fixed
// ball of integer offsets: one wavefront per used voxel probes the ball (coalesced offset table,
// L2-resident hash), keeps what the float predicate accepts, sorts the survivors in LDS (bitonic on
// 64-bit keys d2bits<<32 | id) and writes one contiguous row of the adjacency table.
// Lists are produced for used voxels only: nothing downstream reads an unused voxel's list
// (local graphs VS:376-380, crossValidation VS:2117, closestCheck VS:2199 all start from used voxels).
// When unused voxels are inert in the local graphs (their constant dead-edge weight cannot beat a singleton's
// threshold, checked on the host) the hot-path rows keep only the USED neighbours, still in (d2, id) order, plus
// the full neighbour count (closestCheck needs it: VS:2201, VS:2243).  The full lists of the reference's
// getOneVoxelAdjacency are produced on demand by the FULL instantiation (vgs_get_lists).
#include <algorithm>
#include <vector>

#include "vgs_context.hpp"

// ---- brick table -------------------------------------------------------------------------------------------
// Voxels are sorted by (descending) x-major Morton code, so the voxels of one 4x4x4 brick (code >> 6) are contiguous
// in the voxel array.  One 32-byte entry per brick -- key, occupancy mask, used mask, id of the brick's first voxel --
// answers "which voxel sits in lattice cell c, and is it used?" with one hash probe:
//     id(c) = first + popcount(occupancy >> (local + 1))        (ids ascend while the local code descends)
// About V/9 bricks: the table is a few MB and stays in the 4 MB L2 of each XCD, where the per-voxel hash (24 MB at
// 10 M points) was served from the fabric, and the used flag no longer costs a 64-byte node read per neighbour.
struct Brick { unsigned long long key; unsigned long long occ; unsigned long long used; uint32_t first; uint32_t pad; };

// the packed brick coordinates are highly regular: fold the three fields with odd multipliers before the final multiply
__device__ __forceinline__ uint32_t hash_slot(uint64_t key, uint32_t hbits) {
  const uint32_t bx = (uint32_t)key & 0x1fffffu, by = (uint32_t)(key >> 21) & 0x1fffffu, bz = (uint32_t)(key >> 42);
  uint32_t h = bx * 0x9E3779B1u + by * 0x85EBCA77u + bz * 0xC2B2AE3Du;
  h ^= h >> 15;
  return (h * 0x2C1B3C6Du) >> (32 - hbits);
}

// A brick is named by its lattice coordinates (voxel key >> 2 per axis) packed 21 bits each, + 1 so that 0 means empty;
// a voxel's bit inside the brick is the low 6 bits of its Morton code (z0 y0 x0 z1 y1 x1 from bit 0).  Both are cheap
// to form from (nx, ny, nz): the neighbour search never spreads a full Morton code.
__device__ __forceinline__ unsigned long long brick_key(uint32_t nx, uint32_t ny, uint32_t nz) {
  return (((unsigned long long)(nz >> 2) << 42) | ((unsigned long long)(ny >> 2) << 21) | (unsigned long long)(nx >> 2)) + 1ull;
}
__device__ __forceinline__ int brick_local(uint32_t nx, uint32_t ny, uint32_t nz) {
  return (int)((nz & 1u) | ((ny & 1u) << 1) | ((nx & 1u) << 2) | ((nz & 2u) << 2) | ((ny & 2u) << 3) | ((nx & 2u) << 4));
}

// every voxel finds (or creates) its brick's slot and ORs its bits in; the brick's first voxel is the smallest id
__global__ void k_brick_insert(const uint64_t* __restrict__ vox_code, const NodeRec* __restrict__ node, int64_t V,
                               Brick* __restrict__ table, uint32_t hbits) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const uint64_t code = vox_code[v];
  const unsigned long long key = brick_key(vm_compact21(code >> 2), vm_compact21(code >> 1), vm_compact21(code));  // 0 = empty
  const unsigned long long bit = 1ull << (code & 63ull);  // == brick_local of the same coordinates
  const uint32_t mask = (1u << hbits) - 1u;
  uint32_t s = hash_slot(key, hbits);
  while (true) {
    const unsigned long long prev = atomicCAS(&table[s].key, 0ull, key);
    if (prev == 0ull || prev == key) break;
    s = (s + 1) & mask;
  }
  atomicOr(&table[s].occ, bit);
  if (node[v].flags & VGS_F_EIG) atomicOr(&table[s].used, bit);
  atomicMin(&table[s].first, (uint32_t)v);
}

// voxel id in lattice cell (nx, ny, nz), -1 if empty; *is_used tells whether that voxel has > points_min points
__device__ __forceinline__ int brick_find(const Brick* __restrict__ table, uint32_t hbits, uint32_t nx, uint32_t ny, uint32_t nz, bool* is_used) {
  const unsigned long long key = brick_key(nx, ny, nz);
  const uint32_t mask = (1u << hbits) - 1u;
  uint32_t s = hash_slot(key, hbits);
  while (true) {
    const unsigned long long k = table[s].key;
    if (k == key) break;
    if (k == 0ull) return -1;
    s = (s + 1) & mask;
  }
  const unsigned long long occ = table[s].occ;
  const int local = brick_local(nx, ny, nz);
  if (!((occ >> local) & 1ull)) return -1;
  *is_used = ((table[s].used >> local) & 1ull) != 0;
  const unsigned long long above = (local == 63) ? 0ull : (occ >> (local + 1));
  return (int)(table[s].first + (uint32_t)__popcll(above));
}

// one wavefront (64-thread workgroup) per used voxel
template <int CAP, bool FULL>
__global__ __launch_bounds__(64) void k_adjacency(const uint64_t* __restrict__ vox_code, const uint32_t* __restrict__ used_ids,
                                                  int64_t U, const Brick* __restrict__ bricks,
                                                  uint32_t hbits, const int32_t* __restrict__ offsets, int n_off, int R, int depth,
                                                  float res_f, float min_x, float min_y, float min_z, float r2,
                                                  const NodeRec* __restrict__ node, int adj_stride,
                                                  uint64_t* __restrict__ adj_key, uint32_t* __restrict__ adj_cnt,
                                                  uint32_t* __restrict__ adj_mused) {
  __shared__ uint64_t lst[CAP];
  __shared__ float ctab[3][32];  // voxel centres along each axis for key offsets -R..R (double arithmetic once per wavefront, not per offset)
  const int lane = threadIdx.x;
  const int64_t u = vgs_xcd_item(blockIdx.x, U);
  if (u >= U) return;
  const uint32_t i = used_ids[u];
  const uint64_t code = vox_code[i];
  const uint32_t kx = vm_compact21(code >> 2), ky = vm_compact21(code >> 1), kz = vm_compact21(code);
  const float cx = vm_voxel_center(kx, res_f, min_x), cy = vm_voxel_center(ky, res_f, min_y), cz = vm_voxel_center(kz, res_f, min_z);
  const uint32_t lim = 1u << depth;
  if (lane < 2 * R + 1) {  // keys that wrap below 0 give a meaningless centre, but such cells fail the range test
    ctab[0][lane] = vm_voxel_center(kx + (uint32_t)(lane - R), res_f, min_x);
    ctab[1][lane] = vm_voxel_center(ky + (uint32_t)(lane - R), res_f, min_y);
    ctab[2][lane] = vm_voxel_center(kz + (uint32_t)(lane - R), res_f, min_z);
  }
  __syncthreads();
  int cnt = 0, mused = 0;
  for (int base = 0; base < n_off; base += 64) {
    const int o = base + lane;
    bool keep = false;
    uint64_t key64 = 0;
    bool is_used = false;
    if (o < n_off) {
      const int32_t pk = offsets[o];
      const int dx = (int)(int8_t)(pk & 0xff), dy = (int)(int8_t)((pk >> 8) & 0xff), dz = (int)(int8_t)((pk >> 16) & 0xff);
      const uint32_t nx = kx + (uint32_t)dx, ny = ky + (uint32_t)dy, nz = kz + (uint32_t)dz;  // wraps past 0 fail the range test
      if (nx < lim && ny < lim && nz < lim) {
        const int t = brick_find(bricks, hbits, nx, ny, nz, &is_used);
        if (t >= 0) {
          const float tx = cx - ctab[0][dx + R];
          const float ty = cy - ctab[1][dy + R];
          const float tz = cz - ctab[2][dz + R];
          const float d2 = (tx * tx + ty * ty) + tz * tz;
          if (d2 < r2) {
            keep = true;
            key64 = ((uint64_t)vm_bits(d2) << 32) | (uint32_t)t;
          }
        }
      }
    }
    const unsigned long long mall = __ballot(keep);
    const bool store = FULL ? keep : (keep && is_used);
    const unsigned long long m = __ballot(store);
    if (store) lst[cnt + __popcll(m & ((1ull << lane) - 1ull))] = key64;
    cnt += __popcll(m);
    mused += __popcll(mall);
  }
  // bitonic sort ascending on the next power of two >= cnt
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
        const uint64_t a = lst[lo], b = lst[hi];
        if ((a > b) == up) { lst[lo] = b; lst[hi] = a; }
      }
      __syncthreads();
    }
  }
  uint64_t* row = adj_key + (int64_t)u * adj_stride;
  for (int k = lane; k < cnt; k += 64) row[k] = lst[k];
  if (lane == 0) { adj_cnt[u] = (uint32_t)cnt; adj_mused[u] = (uint32_t)mused; }  // stored entries, all neighbours
}

// an edge that touches an unused voxel carries the constant weight of five distances of 100 (VS:1602-1606);
// if that cannot beat a singleton's threshold 1 - cut the unused voxels can never merge: they are inert (exact)
bool vgs_unused_are_inert(const vgs_params& p) {
  VgsWeightParams W;
  W.inv_sig_p = 1.0f / p.sig_p; W.inv_sig_n = 1.0f / p.sig_n; W.inv_sig_o = 1.0f / p.sig_o;
  W.inv_sig_e = 1.0f / p.sig_e; W.inv_sig_c = 1.0f / p.sig_c;
  W.inv_sig_w2 = 1.0f / (p.sig_w * p.sig_w);
  W.svgs = (p.method == 3) ? 1 : 0;
  const float d100[5] = {100.f, 100.f, 100.f, 100.f, 100.f};
  const float w_dead = vm_distance_weight(d100, W);
  return !(w_dead > vm_cut_threshold(1.0f, p.cut_thred, 1));
}

static vgs_status build_hash_and_offsets(vgs_ctx* c, float* r2_out) {
  const int64_t V = c->V;
  // brick table: the number of bricks is not known without a pass, V/4 slots would already be generous; size by V/2
  uint32_t hbits = 4;
  while ((1ull << hbits) < (uint64_t)(V / 2 + 16)) ++hbits;
  c->hbits = hbits;
  const size_t H = (size_t)1 << hbits;
  VGS_HIP_TRY(c, c->hkey.ensure(H * (sizeof(Brick) / 8)));
  VGS_HIP_TRY(c, hipMemsetAsync(c->hkey.p, 0, H * sizeof(Brick), c->stream));
  {
    // first = min over the brick's voxel ids: start from 0xffffffff
    Brick* tab = (Brick*)c->hkey.p;
    VGS_HIP_TRY(c, hipMemset2DAsync(&tab[0].first, sizeof(Brick), 0xff, sizeof(uint32_t), H, c->stream));
  }
  hipLaunchKernelGGL(k_brick_insert, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, c->stream, c->vox_code.p, c->node.p, V,
                     (Brick*)c->hkey.p, hbits);
  // ball of lattice offsets, ascending integer d2 (a superset of what the float predicate keeps)
  const double r = (double)c->P.graph_size;
  const double res = (double)c->P.voxel_size;
  *r2_out = (float)(r * r);  // static_cast<float>(radius * radius) in pcl::KdTreeFLANN::radiusSearch
  const int R = (int)std::ceil(r / res) + 1;
  if (R > 127) { c->err = "graph_size / voxel_size > 126 voxels"; return VGS_E_UNSUPPORTED; }
  std::vector<std::pair<int, int32_t>> offs;
  const double lim2 = (r / res) * (r / res) * (1.0 + 1e-4) + 1e-3;  // margin over float rounding of centres
  for (int dz = -R; dz <= R; ++dz)
    for (int dy = -R; dy <= R; ++dy)
      for (int dx = -R; dx <= R; ++dx) {
        const int d2 = dx * dx + dy * dy + dz * dz;
        if ((double)d2 <= lim2) offs.emplace_back(d2, (int32_t)((dx & 0xff) | ((dy & 0xff) << 8) | ((dz & 0xff) << 16)));
      }
  std::sort(offs.begin(), offs.end());
  c->n_off = (int)offs.size();
  c->adj_R = R;
  std::vector<int32_t> packed(offs.size());
  for (size_t k = 0; k < offs.size(); ++k) packed[k] = offs[k].second;
  VGS_HIP_TRY(c, c->offsets.ensure(packed.size()));
  VGS_HIP_TRY(c, hipMemcpy(c->offsets.p, packed.data(), packed.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  c->adj_stride = c->n_off;
  return VGS_OK;
}

// full = true: every neighbour (the reference's lists); false: used neighbours only (hot path)
vgs_status vgs_run_adjacency(vgs_ctx* c, bool full, uint64_t* out_key, uint32_t* out_cnt, uint32_t* out_nall, float r2) {
  const int64_t U = c->U;
  const float res_f = c->P.voxel_size;
  const float mnx = (float)c->box.min[0], mny = (float)c->box.min[1], mnz = (float)c->box.min[2];
#define LAUNCH_ADJ(CAPV, FULLV)                                                                                              \
  hipLaunchKernelGGL((k_adjacency<CAPV, FULLV>), dim3(vgs_xcd_grid(U)), dim3(64), 0, c->stream, c->vox_code.p, c->used_ids.p, U,    \
                     (const Brick*)c->hkey.p, c->hbits, c->offsets.p, c->n_off, c->adj_R, c->box.depth, res_f, mnx, mny, mnz, r2, c->node.p, \
                     c->adj_stride, out_key, out_cnt, out_nall)
  if (2 * c->adj_R + 1 > 32) { c->err = "neighbour ball wider than 31 voxels (graph_size / voxel_size > ~12)"; return VGS_E_UNSUPPORTED; }
  if (c->n_off <= 1024) { if (full) LAUNCH_ADJ(1024, true); else LAUNCH_ADJ(1024, false); }
  else if (c->n_off <= 8192) { if (full) LAUNCH_ADJ(8192, true); else LAUNCH_ADJ(8192, false); }
  else { c->err = "neighbour ball larger than 8192 lattice offsets (graph_size / voxel_size > ~12)"; return VGS_E_UNSUPPORTED; }
#undef LAUNCH_ADJ
  VGS_HIP_TRY(c, hipGetLastError());
  return VGS_OK;
}

vgs_status vgs_stage_adjacency(vgs_ctx* c) {
  const int64_t V = c->V, U = c->U;
  c->counts[VGS_N_ADJ] = 0;
  if (V == 0 || U == 0) return VGS_OK;
  float r2 = 0.f;
  vgs_status st = build_hash_and_offsets(c, &r2);
  if (st != VGS_OK) return st;
  c->adj_r2 = r2;
  c->adj_pruned = vgs_unused_are_inert(c->P);
  VGS_HIP_TRY(c, c->adj_key.ensure((size_t)U * c->adj_stride));
  VGS_HIP_TRY(c, c->adj_cnt.ensure(U)); VGS_HIP_TRY(c, c->adj_mused.ensure(U));
  st = vgs_run_adjacency(c, !c->adj_pruned, c->adj_key.p, c->adj_cnt.p, c->adj_mused.p, r2);
  if (st != VGS_OK) return st;
  return VGS_OK;
}
