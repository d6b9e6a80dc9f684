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

#include "brick_table.hpp"

// one wavefront (64-thread workgroup) per used voxel
template <int CAP, bool FULL>
__global__ __launch_bounds__(64) void k_adjacency(const uint64_t* __restrict__ vox_code, const uint32_t* __restrict__ used_ids,
                                                  int64_t U, const Brick* __restrict__ bricks,
                                                  uint32_t hbits, const int32_t* __restrict__ offsets, int n_off, int R, int depth,
                                                  float res_f, float min_x, float min_y, float min_z, float r2,
                                                  const NodeRec* __restrict__ node, int adj_stride,
                                                  uint64_t* __restrict__ adj_key, uint32_t* __restrict__ adj_cnt,
                                                  uint32_t* __restrict__ adj_mused, uint16_t* __restrict__ gtab, int gstride, int ngroups,
                                                  const int32_t* __restrict__ nvals, const uint32_t* __restrict__ redo, unsigned int* __restrict__ n_redo,
                                                  uint32_t* __restrict__ redo_out) {
  // redo != null: second pass over the rows that did not fit the first pass's list (their number is read on the device);
  // redo_out != null: first pass, rows with more than CAP survivors are appended there instead of being written
  __shared__ uint64_t lst[CAP];
  __shared__ uint8_t gl[CAP];    // integer squared length of each survivor's lattice offset (the table is sorted by it)
  __shared__ float ctab[3][32];  // voxel centres along each axis for key offsets -R..R (double arithmetic once per wavefront, not per offset)
  const int lane = threadIdx.x;
  int64_t u;
  if (redo) { if (blockIdx.x >= *n_redo) return; u = (int64_t)redo[blockIdx.x]; }
  else { u = vgs_xcd_item(blockIdx.x, U); if (u >= U) return; }
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
  const float res2 = res_f * res_f;
  bool in_band = true;  // every float d2 lies within half a lattice step^2 of its offset's integer length: groups do not interleave
  for (int base = 0; base < n_off; base += 64) {
    const int o = base + lane;
    bool keep = false;
    uint64_t key64 = 0;
    bool is_used = false;
    int norm = 0;
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
            norm = dx * dx + dy * dy + dz * dz;
            in_band = in_band && (fabsf(d2 - (float)norm * res2) < 0.49f * res2) && (norm < 256);
          }
        }
      }
    }
    const unsigned long long mall = __ballot(keep);
    const bool store = FULL ? keep : (keep && is_used);
    const unsigned long long m = __ballot(store);
    if (store) { const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull)); if (pos < CAP) { lst[pos] = key64; gl[pos] = (uint8_t)norm; } }
    cnt += __popcll(m);
    mused += __popcll(mall);
  }
  uint64_t* row = adj_key + (int64_t)u * adj_stride;
  __syncthreads();
  if (cnt > CAP) {  // dense volumetric neighbourhood: the pass with the full-size list takes this row
    if (lane == 0 && redo_out) redo_out[atomicAdd(n_redo, 1u)] = (uint32_t)u;
    return;
  }
  if (__ballot(!in_band) == 0ull) {
    // The survivors arrive grouped by the integer length of their offset, and the groups cannot interleave in float d2:
    // only the order inside a group (a handful of entries: equal lengths, keys differ in rounding and id) is open.  Each
    // entry counts the smaller keys of its group and goes straight to its final slot of the row.
    for (int p = lane; p < cnt; p += 64) {
      const uint64_t key = lst[p];
      const int g = gl[p];
      int first = p, rank = 0;
      for (int q = p - 1; q >= 0 && gl[q] == g; --q) { first = q; rank += lst[q] < key ? 1 : 0; }
      for (int q = p + 1; q < cnt && gl[q] == g; ++q) rank += lst[q] < key ? 1 : 0;
      row[first + rank] = key;
    }
    if (lane == 0) { adj_cnt[u] = (uint32_t)cnt; adj_mused[u] = (uint32_t)mused; }
    if (gtab) {
      // start of every length group in the row (lower bound of the length in the grouped list); entry ngroups = cnt
      uint16_t* gt = gtab + (int64_t)u * gstride;
      for (int r = lane; r <= ngroups; r += 64) {
        int lo = 0, hi = cnt;
        if (r < ngroups) {
          const int want = nvals[r];
          while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)gl[mid] < want) lo = mid + 1; else hi = mid; }
        } else {
          lo = cnt;
        }
        gt[r] = (uint16_t)lo;
      }
    }
    return;
  }
  if (gtab) for (int r = lane; r <= ngroups; r += 64) gtab[(int64_t)u * gstride + r] = 0xffffu;  // no group table for this row
  // general case (coordinates so large that the rounding of the centres rivals the lattice step): bitonic sort
  // ascending on the next power of two >= cnt
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
  vgs_status bs = vgs_build_bricks(c, c->node.p);
  if (bs != VGS_OK) return bs;
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
  // distinct integer lengths of the offsets (ascending) and the index of each length
  std::vector<int32_t> nvals;
  std::vector<uint8_t> nrank(256, 0xff);
  for (const auto& o : offs)
    if (nvals.empty() || nvals.back() != o.first) nvals.push_back(o.first);
  c->adj_ngroups = (int)nvals.size();
  c->adj_gstride = (c->adj_ngroups + 2) & ~1;   // ngroups + 1 entries, even
  c->adj_have_gtab = (c->adj_ngroups <= 250 && nvals.back() < 256);
  if (c->adj_have_gtab) {
    for (size_t k = 0; k < nvals.size(); ++k) nrank[(size_t)nvals[k]] = (uint8_t)k;
    VGS_HIP_TRY(c, c->adj_nvals.ensure(nvals.size())); VGS_HIP_TRY(c, c->adj_nrank.ensure(256));
    VGS_HIP_TRY(c, hipMemcpy(c->adj_nvals.p, nvals.data(), nvals.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    VGS_HIP_TRY(c, hipMemcpy(c->adj_nrank.p, nrank.data(), 256, hipMemcpyHostToDevice));
  }
  return VGS_OK;
}

// full = true: every neighbour (the reference's lists); false: used neighbours only (hot path)
vgs_status vgs_run_adjacency(vgs_ctx* c, bool full, uint64_t* out_key, uint32_t* out_cnt, uint32_t* out_nall, float r2) {
  const int64_t U = c->U;
  const float res_f = c->P.voxel_size;
  const float mnx = (float)c->box.min[0], mny = (float)c->box.min[1], mnz = (float)c->box.min[2];
  uint16_t* gt = nullptr;   // group tables only for the rows the pipeline keeps
  if (out_key == c->adj_key.p && c->adj_have_gtab) {
    VGS_HIP_TRY(c, c->adj_gtab.ensure((size_t)U * c->adj_gstride));
    gt = c->adj_gtab.p;
  }
#define LAUNCH_ADJ(CAPV, FULLV, GRID, REDO, NREDO, REDO_OUT)                                                                 \
  hipLaunchKernelGGL((k_adjacency<CAPV, FULLV>), dim3(GRID), dim3(64), 0, c->stream, c->vox_code.p, c->used_ids.p, U,            \
                     (const Brick*)c->hkey.p, c->hbits, c->offsets.p, c->n_off, c->adj_R, c->box.depth, res_f, mnx, mny, mnz, r2, c->node.p, \
                     c->adj_stride, out_key, out_cnt, out_nall, gt, c->adj_gstride, c->adj_ngroups, c->adj_nvals.p, REDO, NREDO, REDO_OUT)
  if (2 * c->adj_R + 1 > 32) { c->err = "neighbour ball wider than 31 voxels (graph_size / voxel_size > ~12)"; return VGS_E_UNSUPPORTED; }
  if (c->n_off <= 1024) {
    if (full) LAUNCH_ADJ(1024, true, vgs_xcd_grid(U), nullptr, nullptr, nullptr); else LAUNCH_ADJ(1024, false, vgs_xcd_grid(U), nullptr, nullptr, nullptr);
  } else if (c->n_off <= 8192) {
    // A list for every lattice offset (72 KB of LDS) leaves two wavefronts per CU; surfaces fill a fraction of the ball,
    // so the first pass runs with 2048 slots (18 KB) and hands rows that overflow to a second pass over a device-side list
    VGS_HIP_TRY(c, c->work_ids.ensure((size_t)U + 16)); VGS_HIP_TRY(c, c->counters.ensure(64));
    unsigned int* d_nredo = (unsigned int*)(c->counters.p + 40);
    VGS_HIP_TRY(c, hipMemsetAsync(d_nredo, 0, 4, c->stream));
    if (full) LAUNCH_ADJ(2048, true, vgs_xcd_grid(U), nullptr, d_nredo, c->work_ids.p); else LAUNCH_ADJ(2048, false, vgs_xcd_grid(U), nullptr, d_nredo, c->work_ids.p);
    const unsigned int g2 = (unsigned int)U;   // upper bound; workgroups beyond the list leave at once
    if (full) LAUNCH_ADJ(8192, true, g2, c->work_ids.p, d_nredo, nullptr); else LAUNCH_ADJ(8192, false, g2, c->work_ids.p, d_nredo, nullptr);
  }
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
