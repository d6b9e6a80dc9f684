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
#include <cstdio>
#include <cstdlib>
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
                                                  uint32_t* __restrict__ redo_out, uint16_t* __restrict__ adj_off, unsigned int* __restrict__ n_redo_out) {
  // adj_off != null: the packed lattice offset (dx+16) | (dy+16) << 5 | (dz+16) << 10 of every stored entry goes to a row of
  // its own (the local cut reads it instead of gathering the neighbours' records for their lattice coordinates); a row that
  // is not in band gets 0xffff in its first slot: its order comes from a sort that does not carry the offsets
  // redo != null: second pass over the rows that did not fit the first pass's list (their number is read on the device);
  // redo_out != null: first pass, rows with more than CAP survivors are appended there instead of being written
  __shared__ uint64_t lst[CAP];
  __shared__ uint8_t gl[CAP];    // integer squared length of each survivor's lattice offset (the table is sorted by it)
  __shared__ uint16_t loff[CAP]; // its packed lattice offset
  __shared__ float ctab[3][32];  // voxel centres along each axis for key offsets -R..R (double arithmetic once per wavefront, not per offset)
  const int lane = threadIdx.x;
  auto do_row = [&](const int64_t u) {
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
    uint32_t poff = 0;
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
            poff = (uint32_t)(dx + 16) | ((uint32_t)(dy + 16) << 5) | ((uint32_t)(dz + 16) << 10);
            in_band = in_band && (fabsf(d2 - (float)norm * res2) < 0.49f * res2) && (norm < 256);
          }
        }
      }
    }
    const unsigned long long mall = __ballot(keep);
    const bool store = FULL ? keep : (keep && is_used);
    const unsigned long long m = __ballot(store);
    if (store) { const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull)); if (pos < CAP) { lst[pos] = key64; gl[pos] = (uint8_t)norm; loff[pos] = (uint16_t)poff; } }
    cnt += __popcll(m);
    mused += __popcll(mall);
  }
  uint64_t* row = adj_key + (int64_t)u * adj_stride;
  uint16_t* orow = adj_off ? adj_off + (int64_t)u * adj_stride : nullptr;
  __syncthreads();
  if (cnt > CAP) {  // dense volumetric neighbourhood: the pass with the full-size list takes this row
    if (lane == 0 && redo_out) redo_out[atomicAdd(n_redo_out, 1u)] = (uint32_t)u;   // (its own counter: a middle pass reads one list and fills the next)
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
      if (orow) orow[first + rank] = loff[p];
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
  if (orow && lane == 0) orow[0] = 0xffffu;   // no offsets either
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
  };   // do_row
  if (redo) {
    // a fixed grid strides over the redo list, whose length stays on the device: no host round trip before this launch
    for (unsigned int w = blockIdx.x; w < *n_redo; w += gridDim.x) {
      do_row((int64_t)redo[w]);
      __syncthreads();   // the next row reuses the list
    }
  } else {
    const int64_t u = vgs_xcd_item(blockIdx.x, U);
    if (u < U) do_row(u);
  }
}

// The rows nobody else could take -- more than 2048 used neighbours: a solid volume seen through a ball of ten voxels holds up to 4159 --
// with a WORKGROUP per row (round 6).  k_adjacency<8192> gives such a row one wavefront, and its 88 KB of LDS leave ONE wavefront per CU:
// 66 dependent probe trips and a rank loop over groups of a hundred entries with nothing beside them to hide a single latency (33 ms for
// the 17 k rows of the r = 10 block, 87 % of its step).  Here TB threads share the row: phase A probes the ball's offsets, every thread
// parking its key in the row's own slot of the table (slot = offset index; the same thread reads it back); the survivors of every 64
// offsets are counted, the counts scanned, and phase B moves the keys to their places in the list -- offset order, i.e. grouped by
// integer length as the general kernel has them; the rank inside a group and the group table are the general kernel's, TB entries at a
// time.  Used neighbours only (the hot path); the rows' order, offsets and group tables are the general kernel's, bit for bit.
template <int TB>
__global__ __launch_bounds__(TB) void k_adjacency_wide(const uint64_t* __restrict__ vox_code, const uint32_t* __restrict__ used_ids, const Brick* __restrict__ bricks,
                                                       uint32_t hbits, const int32_t* __restrict__ offsets, int n_off, int R, int depth,
                                                       float res_f, float min_x, float min_y, float min_z, float r2, int adj_stride,
                                                       uint64_t* __restrict__ adj_key, uint32_t* __restrict__ adj_cnt, uint32_t* __restrict__ adj_mused,
                                                       uint16_t* __restrict__ gtab, int gstride, int ngroups, const int32_t* __restrict__ nvals,
                                                       const uint32_t* __restrict__ redo, const unsigned int* __restrict__ n_redo, uint16_t* __restrict__ adj_off) {
  constexpr int CAP = 8192, NCH = CAP / 64;
  __shared__ uint64_t lst[CAP];
  __shared__ uint8_t gl[CAP];
  __shared__ uint16_t loff[CAP];
  __shared__ float ctab[3][32];
  __shared__ int s_ch[NCH + 1];   // survivors of every 64 offsets, then their exclusive prefix sums
  __shared__ int s_mused, s_outband;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto do_row = [&](const int64_t u) {
    const uint32_t i = used_ids[u];
    const uint64_t code = vox_code[i];
    const uint32_t kx = vm_compact21(code >> 2), ky = vm_compact21(code >> 1), kz = vm_compact21(code);
    const float cx = vm_voxel_center(kx, res_f, min_x), cy = vm_voxel_center(ky, res_f, min_y), cz = vm_voxel_center(kz, res_f, min_z);
    const uint32_t lim = 1u << depth;
    if (tid < 2 * R + 1) {
      ctab[0][tid] = vm_voxel_center(kx + (uint32_t)(tid - R), res_f, min_x);
      ctab[1][tid] = vm_voxel_center(ky + (uint32_t)(tid - R), res_f, min_y);
      ctab[2][tid] = vm_voxel_center(kz + (uint32_t)(tid - R), res_f, min_z);
    }
    if (tid == 0) { s_mused = 0; s_outband = 0; }
    for (int k = tid; k <= NCH; k += TB) s_ch[k] = 0;
    __syncthreads();
    uint64_t* row = adj_key + (int64_t)u * adj_stride;
    uint16_t* orow = adj_off ? adj_off + (int64_t)u * adj_stride : nullptr;
    const float res2 = res_f * res_f;
    int mused = 0;
    bool in_band = true;
    // ---- phase A: probe, park the key in the row's slot of this offset ----
    for (int base = wave * 64; base < n_off; base += TB) {
      const int o = base + lane;
      bool keep = false, is_used = false;
      uint64_t key64 = 0;
      if (o < n_off) {
        const int32_t pk = offsets[o];
        const int dx = (int)(int8_t)(pk & 0xff), dy = (int)(int8_t)((pk >> 8) & 0xff), dz = (int)(int8_t)((pk >> 16) & 0xff);
        const uint32_t nx = kx + (uint32_t)dx, ny = ky + (uint32_t)dy, nz = kz + (uint32_t)dz;
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
              const int norm = dx * dx + dy * dy + dz * dz;
              in_band = in_band && (fabsf(d2 - (float)norm * res2) < 0.49f * res2) && (norm < 256);
            }
          }
        }
        row[o] = (keep && is_used) ? key64 : 0ull;   // (a real key is never 0: the voxel itself, d2 = 0, id >= 0 ... id 0 at distance 0 IS 0: see below)
      }
      mused += __popcll(__ballot(keep));
      const unsigned long long m = __ballot(keep && is_used);
      if (lane == 0) s_ch[base >> 6] = __popcll(m);
    }
    if (lane == 0 && mused) atomicAdd(&s_mused, mused);
    if (!in_band) s_outband = 1;
    __syncthreads();
    // ---- the chunks' places (one wavefront; 132 chunks at most) ----
    if (wave == 0) {
      int carry = 0;
      for (int b0 = 0; b0 <= NCH; b0 += 64) {
        const int k = b0 + lane;
        const int x = k <= NCH ? s_ch[k] : 0;
        int incl = x;
        for (int o2 = 1; o2 < 64; o2 <<= 1) { const int y = __shfl_up(incl, o2, 64); if (lane >= o2) incl += y; }
        if (k <= NCH) s_ch[k] = carry + incl - x;
        carry += __shfl(incl, 63, 64);
      }
    }
    __syncthreads();
    const int cnt = s_ch[NCH];
    // ---- phase B: every thread takes its keys back and puts them in place ----
    for (int base = wave * 64; base < n_off; base += TB) {
      const int o = base + lane;
      uint64_t key64 = 0;
      int norm = 0;
      uint32_t poff = 0;
      bool store = false;
      if (o < n_off) {
        const int32_t pk = offsets[o];
        const int dx = (int)(int8_t)(pk & 0xff), dy = (int)(int8_t)((pk >> 8) & 0xff), dz = (int)(int8_t)((pk >> 16) & 0xff);
        key64 = row[o];
        // the voxel with id 0 at distance 0 has the key 0: it is the row's own voxel (offset 0, always used here) -- stored by its offset
        store = key64 != 0ull || (pk & 0xffffff) == 0;
        norm = dx * dx + dy * dy + dz * dz;
        poff = (uint32_t)(dx + 16) | ((uint32_t)(dy + 16) << 5) | ((uint32_t)(dz + 16) << 10);
      }
      const unsigned long long m = __ballot(store);
      if (store) { const int pos = s_ch[base >> 6] + __popcll(m & ((1ull << lane) - 1ull)); lst[pos] = key64; gl[pos] = (uint8_t)norm; loff[pos] = (uint16_t)poff; }
    }
    __syncthreads();
    if (s_outband == 0) {
      for (int p = tid; p < cnt; p += TB) {
        const uint64_t key = lst[p];
        const int g = gl[p];
        int first = p, rank = 0;
        for (int q = p - 1; q >= 0 && gl[q] == g; --q) { first = q; rank += lst[q] < key ? 1 : 0; }
        for (int q = p + 1; q < cnt && gl[q] == g; ++q) rank += lst[q] < key ? 1 : 0;
        row[first + rank] = key;
        if (orow) orow[first + rank] = loff[p];
      }
      if (tid == 0) { adj_cnt[u] = (uint32_t)cnt; adj_mused[u] = (uint32_t)s_mused; }
      if (gtab) {
        uint16_t* gt = gtab + (int64_t)u * gstride;
        for (int r = tid; r <= ngroups; r += TB) {
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
    if (gtab) for (int r = tid; r <= ngroups; r += TB) gtab[(int64_t)u * gstride + r] = 0xffffu;
    if (orow && tid == 0) orow[0] = 0xffffu;
    int np = 64;
    while (np < cnt) np <<= 1;
    for (int k = cnt + tid; k < np; k += TB) lst[k] = ~0ull;
    __syncthreads();
    for (int size = 2; size <= np; size <<= 1) {
      for (int strd = size >> 1; strd > 0; strd >>= 1) {
        for (int t = tid; t < (np >> 1); t += TB) {
          const int lo = ((t / strd) * (strd << 1)) + (t & (strd - 1));
          const int hi = lo + strd;
          const bool up = ((lo & size) == 0);
          const uint64_t a = lst[lo], b = lst[hi];
          if ((a > b) == up) { lst[lo] = b; lst[hi] = a; }
        }
        __syncthreads();
      }
    }
    for (int k = tid; k < cnt; k += TB) row[k] = lst[k];
    if (tid == 0) { adj_cnt[u] = (uint32_t)cnt; adj_mused[u] = (uint32_t)s_mused; }
  };   // do_row
  for (unsigned int w = blockIdx.x; w < *n_redo; w += gridDim.x) {
    do_row((int64_t)redo[w]);
    __syncthreads();   // the next row reuses every array
  }
}

// ---- hot path for small balls: candidates from brick occupancy masks --------------------------------------------
// On a surface about a sixth of the ball's lattice cells are occupied, yet the kernel above probes the brick table for
// every one of them.  Here a lane fetches one BRICK of the (at most 5 x 5 x 5) bricks the ball touches, ANDs its occupancy
// with the precomputed mask of ball cells inside that brick (one mask per position of the voxel in its own brick), and the
// set bits of all lanes are expanded into a dense candidate list: the per-cell work (float predicate, key, group) is done
// for occupied cells only.  Candidates arrive brick by brick, so the survivors are put in order of integer offset length
// with a counting sort (histogram in LDS); inside a length group the final slot is found as above.  Rows that do not fit,
// or whose float distances leave the band of their integer lengths, go to the redo list of the general kernel.
// (round 4: balls up to 12 voxels -- 7 x 7 x 7 bricks, six per lane -- take this path too: config 2's ball of ten voxels went through
// the general kernel's 2048-slot pass, which tests all 4189 offsets of the ball where a planar neighbourhood occupies 305.)
// ADJM_TRIPS candidates per lane are kept in registers (CAPC = 64 * ADJM_TRIPS)
// LISTED (round 5): the rows of a device-side list (the ones the first instantiation passed on: more survivors than its list holds), a fixed
// grid striding over it -- with a 512-entry list they stay on this path instead of the general kernel's nine dependent probe trips per row
template <int CAP, int NB, int ADJM_TRIPS, bool LISTED = false>
__global__ __launch_bounds__(64) void k_adjacency_masks(const uint64_t* __restrict__ vox_code, const uint32_t* __restrict__ used_ids, int64_t U,
                                                        const Brick* __restrict__ bricks, uint32_t hbits, const uint64_t* __restrict__ masks,
                                                        int R, float res_f, float min_x, float min_y, float min_z, float r2,
                                                        int adj_stride, uint64_t* __restrict__ adj_key, uint32_t* __restrict__ adj_cnt,
                                                        uint32_t* __restrict__ adj_mused, uint16_t* __restrict__ gtab, int gstride, int ngroups,
                                                        const int32_t* __restrict__ nvals, unsigned int* __restrict__ n_redo,
                                                        uint32_t* __restrict__ redo_out, uint16_t* __restrict__ adj_off,
                                                        const uint32_t* __restrict__ rows_in = nullptr, const unsigned int* __restrict__ n_rows_in = nullptr) {
  constexpr int NB3 = NB * NB * NB, Bh = NB / 2, CAPC = 64 * ADJM_TRIPS;
  constexpr int BTRIPS = (NB3 + 63) / 64, ADJM_BRICKS = 64 * BTRIPS;   // bricks per lane, table size
  constexpr int ADJM_BRICKS_P2 = ADJM_BRICKS <= 64 ? 64 : (ADJM_BRICKS <= 128 ? 128 : (ADJM_BRICKS <= 256 ? 256 : 512));   // the search's first stride is half of this
  static_assert(NB3 <= 512, "brick slot: nine bits of a candidate");
  __shared__ uint64_t s_occ[ADJM_BRICKS];
  __shared__ uint32_t s_first[ADJM_BRICKS];
  // (round 5) the candidates are not written out as a list any more: a lane of the evaluation finds ITS candidate -- the slot whose run
  // of candidates holds number q (binary search over the slots' prefix sums), then the (q - prefix)-th set bit of that slot's mask
  __shared__ uint16_t s_pref[ADJM_BRICKS];   // exclusive prefix of the slots' candidate counts, slot order j = lane * BTRIPS + trip
  // per slot (index sidx = trip * 64 + lane) the candidate mask and the used mask -- read by the evaluation only, whose survivors stay in
  // registers until two barriers later: the row's list, groups and lattice offsets take the same bytes
  constexpr int ROW_BYTES = CAP * 8 + CAP * 2 + CAP, SLOT_BYTES = ADJM_BRICKS * 16;
  __shared__ __attribute__((aligned(16))) unsigned char s_buf[ROW_BYTES > SLOT_BYTES ? ROW_BYTES : SLOT_BYTES];
  uint64_t* const s_cm = (uint64_t*)s_buf;
  uint64_t* const s_cu = s_cm + ADJM_BRICKS;
  uint64_t* const lst = (uint64_t*)s_buf;
  uint16_t* const loff = (uint16_t*)(s_buf + CAP * 8);   // packed lattice offset of every survivor (see k_adjacency)
  uint8_t* const gl = s_buf + CAP * 8 + CAP * 2;
  __shared__ uint32_t s_hist[128], s_cur[128];   // two 16-bit counters per word: integer length l lives in word l >> 1
  __shared__ float ctab[3][32];
  const int lane = threadIdx.x;
  auto do_row = [&](const int64_t u) {
  const uint32_t i = used_ids[u];
  const uint64_t code = vox_code[i];
  const uint32_t kx = vm_compact21(code >> 2), ky = vm_compact21(code >> 1), kz = vm_compact21(code);
  const float cx = vm_voxel_center(kx, res_f, min_x), cy = vm_voxel_center(ky, res_f, min_y), cz = vm_voxel_center(kz, res_f, min_z);
  if (lane < 2 * R + 1) {
    ctab[0][lane] = vm_voxel_center(kx + (uint32_t)(lane - R), res_f, min_x);
    ctab[1][lane] = vm_voxel_center(ky + (uint32_t)(lane - R), res_f, min_y);
    ctab[2][lane] = vm_voxel_center(kz + (uint32_t)(lane - R), res_f, min_z);
  }
  for (int k = lane; k < 128; k += 64) s_hist[k] = 0;
  const int px = (int)(kx & 3u), py = (int)(ky & 3u), pz = (int)(kz & 3u);
  const uint64_t* mrow = masks + (size_t)((pz * 4 + py) * 4 + px) * (size_t)NB3;
  // ---- one brick per lane and trip: the whole 32-byte table entry comes in one round trip ----
  uint64_t cand[BTRIPS], cusd[BTRIPS];
#pragma unroll
  for (int trip = 0; trip < BTRIPS; ++trip) { cand[trip] = 0ull; cusd[trip] = 0ull; }
#pragma unroll
  for (int trip = 0; trip < BTRIPS; ++trip) {
    const int sidx = trip * 64 + lane;
    uint64_t occ = 0ull;
    uint32_t first = 0u;
    if (sidx < NB3) {
      const uint64_t bm = mrow[sidx];
      const int bi = sidx % NB - Bh, bj = (sidx / NB) % NB - Bh, bk = sidx / (NB * NB) - Bh;
      const int nbx = (int)(kx >> 2) + bi, nby = (int)(ky >> 2) + bj, nbz = (int)(kz >> 2) + bk;
      if (bm != 0ull && nbx >= 0 && nby >= 0 && nbz >= 0) {
        const unsigned long long key = (((unsigned long long)nbz << 42) | ((unsigned long long)nby << 21) | (unsigned long long)nbx) + 1ull;
        const uint32_t mask = (1u << hbits) - 1u;
        uint32_t sl = hash_slot(key, hbits);
        while (true) {
          const ulonglong2 k01 = *(const ulonglong2*)&bricks[sl];          // key, occ
          if (k01.x == key) {
            const ulonglong2 k23 = *((const ulonglong2*)&bricks[sl] + 1);  // used, (first, pad)
            occ = k01.y; first = (uint32_t)k23.y;
            cand[trip] = occ & bm; cusd[trip] = k23.x;
            break;
          }
          if (k01.x == 0ull) break;
          sl = (sl + 1) & mask;
        }
      }
      s_occ[sidx] = occ; s_first[sidx] = first;
    }
  }
#if defined(ADJM_STOP) && ADJM_STOP == 1
  if (lane == 0) adj_cnt[u] = (uint32_t)(cand[0] != 0ull); return;
#endif
  // ---- the slots' prefix sums (round 5: was a serial per-lane loop over each lane's set bits, 0.43 of the stage's 0.92 ms -- a wavefront
  // walked it as often as its fullest brick has cells in the ball, sixteen to thirty times for two or three useful lanes) ----
  int mine = 0;
#pragma unroll
  for (int trip = 0; trip < BTRIPS; ++trip) mine += __popcll(cand[trip]);
  int incl = mine;
  for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
  const int ntot = __shfl(incl, 63, 64);
  auto to_redo = [&]() { if (lane == 0) redo_out[atomicAdd(n_redo, 1u)] = (uint32_t)u; };
  if (ntot > CAPC) { to_redo(); return; }
  {
    int pos = incl - mine;
#pragma unroll
    for (int trip = 0; trip < BTRIPS; ++trip) {
      s_pref[lane * BTRIPS + trip] = (uint16_t)pos;
      pos += __popcll(cand[trip]);
      s_cm[trip * 64 + lane] = cand[trip];
      s_cu[trip * 64 + lane] = cusd[trip];
    }
  }
  __syncthreads();
#if defined(ADJM_STOP) && ADJM_STOP == 2
  if (lane == 0) adj_cnt[u] = (uint32_t)s_pref[1]; return;
#endif
  // ---- per candidate: float predicate, key, integer length; survivors stay in registers until their slots are known ----
  const float res2 = res_f * res_f;
  int nstore = 0, mused = 0;
  bool in_band = true;
  uint64_t rkey[ADJM_TRIPS];
  int rnorm[ADJM_TRIPS];   // -1 = nothing to store; bits 8.. hold the packed lattice offset
#pragma unroll
  for (int tr = 0; tr < ADJM_TRIPS; ++tr) {
    rnorm[tr] = -1; rkey[tr] = 0;
    const int q = tr * 64 + lane;
    if (tr * 64 < ntot) {   // wave-uniform
      bool keep = false, is_used = false;
      if (q < ntot) {
        // slot: the LAST j with prefix <= q (an empty slot shares its prefix with the next one that is not)
        int j = 0;
#pragma unroll
        for (int step = ADJM_BRICKS_P2 / 2; step >= 1; step >>= 1) {
          const int mid = j + step;
          if (mid < ADJM_BRICKS && (int)s_pref[mid] <= q) j = mid;
        }
        const int sidx = (j % BTRIPS) * 64 + j / BTRIPS;
        int kth = q - (int)s_pref[j];
        const uint64_t cmask = s_cm[sidx];
        // the kth set bit of the slot's mask: halves by population count
        uint32_t w = (uint32_t)cmask;
        int bit = 0;
        { const int c = __popc(w); if (kth >= c) { kth -= c; w = (uint32_t)(cmask >> 32); bit = 32; } }
        { const int c = __popc(w & 0xffffu); if (kth >= c) { kth -= c; w >>= 16; bit += 16; } }
        { const int c = __popc(w & 0xffu); if (kth >= c) { kth -= c; w >>= 8; bit += 8; } }
        { const int c = __popc(w & 0xfu); if (kth >= c) { kth -= c; w >>= 4; bit += 4; } }
        { const int c = __popc(w & 0x3u); if (kth >= c) { kth -= c; w >>= 2; bit += 2; } }
        { const int c = (int)(w & 1u); if (kth >= c) { bit += 1; } }
        is_used = ((s_cu[sidx] >> bit) & 1ull) != 0ull;
        const int bi = sidx % NB - Bh, bj = (sidx / NB) % NB - Bh, bk = sidx / (NB * NB) - Bh;
        // bit = z0 y0 x0 z1 y1 x1 (brick_local)
        const int lx = ((bit >> 2) & 1) | (((bit >> 5) & 1) << 1), ly = ((bit >> 1) & 1) | (((bit >> 4) & 1) << 1), lz = (bit & 1) | (((bit >> 3) & 1) << 1);
        const int dx = 4 * bi + lx - px, dy = 4 * bj + ly - py, dz = 4 * bk + lz - pz;
        const uint64_t occ = s_occ[sidx];
        const uint64_t above = (bit == 63) ? 0ull : (occ >> (bit + 1));
        const uint32_t t = s_first[sidx] + (uint32_t)__popcll(above);
        const float tx = cx - ctab[0][dx + R];
        const float ty = cy - ctab[1][dy + R];
        const float tz = cz - ctab[2][dz + R];
        const float d2 = (tx * tx + ty * ty) + tz * tz;
        if (d2 < r2) {
          keep = true;
          const int norm = dx * dx + dy * dy + dz * dz;
          in_band = in_band && (fabsf(d2 - (float)norm * res2) < 0.49f * res2) && (norm < 256);
          if (is_used) {
            rkey[tr] = ((uint64_t)vm_bits(d2) << 32) | t;
            rnorm[tr] = (norm & 255) | (int)(((uint32_t)(dx + 16) | ((uint32_t)(dy + 16) << 5) | ((uint32_t)(dz + 16) << 10)) << 8);
            atomicAdd(&s_hist[(norm & 255) >> 1], 1u << (16 * (norm & 1)));
          }
        }
      }
      nstore += __popcll(__ballot(keep && is_used));
      mused += __popcll(__ballot(keep));
    }
  }
  if (nstore > CAP || __ballot(!in_band) != 0ull) { to_redo(); return; }
  __syncthreads();
#if defined(ADJM_STOP) && ADJM_STOP == 3
  if (lane == 0) adj_cnt[u] = (uint32_t)nstore + (uint32_t)rkey[0]; return;
#endif
  // ---- counting sort by integer length: start of every length's group (four lengths per lane) ----
  {
    const uint32_t w0 = s_hist[2 * lane], w1 = s_hist[2 * lane + 1];
    const uint32_t h[4] = {w0 & 0xffffu, w0 >> 16, w1 & 0xffffu, w1 >> 16};
    const uint32_t sum = h[0] + h[1] + h[2] + h[3];
    uint32_t inc = sum;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64); if (lane >= o) inc += v; }
    const uint32_t s0 = inc - sum, s1 = s0 + h[0], s2 = s1 + h[1], s3 = s2 + h[2];
    const uint32_t p0 = s0 | (s1 << 16), p1 = s2 | (s3 << 16);
    s_hist[2 * lane] = p0; s_hist[2 * lane + 1] = p1;   // starts
    s_cur[2 * lane] = p0; s_cur[2 * lane + 1] = p1;     // cursors
  }
  __syncthreads();
#pragma unroll
  for (int tr = 0; tr < ADJM_TRIPS; ++tr) {
    if (rnorm[tr] >= 0) {
      const int g = rnorm[tr] & 255;
      const uint32_t old = atomicAdd(&s_cur[g >> 1], 1u << (16 * (g & 1)));
      const uint32_t pos = (old >> (16 * (g & 1))) & 0xffffu;
      lst[pos] = rkey[tr]; gl[pos] = (uint8_t)g; loff[pos] = (uint16_t)(rnorm[tr] >> 8);
    }
  }
  __syncthreads();
#if defined(ADJM_STOP) && ADJM_STOP == 4
  if (lane == 0) adj_cnt[u] = (uint32_t)lst[0]; return;
#endif
  // ---- final slot inside the group (the order inside a group is open: equal lengths, keys differ in rounding and id) ----
  uint64_t* row = adj_key + (int64_t)u * adj_stride;
  uint16_t* orow = adj_off ? adj_off + (int64_t)u * adj_stride : nullptr;
  const int cnt = nstore;
  for (int p = lane; p < cnt; p += 64) {
    const uint64_t key = lst[p];
    const int g = gl[p];
    const int first = (int)((s_hist[g >> 1] >> (16 * (g & 1))) & 0xffffu);
    const int last = (int)((s_cur[g >> 1] >> (16 * (g & 1))) & 0xffffu);   // one past the group
    int rank = 0;
    for (int q = first; q < last; ++q) rank += lst[q] < key ? 1 : 0;
    row[first + rank] = key;
    if (orow) orow[first + rank] = loff[p];
  }
  if (lane == 0) { adj_cnt[u] = (uint32_t)cnt; adj_mused[u] = (uint32_t)mused; }
  if (gtab) {
    uint16_t* gt = gtab + (int64_t)u * gstride;
    for (int r = lane; r <= ngroups; r += 64) {
      const int g = r < ngroups ? (nvals[r] & 255) : 0;
      gt[r] = (uint16_t)(r < ngroups ? ((s_hist[g >> 1] >> (16 * (g & 1))) & 0xffffu) : (uint32_t)cnt);
    }
  }
  };   // do_row
  if constexpr (LISTED) {
    for (unsigned int w = blockIdx.x; w < *n_rows_in; w += gridDim.x) {
      do_row((int64_t)rows_in[w]);
      __syncthreads();   // the next row reuses the tables
    }
  } else {
    const int64_t u = vgs_xcd_item(blockIdx.x, U);
    if (u < U) do_row(u);
  }
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
  if (!c->bricks_ready) {   // (the features stage usually has queued it behind its read-back)
    vgs_status bs = vgs_build_bricks(c, c->node.p);
    if (bs != VGS_OK) return bs;
  }
  c->bricks_ready = false;   // good for one adjacency stage: a later one (new graph_size, tiles) builds its own
  // ball of lattice offsets, ascending integer d2 (a superset of what the float predicate keeps)
  const double r = (double)c->P.graph_size;
  const double res = (double)c->P.voxel_size;
  *r2_out = (float)(r * r);  // static_cast<float>(radius * radius) in pcl::KdTreeFLANN::radiusSearch
  // everything below depends on graph_size / voxel_size only: built and uploaded once per parameter pair, not per run
  // (four blocking uploads were 0.2 ms of every step)
  if (c->adj_tab_valid && c->adj_tab_graph == c->P.graph_size && c->adj_tab_voxel == c->P.voxel_size) return VGS_OK;
  c->adj_tab_valid = false;
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
  {
    // ball masks for k_adjacency_masks: for every position p of a voxel inside its 4x4x4 brick and every brick offset b,
    // the cells of that brick that belong to the ball (bit = brick_local of the cell)
    int Rm = 0;
    for (const auto& o : offs) {
      const int32_t pk = o.second;
      const int d[3] = {(int)(int8_t)(pk & 0xff), (int)(int8_t)((pk >> 8) & 0xff), (int)(int8_t)((pk >> 16) & 0xff)};
      for (int a = 0; a < 3; ++a) Rm = std::max(Rm, std::abs(d[a]));
    }
    const int Bh = (Rm + 3) / 4, NB = 2 * Bh + 1, NB3 = NB * NB * NB;
    c->adj_mask_nb = 0;
    if (NB == 3 || NB == 5 || NB == 7) {
      std::vector<uint64_t> masks((size_t)64 * NB3, 0ull);
      for (int p = 0; p < 64; ++p) {
        const int px = p & 3, py = (p >> 2) & 3, pz = p >> 4;
        for (const auto& o : offs) {
          const int32_t pk = o.second;
          const int dx = (int)(int8_t)(pk & 0xff), dy = (int)(int8_t)((pk >> 8) & 0xff), dz = (int)(int8_t)((pk >> 16) & 0xff);
          const int ax = px + dx, ay = py + dy, az = pz + dz;   // cell relative to the origin of the voxel's brick
          const int bi = (ax >= 0 ? ax / 4 : -((3 - ax) / 4)), bj = (ay >= 0 ? ay / 4 : -((3 - ay) / 4)), bk = (az >= 0 ? az / 4 : -((3 - az) / 4));
          const int lx = ax - 4 * bi, ly = ay - 4 * bj, lz = az - 4 * bk;
          const int local = (lz & 1) | ((ly & 1) << 1) | ((lx & 1) << 2) | ((lz & 2) << 2) | ((ly & 2) << 3) | ((lx & 2) << 4);
          masks[(size_t)p * NB3 + (size_t)(((bk + Bh) * NB + (bj + Bh)) * NB + (bi + Bh))] |= 1ull << local;
        }
      }
      VGS_HIP_TRY(c, c->adj_masks.ensure(masks.size()));
      VGS_HIP_TRY(c, hipMemcpy(c->adj_masks.p, masks.data(), masks.size() * 8, hipMemcpyHostToDevice));
      c->adj_mask_nb = NB;
    }
  }
  std::vector<int32_t> packed(offs.size());
  for (size_t k = 0; k < offs.size(); ++k) packed[k] = offs[k].second;
  {
    // connect bits (vgs_context.hpp): one bit per cell of the cube of side 2 Rm + 1 around a voxel, Rm = the ball's largest offset
    int Rm = 0;
    for (const auto& o : offs) {
      const int32_t pk = o.second;
      Rm = std::max(Rm, std::max(std::abs((int)(int8_t)(pk & 0xff)), std::max(std::abs((int)(int8_t)((pk >> 8) & 0xff)), std::abs((int)(int8_t)((pk >> 16) & 0xff)))));
    }
    c->cb_R = Rm;
    const int D = 2 * Rm + 1;
    c->cb_words = (D * D * D + 31) / 32;
  }
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
  c->adj_tab_graph = c->P.graph_size; c->adj_tab_voxel = c->P.voxel_size; c->adj_tab_valid = true;
  return VGS_OK;
}

// full = true: every neighbour (the reference's lists); false: used neighbours only (hot path)
// ids / n_ids (optional): the rows to build instead of the used voxels' (any voxel ids; row k belongs to ids[k])
vgs_status vgs_run_adjacency(vgs_ctx* c, bool full, uint64_t* out_key, uint32_t* out_cnt, uint32_t* out_nall, float r2, const uint32_t* ids,
                             int64_t n_ids) {
  const int64_t U = ids ? n_ids : c->U;
  const uint32_t* row_ids = ids ? ids : c->used_ids.p;
  const float res_f = c->P.voxel_size;
  const float mnx = (float)c->box.min[0], mny = (float)c->box.min[1], mnz = (float)c->box.min[2];
  uint16_t* gt = nullptr;   // group tables only for the rows the pipeline keeps
  uint16_t* off = nullptr;  // the entries' lattice offsets likewise (voxel lattice only: the local cut of method 2 reads them)
  if (out_key == c->adj_key.p && c->adj_have_gtab) {
    VGS_HIP_TRY(c, c->adj_gtab.ensure((size_t)U * c->adj_gstride));
    gt = c->adj_gtab.p;
  }
  if (out_key == c->adj_key.p && c->P.method == 2 && c->adj_R <= 15) {
    VGS_HIP_TRY(c, c->adj_off.ensure((size_t)U * c->adj_stride));
    off = c->adj_off.p;
  }
  c->adj_have_off = (out_key == c->adj_key.p) ? (off != nullptr) : c->adj_have_off;
#define LAUNCH_ADJ(CAPV, FULLV, GRID, REDO, NREDO, REDO_OUT)                                                                 \
  hipLaunchKernelGGL((k_adjacency<CAPV, FULLV>), dim3(GRID), dim3(64), 0, c->stream, c->vox_code.p, row_ids, U,            \
                     (const Brick*)c->hkey.p, c->hbits, c->offsets.p, c->n_off, c->adj_R, c->box.depth, res_f, mnx, mny, mnz, r2, c->node.p, \
                     c->adj_stride, out_key, out_cnt, out_nall, gt, c->adj_gstride, c->adj_ngroups, c->adj_nvals.p, REDO, NREDO, REDO_OUT, off, NREDO)
  if (2 * c->adj_R + 1 > 32) { c->err = "neighbour ball wider than 31 voxels (graph_size / voxel_size > ~12)"; return VGS_E_UNSUPPORTED; }
  if (c->n_off <= 1024 && !full && c->adj_mask_nb > 0 && gt && !c->K.no_adjmasks) {
    // hot path: candidates from the brick occupancy masks; rows it cannot take go through the general kernel
    VGS_HIP_TRY(c, c->work_ids.ensure(2 * (size_t)U + 16)); VGS_HIP_TRY(c, c->counters.ensure(64));
    unsigned int* d_nredo = (unsigned int*)(c->counters.p + 40);
    VGS_HIP_TRY(c, hipMemsetAsync(d_nredo, 0, 8, c->stream));
#define LAUNCH_ADJM(NBV)                                                                                                          \
    hipLaunchKernelGGL((k_adjacency_masks<240, NBV, 5>), dim3(vgs_xcd_grid(U)), dim3(64), 0, c->stream, c->vox_code.p, row_ids, U,  \
                       (const Brick*)c->hkey.p, c->hbits, c->adj_masks.p, c->adj_R, res_f, mnx, mny, mnz, r2, c->adj_stride, out_key,     \
                       out_cnt, out_nall, gt, c->adj_gstride, c->adj_ngroups, c->adj_nvals.p, d_nredo, c->work_ids.p, off)
    if (c->adj_mask_nb == 3) LAUNCH_ADJM(3); else LAUNCH_ADJM(5);
#undef LAUNCH_ADJM
    // rows it passed on (more than 240 survivors: clutter): the same kernel with a 512-entry list over the device-side list of them (round 5:
    // they used to take the general kernel's nine dependent probe trips per row -- 55 us for the launch however few they were); what even
    // that cannot hold (more than 512 survivors, 576 candidates, a row out of band) goes on to the general kernel through a second list
    unsigned int* d_nredo2 = d_nredo + 1;
    uint32_t* list2 = c->work_ids.p + U;
    const unsigned int g2 = (unsigned int)(U < 2048 ? U : 2048);
#define LAUNCH_ADJM2(NBV)                                                                                                         \
    hipLaunchKernelGGL((k_adjacency_masks<512, NBV, 9, true>), dim3(g2), dim3(64), 0, c->stream, c->vox_code.p, row_ids, U,          \
                       (const Brick*)c->hkey.p, c->hbits, c->adj_masks.p, c->adj_R, res_f, mnx, mny, mnz, r2, c->adj_stride, out_key,     \
                       out_cnt, out_nall, gt, c->adj_gstride, c->adj_ngroups, c->adj_nvals.p, d_nredo2, list2, off,                      \
                       (const uint32_t*)c->work_ids.p, (const unsigned int*)d_nredo)
    if (c->adj_mask_nb == 3) LAUNCH_ADJM2(3); else LAUNCH_ADJM2(5);
#undef LAUNCH_ADJM2
    LAUNCH_ADJ(1024, false, g2, list2, d_nredo2, nullptr);
  } else if (c->n_off <= 8192 && !full && c->adj_mask_nb == 7 && gt && !c->K.no_adjmasks) {
    // balls of up to 12 voxels (config 2: ten): candidates from the occupancy masks of 7 x 7 x 7 bricks -- a planar neighbourhood
    // occupies 305 of the 4189 ball cells (up to 832 occupied cells are taken as candidates: the unused voxels count too); rows with more candidates, or more than 512 survivors, go down the general kernel's two passes
    VGS_HIP_TRY(c, c->work_ids.ensure(2 * (size_t)U + 16)); VGS_HIP_TRY(c, c->counters.ensure(64));
    unsigned int* d_n1 = (unsigned int*)(c->counters.p + 40);
    unsigned int* d_n2 = (unsigned int*)(c->counters.p + 41);
    VGS_HIP_TRY(c, hipMemsetAsync(d_n1, 0, 16, c->stream));
    uint32_t* list1 = c->work_ids.p;
    uint32_t* list2 = c->work_ids.p + U;
    hipLaunchKernelGGL((k_adjacency_masks<512, 7, 13>), dim3(vgs_xcd_grid(U)), dim3(64), 0, c->stream, c->vox_code.p, row_ids, U,
                       (const Brick*)c->hkey.p, c->hbits, c->adj_masks.p, c->adj_R, res_f, mnx, mny, mnz, r2, c->adj_stride, out_key,
                       out_cnt, out_nall, gt, c->adj_gstride, c->adj_ngroups, c->adj_nvals.p, d_n1, list1, off);
    const unsigned int g2 = (unsigned int)(U < 4096 ? U : 4096);   // the grids stride over the device-side lists
    hipLaunchKernelGGL((k_adjacency<2048, false>), dim3(g2), dim3(64), 0, c->stream, c->vox_code.p, row_ids, U, (const Brick*)c->hkey.p, c->hbits,
                       c->offsets.p, c->n_off, c->adj_R, c->box.depth, res_f, mnx, mny, mnz, r2, c->node.p, c->adj_stride, out_key, out_cnt, out_nall, gt,
                       c->adj_gstride, c->adj_ngroups, c->adj_nvals.p, list1, d_n1, list2, off, d_n2);
    if (c->K.no_adj_wide)
      hipLaunchKernelGGL((k_adjacency<8192, false>), dim3(g2), dim3(64), 0, c->stream, c->vox_code.p, row_ids, U, (const Brick*)c->hkey.p, c->hbits,
                         c->offsets.p, c->n_off, c->adj_R, c->box.depth, res_f, mnx, mny, mnz, r2, c->node.p, c->adj_stride, out_key, out_cnt, out_nall, gt,
                         c->adj_gstride, c->adj_ngroups, c->adj_nvals.p, list2, d_n2, (uint32_t*)nullptr, off, d_n2);
    else   // a workgroup per row (round 6)
      hipLaunchKernelGGL((k_adjacency_wide<512>), dim3(g2), dim3(512), 0, c->stream, c->vox_code.p, row_ids, (const Brick*)c->hkey.p, c->hbits,
                         c->offsets.p, c->n_off, c->adj_R, c->box.depth, res_f, mnx, mny, mnz, r2, c->adj_stride, out_key, out_cnt, out_nall, gt,
                         c->adj_gstride, c->adj_ngroups, c->adj_nvals.p, (const uint32_t*)list2, (const unsigned int*)d_n2, off);
  } else if (c->n_off <= 1024) {
    if (full) LAUNCH_ADJ(1024, true, vgs_xcd_grid(U), nullptr, nullptr, nullptr); else LAUNCH_ADJ(1024, false, vgs_xcd_grid(U), nullptr, nullptr, nullptr);
  } else if (c->n_off <= 8192) {
    // A list for every lattice offset (72 KB of LDS) leaves two wavefronts per CU; surfaces fill a fraction of the ball,
    // so the first pass runs with 2048 slots (18 KB) and hands rows that overflow to a second pass over a device-side list
    VGS_HIP_TRY(c, c->work_ids.ensure((size_t)U + 16)); VGS_HIP_TRY(c, c->counters.ensure(64));
    unsigned int* d_nredo = (unsigned int*)(c->counters.p + 40);
    VGS_HIP_TRY(c, hipMemsetAsync(d_nredo, 0, 4, c->stream));
    if (full) LAUNCH_ADJ(2048, true, vgs_xcd_grid(U), nullptr, d_nredo, c->work_ids.p); else LAUNCH_ADJ(2048, false, vgs_xcd_grid(U), nullptr, d_nredo, c->work_ids.p);
    const unsigned int g2 = (unsigned int)(U < 4096 ? U : 4096);   // the grid strides over the list
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
  if (V == 0) return VGS_OK;
  float r2 = 0.f;
  vgs_status st = build_hash_and_offsets(c, &r2);   // also without a used voxel: vgs_get_lists builds every voxel's list on request
  if (st != VGS_OK) return st;
  c->adj_r2 = r2;
  if (U == 0) return VGS_OK;
  c->adj_pruned = vgs_unused_are_inert(c->P);
  {
    // The row tables are dense: U rows x (lattice offsets of the search ball) slots -- sorted keys 8 B, lattice offsets 2 B and the
    // connect / mutual flags of the local-cut stage 2 B per slot.  2.6 GB at the 10 M-point config, but U x 4189 x 12 B at
    // a ball of ten voxels: say so BEFORE allocating, in terms the caller can act on, instead of failing inside some hipMalloc.
    // ... plus what the local-cut stage allocates per run whatever the scene: connect bits (cb_words 4-byte words per row), the group
    // table, work lists (16 ids per used voxel), counters per voxel, the near-pair lists (16 entries of 16 B per voxel) and the pair
    // lists' state.  (The pair lists' pool grows with the scene and is not in here; the check is ADVISORY -- free memory is read at
    // this instant and a caching allocator of the same process holds memory it does not show -- and VGS_HIP_TRY's mapping of
    // hipErrorOutOfMemory to VGS_E_NOMEM remains the backstop.)
    const double fixed_rows = (double)U * ((double)(c->cb_words > 0 && c->cb_words <= VGS_CB_MAX_WORDS ? c->cb_words : 0) * 4.0 + (double)c->adj_gstride * 2.0 + 16.0 * 4.0 + 32.0)
                              + (double)V * (16.0 * 16.0 + 8.0 + 10.0);
    const double need = (double)U * (double)c->adj_stride * 12.0 + fixed_rows;
    const double held = (double)c->adj_key.cap * 8.0 + (double)c->adj_off.cap * 2.0 + (double)c->conn.cap + (double)c->conn_bits.cap * 4.0 +
                        (double)c->adj_gtab.cap * 2.0 + (double)c->work_ids.cap * 4.0 + (double)c->nl_ent.cap * 16.0 + (double)c->pl_state.cap;
    size_t free_b = 0, total_b = 0;
    VGS_HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
    if (need > held + (double)free_b * 0.97) {
      char msg[512];
      snprintf(msg, sizeof(msg), "adjacency / connect tables need %.1f GB (%lld used voxels x %d lattice offsets of a search ball of %.1f voxels x 12 B, + per-voxel lists) "
               "but %.1f GB are free on the device (%.1f GB in all): lower graph_size / voxel_size, or split the cloud into spatial tiles (include/vgs_tiles.h)",
               need / 1e9, (long long)U, c->adj_stride, (double)(c->P.graph_size / c->P.voxel_size), ((double)free_b + held) / 1e9, (double)total_b / 1e9);
      c->err = msg;
      return VGS_E_NOMEM;
    }
  }
  VGS_HIP_TRY(c, c->adj_key.ensure((size_t)U * c->adj_stride));
  VGS_HIP_TRY(c, c->adj_cnt.ensure(U)); VGS_HIP_TRY(c, c->adj_mused.ensure(U));
  st = vgs_run_adjacency(c, !c->adj_pruned, c->adj_key.p, c->adj_cnt.p, c->adj_mused.p, r2);
  if (st != VGS_OK) return st;
  return VGS_OK;
}
