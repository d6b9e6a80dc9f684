// merge.hip -- stages a8-a11: mutual-connection filter, isolated-voxel re-attachment, connected
// components, cluster size filter and per-point labels.
// Replaces crossValidation (voxel_segmentation.h:2111-2179), closestCheck (VS:2181-2303),
// clusteringVoxels + recursionSearch (VS:2032-2099) and the cluster filter / point lists of
// drawColorMapofPointsinClusters (VS:963-1009); SVGS twins SS:2142-2305, 2057-2130.
//
// The reference's sequential control flow becomes data-parallel:
//   * crossValidation is the symmetric intersection of the connect lists, one lane per
//     adjacency slot, the reverse slot found by binary search on the (d2, id) sort key (d2 is symmetric);
//   * closestCheck mutates lists while scanning voxels in order; its only order dependence is that a
//     re-attached voxel becomes an eligible target for LATER voxels.  "re-attachment succeeds" is therefore a
//     monotone fixed point over ascending ids, iterated in parallel; each voxel then picks its target among
//     the voxels eligible at its turn (ties: last in scan order, `>=` at VS:2281);
//   * the recursive DFS becomes a lock-free union-find (hook larger root under smaller) over the mutual and
//     re-attachment edges; the root of a component is its smallest voxel id, which is also the order in
//     which the reference discovers clusters (VS:2064).
#include <cstring>
#include <string.h>
#include <utility>

#include <rocprim/rocprim.hpp>

#include "vgs_context.hpp"

struct MgParams {
  VgsWeightParams W;
  int adjacency_min;
  int q7;
  int64_t V;
};

// ------------------------------------------------------------------ crossValidation
// CX_ROWS rows per wavefront, 64 / CX_ROWS lanes each (see k_union_mutual: a row of about 120 entries leaves a whole
// wavefront waiting on its chain of dependent loads; several rows per wavefront keep more of them in flight).
#ifndef CX_ROWS
#define CX_ROWS 2
#endif
__global__ __launch_bounds__(64) void k_cross(const uint32_t* __restrict__ used_ids, const uint32_t* __restrict__ used_rank, int64_t U,
                                              const uint64_t* __restrict__ adj_key, const uint32_t* __restrict__ adj_cnt,
                                              int adj_stride, const uint8_t* __restrict__ conn, uint8_t* __restrict__ mutual,
                                              uint32_t* __restrict__ csize, uint32_t* __restrict__ parent,
                                              const uint16_t* __restrict__ gtab, int gstride, const uint8_t* __restrict__ nrank, float inv_res2,
                                              const uint8_t* __restrict__ pending, uint32_t* __restrict__ defer_list,
                                              unsigned int* __restrict__ n_defer, const uint32_t* __restrict__ work, int n_work,
                                              const uint16_t* __restrict__ adj_off, int cb_R,
                                              const uint32_t* __restrict__ cbits, int cb_words, uint8_t* __restrict__ defer_flag, LcGate gate,
                                              const uint8_t* __restrict__ owned = nullptr) {
  if (!lc_gate_open_early(gate)) return;
  // First pass (pending != null): all rows, while the hand-over kernels of the local cut still run -- a row whose voxel, or
  // one of whose connected neighbours, is handed over is put off (its flags or theirs are not final).  Second pass
  // (work != null): the rows put off.
  constexpr int W = 64 / CX_ROWS;
  const int lane = threadIdx.x, grp = lane / W, sub = lane % W;
  // rows of the voxel lattice carry a table of where each group of equal offset length starts: the reverse entry has
  // the same length, so only that group (a handful of entries) is searched instead of the whole row
  __shared__ uint32_t s_rank4[64];   // length -> group index (256 bytes)
  if (gtab) s_rank4[lane] = ((const uint32_t*)nrank)[lane];
  __syncthreads();
  const uint8_t* s_rank = (const uint8_t*)s_rank4;
  int64_t u;
  if (work) {
    const int w = (int)blockIdx.x * CX_ROWS + grp;
    if (w >= n_work) return;
    u = (int64_t)work[w];
  } else {
    const int64_t ngroups = (U + CX_ROWS - 1) / CX_ROWS;
    const int64_t g = vgs_xcd_item(blockIdx.x, ngroups);   // consecutive rows share a wavefront, consecutive groups an XCD
    u = g * CX_ROWS + grp;
    if (g >= ngroups || u >= U) return;
  }
  if (pending && pending[u]) { if (sub == 0) { defer_list[atomicAdd(n_defer, 1u)] = (uint32_t)u; if (defer_flag) defer_flag[u] = 1; } return; }
  bool touches_pending = false;
  const uint32_t i = used_ids[u];
  const int n = (int)adj_cnt[u];
  const uint64_t* row = adj_key + u * adj_stride;
  const uint8_t* crow = conn + u * adj_stride;
  uint8_t* mrow = mutual + u * adj_stride;
  // a list of length <= 1 is left alone (VS:2120): it is {self}
  int len = 0;
  for (int k = sub; k < n; k += W) len += crow[k] ? 1 : 0;
  for (int o = W / 2; o > 0; o >>= 1) len += __shfl_xor(len, o, 64);   // xor with o < W stays inside the row's lanes
  const bool own_tab = gtab && gtab[u * gstride] != 0xffffu;
  // lattice lookup (round 4): the neighbour t sits at ball offset o from i, so "i in L0(t)" is the bit of the NEGATED offset in t's row
  // of connect bits -- two dependent loads (t's row index, the word) instead of the chain rank -> group table -> binary search over
  // 8-byte keys -> flag.  A row without lattice offsets (its own, or t's: bit 0 clear) takes the search below.
  const uint16_t* orow = (cbits != nullptr && adj_off != nullptr) ? adj_off + u * adj_stride : nullptr;
  const bool own_bits = orow != nullptr && orow[0] != 0xffffu;
  int kept = 0;
  uint32_t best = i;  // first hook of the union-find (see k_cc_init): smallest mutual neighbour below i
  for (int k = sub; k < n; k += W) {
    uint8_t mflag = 0;
    if (crow[k]) {
      const uint64_t key = row[k];
      const uint32_t t = (uint32_t)key;
      if (len <= 1 || t == i) {
        mflag = 1;
      } else {
        const uint32_t ut = used_rank[t];
        bool by_bits = false;
        if (ut != 0xffffffffu && own_bits) {
          const int D = 2 * cb_R + 1;
          const uint32_t centre = (uint32_t)((cb_R * D + cb_R) * D + cb_R);
          const uint32_t idx = (uint32_t)(D * D * D - 1) - vgs_cb_index(orow[k], cb_R);   // the negated offset: point symmetry of the cube
          const uint32_t* trow_bits = cbits + (size_t)ut * (size_t)cb_words;
          const uint32_t wc = trow_bits[centre >> 5];
          if ((wc >> (centre & 31u)) & 1u) {   // t's row has bits (its own voxel is always a member)
            by_bits = true;
            if (pending && pending[ut]) touches_pending = true;
            const uint32_t wd = (idx >> 5) == (centre >> 5) ? wc : trow_bits[idx >> 5];
            mflag = (uint8_t)((wd >> (idx & 31u)) & 1u);
          }
        }
        if (ut != 0xffffffffu && !by_bits) {
          if (pending && pending[ut]) touches_pending = true;
          const uint64_t want = (key & 0xffffffff00000000ull) | (uint64_t)i;
          const uint64_t* trow = adj_key + (int64_t)ut * adj_stride;
          int lo = 0, hi = -1, found = -1;
          bool ranged = false;
          if (own_tab) {
            const int len2 = (int)(vm_from_bits((uint32_t)(key >> 32)) * inv_res2 + 0.5f);  // integer offset length (exact: the row is in band)
            const int r = len2 < 256 ? (int)s_rank[len2] : 255;
            if (r != 255) {
              const uint16_t* gt = gtab + (int64_t)ut * gstride;
              const uint32_t g0 = gt[r], g1 = gt[r + 1];
              if (g0 != 0xffffu) { lo = (int)g0; hi = (int)g1 - 1; ranged = true; }   // 0xffff: t's row has no table
            } else {
              ranged = true;  // a length that no offset has: not in any row
            }
          }
          if (!ranged) { lo = 0; hi = (int)adj_cnt[ut] - 1; }
          while (lo <= hi) {
            const int mid = (lo + hi) >> 1;
            const uint64_t kk = trow[mid];
            if (kk == want) { found = mid; break; }
            if (kk < want) lo = mid + 1; else hi = mid - 1;
          }
          if (found >= 0 && conn[(int64_t)ut * adj_stride + found]) mflag = 1;
        }
      }
      if (mflag && t < best && (!owned || owned[i] || owned[t])) best = t;   // (tiled runs: a connection is trusted only if one endpoint is owned)
    }
    mrow[k] = mflag;
    kept += mflag;
  }
  if (pending) {   // what was written to a row that is put off is overwritten by the second pass
    const unsigned long long mine = (W == 64) ? ~0ull : (((1ull << W) - 1ull) << (grp * W));
    if ((__ballot(touches_pending) & mine) != 0ull) {
      if (sub == 0) { defer_list[atomicAdd(n_defer, 1u)] = (uint32_t)u; if (defer_flag) defer_flag[u] = 1; }
      return;
    }
    if (defer_flag && sub == 0) defer_flag[u] = 0;   // every row of the first pass writes its flag: nobody zeroes the array
  }
  for (int o = W / 2; o > 0; o >>= 1) kept += __shfl_xor(kept, o, 64);
  if (sub == 0) csize[i] = (uint32_t)kept;
  if (parent) {  // single-tile runs: the hook needs no ownership test, so it is taken here instead of re-reading the row
    for (int o = W / 2; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)best, o, 64); best = other < best ? other : best; }
    if (sub == 0) parent[i] = best;
  }
}

// ------------------------------------------------------------------ closestCheck
// flags: bit0 candidate (list == {self} and adjacency vector longer than adjacency_min), bit1 success
__global__ void k_cc_candidates(const uint32_t* __restrict__ used_ids, int64_t U, const uint32_t* __restrict__ adj_nall,
                                const uint32_t* __restrict__ csize, int adjacency_min, uint8_t* __restrict__ flags,
                                uint32_t* __restrict__ cand_list, unsigned int* __restrict__ n_cand) {
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  const uint32_t i = used_ids[u];
  // voxels_adjacency_idx_[i] = [count, idx...]: its size is n + 1 (VS:2201)
  const bool cand = (csize[i] == 1u) && ((int)adj_nall[u] + 1 > adjacency_min);
  if (cand) { flags[i] = 1; cand_list[atomicAdd(n_cand, 1u)] = (uint32_t)u; }
}

// scan entry j of voxel i's raw adjacency vector: j == 0 is the neighbour COUNT read as a voxel id (Q7, VS:2243)
__device__ __forceinline__ int cc_entry(int j, int n_all, const uint64_t* row, const MgParams& P) {
  if (j == 0) {
    if (!P.q7 || (int64_t)n_all >= P.V) return -1;
    return n_all;
  }
  return (int)(uint32_t)row[j - 1];
}

template <bool CHOOSE>
__global__ __launch_bounds__(64) void k_cc_pass(const uint32_t* __restrict__ cand_list, int n_cand, const uint32_t* __restrict__ used_ids,
                                                const uint64_t* __restrict__ adj_key, const uint32_t* __restrict__ adj_cnt,
                                                const uint32_t* __restrict__ adj_nall, int adj_stride,
                                                const NodeRec* __restrict__ node, const uint32_t* __restrict__ csize,
                                                uint8_t* __restrict__ flags, MgParams P, int32_t* __restrict__ attach,
                                                unsigned int* __restrict__ changed) {
  if ((int)blockIdx.x >= n_cand) return;
  const int lane = threadIdx.x;
  const uint32_t u = cand_list[blockIdx.x];
  const uint32_t i = used_ids[u];
  if (!CHOOSE && (flags[i] & 2)) return;
  if (CHOOSE && !(flags[i] & 2)) { if (lane == 0) attach[i] = -1; return; }
  // the row may hold the used neighbours only: an unused voxel is never an eligible target (its list is empty),
  // so skipping it changes neither the choice nor the "last equal weight wins" order
  const int n = (int)adj_cnt[u];
  const int n_all = (int)adj_nall[u];
  const uint64_t* row = adj_key + (int64_t)u * adj_stride;
  const NodeRec me = node[i];
  float best_w = -1.0f;
  int best_j = -1, best_t = -1;
  bool any = false;
  for (int j = lane; j < n + 1; j += 64) {
    const int t = cc_entry(j, n_all, row, P);
    if (t < 0) continue;
    // eligible at voxel i's turn: list longer than one after crossValidation, or an earlier candidate that succeeded
    const bool elig = (csize[t] > 1u) || ((uint32_t)t < i && (flags[t] & 3) == 3);
    if (!elig) continue;
    const float w = vm_pair_weight(me, node[t], P.W);  // distanceProbability == distanceWeight (Q8)
    if (w != w) continue;                               // NaN >= x is false (VS:2281)
    any = true;
    if (CHOOSE && w >= best_w) { best_w = w; best_j = j; best_t = t; }  // ascending j per lane: later equal weights win
  }
  if (!CHOOSE) {
    if (__ballot(any) != 0ull && lane == 0) { flags[i] |= 2; atomicOr(changed, 1u); }
  } else {
    // wave arg-max on (w, j)
    for (int o = 32; o > 0; o >>= 1) {
      const float ow = __shfl_xor(best_w, o, 64);
      const int oj = __shfl_xor(best_j, o, 64);
      const int ot = __shfl_xor(best_t, o, 64);
      if (ow > best_w || (ow == best_w && oj > best_j)) { best_w = ow; best_j = oj; best_t = ot; }
    }
    if (lane == 0) { attach[i] = best_t; if (best_t >= 0) atomicAdd(changed, 1u); }  // `changed` counts re-attachments in this pass
  }
}

// ------------------------------------------------------------------ connected components
__device__ __forceinline__ uint32_t uf_find(uint32_t* parent, uint32_t x) {
  // path halving: parents only ever move towards the root (smaller ids), so a stale write is still an ancestor
  while (true) {
    const uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == x) return x;
    const uint32_t g = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g != p) __hip_atomic_store(&parent[x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x = g;
  }
}
__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
    if (atomicCAS(&parent[hi], hi, lo) == hi) return;
  }
}

// one pass instead of five fills: parent = identity, the per-voxel tables of the stage at their start values
__global__ void k_merge_init(uint32_t* __restrict__ parent, uint32_t* __restrict__ csize, int32_t* __restrict__ attach,
                             uint8_t* __restrict__ cc_flags, uint32_t* __restrict__ csz, int64_t n) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) { parent[v] = (uint32_t)v; csize[v] = 0u; attach[v] = -1; cc_flags[v] = 0; csz[v] = 0u; }
}

// first hook (ECL-CC style): every used voxel points at its smallest trusted neighbour with a smaller id.  Ids only
// decrease along parent links, so this is a forest; it already joins most of every large segment without atomics.
__global__ __launch_bounds__(64) void k_cc_init(const uint32_t* __restrict__ used_ids, int64_t U, const uint64_t* __restrict__ adj_key,
                                                const uint32_t* __restrict__ adj_cnt, int adj_stride, const uint8_t* __restrict__ mutual,
                                                const int32_t* __restrict__ attach, const uint8_t* __restrict__ owned,
                                                uint32_t* __restrict__ parent) {
  const int64_t u = vgs_xcd_item(blockIdx.x, U);
  if (u >= U) return;
  const uint32_t i = used_ids[u];
  const int n = (int)adj_cnt[u];
  const uint64_t* row = adj_key + u * adj_stride;
  const uint8_t* mrow = mutual + u * adj_stride;
  uint32_t best = i;
  for (int k = threadIdx.x; k < n; k += 64) {
    if (!mrow[k]) continue;
    const uint32_t t = (uint32_t)row[k];
    if (t < best && (!owned || owned[i] || owned[t])) best = t;
  }
  if (threadIdx.x == 0) {
    const int32_t t = attach[i];
    if (t >= 0 && (uint32_t)t < best && (!owned || owned[i])) best = (uint32_t)t;
  }
  for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)best, o, 64); best = other < best ? other : best; }
  if (threadIdx.x == 0) parent[i] = best;
}

// UM_ROWS rows per wavefront, 64 / UM_ROWS lanes each: a row holds about 120 entries, so a whole wavefront per row spends
// its time waiting for three dependent loads (row header -> keys and flags -> parents); sharing the wavefront keeps several
// rows' loads in flight at the same occupancy.
#ifndef UM_ROWS
#define UM_ROWS 4
#endif
__global__ __launch_bounds__(64) void k_union_mutual(const uint32_t* __restrict__ used_ids, int64_t U, const uint64_t* __restrict__ adj_key,
                                                     const uint32_t* __restrict__ adj_cnt, int adj_stride,
                                                     const uint8_t* __restrict__ mutual, const int32_t* __restrict__ attach,
                                                     const uint8_t* __restrict__ owned, uint32_t* __restrict__ parent, int do_attach,
                                                     const uint8_t* __restrict__ skip, const uint32_t* __restrict__ work, int n_work, LcGate gate) {
  if (!lc_gate_open_early(gate)) return;
  constexpr int W = 64 / UM_ROWS;
  // plain order: neighbouring voxels on one XCD at the same time contend for the same roots (measured slower)
  int64_t u = (int64_t)blockIdx.x * UM_ROWS + (threadIdx.x / W);
  const int sub = threadIdx.x % W;
  if (work) { if (u >= n_work) return; u = (int64_t)work[u]; }   // the rows of a list (the second pass of crossValidation)
  if (u >= U) return;
  if (skip && skip[u]) return;                                     // rows whose mutual flags are not final yet
  const uint32_t i = used_ids[u];
  const int n = (int)adj_cnt[u];
  const uint64_t* row = adj_key + u * adj_stride;
  const uint8_t* mrow = mutual + u * adj_stride;
  // after the first hook and the pointer jumping most neighbours already hang under the same node: equal parents mean
  // one tree, whatever other wavefronts do meanwhile (links are only ever added), and cost one load instead of two finds
  const uint32_t pi = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int k = sub; k < n; k += W) {
    if (!mrow[k]) continue;
    const uint32_t t = (uint32_t)row[k];
    // tiled runs: a connection is trusted only if one endpoint is owned (both neighbourhoods are then complete)
    if (t > i && (!owned || owned[i] || owned[t])) {  // each mutual edge appears in both rows: union once
      if (__hip_atomic_load(&parent[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pi) continue;
      uf_union(parent, i, t);
    }
  }
  if (sub == 0 && do_attach) {
    const int32_t t = attach[i];
    if (t >= 0 && (!owned || owned[i])) uf_union(parent, i, (uint32_t)t);
  }
}

// the re-attachment edges on their own (closestCheck's candidates only): used when the mutual edges were united while the host
// was still fetching closestCheck's fixed-point flag
__global__ void k_union_attach(const uint32_t* __restrict__ cand, int n_cand, const uint32_t* __restrict__ used_ids, const int32_t* __restrict__ attach,
                               uint32_t* __restrict__ parent, const uint8_t* __restrict__ owned) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_cand) return;
  const uint32_t i = used_ids[cand[j]];
  const int32_t t = attach[i];
  if (t >= 0 && (!owned || owned[i])) uf_union(parent, i, (uint32_t)t);   // (tiled runs: the owner of the isolated voxel decides its re-attachment)
}

// pointer jumping between the first hook and the union pass: every later find starts one hop from a root
__global__ void k_compress(uint32_t* __restrict__ parent, int64_t V) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const uint32_t r = uf_find(parent, (uint32_t)v);
  if (r != (uint32_t)v) __hip_atomic_store(&parent[v], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The unions are complete: every voxel gets its ROOT as parent.  The walk must not write on the way (no path halving
// here): a halving store of another thread that lands after this voxel's final store would leave it pointing at an
// ancestor below the root, and the labels read parent[v] as the root (seen once in ~20 runs of a 90 k-point scene as a
// voxel dropped from its segment).  With read-only walks the only writes are the final ones, each a root.
__device__ __forceinline__ uint32_t uf_root(const uint32_t* parent, uint32_t x) {
  while (true) {
    const uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == x) return x;
    x = p;
  }
}
__device__ __forceinline__ void k_flatten_body(uint32_t* __restrict__ parent, int64_t V, const uint8_t* __restrict__ owned, uint32_t* __restrict__ csz,
                                               uint32_t* s_root, unsigned int* s_cnt);
__global__ void k_flatten(uint32_t* __restrict__ parent, int64_t V, const uint8_t* __restrict__ owned, uint32_t* __restrict__ csz) {
  __shared__ uint32_t s_root[32];
  __shared__ unsigned int s_cnt[32];
  if (threadIdx.x < 32) { s_root[threadIdx.x] = 0xffffffffu; s_cnt[threadIdx.x] = 0u; }
  __syncthreads();
  k_flatten_body(parent, V, owned, csz, s_root, s_cnt);
  __syncthreads();
  if (threadIdx.x < 32 && s_cnt[threadIdx.x]) atomicAdd(&csz[s_root[threadIdx.x]], s_cnt[threadIdx.x]);
}
__device__ __forceinline__ void k_flatten_body(uint32_t* __restrict__ parent, int64_t V, const uint8_t* __restrict__ owned, uint32_t* __restrict__ csz,
                                               uint32_t* s_root, unsigned int* s_cnt) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const uint32_t r = uf_root(parent, (uint32_t)v);
  __hip_atomic_store(&parent[v], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // one atomic per distinct root per wavefront (large segments would otherwise serialise on one address).  A voxel that is its own
  // root counts itself on its own: half of a scanned scene's voxels are unused singletons, 64 distinct roots per wavefront, and
  // walked the loop below 64 times (round 5)
  const bool counted = !owned || owned[v];   // tiled runs count owned voxels only
  const bool self = r == (uint32_t)v;
  if (counted && self) atomicAdd(&csz[r], 1u);
  unsigned long long todo = __ballot(counted && !self);
  if (!counted || self) return;
  const int lane = threadIdx.x & 63;
  while (todo) {
    const int l0 = __ffsll((long long)todo) - 1;
    const uint32_t r0 = (uint32_t)__shfl((int)r, l0, 64);
    const unsigned long long same = __ballot(r == r0) & todo;
    if (lane == l0) {
      // (round 6) through a small table of the workgroup first: the ground of a scanned scene is one root for a fifth of the voxels, and
      // one global atomic per wavefront on that one address was most of this kernel's 100 us
      const unsigned int n = (unsigned int)__popcll(same);
      unsigned int h = (r0 * 2654435761u) >> 27;   // 32 slots
      bool put = false;
      for (int probe = 0; probe < 4 && !put; ++probe, h = (h + 1u) & 31u) {
        const uint32_t prev = atomicCAS(&s_root[h], 0xffffffffu, r0);
        if (prev == 0xffffffffu || prev == r0) { atomicAdd(&s_cnt[h], n); put = true; }
      }
      if (!put) atomicAdd(&csz[r0], n);
    }
    todo &= ~same;
    if ((same >> lane) & 1ull) break;
  }
}

__global__ __launch_bounds__(1024) void k_root_flags(const uint32_t* __restrict__ parent, const uint32_t* __restrict__ csz, int64_t V, int voxels_min,
                                                     uint32_t* __restrict__ keep_flag, unsigned int* __restrict__ n_roots) {
  __shared__ unsigned int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool root = false;
  if (v < V) {
    root = parent[v] == (uint32_t)v;
    keep_flag[v] = (root && (int)csz[v] > voxels_min) ? 1u : 0u;  // clusters_voxel_idx_[m].size() > cluster_voxels_min_ (VS:969)
  }
  // one global atomic per 1024 voxels (same-address atomics serialise: one per wavefront cost 0.15 ms at 10^6 voxels)
  const unsigned long long m = __ballot(root);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cnt, (unsigned int)__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) atomicAdd(n_roots, s_cnt);
}

__global__ void k_voxel_labels(const uint32_t* __restrict__ parent, const uint32_t* __restrict__ keep_flag,
                               const uint32_t* __restrict__ keep_rank, int64_t V, int32_t* __restrict__ vox_label,
                               unsigned int* __restrict__ n_kept) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const uint32_t r = parent[v];
  vox_label[v] = keep_flag[r] ? (int32_t)keep_rank[r] : -1;
  if (v == V - 1) *n_kept = keep_rank[v] + keep_flag[v];   // number of kept segments, for the stage's one read-back
}

__global__ void k_point_labels(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ pt_vox, const int32_t* __restrict__ vox_label,
                               int64_t N, int32_t* __restrict__ label) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const uint32_t v = pt_vox[j];
  label[perm[j]] = (v == 0xffffffffu) ? -1 : vox_label[v];
}

// the per-point labels of a context whose merge stage left them to be asked for (a tile between vgs_segment and the label hand-back)
vgs_status vgs_ensure_point_labels(vgs_ctx* c) {
  if (!c->pt_labels_pending) return VGS_OK;
  c->pt_labels_pending = false;
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  if (c->d2h_open && c->d2h_src == c->pt_label.p) { VGS_HIP_TRY(c, hipEventSynchronize(c->ev_d2h)); c->d2h_open = false; }
  const int TB = 256;
  hipLaunchKernelGGL(k_point_labels, dim3((unsigned)((c->N + TB - 1) / TB)), dim3(TB), 0, c->stream, c->perm_b.p, c->pt_vox.p, c->vox_label.p,
                     c->N, c->pt_label.p);
  VGS_HIP_TRY(c, hipEventRecord(c->ev[15], c->stream));
  c->labels_event_valid = true;
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VGS_OK;
}

static VgsWeightParams make_weight_params_m(const vgs_params& p) {
  VgsWeightParams W;
  W.inv_sig_p = 1.0f / p.sig_p; W.inv_sig_n = 1.0f / p.sig_n; W.inv_sig_o = 1.0f / p.sig_o;
  W.inv_sig_e = 1.0f / p.sig_e; W.inv_sig_c = 1.0f / p.sig_c;
  W.inv_sig_w2 = 1.0f / (p.sig_w * p.sig_w);
  W.svgs = (p.method == 3) ? 1 : 0;
  return W;
}

// ------------------------------------------------------------------ diagnostics: the affinity matrix of one local graph
// W[a * n + b] = distanceWeight(measuringDistance(row[a], row[b])) with row[a] as the FIRST argument (VS:1796-1910): the
// n x n matrix buildAdjacencyGraph fills for node i, over the entries of the stored adjacency row.
__global__ void k_local_weights(const uint64_t* __restrict__ row, int n, const NodeRec* __restrict__ node, VgsWeightParams W,
                                int32_t* __restrict__ ids, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) ids[t] = (int32_t)(uint32_t)row[t];
  if (t >= (int64_t)n * n) return;
  const int a = (int)(t / n), b = (int)(t % n);
  out[t] = vm_pair_weight(node[(uint32_t)row[a]], node[(uint32_t)row[b]], W);
}

extern "C" vgs_status vgs_get_local_weights(vgs_ctx* c, int32_t node_id, int32_t* n_out, int32_t* ids, float* weights) {
  if (!c || !n_out) return VGS_E_ARG;
  if (c->stage < ST_ADJACENCY) { c->err = "vgs_get_local_weights: adjacency first"; return VGS_E_STATE; }
  if (node_id < 0 || node_id >= c->V) { c->err = "vgs_get_local_weights: node id out of range"; return VGS_E_ARG; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  uint32_t u = 0xffffffffu;
  VGS_HIP_TRY(c, hipMemcpy(&u, c->used_rank.p + node_id, 4, hipMemcpyDeviceToHost));
  if (u == 0xffffffffu) { *n_out = 0; return VGS_OK; }   // an unused voxel has no local graph (VS:384)
  uint32_t n = 0;
  VGS_HIP_TRY(c, hipMemcpy(&n, c->adj_cnt.p + u, 4, hipMemcpyDeviceToHost));
  *n_out = (int32_t)n;
  if (!ids || !weights || n == 0) return VGS_OK;
  DevBuf<float> d_w; DevBuf<int32_t> d_ids;
  VGS_HIP_TRY(c, d_w.ensure((size_t)n * n)); VGS_HIP_TRY(c, d_ids.ensure(n));
  const int64_t total = (int64_t)n * n;
  hipLaunchKernelGGL(k_local_weights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->adj_key.p + (size_t)u * c->adj_stride, (int)n,
                     c->node.p, make_weight_params_m(c->P), d_ids.p, d_w.p);
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(weights, d_w.p, (size_t)total * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(ids, d_ids.p, (size_t)n * 4, hipMemcpyDeviceToHost);
  d_w.release(); d_ids.release();
  if (e != hipSuccess) { c->err = std::string("vgs_get_local_weights: ") + hipGetErrorString(e); return VGS_E_HIP; }
  return VGS_OK;
}

vgs_status vgs_stage_merge(vgs_ctx* c) {
  c->cl_valid = false;   // (clusters.hip: the cluster lists on the device belong to the labels of the last run)
  const int64_t V = c->V, U = c->U, N = c->N;
  c->bnd_unique = -1;  // tile protocol results belong to the previous segmentation
  c->counts[VGS_N_CLUSTERS] = 0; c->counts[VGS_N_KEPT] = 0; c->counts[VGS_N_ISOLATED] = 0; c->counts[VGS_N_REATTACHED] = 0;
  // an asynchronous download (vgs_get_point_labels_async) may still read the last labels: this run writes the other buffer
  if (c->d2h_open && c->d2h_src == c->pt_label.p) std::swap(c->pt_label, c->pt_label_alt);
  VGS_HIP_TRY(c, c->pt_label.ensure(N > 0 ? N : 1));
  if (V == 0) {
    if (N > 0) VGS_HIP_TRY(c, hipMemsetAsync(c->pt_label.p, 0xff, N * sizeof(int32_t), c->stream));
    return VGS_OK;
  }
  MgParams MP;
  MP.W = make_weight_params_m(c->P);
  MP.adjacency_min = c->P.adjacency_min;
  MP.q7 = c->P.q7_count_as_index;
  MP.V = V;
  const int TB = 256;
  const unsigned nbV = (unsigned)((V + TB - 1) / TB);
  VGS_HIP_TRY(c, c->csize.ensure(V)); VGS_HIP_TRY(c, c->attach.ensure(V)); VGS_HIP_TRY(c, c->cc_flags.ensure(V));
  VGS_HIP_TRY(c, c->parent.ensure(V)); VGS_HIP_TRY(c, c->csz.ensure(V)); VGS_HIP_TRY(c, c->kept_rank.ensure(V + 1));
  VGS_HIP_TRY(c, c->vox_label.ensure(V));
  // this stage's counters live in words 48-55: words 0-13 still belong to the local cut, whose hand-over kernels may be running
  VGS_HIP_TRY(c, c->counters.ensure(64));
  uint64_t* mcnt = c->counters.p + 48;
  VGS_HIP_TRY(c, hipMemsetAsync(mcnt, 0, 8 * sizeof(uint64_t), c->stream));
  hipLaunchKernelGGL(k_merge_init, dim3(nbV), dim3(TB), 0, c->stream, c->parent.p, c->csize.p, c->attach.p, c->cc_flags.p, c->csz.p, V);
  uint8_t* mutual = nullptr;
  unsigned int n_cand = 0, n_succ = 0;
  bool compressed = false, united = false;   // k_compress / the mutual unions already ran behind a read-back of closestCheck
  if (U > 0) {
    if (c->conn.cap < 2 * (size_t)U * c->adj_stride) { c->err = "connect buffer missing (local cut stage not run)"; return VGS_E_STATE; }
    mutual = c->conn.p + (size_t)U * c->adj_stride;  // second half holds the mutual flags
    const uint16_t* gt = (c->P.method == 2 && c->adj_have_gtab) ? c->adj_gtab.p : nullptr;
    const float inv_res2 = 1.0f / (c->P.voxel_size * c->P.voxel_size);
    // Tiled runs (round 5): which voxels this rank owns depends on the voxel lattice, the region and the points' sources only, so it is
    // known NOW, not behind closestCheck -- the first hook is taken in k_cross under the ownership test and the unions of the final rows run
    // beside the hand-over kernels as in a single-context run (the native driver's merge stage was 0.35 ms longer than the plain one's).
    const bool tile_early = c->have_region && !c->K.no_tile_early;
    if (tile_early) { vgs_status so = vgs_compute_owned(c); if (so != VGS_OK) return so; }
    const uint8_t* owned_early = tile_early ? c->owned.p : (const uint8_t*)nullptr;
    uint32_t* cross_parent = (c->have_region && !tile_early) ? nullptr : c->parent.p;
    unsigned int* d_ndefer = (unsigned int*)(c->counters.p + 13);   // zeroed with the local cut's counters
    // runs whose hand-over kernels are still running: the unions of the final rows go beside them (see below)
    const bool early_union = (!c->have_region || tile_early) && c->lc_tail.open && !c->K.no_overlap && !c->K.no_early_union;
    if (early_union) VGS_HIP_TRY(c, c->lc_defer_flag.ensure((size_t)U));
    // connect bits of the cuts (method 2, rows with lattice offsets): the lattice lookup of k_cross
    const bool use_bits = c->cb_enabled && c->P.method == 2 && c->adj_have_off;
    const uint16_t* cb_off = use_bits ? c->adj_off.p : (const uint16_t*)nullptr;
    const int cb_lut = c->cb_R;
    const uint32_t* cb_bits = use_bits ? c->conn_bits.p : (const uint32_t*)nullptr;
    // crossValidation starts while the hand-over kernels of the local cut still run (vgs_stage_localcut): rows that touch a
    // handed-over voxel are put off, ...
    // (When the hand-overs are MANY -- LcGate's word, written on the device by k_ho_lists -- this first pass would put off every row, one
    // atomic each, and the unions behind it would find nothing final: both return at once, and the plain pass over all rows follows
    // below, behind the hand-over kernels.)
    const LcGate g_first = {(c->lc_tail.open && c->lc_tail.gated) ? (const unsigned int*)(c->counters.p + 57) : (const unsigned int*)nullptr, LC_FEW};
    const size_t first_lds = (c->lc_tail.open && c->K.cross_lds_kb > 0) ? (size_t)c->K.cross_lds_kb * 1024 : 0;
    hipLaunchKernelGGL(k_cross, dim3(vgs_xcd_grid((U + CX_ROWS - 1) / CX_ROWS)), dim3(64), first_lds, c->stream, c->used_ids.p, c->used_rank.p, U, c->adj_key.p, c->adj_cnt.p,
                       c->adj_stride, c->conn.p, mutual, c->csize.p, cross_parent, gt, c->adj_gstride, c->adj_nrank.p, inv_res2,
                       c->lc_tail.open ? c->lc_pending.p : (const uint8_t*)nullptr, c->lc_defer.p, d_ndefer, (const uint32_t*)nullptr, 0,
                       cb_off, cb_lut, cb_bits, c->cb_words, early_union ? c->lc_defer_flag.p : (uint8_t*)nullptr, g_first, owned_early);
    if (early_union) {
      // The unions of the rows that are final go here, beside the hand-over kernels of the local cut (which leave most of the GPU
      // idle and end the critical path of the stage): pointer jumping over the first hooks, then every mutual edge of a row that
      // was not put off.  The rows put off follow behind their second crossValidation pass, without a first hook (their parents
      // may have moved by then).
      hipLaunchKernelGGL(k_compress, dim3(nbV), dim3(TB), 0, c->stream, c->parent.p, V);
      hipLaunchKernelGGL(k_union_mutual, dim3((unsigned)((U + UM_ROWS - 1) / UM_ROWS)), dim3(64), first_lds, c->stream, c->used_ids.p, U, c->adj_key.p, c->adj_cnt.p,
                         c->adj_stride, mutual, c->attach.p, owned_early, c->parent.p, 0, c->lc_defer_flag.p, (const uint32_t*)nullptr, 0, g_first);
      compressed = true; united = true;
    }
    // ... then the local cut is completed (its flags and list lengths read back) and the rows put off follow
    unsigned int n_defer = 0;
    {
      vgs_status sf = vgs_localcut_finish(c, &n_defer);
      if (sf != VGS_OK) return sf;
    }
    if (c->lc_tail.many) {
      // the first pass and its unions did not run (see above) -- or ran in part, for workgroups that started before the word was written:
      // whatever they left (hooks, unions of rows whose flags nobody wrote this run) goes, and crossValidation takes every row now
      hipLaunchKernelGGL(k_merge_init, dim3(nbV), dim3(TB), 0, c->stream, c->parent.p, c->csize.p, c->attach.p, c->cc_flags.p, c->csz.p, V);
      hipLaunchKernelGGL(k_cross, dim3(vgs_xcd_grid((U + CX_ROWS - 1) / CX_ROWS)), dim3(64), 0, c->stream, c->used_ids.p, c->used_rank.p, U, c->adj_key.p, c->adj_cnt.p,
                         c->adj_stride, c->conn.p, mutual, c->csize.p, cross_parent, gt, c->adj_gstride, c->adj_nrank.p, inv_res2,
                         (const uint8_t*)nullptr, c->lc_defer.p, d_ndefer, (const uint32_t*)nullptr, 0, cb_off, cb_lut, cb_bits, c->cb_words, (uint8_t*)nullptr,
                         LcGate{nullptr, 0u}, owned_early);
      compressed = false; united = false;
      n_defer = 0;
    }
    if (n_defer > 0) {
      hipLaunchKernelGGL(k_cross, dim3((n_defer + CX_ROWS - 1) / CX_ROWS), dim3(64), 0, c->stream, c->used_ids.p, c->used_rank.p, U, c->adj_key.p, c->adj_cnt.p,
                         c->adj_stride, c->conn.p, mutual, c->csize.p, early_union ? (uint32_t*)nullptr : cross_parent, gt, c->adj_gstride, c->adj_nrank.p, inv_res2,
                         (const uint8_t*)nullptr, c->lc_defer.p, d_ndefer, c->lc_defer.p, (int)n_defer, cb_off, cb_lut, cb_bits, c->cb_words, (uint8_t*)nullptr, LcGate{nullptr, 0u});
      if (early_union)
        hipLaunchKernelGGL(k_union_mutual, dim3((n_defer + UM_ROWS - 1) / UM_ROWS), dim3(64), 0, c->stream, c->used_ids.p, U, c->adj_key.p, c->adj_cnt.p,
                           c->adj_stride, mutual, c->attach.p, owned_early, c->parent.p, 0, (const uint8_t*)nullptr, c->lc_defer.p, (int)n_defer, LcGate{nullptr, 0u});
    }
    // closestCheck
    VGS_HIP_TRY(c, c->work_ids.ensure((size_t)U + 16));
    unsigned int* d_ncand = (unsigned int*)(mcnt + 0);
    unsigned int* d_changed = (unsigned int*)(mcnt + 1);
    hipLaunchKernelGGL(k_cc_candidates, dim3((unsigned)((U + TB - 1) / TB)), dim3(TB), 0, c->stream, c->used_ids.p, U, c->adj_mused.p,
                       c->csize.p, MP.adjacency_min, c->cc_flags.p, c->work_ids.p, d_ncand);
    // Two host round trips of closestCheck (the number of candidates, the fixed-point flag) hide behind work that does not depend
    // on it: the pointer jumping over the first hooks, and the unions of the mutual edges (round 4; single-context runs only --
    // a tile's unions depend on ownership, which is computed behind closestCheck)
    const bool hide = !c->have_region && vgs_can_split_readback(c) && !early_union;
    if (hide) {
      vgs_status sb = vgs_readback_begin(c, d_ncand, 4);
      if (sb != VGS_OK) return sb;
      hipLaunchKernelGGL(k_compress, dim3(nbV), dim3(TB), 0, c->stream, c->parent.p, V);
      compressed = true;
      vgs_status se = vgs_readback_end(c, &n_cand, 4);
      if (se != VGS_OK) return se;
    } else {
      VGS_READBACK(c, &n_cand, d_ncand, 4);
    }
    if (n_cand > 0) {
      // fixed point of "re-attachment succeeds": passes are queued four at a time, each with its own change counter, and
      // only the last counter is read back (a pass after the fixed point changes nothing and costs microseconds; a host
      // round trip per pass cost more than the passes)
      unsigned int* d_chg4 = (unsigned int*)(mcnt + 4);   // words 4-5: four 32-bit counters
      for (int round = 0; round < 1 << 18; ++round) {
        VGS_HIP_TRY(c, hipMemsetAsync(d_chg4, 0, 16, c->stream));
        for (int q = 0; q < 4; ++q)
          hipLaunchKernelGGL((k_cc_pass<false>), dim3(n_cand), dim3(64), 0, c->stream, c->work_ids.p, (int)n_cand, c->used_ids.p,
                             c->adj_key.p, c->adj_cnt.p, c->adj_mused.p, c->adj_stride, c->node.p, c->csize.p, c->cc_flags.p, MP, c->attach.p, d_chg4 + q);
        unsigned int ch = 0;
        if (hide && !united) {
          vgs_status sb = vgs_readback_begin(c, d_chg4 + 3, 4);
          if (sb != VGS_OK) return sb;
          hipLaunchKernelGGL(k_union_mutual, dim3((unsigned)((U + UM_ROWS - 1) / UM_ROWS)), dim3(64), 0, c->stream, c->used_ids.p, U, c->adj_key.p, c->adj_cnt.p,
                             c->adj_stride, mutual, c->attach.p, (const uint8_t*)nullptr, c->parent.p, 0, (const uint8_t*)nullptr, (const uint32_t*)nullptr, 0, LcGate{nullptr, 0u});
          united = true;
          vgs_status se = vgs_readback_end(c, &ch, 4);
          if (se != VGS_OK) return se;
        } else {
          VGS_READBACK(c, &ch, d_chg4 + 3, 4);
        }
        if (!ch) break;
      }
      hipLaunchKernelGGL((k_cc_pass<true>), dim3(n_cand), dim3(64), 0, c->stream, c->work_ids.p, (int)n_cand, c->used_ids.p, c->adj_key.p,
                         c->adj_cnt.p, c->adj_mused.p, c->adj_stride, c->node.p, c->csize.p, c->cc_flags.p, MP, c->attach.p, d_changed);
    }
  } else {
    VGS_HIP_TRY(c, c->conn.ensure(16));
  }
  // connected components
  const bool owned_known = c->have_region && U > 0 && !c->K.no_tile_early;   // (computed at the head of the stage, first hooks taken by k_cross)
  if (c->have_region && !owned_known) { vgs_status so = vgs_compute_owned(c); if (so != VGS_OK) return so; }
  if (U > 0 && c->have_region && !owned_known)
    hipLaunchKernelGGL(k_cc_init, dim3(vgs_xcd_grid(U)), dim3(64), 0, c->stream, c->used_ids.p, U, c->adj_key.p, c->adj_cnt.p, c->adj_stride, mutual,
                       c->attach.p, c->have_region ? c->owned.p : nullptr, c->parent.p);
  if (U > 0 && !compressed) hipLaunchKernelGGL(k_compress, dim3(nbV), dim3(TB), 0, c->stream, c->parent.p, V);
  if (U > 0 && !united)
    hipLaunchKernelGGL(k_union_mutual, dim3((unsigned)((U + UM_ROWS - 1) / UM_ROWS)), dim3(64), 0, c->stream, c->used_ids.p, U, c->adj_key.p, c->adj_cnt.p,
                       c->adj_stride, mutual, c->attach.p, c->have_region ? c->owned.p : nullptr, c->parent.p, 1, (const uint8_t*)nullptr, (const uint32_t*)nullptr, 0, LcGate{nullptr, 0u});
  else if (U > 0 && n_cand > 0)
    hipLaunchKernelGGL(k_union_attach, dim3((n_cand + TB - 1) / TB), dim3(TB), 0, c->stream, c->work_ids.p, (int)n_cand, c->used_ids.p, c->attach.p, c->parent.p,
                       c->have_region ? c->owned.p : (const uint8_t*)nullptr);
  hipLaunchKernelGGL(k_flatten, dim3(nbV), dim3(TB), 0, c->stream, c->parent.p, V, c->have_region ? c->owned.p : nullptr, c->csz.p);
  // cluster filter + labels (VGS_T_LABELS: this tail of the stage, measured on its own; it is part of VGS_T_MERGE)
  VGS_HIP_TRY(c, hipEventRecord(c->ev[14], c->stream));   // ev[14], ev[15]: this tail's own pair
  uint32_t* keep_flag = c->head_flag.p;  // >= N >= V entries, free after features
  VGS_HIP_TRY(c, c->head_flag.ensure(V + 1));
  keep_flag = c->head_flag.p;
  unsigned int* d_nroots = (unsigned int*)(mcnt + 2);
  hipLaunchKernelGGL(k_root_flags, dim3((unsigned)((V + 1023) / 1024)), dim3(1024), 0, c->stream, c->parent.p, c->csz.p, V, c->P.method == 3 ? -1 : c->P.voxels_min,
                     keep_flag, d_nroots);
  size_t bytes = 0;
  VGS_HIP_TRY(c, rocprim::exclusive_scan(nullptr, bytes, keep_flag, c->kept_rank.p, 0u, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
  VGS_HIP_TRY(c, c->sort_tmp.ensure(bytes));
  VGS_HIP_TRY(c, rocprim::exclusive_scan(c->sort_tmp.p, bytes, keep_flag, c->kept_rank.p, 0u, (size_t)V, rocprim::plus<uint32_t>(), c->stream));
  hipLaunchKernelGGL(k_voxel_labels, dim3(nbV), dim3(TB), 0, c->stream, c->parent.p, keep_flag, c->kept_rank.p, V, c->vox_label.p,
                     (unsigned int*)(mcnt + 3));
  // (Round 4 measured the scatter-free alternative -- the point's octree key formed again from its coordinates, its voxel from the
  // brick table, label[p] written in input order: 0.26 ms against this kernel's 0.14 ms at 10 M points.  PCL's key arithmetic is
  // double precision -- three fp64 divisions per point -- and that costs more than the 4-byte scatter saves.)
  // A tile's labels are local names: the driver hands the global ones back (vgs_apply_tile_labels / vgs_apply_global_labels rewrite every
  // point's label), so the 40 MB scatter is left to whoever asks for point labels BEFORE that happens (vgs_ensure_point_labels; round 5)
  c->pt_labels_pending = c->have_region && N > 0;
  if (!c->pt_labels_pending)
    hipLaunchKernelGGL(k_point_labels, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, c->stream, c->perm_b.p, c->pt_vox.p, c->vox_label.p,
                       N, c->pt_label.p);
  VGS_HIP_TRY(c, hipEventRecord(c->ev[15], c->stream));
  c->labels_event_valid = true;
  VGS_HIP_TRY(c, hipGetLastError());
  // one read-back: words 1-3 = re-attachments, roots, kept segments
  uint64_t hm[4] = {0, 0, 0, 0};
  VGS_READBACK(c, hm, mcnt, sizeof(hm));
  { float lms = 0.f; if (hipEventElapsedTime(&lms, c->ev[14], c->ev[15]) == hipSuccess) c->times[VGS_T_LABELS] = lms; }
  const unsigned int n_roots = (unsigned int)hm[2];
  if (U > 0 && n_cand > 0) n_succ = (unsigned int)hm[1];
  c->counts[VGS_N_REATTACHED] = n_succ;
  c->counts[VGS_N_CLUSTERS] = n_roots;
  c->counts[VGS_N_KEPT] = (int64_t)(unsigned int)hm[3];
  c->counts[VGS_N_ISOLATED] = n_cand;
  return VGS_OK;
}
