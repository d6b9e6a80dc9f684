// pairlist.hip -- builder of the pair lists (pairlist.hpp): every heavy pair of a voxel's search ball, evaluated once per step.
// Part of the local-cut stage (stages a5-a7): replaces, for the neighbourhoods cut by k_localcut_pg, the per-neighbourhood
// buildAdjacencyGraph -> measuringDistance -> distanceWeight of voxel_segmentation.h:1796-1910, 1597-1740.
//
// One workgroup per wanted row a:
//   1. the entries of a's adjacency row at a lexicographically positive lattice offset (adj_off) are its candidates -- each pair of
//      the step lives in exactly one row;
//   2. screening as in the dense hand-over kernel (localcut_dense.hpp): proximity alone (d2 >= d2_stop), then the normals' dot
//      product against the table of cosines by distance bin -- a pair that fails either weighs <= 1 - cut;
//   3. both orientations of the weight for the survivors, on full wavefronts;
//   4. the pairs with a heavy orientation, sorted by max(w(a, b), w(b, a)) descending (regsort.hpp), go to a chunk of the pool
//      handed out by ONE atomic per row.
#include <cstdio>
#include <cstdlib>

#include "vgs_context.hpp"

#include "pairlist.hpp"
#include "regsort.hpp"

#define PL_TBINS 64   // = LC_TBINS (localcut.hip): bins of the screening table

struct PlParams {
  VgsWeightParams W;
  float thr0;        // 1 - cut
  float d2_stop;     // squared centroid distance from which proximity alone proves w <= thr0 (+inf: never)
  float ctab_scale;
  const float* ctab; // PL_TBINS cosines (lc_screen_table)
  float res_f, min_x, min_y, min_z, cube_tol;
};

#define PL_CHUNK_OF(cap) ((cap) <= 512 ? 1024u : 4u * (unsigned int)(cap))   // entries a workgroup takes from the pool at a time
#define PL_GRID_SMALL 8192   // workgroups of the one-wavefront builders (they stride over the work list)
#define PL_GRID_BIG 512

// pool bookkeeping in the context's counter words (zeroed with the local cut's counters at the start of the stage)
#define PL_W_CURSOR 58   // entries handed out
#define PL_W_WORK 59     // length of the work list (low half), of the redo list (high half)
#define PL_W_FULL 60     // rows that found the pool exhausted
#define PL_W_WORK2 61    // the second build's work / redo list lengths
#define PL_W_WORK3 42    // the third's (three builds of one run may overlap on their streams: the wide classes' rows, the many hand-overs' rows,
                         // the rows of the neighbourhoods the dense kernel passes on)

// (barriers that order LDS traffic only: see pg_barrier in localcut_pg.hpp; one wavefront needs no s_barrier at all)
template <int NWV>
__device__ __forceinline__ void pl_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  if constexpr (NWV == 1) __builtin_amdgcn_wave_barrier(); else __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int NWV, int CAP>
__global__ __launch_bounds__(64 * NWV) void k_pair_lists(const uint32_t* __restrict__ work, const unsigned int* __restrict__ n_work_dev, int n_work,
                                                         const uint32_t* __restrict__ used_ids, const uint64_t* __restrict__ adj_key,
                                                         const uint32_t* __restrict__ adj_cnt, int adj_stride, const uint16_t* __restrict__ adj_off,
                                                         const NodeRec* __restrict__ node, const uint64_t* __restrict__ vox_code, PlParams P,
                                                         uint2* __restrict__ idx, float4* __restrict__ ent, uint8_t* __restrict__ any,
                                                         unsigned int* __restrict__ cursor, unsigned int pool_cap, unsigned long long* __restrict__ n_full,
                                                         uint32_t* __restrict__ redo, unsigned int* __restrict__ n_redo) {
  constexpr int TB = 64 * NWV;
  __shared__ uint64_t lk[CAP];        // sort keys of the kept pairs: bits of max(w1, w2) << 32 | ~(candidate index)
  __shared__ float wab[CAP], wba[CAP];
  __shared__ uint16_t c_slot[CAP];    // candidate -> position in the row
  __shared__ uint32_t c_tid[CAP];     // candidate -> voxel id (the row is read once)
  __shared__ uint16_t s_cand[CAP];    // survivor of the screen -> candidate
  __shared__ float s_ctab[PL_TBINS];
  __shared__ int s_nc, s_ns, s_nk;
  __shared__ unsigned int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  if (tid < PL_TBINS) s_ctab[tid] = P.ctab[tid];
  const unsigned int n_items = n_work_dev ? *n_work_dev : (unsigned int)n_work;
  unsigned int chunk_pos = 0u, chunk_end = 0u;   // this workgroup's piece of the pool (uniform)
  bool pool_done = false;                        // ... it lies behind the pool's end: no further chunk is taken (uniform)
  for (unsigned int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int64_t u = work ? (int64_t)work[item] : (int64_t)item;
    const uint32_t vid = used_ids[u];
    const int n = (int)adj_cnt[u];
    const uint64_t* row = adj_key + u * adj_stride;
    const uint16_t* orow = adj_off + u * adj_stride;
    pl_barrier<NWV>();   // the previous row's arrays are free
    if (tid == 0) { s_nc = 0; s_ns = 0; s_nk = 0; }
    const NodeRec& A = node[vid];
    bool usable = orow[0] != 0xffffu;
    {
      // the ring bound (pairlist.hpp) and the cut's reading of offsets want the centroid inside the voxel's cube (as nearlist.hip)
      const uint64_t code = vox_code[vid];
      const float fx = vm_voxel_center(vm_compact21(code >> 2), P.res_f, P.min_x), fy = vm_voxel_center(vm_compact21(code >> 1), P.res_f, P.min_y),
                  fz = vm_voxel_center(vm_compact21(code), P.res_f, P.min_z);
      const float lim = 0.5f * P.res_f + P.cube_tol;
      const float ulp = 1.2e-7f;
      const bool inside = fabsf(A.c[0] - fx) + fabsf(fx) * ulp <= lim && fabsf(A.c[1] - fy) + fabsf(fy) * ulp <= lim && fabsf(A.c[2] - fz) + fabsf(fz) * ulp <= lim;
      usable = usable && (inside || !(A.flags & VGS_F_POS));
    }
    if (!usable) {   // uniform
      if (tid == 0) idx[vid] = make_uint2(0u, PL_UNUSABLE);
      continue;
    }
    pl_barrier<NWV>();
    // ---- 1. candidates: row entries at a positive lattice offset ----
    for (int base = 1; base < n; base += TB) {   // entry 0 is the voxel itself
      const int k = base + tid;
      bool pos = false;
      if (k < n) {
        const uint32_t pk = orow[k];
        const int dx = (int)(pk & 31u) - 16, dy = (int)((pk >> 5) & 31u) - 16, dz = (int)((pk >> 10) & 31u) - 16;
        pos = dz > 0 || (dz == 0 && (dy > 0 || (dy == 0 && dx > 0)));
      }
      const unsigned long long mk = __ballot(pos);
      int b = 0;
      if (mk != 0ull && lane == 0) b = atomicAdd(&s_nc, __popcll(mk));
      b = __builtin_amdgcn_readfirstlane(b);
      if (pos) { const int at = b + __popcll(mk & lt); if (at < CAP) { c_slot[at] = (uint16_t)k; c_tid[at] = (uint32_t)row[k]; } }
    }
    pl_barrier<NWV>();
    const int nc = s_nc;
    if (nc > CAP) {   // uniform: a row this kernel's arrays cannot hold goes to the next larger instantiation, or has no list
      if (tid == 0) { if (redo) redo[atomicAdd(n_redo, 1u)] = (uint32_t)u; else idx[vid] = make_uint2(0u, PL_UNUSABLE); }
      continue;
    }
    // ---- 2. screen ----
    const float ax = A.c[0], ay = A.c[1], az = A.c[2];
    for (int base = 0; base < nc; base += TB) {
      const int q = base + tid;
      bool keep = false;
      if (q < nc) {
        const NodeRec& B = node[c_tid[q]];
        const float dx = ax - B.c[0], dy = ay - B.c[1], dz = az - B.c[2];
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        const uint32_t both = A.flags & B.flags;
        if ((both & VGS_F_POS) != 0u && d2 >= P.d2_stop) {
          keep = false;   // proximity alone: w <= bound(d2) <= bound(d2_stop) <= thr0
        } else if ((both & (VGS_F_POS | VGS_F_NRM)) == (VGS_F_POS | VGS_F_NRM) && d2 > 0.0f) {
          int kb = (int)(d2 * P.ctab_scale);
          kb = kb > PL_TBINS - 1 ? PL_TBINS - 1 : kb;
          const float dot = vm_dot3(A.n, B.n);
          keep = !(dot <= s_ctab[kb] && dot >= -1.0f);
        } else {
          keep = !(vm_weight_bound_da(A, B, P.W) <= P.thr0);
        }
      }
      const unsigned long long mk = __ballot(keep);
      int b = 0;
      if (mk != 0ull && lane == 0) b = atomicAdd(&s_ns, __popcll(mk));
      b = __builtin_amdgcn_readfirstlane(b);
      if (keep) s_cand[b + __popcll(mk & lt)] = (uint16_t)q;
    }
    pl_barrier<NWV>();
    const int ns = s_ns;
    // ---- 3. both orientations of the survivors' weights ----
    // (one lane per pair: only the convexity distance depends on the order of the two voxels -- vgs_math.h: vm_pair_weight_both)
    for (int e = tid; e < ns; e += TB) {
      const int q = s_cand[e];
      const NodeRec& B = node[c_tid[q]];
      float w12, w21;
      vm_pair_weight_both(A, B, P.W, &w12, &w21);
      wab[q] = w12; wba[q] = w21;
    }
    pl_barrier<NWV>();
    // ---- 4. the heavy ones, sorted by their heavier orientation ----
    for (int base = 0; base < ns; base += TB) {
      const int s = base + tid;
      bool heavy = false;
      uint64_t key = 0;
      if (s < ns) {
        const int q = s_cand[s];
        float w1 = wab[q], w2 = wba[q];
        heavy = (w1 > P.thr0) || (w2 > P.thr0);   // NaN compares false
        if (heavy) {
          if (!(w1 == w1)) { w1 = 0.0f; wab[q] = 0.0f; }   // a NaN orientation never merges (Q3): stored as 0, below every threshold
          if (!(w2 == w2)) { w2 = 0.0f; wba[q] = 0.0f; }
          const float wm = w1 > w2 ? w1 : w2;
          key = ((uint64_t)vm_bits(wm) << 32) | (uint64_t)(0xffffffffu - (uint32_t)q);   // ties: ascending row position
        }
      }
      const unsigned long long mk = __ballot(heavy);
      int b = 0;
      if (mk != 0ull && lane == 0) b = atomicAdd(&s_nk, __popcll(mk));
      b = __builtin_amdgcn_readfirstlane(b);
      if (heavy) lk[b + __popcll(mk & lt)] = key;
    }
    pl_barrier<NWV>();
    const int nk = s_nk;
    if (nk > 1) {
      if constexpr (NWV == 1) {
        regsort::lds_fence();
        if (nk <= 64) regsort::sort_desc<1>(lk, nk, lane);
        else if (nk <= 128) regsort::sort_desc<2>(lk, nk, lane);
        else if (nk <= 256) regsort::sort_desc<4>(lk, nk, lane);
        else regsort::sort_desc<8>(lk, nk, lane);
        regsort::lds_fence();
      } else {
        regsort::sort_desc_block<NWV>(lk, nk, wave, lane, [&]() { pl_barrier<NWV>(); });
      }
    }
    // The pool is handed out in chunks of PL_CHUNK entries, one atomic per chunk: a workgroup fills its chunk row by row and takes
    // a new one when the next row does not fit (what is left of the old one is lost: less than a row in PL_CHUNK).  One atomic per
    // ROW was 133 k returning atomics on one address on the noisy surface -- they serialise at ~15 ns each and took two of the
    // kernel's 2.0 ms.
    constexpr unsigned int PL_CHUNK = PL_CHUNK_OF(CAP);
    if (nk > 0 && !pool_done && chunk_pos + (unsigned int)nk > chunk_end) {   // uniform
      if (tid == 0) s_base = atomicAdd(cursor, PL_CHUNK);
      pl_barrier<NWV>();
      chunk_pos = s_base;
      chunk_end = s_base + PL_CHUNK;
      pl_barrier<NWV>();
    }
    const unsigned int base_e = chunk_pos;
    chunk_pos += (unsigned int)nk;
    // (a workgroup that has found the pool exhausted takes no further chunk: the 32-bit cursor ends at most one chunk per workgroup behind
    // the pool's end -- the pool is capped at 4.0e9 entries -- instead of advancing with every row of the run and wrapping: ADVICE r5)
    if (nk > 0 && (pool_done || (unsigned long long)chunk_end > (unsigned long long)pool_cap)) {   // uniform: the chunk lies (partly) behind the pool's end
      pool_done = true;
      if (tid == 0) { idx[vid] = make_uint2(0u, PL_UNUSABLE); atomicAdd(n_full, 1ull); }
      continue;
    }
    for (int r = tid; r < nk; r += TB) {
      const int q = (int)(0xffffffffu - (uint32_t)lk[r]);
      const int slot = c_slot[q];
      const uint32_t t = c_tid[q];
      ent[(size_t)base_e + (size_t)r] = make_float4(wab[q], wba[q], __uint_as_float((uint32_t)orow[slot]), __uint_as_float(t));
      any[t] = 1;
    }
    if (tid == 0) { idx[vid] = make_uint2(base_e, (unsigned int)nk); if (nk) any[vid] = 1; }
  }
}

// the rows of the neighbourhoods the pair-list kernel is going to cut: every voxel of their adjacency rows is wanted
struct PlWho { const uint32_t* ids[6]; const unsigned int* n_dev[6]; unsigned int n_host[6]; int n_lists; };   // n_dev[k] null: n_host[k]
__global__ __launch_bounds__(256) void k_pl_mark(PlWho who, const uint64_t* __restrict__ adj_key, const uint32_t* __restrict__ adj_cnt, int adj_stride,
                                                 uint8_t* __restrict__ need, LcGate gate) {
  if (!lc_gate_open(gate)) return;
  const int lane = threadIdx.x & 63;
  const unsigned int wv = blockIdx.x * 4u + (threadIdx.x >> 6), nwv = gridDim.x * 4u;
  for (int k = 0; k < who.n_lists; ++k) {
    const unsigned int nk = who.n_dev[k] ? *who.n_dev[k] : who.n_host[k];
    for (unsigned int it = wv; it < nk; it += nwv) {
      const int64_t u = (int64_t)who.ids[k][it];
      const int n = (int)adj_cnt[u];
      const uint64_t* row = adj_key + u * adj_stride;
      for (int e = lane; e < n; e += 64) need[(uint32_t)row[e]] = 1;
    }
  }
}
// ... and of those the ones without a list so far, as a work list of rows (used-voxel indices) in voxel order
// (256 threads: a closed gate makes this an empty launch beside the dense kernel's hand-overs, and a 16-wavefront workgroup waits for a
// CU to drain before it can find that out -- with the dispatcher holding the CU for it meanwhile: round 6 timeline, 0.5 ms)
__global__ __launch_bounds__(256) void k_pl_worklist(const uint32_t* __restrict__ used_ids, int64_t U, const uint8_t* __restrict__ need,
                                                      const uint2* __restrict__ idx, uint32_t* __restrict__ work, unsigned int* __restrict__ n_work, LcGate gate) {
  __shared__ unsigned int s_cnt[4], s_base;
  if (!lc_gate_open(gate)) return;   // (the work list stays empty: the builders behind this launch find nothing to do)
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool want = false;
  if (u < U) { const uint32_t v = used_ids[u]; want = (need == nullptr || need[v] == 1) && idx[v].y == PL_NOT_BUILT; }
  const unsigned long long mk = __ballot(want);
  if (lane == 0) s_cnt[wave] = (unsigned int)__popcll(mk);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int tot = 0;
    for (int w = 0; w < 4; ++w) { const unsigned int x = s_cnt[w]; s_cnt[w] = tot; tot += x; }
    s_base = tot ? atomicAdd(n_work, tot) : 0u;
  }
  __syncthreads();
  if (want) work[s_base + s_cnt[wave] + __popcll(mk & ((1ull << lane) - 1ull))] = (uint32_t)u;
}

// Start of a run: can the lists exist for this context's data (voxel lattice, rows with lattice offsets and without inert unused
// voxels)?  One fill resets the per-voxel state -- index (not built), wanted marks, "part of a heavy pair" flags are ONE buffer.
vgs_status vgs_pairlists_begin(vgs_ctx* c, hipStream_t strm) {
  c->pl_enabled = false;
  if (c->P.method != 2 || !c->adj_pruned || !c->adj_have_off || c->adj_R > 15 || c->U == 0 || c->K.no_pairlists) return VGS_OK;
  const int64_t V = c->V;
  VGS_HIP_TRY(c, c->pl_state.ensure((size_t)V * 10));
  VGS_HIP_TRY(c, hipMemsetAsync(c->pl_state.p, 0xff, (size_t)V * 10, strm));
  VGS_HIP_TRY(c, c->pl_work.ensure(6 * (size_t)c->U + 16));
  {
    // Pairs of one neighbourhood that are not in each other's ball: their centres are at least sqrt(r2) apart (the float predicate
    // of adjacency.hip failed), their centroids lie in their cubes widened by the tolerance the builder checks, so the centroids are
    // at least d apart and fact (U) bounds their weight.  The margins cover the float evaluation of either distance.
    const float res = c->P.voxel_size;
    const float d = vm_sqrt(c->adj_r2) * (1.0f - 2.0e-6f) - 1.7320509f * (1.0f + 2.0f * PL_CUBE_TOL + 4.0e-6f) * res;
    VgsWeightParams W;
    W.inv_sig_p = 1.0f / c->P.sig_p; W.inv_sig_n = 1.0f / c->P.sig_n; W.inv_sig_o = 1.0f / c->P.sig_o; W.inv_sig_e = 1.0f / c->P.sig_e;
    W.inv_sig_c = 1.0f / c->P.sig_c; W.inv_sig_w2 = 1.0f / (c->P.sig_w * c->P.sig_w); W.svgs = 0;
    c->pl_w_ring = d > 0.0f ? vm_weight_bound_d(d * d * (1.0f - 2.0e-6f), W) : __builtin_huge_valf();
  }
  // The pool, sized ONCE per run (up to three builds of a run, on different streams, each used to look at the free memory again,
  // and a build that found more of it re-allocated the pool under the index entries and the launches of the earlier ones): at most half of a
  // row's entries are at positive offsets; never more than the 32-bit index of an entry can name.
  {
    const int64_t U = c->U;
    // (every builder workgroup may leave one chunk partly used)
    const double want = (double)U * (double)((c->adj_stride - 1) / 2 + 1) + (double)PL_GRID_SMALL * PL_CHUNK_OF(512) + (double)PL_GRID_BIG * PL_CHUNK_OF(4096);
    const size_t cap = (size_t)(want < 4.0e9 ? want : 4.0e9);
    if (c->pl_ent.cap < cap) {
      size_t freeb = 0, totb = 0;
      (void)hipMemGetInfo(&freeb, &totb);
      size_t take = cap;
      // the lists are a cache of weights: when the device cannot hold one for every ball offset the pool is what fits, and rows that
      // find it exhausted keep the paths of round 4 (counted in PL_W_FULL)
      if (take * sizeof(float4) > freeb / 2) take = freeb / 2 / sizeof(float4);
      if (take < (size_t)U + (size_t)PL_GRID_SMALL * PL_CHUNK_OF(512) + (size_t)PL_GRID_BIG * PL_CHUNK_OF(4096)) return VGS_OK;   // (not enabled)
      if (c->pl_ent.cap < take) {
        // the previous run's launches may still read the old pool on the side streams
        VGS_HIP_TRY(c, hipDeviceSynchronize());
        VGS_HIP_TRY(c, c->pl_ent.ensure(take));
      }
    }
  }
  c->pl_enabled = true;
  return VGS_OK;
}

// Builds the lists of the rows the given neighbourhoods need (all_rows: of every used voxel) on `strm`, which must be ordered behind
// vgs_pairlists_begin's fill and behind whatever wrote the hand-over lists.  Rows built by an earlier call of the run are skipped.
vgs_status vgs_pairlists_build(vgs_ctx* c, hipStream_t strm, const uint32_t* const* ids, const unsigned int* const* n_dev, const unsigned int* n_host,
                               int n_lists, bool all_rows, const float* ctab, float ctab_scale, float d2_stop, int slot, const LcGate& gate,
                               bool big_rows_too) {
  if (!c->pl_enabled) return VGS_E_STATE;
  const int64_t U = c->U, V = c->V;
  uint2* idx = (uint2*)c->pl_state.p;
  uint8_t* need = c->pl_state.p + (size_t)V * 8;
  uint8_t* any = c->pl_state.p + (size_t)V * 9;
  unsigned int* cursor = (unsigned int*)(c->counters.p + PL_W_CURSOR);
  // two builds of one run (the wide classes' rows, the hand-overs' rows) may overlap on their streams: each has its own work lists
  unsigned int* n_work = (unsigned int*)(c->counters.p + (slot == 2 ? PL_W_WORK3 : (slot ? PL_W_WORK2 : PL_W_WORK)));
  unsigned int* n_redo = n_work + 1;
  uint32_t* const wl = c->pl_work.p + (size_t)slot * 2 * (size_t)U;
  // (the pool was sized by vgs_pairlists_begin: it is never re-allocated between two builds of one run -- index entries and launches in
  // flight hold its address)
  const unsigned int pool_cap = (unsigned int)(c->pl_ent.cap < 0xfffffff0ull ? c->pl_ent.cap : 0xfffffff0ull);
  VGS_HIP_TRY(c, hipMemsetAsync(n_work, 0, 8, strm));
  if (!all_rows) {
    PlWho who;
    who.n_lists = n_lists;
    unsigned int upper = 0;
    for (int k = 0; k < 6; ++k) {
      who.ids[k] = k < n_lists ? ids[k] : nullptr; who.n_dev[k] = (k < n_lists && n_dev) ? n_dev[k] : nullptr; who.n_host[k] = (k < n_lists && n_host) ? n_host[k] : 0u;
      if (k < n_lists) upper += who.n_dev[k] ? (unsigned int)U : who.n_host[k];
    }
    const unsigned int grid = std::min<unsigned int>(4096u, std::max<unsigned int>(1u, (upper + 3u) / 4u));
    hipLaunchKernelGGL(k_pl_mark, dim3(grid), dim3(256), 0, strm, who, c->adj_key.p, c->adj_cnt.p, c->adj_stride, need, gate);
  }
  hipLaunchKernelGGL(k_pl_worklist, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, strm, c->used_ids.p, U, all_rows ? (const uint8_t*)nullptr : need, idx,
                     wl, n_work, gate);
  PlParams P;
  P.W.inv_sig_p = 1.0f / c->P.sig_p; P.W.inv_sig_n = 1.0f / c->P.sig_n; P.W.inv_sig_o = 1.0f / c->P.sig_o; P.W.inv_sig_e = 1.0f / c->P.sig_e;
  P.W.inv_sig_c = 1.0f / c->P.sig_c; P.W.inv_sig_w2 = 1.0f / (c->P.sig_w * c->P.sig_w); P.W.svgs = 0;
  P.thr0 = vm_cut_threshold(1.0f, c->P.cut_thred, 1);
  P.d2_stop = d2_stop; P.ctab = ctab; P.ctab_scale = ctab_scale;
  P.res_f = c->P.voxel_size; P.min_x = (float)c->box.min[0]; P.min_y = (float)c->box.min[1]; P.min_z = (float)c->box.min[2];
  P.cube_tol = PL_CUBE_TOL * c->P.voxel_size;
  unsigned long long* n_full = (unsigned long long*)(c->counters.p + PL_W_FULL);
  // rows of up to 1024 entries (512 at positive offsets) in a one-wavefront workgroup; longer ones are passed on to eight wavefronts
  const unsigned int grid1 = (unsigned int)std::min<int64_t>(U, PL_GRID_SMALL);
  // (big_rows_too false: rows with more than 512 candidates get no list and their neighbourhoods keep the kernels of round 4 -- the
  // hand-overs of the one-wavefront classes, whose vertices rarely have such rows, do not pay for the eight-wavefront launch)
  const bool big_rows = big_rows_too && (c->adj_stride - 1) / 2 > 512;
  if ((c->adj_stride - 1) / 2 <= 272)   // (a ball of five voxels: 515 offsets, 257 of them positive -- half the LDS, twice the wavefronts per CU)
    hipLaunchKernelGGL((k_pair_lists<1, 272>), dim3(grid1), dim3(64), 0, strm, wl, n_work, 0, c->used_ids.p, c->adj_key.p, c->adj_cnt.p, c->adj_stride,
                       c->adj_off.p, c->node.p, c->vox_code.p, P, idx, c->pl_ent.p, any, cursor, pool_cap, n_full, (uint32_t*)nullptr, n_redo);
  else
    hipLaunchKernelGGL((k_pair_lists<1, 512>), dim3(grid1), dim3(64), 0, strm, wl, n_work, 0, c->used_ids.p, c->adj_key.p, c->adj_cnt.p, c->adj_stride,
                       c->adj_off.p, c->node.p, c->vox_code.p, P, idx, c->pl_ent.p, any, cursor, pool_cap, n_full, big_rows ? wl + U : (uint32_t*)nullptr, n_redo);
  if (big_rows)
    hipLaunchKernelGGL((k_pair_lists<8, 4096>), dim3((unsigned int)std::min<int64_t>(U, PL_GRID_BIG)), dim3(512), 0, strm, wl + U, n_redo, 0, c->used_ids.p, c->adj_key.p,
                       c->adj_cnt.p, c->adj_stride, c->adj_off.p, c->node.p, c->vox_code.p, P, idx, c->pl_ent.p, any, cursor, pool_cap, n_full, (uint32_t*)nullptr,
                       (unsigned int*)nullptr);
  VGS_HIP_TRY(c, hipGetLastError());
  return VGS_OK;
}
