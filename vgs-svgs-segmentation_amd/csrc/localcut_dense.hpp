// localcut_dense.hpp -- the local cut of the voxels the one-wavefront kernel hands over: ONE WORKGROUP PER VOXEL, every
// pair examined once.  Included by localcut.hip behind localcut_wave.hpp (LcParams, wave_sync, TB from there).
//
// Who gets here: neighbourhoods in clutter (vegetation, edges of scan shadows).  No segment forms early and freezes, so the
// lazy schedule of localcut_wave.hpp ends up looking at almost every pair, shell after shell, and gives up.  The plain order
// suits them better -- all n(n-1)/2 pairs once, but cheaply:
//   * the neighbour records sit in LDS as one array per field (a lane walking consecutive vertices reads consecutive
//     banks), so no pair costs an L2 round trip;
//   * pass 1 puts every pair through the proximity + angle bound (vm_weight_bound_da) and queues the survivors, pass 2
//     evaluates the queue on full wavefronts and keeps the edges heavier than a singleton's threshold (fact S of
//     localcut_wave.hpp); the pairs are taken in blocks of QCAP so that the queue cannot overflow;
//   * the sequential merge is the claim scheme of the one-wavefront kernel (all edges of a 64-edge step that touch no
//     segment of an earlier undecided edge act at once), and it stops as soon as the voxel's own segment is frozen (fact F);
//   * phase B (pairs between non-singleton segments still below the singleton threshold) only when the voxel's segment
//     is one of them.
// Same sequential semantics as k_localcut; a list that overflows sends the voxel on to that kernel.
#ifndef LOCALCUT_DENSE_HPP_
#define LOCALCUT_DENSE_HPP_

#include "pairlist.hpp"   // LcGate

// Two instantiations: <128, 2048, 2048, 256> for the one-wavefront classes (30 KB of LDS, five voxels per CU) and
// <512, 4096, 2048, 512> for the hand-overs of the classes above (78 KB, two per CU).  MAXM neighbours at most; LCAP edges
// heavier than the singleton threshold a neighbourhood may hold (phase B: pairs at or below it); QCAP pairs per block of
// pass 1 = queue slots; TB threads.  The grid is fixed and strides over the hand-over list, whose length is on the device.
// the sort's chunk-local stages in registers (regsort.hpp); 0: every stage through LDS
#ifndef LD_REG_SORT
#define LD_REG_SORT 1
#endif
// workgroups per CU the small instantiation is compiled for (registers: 512 / this per lane)
#ifndef LD_SMALL_WG_PER_CU
#define LD_SMALL_WG_PER_CU 5
#endif
#define DN_NBIN 1024    // histogram bins of the banded phase B (they reuse the queue's bytes)
#define DN_ABIN 256     // histogram bins of the banded phase A, over (thr0, 1]
#define DN_NF 15        // words of a record kept in LDS: c[3], n[3], f[8], flags

#ifdef VGS_PROF
__device__ unsigned long long g_dn_prof[16];
__device__ unsigned long long g_dn_trace[4 * 16384 + 1];   // per row: wall begin, wall end, HW_ID | XCC_ID << 32, m | MAXM << 32
#define DNP_T0() long long _dt0 = clock64()
#define DNP_ACC(slot) do { long long _dt1 = clock64(); if (tid == 0) atomicAdd(&g_dn_prof[slot], (unsigned long long)(_dt1 - _dt0)); _dt0 = _dt1; } while (0)
#else
#define DNP_T0() do {} while (0)
#define DNP_ACC(slot) do {} while (0)
#endif

template <int MAXM, int LCAP, int QCAP, int TB>
__global__ __launch_bounds__(TB, (MAXM <= 255 ? LD_SMALL_WG_PER_CU : 4)) void k_localcut_dense(const uint32_t* __restrict__ work, int work_stride, int n_lists, const unsigned int* __restrict__ n_work_dev,
                                                          const uint64_t* __restrict__ adj_key, const uint32_t* __restrict__ adj_cnt,
                                                          int adj_stride, const NodeRec* __restrict__ node, LcParams P,
                                                          uint8_t* __restrict__ conn, unsigned long long* __restrict__ counters,
                                                          uint32_t* __restrict__ fallback, unsigned int* __restrict__ n_fallback,
                                                          uint32_t* __restrict__ evals_out, LcGate gate,
                                                          uint32_t* __restrict__ to_pg, unsigned int* __restrict__ n_to_pg) {
  if (!lc_gate_open(gate)) return;
  constexpr bool SMALL = MAXM <= 255;   // vertex indices and segment sizes fit a byte, pair ids 16 bits
  typedef typename std::conditional<SMALL, uint8_t, uint16_t>::type idx_t;
  typedef typename std::conditional<SMALL, uint16_t, uint32_t>::type q_t;
  constexpr int PSH = SMALL ? 8 : 16;   // pair id = (a << PSH) | b, a < b
  constexpr uint32_t PMASK = (1u << PSH) - 1u;
  constexpr uint32_t PCOMP = SMALL ? 0xffffu : 0xffffffffu;
  __shared__ uint64_t lk[LCAP];               // weight bits << 32 | ~pair id: one compare orders (w desc, pair asc)
  __shared__ q_t queue[QCAP];                 // pair ids that passed the bound
  static_assert(DN_NBIN * 4 <= QCAP * sizeof(q_t), "the histogram lives in the queue");
  __shared__ uint32_t rec[DN_NF][MAXM];       // neighbour records, one array per field
  __shared__ float thr[MAXM];
  __shared__ uint32_t claim[MAXM];
  __shared__ idx_t seg[MAXM], rep[MAXM], ssz[MAXM], alist[MAXM];
  __shared__ int s_nq, s_nlist, s_flag, s_nb;
  __shared__ uint32_t s_hist_a[DN_ABIN];      // banded phase A
  __shared__ float s_ctab[LC_TBINS];          // LcParams::ctab where a lane can index it

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const VgsWeightParams& W = P.W;
  const float cut = P.cut;
  const float thr0 = vm_cut_threshold(1.0f, cut, 1);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  if (tid < LC_TBINS) s_ctab[tid] = P.ctab[tid];
  // n_lists hand-over lists (largest neighbourhoods first), work_stride apart, taken one behind the other; their lengths
  // are on the device and the fixed grid strides over them
  auto process = [&](const uint32_t u) {
  const int m = __builtin_amdgcn_readfirstlane((int)adj_cnt[u]);
  const uint64_t* row = adj_key + (int64_t)u * adj_stride;
  uint8_t* crow = conn + (int64_t)u * adj_stride;
  DNP_T0();
#ifdef VGS_PROF
  const long long t_begin = clock64();
  const long long w_begin = wall_clock64();
#endif

  auto hand_on = [&]() {   // all threads
    if (tid == 0) { fallback[atomicAdd(n_fallback, 1u)] = u; atomicAdd(&counters[7], 1ull); }
  };
  if (m > MAXM) { hand_on(); return; }   // not a one-wavefront voxel (cannot happen from the hand-over list)

  // ---- records: 4 lanes per record read its four 16-byte quads, each word goes to its field array ----
  for (int e = tid; e < m * 4; e += TB) {
    const int v = e >> 2, qd = e & 3;
    const uint4 x = ((const uint4*)node)[(size_t)(uint32_t)row[v] * 4 + qd];
    const int f0 = qd * 4;
    rec[f0][v] = x.x;
    if (f0 + 1 < DN_NF) rec[f0 + 1][v] = x.y;
    if (f0 + 2 < DN_NF) rec[f0 + 2][v] = x.z;
    if (f0 + 3 < DN_NF) rec[f0 + 3][v] = x.w;
  }
  for (int c = tid; c < m; c += TB) { seg[c] = (idx_t)c; rep[c] = (idx_t)c; ssz[c] = 1; thr[c] = thr0; claim[c] = 0xffffffffu; }
  if (tid == 0) { s_nq = 0; s_nlist = 0; s_flag = 0; s_nb = 0; }
  __syncthreads();
  auto load = [&](int v) -> NodeRec {
    NodeRec r;
    r.c[0] = __uint_as_float(rec[0][v]); r.c[1] = __uint_as_float(rec[1][v]); r.c[2] = __uint_as_float(rec[2][v]);
    r.n[0] = __uint_as_float(rec[3][v]); r.n[1] = __uint_as_float(rec[4][v]); r.n[2] = __uint_as_float(rec[5][v]);
#pragma unroll
    for (int k = 0; k < 8; ++k) r.f[k] = __uint_as_float(rec[6 + k][v]);
    r.flags = rec[14][v];
    r.pad = 0;
    return r;
  };
  auto load_cn = [&](int v) -> NodeRec {   // what the bound reads
    NodeRec r;
    r.c[0] = __uint_as_float(rec[0][v]); r.c[1] = __uint_as_float(rec[1][v]); r.c[2] = __uint_as_float(rec[2][v]);
    r.n[0] = __uint_as_float(rec[3][v]); r.n[1] = __uint_as_float(rec[4][v]); r.n[2] = __uint_as_float(rec[5][v]);
    r.flags = rec[14][v];
    return r;
  };
  // pair p of the row-major triangle over nv vertices (a < b), decoded from the end (localcut_wave.hpp: enum_section)
  auto decode = [&](uint32_t p, int nv, uint32_t Pn, int& a, int& b) {
    const uint32_t q = Pn - 1u - p;
    uint32_t r = (uint32_t)((__builtin_amdgcn_sqrtf((float)(8u * q + 1u)) - 1.0f) * 0.5f);
    r += (((r + 1u) * (r + 2u)) >> 1) <= q ? 1u : 0u;
    r -= ((r * (r + 1u)) >> 1) > q ? 1u : 0u;
    a = nv - 2 - (int)r;
    b = nv - 1 - (int)(q - ((r * (r + 1u)) >> 1));
  };
  // descending sort of lk[0, cnt): one-direction bitonic network, slots >= cnt never move (localcut_wave.hpp: sort_section)
  auto sort_list = [&](int cnt) __attribute__((always_inline)) {
    if constexpr (LD_REG_SORT != 0) {   // 512 keys per wavefront in registers, the widest strides through LDS (regsort.hpp)
      static_assert(LCAP <= 512 * (TB / 64), "one block of 512 keys per wavefront");
      regsort::sort_desc_block<TB / 64>(lk, cnt, wave, lane, [&]() { __syncthreads(); });
      return;
    }
    int np = 128;
    while (np < cnt) np <<= 1;
    auto cmpx = [&](int lo, int hi) {
      if (hi >= cnt) return;
      const uint64_t x = lk[lo], y = lk[hi];
      if (x < y) { lk[lo] = y; lk[hi] = x; }
    };
    // comparators at distance < 128 stay inside a 128-key chunk: a wavefront runs them for its chunks back to back with
    // no workgroup barrier in between (64 comparators per stage = one per lane)
    auto chunk_tail = [&](int first_sl) {   // strides 2^first_sl .. 1 in every chunk
      for (int ch = wave; ch < (np >> 7); ch += TB / 64) {
        if ((ch << 7) >= cnt) break;
        for (int sl = first_sl; sl >= 0; --sl) {
          const int lo = (ch << 7) + (((lane >> sl) << (sl + 1)) | (lane & ((1 << sl) - 1)));
          cmpx(lo, lo + (1 << sl));
          wave_sync();
        }
      }
    };
    __syncthreads();
    // sizes 2 .. 128: entirely chunk-local
    for (int ch = wave; ch < (np >> 7); ch += TB / 64) {
      if ((ch << 7) >= cnt) break;
      for (int size = 2, sbit = 1; size <= 128; size <<= 1, ++sbit) {
        {
          const int blk = lane >> (sbit - 1), i = lane & ((size >> 1) - 1);
          cmpx((ch << 7) + (blk << sbit) + i, (ch << 7) + (blk << sbit) + size - 1 - i);
          wave_sync();
        }
        for (int sl = sbit - 2; sl >= 0; --sl) {
          const int lo = (ch << 7) + (((lane >> sl) << (sl + 1)) | (lane & ((1 << sl) - 1)));
          cmpx(lo, lo + (1 << sl));
          wave_sync();
        }
      }
    }
    __syncthreads();
    for (int size = 256, sbit = 8; size <= np; size <<= 1, ++sbit) {
      for (int t = tid; t < (np >> 1); t += TB) {
        const int blk = t >> (sbit - 1), i = t & ((size >> 1) - 1);
        cmpx((blk << sbit) + i, (blk << sbit) + size - 1 - i);
      }
      __syncthreads();
      for (int sl = sbit - 2; sl >= 7; --sl) {
        const int strd = 1 << sl;
        for (int t = tid; t < (np >> 1); t += TB) {
          const int lo = ((t >> sl) << (sl + 1)) | (t & (strd - 1));
          cmpx(lo, lo + strd);
        }
        __syncthreads();
      }
      chunk_tail(6);
      __syncthreads();
    }
  };
  // Sequential merge of the sorted list on wavefront 0 (localcut_wave.hpp: merge_list, to the end of the list); stops early
  // when one segment is left or the voxel's own segment can no longer merge.  Ends with a workgroup barrier.
  int merges = 0;   // meaningful on wavefront 0
  auto merge_list = [&](int cnt, bool last_list) {   // last_list: no further list of this phase follows
    if (wave == 0) {
      // while the voxel is alone it can only merge through an edge of its own (vertex 0 is the first vertex of its pairs)
      int last_own = -1;
      for (int base = 0; base < cnt; base += 64) {
        const int e = base + lane;
        const bool own = e < cnt && ((PCOMP - (uint32_t)lk[e]) >> PSH) == 0u;
        const unsigned long long mk = __ballot(own);
        if (mk != 0ull) last_own = base + 63 - __builtin_clzll(mk);
      }
      int pos = 0;
      while (pos < cnt) {
        const int e = pos + lane;
        float w = 0.f;
        int sa = 0, sb = 0;
        bool alive = false;
        if (e < cnt) {
          const uint64_t key = lk[e];
          w = vm_from_bits((uint32_t)(key >> 32));
          const uint32_t pid = PCOMP - (uint32_t)key;
          sa = seg[pid >> PSH];
          sb = seg[pid & PMASK];
          alive = true;
        }
        while (true) {
          // One wavefront of the workgroup merges and nothing hides its LDS round trips: both representative chains are walked
          // at once, and every step brings the size and the threshold of the node reached along (the chains are about one step
          // long), so an iteration waits for the walk, the claims, their read-back and the stores -- four trips instead of seven.
          float ta = 0.f, tb = 0.f;
          int nsz = 1;
          if (alive) {
            int r = rep[sa], rb = rep[sb], za = (int)ssz[sa], zb = (int)ssz[sb];
            ta = thr[sa]; tb = thr[sb];
            while (r != sa || rb != sb) { sa = r; sb = rb; r = rep[sa]; rb = rep[sb]; za = (int)ssz[sa]; zb = (int)ssz[sb]; ta = thr[sa]; tb = thr[sb]; }
            nsz = za + zb;
            alive = sa != sb;
          }
          if (__ballot(alive) == 0ull) break;
          if (alive) { atomicMin(&claim[sa], (uint32_t)lane); atomicMin(&claim[sb], (uint32_t)lane); }
          wave_sync();
          bool decided = false;
          if (alive) {
            const uint32_t ca = claim[sa], cb = claim[sb];   // both loads before either compare
            decided = (ca == (uint32_t)lane) & (cb == (uint32_t)lane);
          }
          wave_sync();
          if (alive) { claim[sa] = 0xffffffffu; claim[sb] = 0xffffffffu; }
          const bool pass = decided && (w > ta) && (w > tb);
          if (pass) {
            const int keep = (ta >= tb) ? sa : sb;   // VS:1972-1983: the segment with the larger threshold survives
            const int gone = sa ^ sb ^ keep;          // see localcut_wave.hpp
            rep[gone] = (idx_t)keep;
            thr[keep] = vm_cut_threshold(w, cut, nsz);   // seg_int = w (VS:1988)
            ssz[keep] = (idx_t)nsz;
            ssz[gone] = 0;
          }
          merges += __popcll(__ballot(pass));
          alive = alive && !decided;
          wave_sync();
        }
        pos += 64;
        if (merges >= m - 1 || pos >= cnt) break;
        // every edge from here on weighs at most wn: a segment whose threshold is not below that is frozen (fact F), and
        // only the voxel's own segment is reported
        const float wn = vm_from_bits((uint32_t)(lk[pos] >> 32));
        int s0 = 0, r;
        while ((r = rep[s0]) != s0) s0 = r;
        if (!(thr[s0] < wn)) break;
        if (last_list && ssz[s0] == 1 && pos > last_own) break;   // (the other segments' states are left unfinished)
      }
      for (int c = lane; c < m; c += 64) {
        int s = seg[c];
        while (rep[s] != s) s = rep[s];
        seg[c] = (idx_t)s;
      }
    }
    __syncthreads();
  };

  unsigned int my_pairs = 0;
  bool done = (m < 2);
  bool handed = false;
  // ---- 1. can the voxel merge at all?  (an incident edge heavier than a singleton's threshold) ----
  if (!done) {
    bool any = false;
    const NodeRec A = load(0);
    for (int x = 1 + tid; x < m; x += TB) {
      const NodeRec B = load(x);
      ++my_pairs;
      if (!(vm_weight_bound_da(A, B, W) <= thr0)) any = any || (vm_pair_weight_regs(A, B, W) > thr0);
    }
    if (any) s_flag = 1;
    __syncthreads();
    done = s_flag == 0;
  }
  DNP_ACC(1);
  if (!done) {
    // ---- 2. phase A: every edge heavier than thr0 ----
    const uint32_t Pall = (uint32_t)(m * (m - 1) / 2);
    // weights above thr0 in DN_ABIN bins (banded phase A)
    const float scale_a = (float)DN_ABIN / (1.0f - thr0);
    auto bin_a = [&](float w) -> int { const int bb = (int)((w - thr0) * scale_a); return bb < 0 ? 0 : (bb > DN_ABIN - 1 ? DN_ABIN - 1 : bb); };
    // One sweep over all pairs: bound -> queue -> weight, and for the edges heavier than thr0
    //   mode 0: into the list (stops once the list overflows);  mode 1: count them per weight bin;
    //   mode 2: into the list if their bin is in [lo, top).
    auto sweep = [&](int mode, int lo, int top) {
      if (tid == 0) s_nq = 0;
      __syncthreads();
      for (uint32_t base = 0; base < Pall; base += QCAP) {
        for (uint32_t p = base + (uint32_t)tid; p < base + QCAP; p += TB) {   // same trip count for every thread of a wavefront
          bool keep = false;
          int a = 0, b = 0;
          if (p < Pall) {
            decode(p, m, Pall, a, b);
            ++my_pairs;
            const NodeRec A = load_cn(a), B = load_cn(b);
            const float dx = A.c[0] - B.c[0], dy = A.c[1] - B.c[1], dz = A.c[2] - B.c[2];
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            const uint32_t both = A.flags & B.flags;
            if ((both & VGS_F_POS) != 0u && d2 >= P.d2_stop) {
              keep = false;   // proximity alone: w <= bound(d2) <= bound(d2_stop) <= thr0
            } else if ((both & (VGS_F_POS | VGS_F_NRM)) == (VGS_F_POS | VGS_F_NRM) && d2 > 0.0f) {
              // the proximity + angle bound read from a table by distance (LcParams::ctab): a tenth of its cost
              int k = (int)(d2 * P.ctab_scale);
              k = k > LC_TBINS - 1 ? LC_TBINS - 1 : k;
              const float dot = vm_dot3(A.n, B.n);
              keep = !(dot <= s_ctab[k] && dot >= -1.0f);
            } else {
              keep = !(vm_weight_bound_da(A, B, W) <= thr0);
            }
          }
          const unsigned long long mk = __ballot(keep);
          if (mk != 0ull) {
            int qb = 0;
            if (lane == 0) qb = atomicAdd(&s_nq, __popcll(mk));
            qb = __shfl(qb, 0, 64);
            if (keep) queue[qb + __popcll(mk & lt_mask)] = (q_t)(((uint32_t)a << PSH) | (uint32_t)b);
          }
        }
        __syncthreads();
        const int nq = s_nq;
        for (int e = tid; e < nq; e += TB) {
          const uint32_t pid = queue[e];
          const float w = vm_pair_weight_regs(load((int)(pid >> PSH)), load((int)(pid & PMASK)), W);
          if (w > thr0) {
            bool store = mode == 0;
            if (mode != 0) {
              const int bb = bin_a(w);
              if (mode == 1) atomicAdd(&s_hist_a[bb], 1u);
              else store = bb >= lo && bb < top;
            }
            if (store) {
              const int pos = atomicAdd(&s_nlist, 1);
              if (pos < LCAP) lk[pos] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(PCOMP - pid);
            }
          }
        }
        __syncthreads();
        if (tid == 0) s_nq = 0;
        if (mode == 0 && s_nlist > LCAP) break;   // uniform: read after the barrier, written before it
        __syncthreads();
      }
    };
    sweep(0, 0, 0);
    DNP_ACC(2);
    const int nlA = s_nlist;
    __syncthreads();
    if (nlA <= LCAP) {
      sort_list(nlA);
      DNP_ACC(3);
      merge_list(nlA, true);
      DNP_ACC(4);
    } else {
      // More heavy edges than the list holds (a large neighbourhood on one smooth surface that the lazy schedule could not
      // finish): bands of descending weight as in phase B below, from a histogram of the weights above thr0.  Each band
      // costs another sweep, but the voxel's segment usually freezes in the first.
      if (tid == 0) atomicAdd(&counters[0], 1ull);
      for (int k = tid; k < DN_ABIN; k += TB) s_hist_a[k] = 0u;
      __syncthreads();
      sweep(1, 0, 0);
      int top = DN_ABIN;
      while (true) {
        if (tid == 0) {
          unsigned int acc = 0;
          int lo = top;
          while (lo > 0 && acc + s_hist_a[lo - 1] <= (unsigned int)LCAP) { --lo; acc += s_hist_a[lo]; }
          s_nb = lo;
          s_nlist = 0;
        }
        __syncthreads();
        const int lo = s_nb;
        if (lo == top) { hand_on(); handed = true; break; }   // one bin alone overflows the list
        sweep(2, lo, top);
        const int nband = s_nlist < LCAP ? s_nlist : LCAP;
        __syncthreads();
        sort_list(nband);
        merge_list(nband, lo == 0);
        if (lo == 0) break;
        top = lo;
        const float wub = thr0 + ((float)top / scale_a) * 1.0001f + 1.0e-6f;   // every edge left weighs less
        const int r0 = seg[0];
        if (!(thr[r0] < wub) || (int)ssz[r0] == m) break;   // frozen above everything that is left (fact F)
        __syncthreads();
      }
      DNP_ACC(4);
    }
    if (!handed) {
#ifdef VGS_PROF
      if (tid == 0) { atomicAdd(&g_dn_prof[10], (unsigned long long)nlA); atomicAdd(&g_dn_prof[11], 1ull); }
#endif
      // ---- 3. phase B: non-singleton segments still below thr0 merge through edges at or below thr0 ----
      const int s0 = seg[0];
      const bool s0_active = (ssz[s0] >= 2) && (thr[s0] < thr0);
      if (s0_active) {   // uniform
        if (tid == 0) s_nlist = 0;
        if (wave == 0) {
          // the vertices of the active segments, the voxel's own segment first
          int n0 = 0;
          for (int base = 0; base < m; base += 64) {
            const int v = base + lane;
            const bool own = v < m && seg[v] == s0;
            const unsigned long long mk = __ballot(own);
            if (own) alist[n0 + __popcll(mk & lt_mask)] = (idx_t)v;
            n0 += __popcll(mk);
          }
          int nb = n0;
          for (int base = 0; base < m; base += 64) {
            const int v = base + lane;
            bool act = false;
            if (v < m) { const int sv = seg[v]; act = sv != s0 && (ssz[sv] >= 2) && (thr[sv] < thr0); }
            const unsigned long long mk = __ballot(act);
            if (act) alist[nb + __popcll(mk & lt_mask)] = (idx_t)v;
            nb += __popcll(mk);
          }
          if (lane == 0) { s_nb = nb; s_nq = n0; }
        }
        __syncthreads();
        const int nb = s_nb, n0 = s_nq;
        // The voxel's segment changes only through an edge of its own heavier than its threshold L.  If no pair between
        // it and another active segment weighs more than L (and at most thr0), nothing the other segments do among
        // themselves can reach it: phase B is over before it starts.  n0 * (nb - n0) pairs instead of nb^2 / 2.
        {
          const float L = thr[s0];
          const int no = nb - n0;
          bool hit = false;
          for (int idx = tid; idx < n0 * no; idx += TB) {
            const int x = alist[idx / no], y = alist[n0 + idx % no];
            const int a = x < y ? x : y, b = x < y ? y : x;
            const NodeRec A = load(a), B = load(b);
            ++my_pairs;
            if (!(vm_weight_bound_da(A, B, W) <= L)) {
              const float w = vm_pair_weight_regs(A, B, W);
              hit = hit || (w > L && w <= thr0);
            }
          }
          if (hit) s_flag = 2;
          __syncthreads();
        }
        const bool reachable = s_flag == 2;
#ifdef VGS_PROF
        if (tid == 0 && !reachable) atomicAdd(&g_dn_prof[8], 1ull);
#endif
        if (reachable) {
        const uint32_t Pb = (uint32_t)(nb * (nb - 1) / 2);
        for (uint32_t p = (uint32_t)tid; p < Pb; p += TB) {
          int ia, ib;
          decode(p, nb, Pb, ia, ib);
          const int xa = alist[ia], xb = alist[ib];
          const int a = xa < xb ? xa : xb, b = xa < xb ? xb : xa;   // the list is not in vertex order
          if (seg[a] != seg[b]) {
            const float w = vm_pair_weight_regs(load(a), load(b), W);
            ++my_pairs;
            if (w <= thr0) {   // heavier edges were examined in phase A; NaN compares false
              const int pos = atomicAdd(&s_nlist, 1);
              if (pos < LCAP) lk[pos] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(PCOMP - (((uint32_t)a << PSH) | (uint32_t)b));
            }
          }
        }
        __syncthreads();
        const int nlB = s_nlist;
        DNP_ACC(5);
        if (nlB <= LCAP) {
          sort_list(nlB);
          DNP_ACC(6);
          merge_list(nlB, true);
          DNP_ACC(7);
#ifdef VGS_PROF
          if (tid == 0) { atomicAdd(&g_dn_prof[12], (unsigned long long)nlB); atomicAdd(&g_dn_prof[13], 1ull); }
#endif
        } else {
          // More such pairs than the list holds (large neighbourhoods whose segments all stay below thr0): the scan takes
          // them in bands of descending weight.  One pass histograms the weights (DN_NBIN bins over [0, thr0]; the
          // queue's slots are free now), then every band -- the heaviest whole bins that fit the list -- is collected by a
          // pass of its own, sorted and merged.  Pairs merged away meanwhile only make later bands shorter.  The bands end
          // when the voxel's own segment is frozen below the next band (fact F).
          if (tid == 0) atomicAdd(&counters[0], 1ull);
          uint32_t* hist = (uint32_t*)queue;
          const float scale = (float)DN_NBIN / thr0;
          auto bin_of = [&](float w) -> int { const int bb = (int)(w * scale); return bb < 0 ? 0 : (bb > DN_NBIN - 1 ? DN_NBIN - 1 : bb); };
          for (int k = tid; k < DN_NBIN; k += TB) hist[k] = 0u;
          __syncthreads();
          for (uint32_t p = (uint32_t)tid; p < Pb; p += TB) {
            int ia, ib;
            decode(p, nb, Pb, ia, ib);
            const int xa = alist[ia], xb = alist[ib];
          const int a = xa < xb ? xa : xb, b = xa < xb ? xb : xa;   // the list is not in vertex order
            if (seg[a] != seg[b]) {
              const float w = vm_pair_weight_regs(load(a), load(b), W);
              ++my_pairs;
              if (w <= thr0) atomicAdd(&hist[bin_of(w)], 1u);
            }
          }
          __syncthreads();
          int top = DN_NBIN;   // bins [top, DN_NBIN) are done
          while (true) {
            if (tid == 0) {
              unsigned int acc = 0;
              int lo = top;
              while (lo > 0 && acc + hist[lo - 1] <= (unsigned int)LCAP) { --lo; acc += hist[lo]; }
              s_nb = lo;
              s_nlist = 0;
            }
            __syncthreads();
            const int lo = s_nb;
            if (lo == top) { hand_on(); handed = true; break; }   // one bin alone overflows the list: degenerate ties
            for (uint32_t p = (uint32_t)tid; p < Pb; p += TB) {
              int ia, ib;
              decode(p, nb, Pb, ia, ib);
              const int xa = alist[ia], xb = alist[ib];
          const int a = xa < xb ? xa : xb, b = xa < xb ? xb : xa;   // the list is not in vertex order
              if (seg[a] != seg[b]) {
                const float w = vm_pair_weight_regs(load(a), load(b), W);
                ++my_pairs;
                if (w <= thr0) {
                  const int bb = bin_of(w);
                  if (bb >= lo && bb < top) {
                    const int pos = atomicAdd(&s_nlist, 1);
                    if (pos < LCAP) lk[pos] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(PCOMP - (((uint32_t)a << PSH) | (uint32_t)b));
                  }
                }
              }
            }
            __syncthreads();
            const int nband = s_nlist < LCAP ? s_nlist : LCAP;   // <= the histogram's count of these bins
            sort_list(nband);
            merge_list(nband, lo == 0);
            if (lo == 0) break;
            top = lo;
            // every pair left has a bin below `top`, so it weighs less than this
            const float wub = (float)top / scale * 1.0001f;
            const int r0 = seg[0];   // flattened by merge_list
            if (!(thr[r0] < wub) || (int)ssz[r0] == m) break;
            __syncthreads();   // s_nb / s_nlist are rewritten at the top
          }
          DNP_ACC(8);
        }
        }   // reachable
      }
    }
  }
  if (handed) return;   // k_localcut writes the row and the count
  // ---- result: the segment of the voxel itself, the whole row (nobody zeroes the table first) ----
  {
    const int s0 = seg[0];
    for (int c = tid; c < m; c += TB) crow[c] = (seg[c] == s0) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) my_pairs += __shfl_xor(my_pairs, o, 64);
  if (tid == 0) evals_out[u] = 0;
  __syncthreads();
  if (lane == 0 && my_pairs) atomicAdd(&evals_out[u], my_pairs);
#ifdef VGS_PROF
  if (tid == 0) {
    atomicAdd(&g_dn_prof[0], 1ull);
    const unsigned long long tt = (unsigned long long)(clock64() - t_begin);
    atomicMax(&g_dn_prof[14], tt);
    atomicMax(&g_dn_prof[9], (unsigned long long)(wall_clock64() - w_begin));
    if (tt > 400000ull) atomicAdd(&g_dn_prof[15], 1ull);
    const unsigned long long slot = atomicAdd(&g_dn_trace[4 * 16384], 1ull);
    if (slot < 16384ull) {
      g_dn_trace[4 * slot] = (unsigned long long)w_begin; g_dn_trace[4 * slot + 1] = (unsigned long long)wall_clock64();
      g_dn_trace[4 * slot + 2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
      g_dn_trace[4 * slot + 3] = (unsigned long long)m | ((unsigned long long)MAXM << 32);
    }
  }
#endif
  };   // process

  for (unsigned int wi = blockIdx.x; ; wi += gridDim.x) {
    unsigned int wpos = wi;
    int wbin = 0;
    while (wbin < n_lists && wpos >= n_work_dev[wbin]) { wpos -= n_work_dev[wbin]; ++wbin; }
    if (wbin == n_lists) break;
    process((uint32_t)__builtin_amdgcn_readfirstlane((int)work[(size_t)wbin * work_stride + wpos]));   // uniform, and the compiler should know
    __syncthreads();   // the next voxel reuses every array
  }
}

#endif
