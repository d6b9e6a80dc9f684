// localcut.hip -- stages a5-a7: per-voxel local affinity graph and threshold-merge cut.
// Replaces the hot loop of segmentVoxelCloudWithGraphModel (voxel_segmentation.h:376-412):
// getOneVoxelAdjacency -> buildAdjacencyGraph (VS:1796-1910) -> measuringDistance (VS:1597-1720) ->
// distanceWeight (VS:1722-1740) -> cutGraphSegmentation (VS:1913-2029); SVGS twins SS:384-413.
//
// Semantics kept exactly: edges are examined in descending weight order (ties: ascending
// k = a*n + b, a < b in adjacency order); an edge merges two segments iff w > max(seg_int - cut/size);
// the result is the segment that holds vertex 0 (the voxel itself).
// What changes is the data flow:
//   * unique pairs only (the n x n matrix is symmetric up to an ulp), unused voxels pruned when their
//     constant dead-edge weight cannot beat a singleton's threshold (checked on the host);
//   * one workgroup per voxel: neighbour records staged in LDS, weights evaluated by all lanes,
//     64-bit keys (weight bits | tie-break) sorted by an LDS bitonic network;
//   * the sequential merge runs on one wavefront but examines 64 sorted edges per step: a step that
//     finds no mergeable edge is skipped whole (the state did not change, so no edge in it can merge),
//     otherwise the first mergeable edge is applied and the scan resumes right behind it;
//   * the cut stops as soon as fewer than two segments can still merge (thresholds only fall when a
//     merge happens, and every later edge is lighter);
//   * neighbourhoods with more pairs than the LDS list holds are processed in rounds: a histogram of the
//     not-yet-examined candidate weights picks the heaviest <= CAP edges, which are collected, sorted and
//     merged before the next round (weights are recomputed instead of stored: flops are cheaper than HBM).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include <cstdio>
#include <cstdlib>

#include "vgs_context.hpp"

#ifndef LC_TB
#define LC_TB 256
#endif
#ifndef LC_SMALL_CAP
#define LC_SMALL_CAP 4096   // edge list of the hand-over kernel for up to 128 neighbours
#endif
#define LC_NBIN 2048

#define LC_TBINS 64
struct LcParams {
  VgsWeightParams W;
  float cut;
  int prune_unused;
  float d2_stop;   // squared centroid distance from which proximity alone proves w <= a singleton's threshold (+inf: never)
  // Screening table of the dense kernel: a pair of valid positions and normals with squared centroid distance in bin k
  // (k = int(d2 * ctab_scale)) and dot(n1, n2) in [-1, ctab[k]] weighs <= 1 - cut (lc_screen_table)
  float ctab_scale;
  const float* ctab;   // LC_TBINS floats on the device
  int xl_from;         // tests (VGS_DBG_XL_FROM): a kernel with an overflow list passes neighbourhoods above this on (0: its own limit only)
  int dbg_max_m;       // tests (VGS_DBG_MAXM): the pair-list kernel hands neighbourhoods above this on, as the shell classes do (0: its own limit only)
};

__device__ __forceinline__ int lc_bin1(float w) {
  int b = (int)(w * (float)LC_NBIN);
  return b < 0 ? 0 : (b > LC_NBIN - 1 ? LC_NBIN - 1 : b);
}
__device__ __forceinline__ int lc_bin2(float w, int b1) {
  float f = w * (float)LC_NBIN - (float)b1;
  int b = (int)(f * (float)LC_NBIN);
  return b < 0 ? 0 : (b > LC_NBIN - 1 ? LC_NBIN - 1 : b);
}
// monotone non-decreasing rank of a weight in [0,1]
__device__ __forceinline__ int lc_rank(float w) {
  int b1 = lc_bin1(w);
  return b1 * LC_NBIN + lc_bin2(w, b1);
}

// LDS traffic inside a single-wavefront region: make earlier LDS writes of all lanes visible and stop
// the compiler from moving accesses across (lanes run in lockstep, so no s_barrier is needed)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

#ifdef VGS_PROF
__device__ unsigned long long g_lc_prof[2][16];   // per instantiation (small, large): cycles of wavefront 0 per phase
#define LCP_T0() long long _lt0 = clock64()
#define LCP_ACC(slot) do { long long _lt1 = clock64(); if (tid == 0) atomicAdd(&g_lc_prof[MAXM > 128 ? 1 : 0][slot], (unsigned long long)(_lt1 - _lt0)); _lt0 = _lt1; } while (0)
#else
#define LCP_T0() do {} while (0)
#define LCP_ACC(slot) do {} while (0)
#endif

template <int MAXM, int CAP, bool NODES_LDS>
__global__ __launch_bounds__(LC_TB) void k_localcut(const uint32_t* __restrict__ work, int n_work, const unsigned int* __restrict__ n_work_dev,
                                                    const uint64_t* __restrict__ adj_key, const uint32_t* __restrict__ adj_cnt,
                                                    int adj_stride, const NodeRec* __restrict__ node, LcParams P,
                                                    uint8_t* __restrict__ conn, unsigned long long* __restrict__ counters,
                                                    uint32_t* __restrict__ evals_out, uint32_t* __restrict__ over_ids) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* list = (uint64_t*)smem;                 // CAP keys, descending after the sort
  uint32_t* hist = (uint32_t*)smem;                 // aliases the list between rounds
  unsigned char* q = smem + (size_t)CAP * 8;
  NodeRec* lnode = (NodeRec*)q;                     // MAXM records (NODES_LDS only)
  if (NODES_LDS) q += (size_t)MAXM * sizeof(NodeRec);
  uint32_t* gid = (uint32_t*)q; q += (size_t)MAXM * 4;   // global voxel id of compact vertex
  float* thr = (float*)q; q += (size_t)MAXM * 4;         // threshold of the segment represented by this vertex
  uint16_t* loc = (uint16_t*)q; q += (size_t)MAXM * 2;   // position in the adjacency row
  uint16_t* seg = (uint16_t*)q; q += (size_t)MAXM * 2;   // vertex -> segment representative
  uint16_t* ssize = (uint16_t*)q; q += (size_t)MAXM * 2; // size of the segment represented by this vertex
  __shared__ int s_m, s_nlist, s_rdone, s_sel, s_err, s_frozen, s_act[LC_TB / 64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // a hand-over list is launched with a fixed grid while its length is still on the device: n_work_dev, when given,
  // holds the list length and n_work is the offset of this launch in the list
  if (n_work_dev) { if ((unsigned int)n_work + blockIdx.x >= *n_work_dev) return; work += n_work; }
  else if ((int)blockIdx.x >= n_work) return;
  const uint32_t u = work[blockIdx.x];
  const int n = (int)adj_cnt[u];
  const uint64_t* row = adj_key + (int64_t)u * adj_stride;
  uint8_t* crow = conn + (int64_t)u * adj_stride;

  // ---- gather the (used) neighbours in adjacency order ----
  if (tid == 0) { s_m = 0; s_err = 0; s_frozen = 0; s_rdone = LC_NBIN * LC_NBIN; evals_out[u] = 0; }
  for (int k = tid; k < n; k += LC_TB) crow[k] = 0;
  __syncthreads();
  if (wave == 0) {
    int mm = 0;
    for (int base = 0; base < n; base += 64) {
      const int k = base + lane;
      bool keep = false;
      uint32_t t = 0;
      if (k < n) {
        t = (uint32_t)row[k];
        keep = P.prune_unused ? ((node[t].flags & VGS_F_EIG) != 0) : true;
      }
      const unsigned long long mk = __ballot(keep);
      const int pos = mm + __popcll(mk & ((1ull << lane) - 1ull));
      if (keep && pos < MAXM) { gid[pos] = t; loc[pos] = (uint16_t)k; }
      mm += __popcll(mk);
    }
    if (lane == 0) s_m = mm;
  }
  __syncthreads();
  const int m = s_m;
  if (m > MAXM || (over_ids && P.xl_from > 0 && m > P.xl_from)) {
    // outside this instantiation's limits: counted, and queued for the next larger instantiation when there is one (the host reads
    // the count back and launches it: vgs_localcut_finish); the row keeps only the self connection until then
    if (tid == 0) {
      const unsigned long long at = atomicAdd(&counters[1], 1ull);
      if (over_ids) over_ids[at] = u;
      crow[0] = 1;
    }
    return;
  }
  for (int c = tid; c < m; c += LC_TB) {
    seg[c] = (uint16_t)c;
    ssize[c] = 1;
    thr[c] = vm_cut_threshold(1.0f, P.cut, 1);
  }
  if (NODES_LDS) {
    // 64-byte records copied as 16-byte quads: 4 lanes per record
    const float4* src = (const float4*)node;
    float4* dst = (float4*)lnode;
    for (int e = tid; e < m * 4; e += LC_TB) dst[e] = src[(size_t)gid[e >> 2] * 4 + (e & 3)];
  }
  __syncthreads();

  const long long Ptot = (long long)m * (m - 1) / 2;
  unsigned long long my_pairs = 0;
  LCP_T0();
  LCP_ACC(0);   // (starts the clock; slot 0 = rows)

  // every thread walks the pairs (a < b) with stride LC_TB; f(a, b, w) sees each candidate pair whose
  // endpoints are in different segments and whose weight is not NaN (a NaN edge never merges, Q3)
  auto for_pairs = [&](auto&& f) {
    int a = 0, qq = tid;
    while (true) {
      while (a < m - 1 && qq >= m - 1 - a) { qq -= (m - 1 - a); ++a; }
      if (a >= m - 1) break;
      const int b = a + 1 + qq;
      if (seg[a] != seg[b]) {
        float w;
        if (NODES_LDS) w = vm_pair_weight(lnode[a], lnode[b], P.W);
        else w = vm_pair_weight(node[gid[a]], node[gid[b]], P.W);
        ++my_pairs;
        if (!(w != w)) f(a, b, w);
      }
      qq += LC_TB;
    }
  };

  // descending bitonic sort of list[0, nl): weight first, then ascending (a, b); every thread takes part
  auto sort_desc = [&](int nl) {
    int np = 64;
    while (np < nl) np <<= 1;
    for (int k = nl + tid; k < np; k += LC_TB) list[k] = 0ull;
    __syncthreads();
    for (int size = 2, sbit = 1; size <= np; size <<= 1, ++sbit) {
      for (int sl = sbit - 1; sl >= 0; --sl) {
        const int strd = 1 << sl;
        for (int t = tid; t < (np >> 1); t += LC_TB) {
          const int lo = ((t >> sl) << (sl + 1)) | (t & (strd - 1));
          const int hi = lo + strd;
          const bool dn = ((lo & size) == 0);
          const uint64_t x = list[lo], y = list[hi];
          if ((x < y) == dn) { list[lo] = y; list[hi] = x; }
        }
        __syncthreads();
      }
    }
  };
  // sequential merge of the sorted list on one wavefront, 64 edges per step; ends with a workgroup barrier
  auto merge_all = [&](int nl) {
    if (wave == 0) {
      int pos = 0;
      while (pos < nl) {
        const int e = pos + lane;
        bool pass = false;
        float w = 0.f;
        int sa = 0, sb = 0;
        if (e < nl) {
          const uint64_t key = list[e];
          w = vm_from_bits((uint32_t)(key >> 32));
          const uint32_t tb = 0xffffffffu - (uint32_t)key;
          sa = seg[tb >> 16];
          sb = seg[tb & 0xffffu];
          pass = (sa != sb) && (w > thr[sa]) && (w > thr[sb]);
        }
        const unsigned long long mk = __ballot(pass);
        if (mk == 0ull) { pos += 64; continue; }  // nothing in these 64 edges can merge in the current state
        const int f = __ffsll((long long)mk) - 1;  // first mergeable edge in order: it merges (state unchanged before it)
        const float wf = __shfl(w, f, 64);
        const int s1 = __shfl(sa, f, 64), s2 = __shfl(sb, f, 64);
        const float t1 = thr[s1], t2 = thr[s2];
        const int keep = (t1 >= t2) ? s1 : s2;   // VS:1972-1983: the segment with the larger threshold survives
        const int gone = (t1 >= t2) ? s2 : s1;
        const int nsz = (int)ssize[keep] + (int)ssize[gone];
        wave_sync();
        if (lane == 0) {
          ssize[keep] = (uint16_t)nsz;
          ssize[gone] = 0;
          thr[keep] = vm_cut_threshold(wf, P.cut, nsz);  // seg_int = w (VS:1988)
        }
        for (int c = lane; c < m; c += 64)
          if (seg[c] == gone) seg[c] = (uint16_t)keep;
        wave_sync();
        pos += f + 1;
        // stop when fewer than two segments can still merge at the next (and every later) weight
        const float wn = (pos < nl) ? vm_from_bits((uint32_t)(list[pos] >> 32)) : 0.f;
        // ... or when the voxel's own segment is frozen (fact F of localcut_wave.hpp): its threshold moves only when it merges,
        // every later edge weighs <= wn, so with thr >= wn no later edge joins it -- what the other segments still do is not asked
        if (pos < nl && !(thr[seg[0]] < wn)) { if (lane == 0) s_frozen = 1; break; }
        int active = 0;
        for (int c = lane; c < m; c += 64) active += (ssize[c] != 0 && thr[c] < wn) ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) active += __shfl_xor(active, o, 64);
        if (active < 2) break;
      }
    }
    __syncthreads();
  };

  // ======================= fast path (facts S and F of localcut_wave.hpp) =======================
  // 1. the voxel can only ever merge through an incident edge heavier than a singleton's threshold thr0;
  // 2. all edges heavier than thr0, found with the cheap proximity+angle bound in front of the full weight;
  // 3. the few pairs between non-singleton segments still below thr0.  Falls back to the histogram rounds below
  //    (from a clean state) if a list overflows.
  bool fast_done = false;
#ifdef VGS_PROF
  int nq_dbg = 0;
#endif
  if (m >= 2) {
    const float thr0 = vm_cut_threshold(1.0f, P.cut, 1);
    auto nd = [&](int a) -> const NodeRec& { return NODES_LDS ? lnode[a] : node[gid[a]]; };
    if (tid == 0) { s_sel = 0; s_nlist = 0; }
    __syncthreads();
    for (int x = 1 + tid; x < m; x += LC_TB) {
      const NodeRec& A = nd(0);
      const NodeRec& B = nd(x);
      const float ub = vm_weight_bound_da(A, B, P.W);
      ++my_pairs;
      if (!(ub <= thr0)) { const float w = vm_pair_weight(A, B, P.W); if (w > thr0) s_sel = 1; }
    }
    __syncthreads();
    LCP_ACC(1);   // incident edges
    if (s_sel == 0) {
      fast_done = true;  // the voxel stays alone
    } else {
      {
        // Two passes, so that the expensive full weight runs on full wavefronts: (1) every pair gets the cheap proximity +
        // angle bound, the survivors are queued (pair ids in the upper half of the list array: CAP 32-bit slots);
        // (2) the queue is evaluated densely, edges heavier than thr0 go to the list.  In clutter -- where this kernel
        // is used -- four of five pairs leave in pass 1; evaluated in place they would idle beside a lane that stays.
        uint32_t* queue = (uint32_t*)(list + CAP / 2);
        int* q_n = &s_rdone;   // free until the histogram rounds (restored below)
        if (tid == 0) *q_n = 0;
        __syncthreads();
        int a = 0, qq = tid;
        while (true) {
          while (a < m - 1 && qq >= m - 1 - a) { qq -= (m - 1 - a); ++a; }
          const bool live = a < m - 1;
          bool keep = false;
          int b = 0;
          if (live) {
            b = a + 1 + qq;
            ++my_pairs;
            keep = !(vm_weight_bound_da(nd(a), nd(b), P.W) <= thr0);
          }
          const unsigned long long mk = __ballot(keep);
          if (mk != 0ull) {
            int base = 0;
            if (lane == 0) base = atomicAdd(q_n, __popcll(mk));
            base = __shfl(base, 0, 64);
            if (keep) { const int pos = base + __popcll(mk & ((1ull << lane) - 1ull)); if (pos < CAP) queue[pos] = ((uint32_t)a << 16) | (uint32_t)b; }
          }
          if (__ballot(live) == 0ull) break;
          qq += LC_TB;
        }
        __syncthreads();
        LCP_ACC(2);   // screening pass
        const int nq = *q_n;
#ifdef VGS_PROF
        nq_dbg = nq;
#endif
        if (nq > CAP) {
          if (tid == 0) s_nlist = CAP + 1;   // queue overflow: take the general rounds
        } else {
          for (int e = tid; e < nq; e += LC_TB) {
            const uint32_t pid = queue[e];
            const int pa = (int)(pid >> 16), pb = (int)(pid & 0xffffu);
            const float w = vm_pair_weight(nd(pa), nd(pb), P.W);
            if (w > thr0) {
              const int pos = atomicAdd(&s_nlist, 1);
              // the list grows from the bottom while the queue sits in the upper half: more than CAP / 2 edges count as overflow
              if (pos < CAP / 2) list[pos] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(0xffffffffu - pid);
              else s_nlist = CAP + 1;
            }
          }
        }
        __syncthreads();
        LCP_ACC(3);   // dense evaluation
        if (tid == 0) s_rdone = LC_NBIN * LC_NBIN;
      }
      __syncthreads();
      const int nlA = s_nlist;
      __syncthreads();
      if (nlA <= CAP) {
        sort_desc(nlA);
        LCP_ACC(4);   // sort A
        merge_all(nlA);
        LCP_ACC(5);   // merge A
#ifdef VGS_PROF
        if (tid == 0) { atomicAdd(&g_lc_prof[MAXM > 128 ? 1 : 0][10], (unsigned long long)nlA); atomicAdd(&g_lc_prof[MAXM > 128 ? 1 : 0][11], (unsigned long long)nq_dbg); atomicAdd(&g_lc_prof[MAXM > 128 ? 1 : 0][12], 1ull); }
#endif
        // phase B: vertices of non-singleton segments that can still merge below thr0
        const int s0 = seg[0];
        int active = 0;
        for (int c = tid; c < m; c += LC_TB) active += (ssize[c] >= 2 && thr[c] < thr0) ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) active += __shfl_xor(active, o, 64);
        if (lane == 0) s_act[wave] = active;
        if (tid == 0) s_nlist = 0;
        __syncthreads();
        int tot = 0;
        for (int x = 0; x < LC_TB / 64; ++x) tot += s_act[x];
        const bool need_b = (tot >= 2) && (ssize[s0] >= 2) && (thr[s0] < thr0);
        __syncthreads();
        if (!need_b) {
          fast_done = true;
        } else {
          int a = 0, qq = tid;
          while (true) {
            while (a < m - 1 && qq >= m - 1 - a) { qq -= (m - 1 - a); ++a; }
            if (a >= m - 1) break;
            const int b = a + 1 + qq;
            const int sa = seg[a], sb = seg[b];
            if (sa != sb && ssize[sa] >= 2 && ssize[sb] >= 2 && thr[sa] < thr0 && thr[sb] < thr0) {
              const float w = vm_pair_weight(nd(a), nd(b), P.W);
              ++my_pairs;
              if (w <= thr0) {
                const int pos = atomicAdd(&s_nlist, 1);
                if (pos < CAP) list[pos] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(0xffffffffu - (((uint32_t)a << 16) | (uint32_t)b));
              }
            }
            qq += LC_TB;
          }
          __syncthreads();
          const int nlB = s_nlist;
          __syncthreads();
          LCP_ACC(6);   // phase B pass
          if (nlB <= CAP) {
            sort_desc(nlB);
            LCP_ACC(7);   // sort B
            merge_all(nlB);
            LCP_ACC(8);   // merge B
#ifdef VGS_PROF
            if (tid == 0) { atomicAdd(&g_lc_prof[MAXM > 128 ? 1 : 0][13], (unsigned long long)nlB); atomicAdd(&g_lc_prof[MAXM > 128 ? 1 : 0][14], 1ull); }
#endif
            fast_done = true;
          }
        }
      }
      if (!fast_done) {  // a list overflowed: start again from a clean state with the general rounds
        for (int c = tid; c < m; c += LC_TB) { seg[c] = (uint16_t)c; ssize[c] = 1; thr[c] = thr0; }
        if (tid == 0) atomicAdd(&counters[7], 1ull);
        __syncthreads();
      }
    }
  }

  for (int round = 0; m >= 2 && !fast_done; ++round) {
    const int rdone = s_rdone;  // edges with rank >= rdone have been examined
    const bool single = (round == 0 && Ptot <= (long long)CAP);
    int take_from = 0;          // this round examines ranks [take_from, rdone)
    if (!single) {
      // pass A: histogram (level 1) of the not yet examined candidate edges
      for (int b = tid; b < LC_NBIN; b += LC_TB) hist[b] = 0;
      __syncthreads();
      for_pairs([&](int, int, float w) {
        const int r = lc_rank(w);
        if (r < rdone) atomicAdd(&hist[r / LC_NBIN], 1u);
      });
      __syncthreads();
      if (tid == 0) {
        // heaviest whole bins that fit; s_sel: >= 0 rank to take from, -1 nothing left, <= -2 refine bin (-2 - s_sel)
        unsigned acc = 0;
        int sel = -1;
        for (int b = LC_NBIN - 1; b >= 0; --b) {
          const unsigned h = hist[b];
          if (h == 0) continue;
          if (acc + h <= (unsigned)CAP) { acc += h; sel = b * LC_NBIN; }
          else { if (acc == 0) sel = -2 - b; break; }
        }
        s_sel = sel;
      }
      __syncthreads();
      int sel = s_sel;
      if (sel == -1) break;  // no candidate edge left
      if (sel <= -2) {
        // pass A': the heaviest non-empty bin alone exceeds the list: histogram (level 2) inside it
        const int bt = -2 - sel;
        __syncthreads();
        for (int b = tid; b < LC_NBIN; b += LC_TB) hist[b] = 0;
        __syncthreads();
        for_pairs([&](int, int, float w) {
          const int r = lc_rank(w);
          if (r < rdone && r / LC_NBIN == bt) atomicAdd(&hist[r % LC_NBIN], 1u);
        });
        __syncthreads();
        if (tid == 0) {
          unsigned acc = 0;
          int s2 = -1;
          for (int b = LC_NBIN - 1; b >= 0; --b) {
            const unsigned h = hist[b];
            if (h == 0) continue;
            if (acc + h <= (unsigned)CAP) { acc += h; s2 = bt * LC_NBIN + b; }
            else break;
          }
          if (s2 < 0) s_err = 1;  // > CAP weights inside one 2^-22 interval: degenerate ties
          s_sel = s2;
        }
        __syncthreads();
        sel = s_sel;
        if (sel < 0) break;
      }
      take_from = sel;
      __syncthreads();
    }
    // pass B: collect the edges of this round
    if (tid == 0) s_nlist = 0;
    __syncthreads();
    for_pairs([&](int a, int b, float w) {
      bool in = single;
      if (!single) { const int r = lc_rank(w); in = (r >= take_from && r < rdone); }
      if (in) {
        const int pos = atomicAdd(&s_nlist, 1);
        if (pos < CAP) list[pos] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(0xffffffffu - (((uint32_t)a << 16) | (uint32_t)b));
      }
    });
    __syncthreads();
    const int nl = s_nlist < CAP ? s_nlist : CAP;
    sort_desc(nl);
    merge_all(nl);
    __syncthreads();
    if (single || take_from <= 0 || s_frozen) break;
    // every unexamined edge has rank < take_from, i.e. weight below wub: go on only if two segments can merge there
    const float wub = (float)(take_from + 1) / ((float)LC_NBIN * (float)LC_NBIN) * 1.0001f;
    if (!(thr[seg[0]] < wub)) break;   // the voxel's own segment is frozen (fact F)
    int active = 0;
    for (int c = tid; c < m; c += LC_TB) active += (ssize[c] != 0 && thr[c] < wub) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) active += __shfl_xor(active, o, 64);
    if (lane == 0) s_act[wave] = active;
    if (tid == 0) s_rdone = take_from;
    __syncthreads();
    int tot = 0;
    for (int x = 0; x < LC_TB / 64; ++x) tot += s_act[x];
    __syncthreads();
    if (tot < 2) break;
  }
  __syncthreads();
  LCP_ACC(9);   // general rounds (when the fast path gave up)
#ifdef VGS_PROF
  if (tid == 0) atomicAdd(&g_lc_prof[MAXM > 128 ? 1 : 0][0], 1ull << 32);
#endif
  // ---- result: the segment of vertex 0 (the voxel itself: first entry of its sorted adjacency row) ----
  if (m >= 1) {
    const uint16_t s0 = seg[0];
    for (int c = tid; c < m; c += LC_TB)
      if (seg[c] == s0) crow[loc[c]] = 1;
  }
  for (int o = 32; o > 0; o >>= 1) my_pairs += __shfl_xor(my_pairs, o, 64);
  if (lane == 0 && my_pairs) atomicAdd(&evals_out[u], (uint32_t)my_pairs);  // 4 waves, one voxel-private word
  if (tid == 0 && s_err) atomicAdd(&counters[2], 1ull);
}

#include "localcut_wave.hpp"
#include "localcut_dense.hpp"
#include "localcut_pg.hpp"
#ifndef LD_SMALL_LCAP
#define LD_SMALL_LCAP 2048
#endif
#define DN_SMALL 128, LD_SMALL_LCAP, 2048, 256   // hand-overs of the one-wavefront classes
#define DN_LARGE 512, 4096, 2048, 512   // hand-overs of classes C and D up to 512 neighbours
#ifndef PG_SMALL_OCC
#define PG_SMALL_OCC 5
#endif
#define PG_SMALL 128, 2048, 4, PG_SMALL_OCC, true   // the pair-list kernel for the hand-overs of the one-wavefront classes
#define PG_C0 320, 2048, 4, 5, false            // ... for neighbourhoods of 129-320 voxels (29 KB of LDS: five workgroups per CU)
#define PG_C 512, 2048, 4, 4, false             // ... of up to 512 (34 KB: four)
#define PG_D 1024, 4096, 8, 4, false            // ... of up to 1024 (68 KB: two workgroups of eight wavefronts)
#ifndef PG_XL_NW
#define PG_XL_NW 16   // (round 6: 4 -> 16 wavefronts: the block scene 49.5 -> 36.8 ms, config 2 4.94 -> 4.89)
#endif
#define PG_XL 4224, 2048, PG_XL_NW, 1, false, 21        // ... of whole balls of up to ten voxels (4189 offsets; 154 KB: one workgroup of sixteen wavefronts per CU)

// Connect bits of the voxels the hand-over kernels cut (k_localcut_dense, k_localcut: they write the connect row only): one wavefront
// per pending voxel turns its row into the bit-per-ball-offset form the wave kernels write themselves (localcut_wave.hpp, result).
// lists != null: workgroup b takes entry b of the concatenation of the (up to 5) hand-over lists; otherwise every pending row of [0, U)
struct CbLists { const uint32_t* ids[5]; unsigned int end[5]; };   // end[k] = number of entries in lists 0 .. k
__global__ __launch_bounds__(64) void k_conn_bits(const uint8_t* __restrict__ pending, int64_t U, const uint32_t* __restrict__ adj_cnt, int adj_stride,
                                                  const uint8_t* __restrict__ conn, const uint16_t* __restrict__ adj_off,
                                                  int cb_R, int cb_words, uint32_t* __restrict__ cbits, CbLists L, int use_lists) {
  __shared__ uint32_t cb[VGS_CB_MAX_WORDS];
  int64_t u = (int64_t)blockIdx.x;
  if (use_lists) {
    const unsigned int b = blockIdx.x;
    int k = 0;
    while (k < 4 && b >= L.end[k]) ++k;
    if (b >= L.end[k]) return;
    u = (int64_t)L.ids[k][b - (k ? L.end[k - 1] : 0u)];
  } else if (u >= U || !pending[u]) return;
  const int lane = threadIdx.x;
  for (int k = lane; k < cb_words; k += 64) cb[k] = 0u;
  __syncthreads();
  const int n = (int)adj_cnt[u];
  const uint8_t* crow = conn + u * adj_stride;
  const uint16_t* orow = adj_off + u * adj_stride;
  if (orow[0] != 0xffffu)
    for (int c = lane; c < n; c += 64)
      if (crow[c]) { const uint32_t idx = vgs_cb_index(orow[c], cb_R); atomicOr(&cb[idx >> 5], 1u << (idx & 31u)); }
  __syncthreads();
  for (int k = lane; k < cb_words; k += 64) cbits[(size_t)u * (size_t)cb_words + k] = cb[k];
}

// The hand-over lists of the one-wavefront classes, built from the marks their kernels left (pending[u] = 1 + list + LW_HO_BINS * why,
// localcut_wave.hpp): voxel order, one atomic per list and 1024 voxels; the reasons are counted into the schedule counters on the way.
// The marks become plain "pending" flags for the merge stage.
__global__ __launch_bounds__(256) void k_ho_lists(uint8_t* __restrict__ pending, int64_t U, uint32_t* __restrict__ ids, int64_t stride,
                                                  unsigned int* __restrict__ n_lists, unsigned long long* __restrict__ counters,
                                                  unsigned int* __restrict__ gate /* [0] the word of LcGate, [1] ticket */, unsigned int many) {
  // a thread takes four consecutive voxels (one 32-bit load of their marks); slot k of the thread = voxel 4 * t + k
  __shared__ unsigned int s_cnt[4][4][LW_HO_BINS];   // [wave][slot][list]: count, then offset inside the workgroup's piece of the list
  __shared__ unsigned int s_base[LW_HO_BINS];
  __shared__ unsigned int s_why[LW_N_WHY];
  if (threadIdx.x < LW_N_WHY) s_why[threadIdx.x] = 0u;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t marks = 0u;
  if (4 * t + 3 < U) marks = ((const uint32_t*)pending)[t];
  else for (int k = 0; k < 4; ++k) if (4 * t + k < U) marks |= (uint32_t)pending[4 * t + k] << (8 * k);
  int bin[4], why[4];
  for (int k = 0; k < 4; ++k) {
    const int p = (int)((marks >> (8 * k)) & 0xffu);
    bin[k] = -1; why[k] = -1;
    if (p != 0 && p != LW_PENDING_LISTED) { bin[k] = (p - 1) % LW_HO_BINS; why[k] = (p - 1) / LW_HO_BINS; }
  }
  // (byte stores, and only of the marks taken: the multi-wavefront classes may still be running and set their own voxels' bytes)
  for (int k = 0; k < 4; ++k) if (bin[k] >= 0) pending[4 * t + k] = (uint8_t)LW_PENDING_LISTED;
  __syncthreads();
  unsigned long long mk[4][LW_HO_BINS];
  for (int k = 0; k < 4; ++k)
    for (int b = 0; b < LW_HO_BINS; ++b) {
      mk[k][b] = __ballot(bin[k] == b);
      if (lane == 0) s_cnt[wave][k][b] = (unsigned int)__popcll(mk[k][b]);
    }
  for (int y = 0; y < LW_N_WHY; ++y) {
    unsigned int n = 0;
    for (int k = 0; k < 4; ++k) n += (unsigned int)__popcll(__ballot(why[k] == y));
    if (lane == 0 && n) atomicAdd(&s_why[y], n);
  }
  __syncthreads();
  if (threadIdx.x < LW_HO_BINS) {
    const int b = threadIdx.x;
    unsigned int tot = 0;
    for (int w = 0; w < 4; ++w) for (int k = 0; k < 4; ++k) { const unsigned int x = s_cnt[w][k][b]; s_cnt[w][k][b] = tot; tot += x; }
    s_base[b] = tot ? atomicAdd(&n_lists[b], tot) : 0u;
  }
  __syncthreads();
  // (inside a workgroup's piece the order is wave, slot, lane -- not quite voxel order: nobody depends on it)
  for (int k = 0; k < 4; ++k)
    if (bin[k] >= 0) ids[(int64_t)bin[k] * stride + s_base[bin[k]] + s_cnt[wave][k][bin[k]] + __popcll(mk[k][bin[k]] & ((1ull << lane) - 1ull))] = (uint32_t)(4 * t + k);
  if (threadIdx.x < LW_N_WHY && s_why[threadIdx.x] != 0u) {
    const int word[LW_N_WHY] = {3, 62, -1};   // the lazy schedule gave the voxel up (whatever the reason), voted over, (size: not counted)
    if (word[threadIdx.x] >= 0) atomicAdd(&counters[word[threadIdx.x]], (unsigned long long)s_why[threadIdx.x]);
  }
  // the last workgroup through sees every list's length complete (the counts travel in device-scope atomics: no fence) and says which way
  // the hand-overs go (LcGate)
  if (threadIdx.x == 0 && __hip_atomic_fetch_add(&gate[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
    unsigned int tot = 0;
    for (int k = 0; k < LW_HO_BINS; ++k) tot += __hip_atomic_load(&n_lists[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&gate[0], tot > many ? LC_MANY : LC_FEW, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// split the used voxels into classes by the number of neighbours; order inside a class follows the voxel order.
// Class A (the bulk) is split once more: voxels with few heavy near pairs of their own (short near-pair list) are the
// ones the lazy schedule works on for long or gives up on, so they form class A1, the part of the bulk launch that is
// dealt out first.  (Any voxel may go to any class: the split only schedules.)
// One global atomic per class and 1024 voxels (same-address atomics serialise).
#define LC_NCLASS 7   // A, B, C, D, A1, C0 (the part of class C up to max_c0 neighbours: a smaller LDS footprint, one more workgroup per CU), S (samples of A, A1, B: LwParams::ho_bins)
__global__ __launch_bounds__(1024) void k_classify(const uint32_t* __restrict__ adj_mused, const uint32_t* __restrict__ adj_cnt, int64_t U, int prune,
                                                   int max_a, int max_b, int max_c, const uint32_t* __restrict__ used_ids,
                                                   const uint8_t* __restrict__ nl_cnt, const uint32_t* __restrict__ nl_tot, int a1_max, uint32_t* __restrict__ ids_a,
                                                   uint32_t* __restrict__ ids_b, uint32_t* __restrict__ ids_c, uint32_t* __restrict__ ids_d,
                                                   uint32_t* __restrict__ ids_a1, unsigned int* __restrict__ n_abc, int max_c0, uint32_t* __restrict__ ids_c0,
                                                   int sample, uint32_t* __restrict__ ids_s) {
  __shared__ unsigned int s_cnt[16][LC_NCLASS];   // per wavefront and class: count, then base
  __shared__ unsigned int s_base[LC_NCLASS];
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int cls = -1;
  if (u < U) {
    const int m = (int)(prune ? adj_mused[u] : adj_cnt[u]);
    cls = m <= max_a ? 0 : (m <= max_b ? 1 : (m <= max_c0 ? 5 : (m <= max_c ? 2 : 3)));
    if (sample && cls <= 1 && (u & (int64_t)(sample - 1)) == 0) cls = 6;   // every sample-th voxel of the one-wavefront classes (k_count_classes counts alike)
    if (cls == 0 && nl_cnt && nl_cnt[used_ids[u]] != NL_NONE && (int)nl_tot[used_ids[u]] <= a1_max) cls = 4;   // a voxel without a list stays in A   // every sixteenth voxel of the one-wavefront classes (before the A1 split: k_count_classes counts alike)
  }
  unsigned long long mk[LC_NCLASS];
  for (int k = 0; k < LC_NCLASS; ++k) {
    mk[k] = __ballot(cls == k);
    if (lane == 0) s_cnt[wave][k] = (unsigned int)__popcll(mk[k]);
  }
  __syncthreads();
  if (threadIdx.x < LC_NCLASS) {
    const int k = threadIdx.x;
    unsigned int tot = 0;
    for (int w = 0; w < 16; ++w) { const unsigned int x = s_cnt[w][k]; s_cnt[w][k] = tot; tot += x; }
    s_base[k] = tot ? atomicAdd(&n_abc[k], tot) : 0u;
  }
  __syncthreads();
  uint32_t* const outs[LC_NCLASS] = {ids_a, ids_b, ids_c, ids_d, ids_a1, ids_c0, ids_s};
  for (int k = 0; k < LC_NCLASS; ++k)
    if (cls == k) outs[k][s_base[k] + s_cnt[wave][k] + __popcll(mk[k] & ((1ull << lane) - 1ull))] = (uint32_t)u;
}

// Class sizes alone (A and A1 together): what the host needs to size the launches does not depend on the near-pair lists, so it is
// counted BEFORE they are built and fetched while they are (vgs_stage_localcut).
__global__ __launch_bounds__(1024) void k_count_classes(const uint32_t* __restrict__ adj_cnt, int64_t U, int max_a, int max_b, int max_c, int max_c0,
                                                        unsigned int* __restrict__ n_cls /* 7: A + A1, B, C, D, C0, S, of D those above max_d */, int sample,
                                                        int max_d) {
  __shared__ unsigned int s_cnt[7];
  if (threadIdx.x < 7) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int cls = -1;
  bool xl = false;
  if (u < U) { const int m = (int)adj_cnt[u]; cls = m <= max_a ? 0 : (m <= max_b ? 1 : (m <= max_c0 ? 4 : (m <= max_c ? 2 : 3))); xl = m > max_d; if (sample && cls <= 1 && (u & (int64_t)(sample - 1)) == 0) cls = 5; }
  for (int k = 0; k < 7; ++k) {
    const unsigned long long mk = __ballot(k < 6 ? cls == k : xl);
    if ((threadIdx.x & 63) == 0 && mk) atomicAdd(&s_cnt[k], (unsigned int)__popcll(mk));
  }
  __syncthreads();
  if (threadIdx.x < 7 && s_cnt[threadIdx.x]) atomicAdd(&n_cls[threadIdx.x], s_cnt[threadIdx.x]);
}

// Smallest squared distance (to the bisection's resolution) whose weight bound is at or below a singleton's threshold:
// vm_weight_bound_d is monotone and host and device evaluate it alike (DevMath), so every pair at least this far apart
// weighs <= bound(d2) <= bound(d2_stop) <= 1 - cut.
static float lc_d2_stop(const VgsWeightParams& W, float cut, float d2_all) {
  const float thr0 = vm_cut_threshold(1.0f, cut, 1);
  if (!(vm_weight_bound_d(d2_all, W) <= thr0)) return __builtin_huge_valf();
  float lo = 0.0f, hi = d2_all;
  for (int it = 0; it < 24; ++it) {
    const float mid = 0.5f * (lo + hi);
    if (vm_weight_bound_d(mid, W) <= thr0) hi = mid; else lo = mid;
  }
  return hi;
}

// Screening table (LcParams::ctab).  A pair in bin k is at least d_k = sqrt(k / scale) apart; A_k is an angle with
// bound_sa(d_k, A_k) <= thr0 (bisection on the float formula that vm_weight_bound_da ends in, monotone in both distances;
// the 2e-6 on the threshold covers vm_exp's last-bit wobble), and ctab[k] a cosine no angle below A_k + 2e-6 reaches
// (vm_acos is within an ulp of acos, 2.4e-7 at pi).  So dot <= ctab[k] -- with dot >= -1, or vm_acos gives NaN and the bound
// says nothing -- means dist_angle >= A_k, hence w <= bound_da(pair) <= bound_sa(d_k, A_k) <= thr0.  ctab[k] = -2: no angle
// settles bin k.
static void lc_screen_table(const VgsWeightParams& W, float cut, float d2_stop, float* ctab, float* ctab_scale) {
  const float thr0 = vm_cut_threshold(1.0f, cut, 1) * (1.0f - 2.0e-6f);
  const bool bounded = d2_stop < __builtin_huge_valf();
  const double width = bounded ? (double)d2_stop / LC_TBINS : 0.0;
  *ctab_scale = bounded ? (float)(1.0 / width) : 0.0f;   // unbounded: every pair reads bin 0 (distance 0)
  for (int k = 0; k < LC_TBINS; ++k) {
    // d2 * scale >= k in float => d2 >= k * width up to rounding: a relative 1e-6 below
    const float dk = vm_sqrt((float)((double)k * width * (1.0 - 1.0e-6)));
    float ck = -2.0f;
    if (vm_weight_bound_sa(dk, 3.1415925f, W) <= thr0) {
      float lo = 0.0f, hi = 3.1415925f;
      if (vm_weight_bound_sa(dk, 0.0f, W) <= thr0) hi = 0.0f;
      else
        for (int it = 0; it < 30; ++it) {
          const float mid = 0.5f * (lo + hi);
          if (vm_weight_bound_sa(dk, mid, W) <= thr0) hi = mid; else lo = mid;
        }
      ck = (float)(cos((double)hi + 2.0e-6) - 1.0e-6);
      if (!((double)hi + 2.0e-6 < 3.14159)) ck = -2.0f;
    }
    ctab[k] = ck;
  }
}

static VgsWeightParams make_weight_params(const vgs_params& p) {
  VgsWeightParams W;
  W.inv_sig_p = 1.0f / p.sig_p; W.inv_sig_n = 1.0f / p.sig_n; W.inv_sig_o = 1.0f / p.sig_o;
  W.inv_sig_e = 1.0f / p.sig_e; W.inv_sig_c = 1.0f / p.sig_c;
  W.inv_sig_w2 = 1.0f / (p.sig_w * p.sig_w);
  W.svgs = (p.method == 3) ? 1 : 0;
  return W;
}

// include/vgs.h: the screening table of the dense hand-over kernels for a parameter set (host arithmetic only)
vgs_status vgs_screen_table(const vgs_params* p, float* d2_stop, float* ctab_scale, float* ctab) {
  if (!p || !d2_stop || !ctab_scale || !ctab) return VGS_E_ARG;
  const VgsWeightParams W = make_weight_params(*p);
  const float reach = 2.0f * p->graph_size + 4.0f * p->voxel_size;   // as vgs_stage_localcut: no pair of a neighbourhood is farther apart
  *d2_stop = lc_d2_stop(W, p->cut_thred, reach * reach * 1.01f);
  lc_screen_table(W, p->cut_thred, *d2_stop, ctab, ctab_scale);
  return VGS_OK;
}

template <int MAXM, int CAP, bool NODES_LDS>
static size_t lc_smem_bytes() {
  return (size_t)CAP * 8 + (NODES_LDS ? (size_t)MAXM * sizeof(NodeRec) : 0) + (size_t)MAXM * (4 + 4 + 2 + 2 + 2) + 64;
}

#ifdef VGS_PROF
static DevBuf<uint32_t> s_dbg;   // per-voxel cycle / round counters of the diagnostics build
#endif

vgs_status vgs_stage_localcut(vgs_ctx* c) {
  const int64_t U = c->U;
  c->counts[VGS_N_PAIRS] = 0;
  if (c->lc_tail.open) {
    // a previous run of this stage was never completed by the merge stage (an error in between): its hand-over kernels may
    // still be running on the side streams, and they use the counters and lists this run is about to reset
    VGS_HIP_TRY(c, hipStreamSynchronize(c->stream2)); VGS_HIP_TRY(c, hipStreamSynchronize(c->stream3)); VGS_HIP_TRY(c, hipStreamSynchronize(c->stream4));
    c->lc_tail.open = false;
  }
  if (U == 0) return VGS_OK;
  LcParams LP;
  LP.W = make_weight_params(c->P);
  LP.cut = c->P.cut_thred;
  // an edge that touches an unused voxel carries the constant weight of five distances of 100 (VS:1602-1606);
  // if that cannot beat a singleton's threshold the unused voxels are inert and are pruned (exact)
  // (the adjacency stage already dropped the unused neighbours from the rows when that is the case)
  if (vgs_unused_are_inert(c->P) != c->adj_pruned) { c->err = "adjacency rows do not match the current sigma/cut parameters"; return VGS_E_STATE; }
  LP.prune_unused = 0;
  LP.xl_from = c->K.dbg_xl_from;
  LP.dbg_max_m = c->K.dbg_max_m;
  LP.d2_stop = __builtin_huge_valf();   // set below, once the neighbourhood's reach is known
  LP.ctab = nullptr; LP.ctab_scale = 0.0f;

  VGS_HIP_TRY(c, c->conn.ensure(2 * (size_t)U * c->adj_stride));  // [0,U*stride) connect flags, second half: mutual flags (merge stage)
  VGS_HIP_TRY(c, c->work_ids.ensure((12 + LW_HO_BINS) * (size_t)U + 16));
  VGS_HIP_TRY(c, c->evals.ensure((size_t)U));  // per-voxel evaluation counters (index u), summed on request (vgs_get_counts)
  VGS_HIP_TRY(c, c->counters.ensure(128));   // words 64-127: the one-wavefront classes' samples (LwParams::vote)
  VGS_HIP_TRY(c, hipMemsetAsync(c->counters.p, 0, 128 * sizeof(uint64_t), c->stream));
  {
    // pair lists (pairlist.hpp): state reset here, rows built when a class asks for them
    vgs_status sp = vgs_pairlists_begin(c, c->stream);
    if (sp != VGS_OK) return sp;
  }
  uint32_t* ids_a = c->work_ids.p;            // m <= WAVE_A: one wavefront per voxel, records in LDS, small footprint
  uint32_t* ids_b = c->work_ids.p + U;        // m <= WAVE_B: one wavefront per voxel, records in LDS
  uint32_t* ids_c = c->work_ids.p + 2 * U;    // m <= WAVE_C: one wavefront per voxel, centroids in LDS, records through L2
  uint32_t* ids_d = c->work_ids.p + 3 * U;    // the rest: one workgroup per voxel (k_localcut)
  uint32_t* ids_f = c->work_ids.p + 8 * U;    // handed over by the A/B wave kernels (m <= WAVE_B): LW_HO_BINS lists of U slots, by neighbourhood size
  uint32_t* ids_g = c->work_ids.p + 5 * U;    // handed over by the C wave kernel
  uint32_t* ids_a1 = c->work_ids.p + 6 * U;   // class A voxels with a short near-pair list of their own: they run first
  uint32_t* ids_c0 = c->work_ids.p + (8 + LW_HO_BINS) * U;   // class C0: WAVE_B < m <= WAVE_C0
  uint32_t* ids_f2 = c->work_ids.p + 7 * U;   // sent on by the dense hand-over kernel (a list overflowed)
  uint32_t* ids_s = c->work_ids.p + (9 + LW_HO_BINS) * U;   // samples of the one-wavefront classes (LwParams::ho_bins)
  unsigned int* d_nabc = (unsigned int*)(c->counters.p + 8);   // 5 class counters (words 8-10)
  unsigned int* d_nf = (unsigned int*)(c->counters.p + 14);        // lengths of the LW_HO_BINS lists (words 14-15)
  unsigned int* d_ng = (unsigned int*)(c->counters.p + 11) + 1;    // word 11, upper half
  unsigned int* d_nf2 = (unsigned int*)(c->counters.p + 12);       // sent on by the dense kernels: word 12, small | large
  unsigned int* d_gate = (unsigned int*)(c->counters.p + 57);      // LcGate's word (low half) and k_ho_lists' ticket (high half)
  unsigned int* d_ng2 = d_nf2 + 1;
  uint32_t* ids_g2 = c->work_ids.p + 4 * U;   // sent on by the large dense kernel
  static_assert(LW_HO_BINS == 4, "four 32-bit list lengths in counter words 14-15");

  constexpr int WAVE_A = 96, WAVE_B = 128, WAVE_C = 512;
#ifndef LW_LCAP_C
#define LW_LCAP_C 2048
#endif
  constexpr int LCAP_A = 448, LCAP_B = 312, LCAP_C = LW_LCAP_C;
  constexpr unsigned int GRID_F = 16384, GRID_G = 1024;
  constexpr int NW_C = 4;  // wavefronts per voxel in class C (they share 33 KB of LDS)
  // Class C0 (round 4): the planar neighbourhoods of a ball of ten voxels hold pi * 100 = 314 voxels -- 61 % of config 2's class C are
  // at most 320 -- and class C is latency bound at the five 31.8 KB workgroups a CU holds.  The same kernel for up to 320 vertices
  // with a 2032-edge list takes 26.8 KB = 21 LDS granules: SIX per CU.
  constexpr int WAVE_C0 = 320, LCAP_C0 = 2032;
  constexpr int WAVE_D = 1024, LCAP_D = 4096, NW_D = 8;  // class D: 66 KB of LDS per voxel, two voxels per CU  // fixed grids of the hand-over launches  // A and B: exactly 5 KB of LDS per wavefront (32 wavefronts per CU)
    constexpr int SMALL_M = 128, SMALL_CAP = LC_SMALL_CAP;
  constexpr int LARGE_M = 2048, LARGE_CAP = 8192;
  // The host sizes the launches from the class sizes; only the split of the bulk class into A1 / A needs the near-pair lists.  So
  // the sizes are counted first and fetched WHILE the lists are built (0.4 ms), and the bulk launch reads its split on the device
  // (round 4: the read-back behind k_classify left the GPU idle for 50 us in front of the bulk kernel).
#ifdef VGS_PROF
  const bool early_sizes = false;   // (VGS_ONLY_CLASS edits the host's counts)
#else
  const bool early_sizes = vgs_can_split_readback(c);
#endif
  unsigned int* d_ncls = (unsigned int*)(c->counters.p + 44);   // words 44-46: A + A1, B, C, D, C0
  const int max_c0 = c->K.no_c0 ? WAVE_B : WAVE_C0;   // VGS_NO_C0: class C takes them all
  const bool dense_ = !c->K.no_dense;
  const int sample = (c->pl_enabled && dense_ && !c->K.no_vote) ? c->K.vote_period : 0;   // a power of two: one voxel in so many is a sample
  if (early_sizes) {
    hipLaunchKernelGGL(k_count_classes, dim3((unsigned)((U + 1023) / 1024)), dim3(1024), 0, c->stream, c->adj_cnt.p, U, WAVE_A, WAVE_B, WAVE_C, max_c0, d_ncls, sample, 1024);
    vgs_status sb = vgs_readback_begin(c, d_ncls, 28);
    if (sb != VGS_OK) return sb;
  }
  {
    // near-pair lists for the shells of the one-wavefront classes closest to the voxel (nearlist.hip); built on the main
    // stream before the classes are formed (the split of class A reads the list lengths) and before the side streams fork
    vgs_status sn = vgs_stage_nearlists(c);
    if (sn != VGS_OK) return sn;
  }
  const int a1_max = c->K.a1_max;
  hipLaunchKernelGGL(k_classify, dim3((unsigned)((U + 1023) / 1024)), dim3(1024), 0, c->stream, c->adj_cnt.p, c->adj_cnt.p, U,
                     0, WAVE_A, WAVE_B, WAVE_C, c->used_ids.p, c->nl_enabled ? c->nl_cnt.p : (const uint8_t*)nullptr, c->nl_tot.p, a1_max, ids_a, ids_b, ids_c,
                     ids_d, ids_a1, d_nabc, max_c0, ids_c0, sample, ids_s);
  unsigned int nabc[LC_NCLASS] = {0, 0, 0, 0, 0, 0, 0};
  unsigned int n_bulk = 0;   // A + A1
  unsigned int n_above_d = ~0u;   // neighbourhoods above 1024 voxels (~0: not counted -- the diagnostics path without the early counts)
  if (early_sizes) {
    unsigned int ncls[7] = {0, 0, 0, 0, 0, 0, 0};
    vgs_status se = vgs_readback_end(c, ncls, 28);
    if (se != VGS_OK) return se;
    n_bulk = ncls[0]; nabc[1] = ncls[1]; nabc[2] = ncls[2]; nabc[3] = ncls[3]; nabc[5] = ncls[4]; nabc[6] = ncls[5];
    nabc[0] = n_bulk; nabc[4] = 0;   // (host-side bookkeeping only: the kernel reads the split from d_nabc)
    n_above_d = ncls[6];
  } else {
    VGS_READBACK(c, nabc, d_nabc, sizeof(nabc));
    n_bulk = nabc[0] + nabc[4];
  }
#ifdef VGS_PROF
  if (c->K.only_class >= 0) {  // diagnostics: run a single class (results are incomplete)
    for (int k = 0; k < LC_NCLASS; ++k) if (k != c->K.only_class) nabc[k] = 0;
  }
#endif
  unsigned long long* cnt = (unsigned long long*)c->counters.p;
  LwParams WP;
  WP.lc = LP;
  WP.r2_graph = c->P.graph_size * c->P.graph_size;
  {
    // no two members of one neighbourhood are farther apart than this (centres within graph_size of the voxel,
    // centroids within one voxel diagonal of their centres; SVGS: centroids within graph_size)
    const float reach = 2.0f * c->P.graph_size + 4.0f * c->P.voxel_size;
    WP.d2_all = reach * reach * 1.01f;
    LP.d2_stop = lc_d2_stop(LP.W, LP.cut, WP.d2_all);
    // the screening table depends on the parameters only: rebuilt and sent when they change
    const float key[8] = {LP.W.inv_sig_p, LP.W.inv_sig_n, LP.W.inv_sig_w2, LP.cut, LP.d2_stop, (float)LP.W.svgs, 0.f, 0.f};
    VGS_HIP_TRY(c, c->lc_ctab.ensure(LC_TBINS));
    if (!c->lc_ctab_valid || memcmp(key, c->lc_ctab_key, sizeof(key)) != 0) {
      float tab[LC_TBINS];
      lc_screen_table(LP.W, LP.cut, LP.d2_stop, tab, &c->lc_ctab_scale);
      VGS_HIP_TRY(c, hipMemcpyAsync(c->lc_ctab.p, tab, sizeof(tab), hipMemcpyHostToDevice, c->stream));
      VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));   // tab is on the stack
      memcpy(c->lc_ctab_key, key, sizeof(key));
      c->lc_ctab_valid = true;
    }
    LP.ctab = c->lc_ctab.p; LP.ctab_scale = c->lc_ctab_scale;
    WP.lc = LP;
  }
  WP.shell0 = c->K.shell0;
  WP.grow = 2.25f;
  WP.cap_frac = c->K.cap_frac;
  WP.dbg_stop = c->K.dbg_stop;
  WP.max_rounds = c->K.max_rounds;
  WP.dbg_max_m = c->K.dbg_max_m;
  VGS_HIP_TRY(c, c->lc_pending.ensure((size_t)U)); VGS_HIP_TRY(c, c->lc_defer.ensure((size_t)U));
  VGS_HIP_TRY(c, hipMemsetAsync(c->lc_pending.p, 0, (size_t)U, c->stream));
  WP.pending = c->lc_pending.p;
  const bool dense = !c->K.no_dense;   // diagnostics: the general workgroup kernel takes the hand-overs (one list)
  WP.near_min_own = c->K.near_min_own;
  // connect bits for crossValidation's lattice lookup: voxel lattice (method 2), rows with lattice offsets, a ball that fits the LUT
  c->cb_enabled = c->P.method == 2 && c->adj_have_off && c->cb_words > 0 && c->cb_words <= VGS_CB_MAX_WORDS && c->adj_R <= 15 && !c->K.no_connbits;
  WP.cbits = nullptr; WP.cb_R = 0; WP.cb_words = 0;
  WP.n_first_dev = nullptr; WP.n_main_dev = nullptr;
  if (c->cb_enabled) {
    VGS_HIP_TRY(c, c->conn_bits.ensure((size_t)U * (size_t)c->cb_words));
    WP.cbits = c->conn_bits.p; WP.cb_R = c->cb_R; WP.cb_words = c->cb_words;
  }
  WP.ho_bins = (dense ? LW_HO_BINS : 1) | (sample ? LW_HO_VOTE : 0);
  WP.ho_stride = (int)U;
  {
    // Shells up to (NL_REACH voxels)^2 are complete in the near-pair lists; the margin covers centroids that float
    // rounding puts a hair outside their voxel's cube.
    const int steps = c->nl_enabled ? c->nl_reach_steps : NL_REACH;
    const float reach = (float)steps * c->P.voxel_size;
    WP.near.cnt = c->nl_cnt.p; WP.near.tot = c->nl_tot.p; WP.near.ent = c->nl_ent.p;
    // centroids may sit NL_CUBE_TOL voxels outside their cubes: (1 - 2 * NL_CUBE_TOL / steps)^2, rounded down
    WP.near.d2max = reach * reach * (steps >= 2 ? NL_D2_SLACK : 0.996f);
    WP.near.enabled = c->nl_enabled ? 1 : 0;
    WP.near.direct = (c->nl_enabled && c->nl_direct) ? 1 : 0;
  }
  // general kernel: neighbour records in LDS up to SMALL_M, from L2 beyond.  With n_dev the list length is read on the
  // device (fixed grid of nw workgroups starting at list position `offset`); otherwise nw is the length.
  auto launch_block = [&](hipStream_t strm, const uint32_t* ids, unsigned int nw, bool mid, const unsigned int* n_dev = nullptr, unsigned int offset = 0) -> vgs_status {
    if (nw == 0) return VGS_OK;
    const int arg_n = n_dev ? (int)offset : (int)nw;
    if (mid) {
      // records through L2 (NODES_LDS = false): 34 KB instead of 42 KB of LDS per workgroup, measured slightly faster
      auto kern = k_localcut<SMALL_M, SMALL_CAP, false>;
      const size_t sm = lc_smem_bytes<SMALL_M, SMALL_CAP, false>();
      VGS_HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
      hipLaunchKernelGGL(kern, dim3(nw), dim3(LC_TB), sm, strm, ids, arg_n, n_dev, c->adj_key.p, c->adj_cnt.p, c->adj_stride,
                         c->node.p, LP, c->conn.p, cnt, c->evals.p, (uint32_t*)nullptr);
    } else {
      auto kern = k_localcut<LARGE_M, LARGE_CAP, false>;
      const size_t sm = lc_smem_bytes<LARGE_M, LARGE_CAP, false>();
      VGS_HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
      hipLaunchKernelGGL(kern, dim3(nw), dim3(LC_TB), sm, strm, ids, arg_n, n_dev, c->adj_key.p, c->adj_cnt.p, c->adj_stride,
                         c->node.p, LP, c->conn.p, cnt, c->evals.p, ids_d /* overflow list: class D's own list is consumed by then */);
    }
    return VGS_OK;
  };
  static_assert(WAVE_B <= SMALL_M, "wave-kernel hand-overs must fit the mid-size workgroup kernel");
  uint32_t* dbg_buf = nullptr;
#ifdef VGS_PROF
  VGS_HIP_TRY(c, s_dbg.ensure(4 * (size_t)U));
  VGS_HIP_TRY(c, hipMemset(s_dbg.p, 0, 16 * (size_t)U));
  dbg_buf = s_dbg.p;
#endif
  VGS_HIP_TRY(c, hipEventRecord(c->ev[6], c->stream));
  // every kernel writes the whole row of its voxel (0.2 GB that need no memset); only the diagnostic early exits leave rows behind
  if (WP.dbg_stop != 0 || WP.shell0 < 0.0f) VGS_HIP_TRY(c, hipMemsetAsync(c->conn.p, 0, (size_t)U * c->adj_stride, c->stream));
  // The heavy classes (few, long-running wavefronts with a large LDS footprint) run on two side streams and are
  // launched BEFORE the bulk class: once the bulk class has filled every CU's LDS with its small workgroups a 35 KB
  // workgroup waits for a contiguous hole for milliseconds (measured: 291 class-C voxels took 8.8 ms behind class A).
  // from here on the side streams carry work of this run: a failure below must make the next run wait for them
  c->pl_enabled_at_launch = false;
  c->lc_tail.gated = false; c->lc_tail.pg_xl_queued = false;
  c->lc_tail.open = true; c->lc_tail.dense = dense; c->lc_tail.grid_f = 0; c->lc_tail.grid_g = GRID_G; c->lc_tail.tail_ms = 0.f;
  VGS_HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
  VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev[2], 0));
  VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream3, c->ev[2], 0));
  // the bulk class waits on an event that crosses queues once more, so it starts a few microseconds after the others
  VGS_HIP_TRY(c, hipEventRecord(c->ev[9], c->stream3));
  VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev[9], 0));
  // Neighbourhoods above 128 voxels from the PAIR LISTS (round 5) when they are a real share of the scene (a ball of ten voxels: all of
  // config 2): every row is built once -- 12 M weight evaluations where the shell classes did 68 M -- and the classes below read bands
  // of descending weight instead of walking distance shells and evaluating what the near-pair lists do not reach.  What the kernel
  // cannot take (a row without a list, a list that overflows at one entry per vertex) joins the hand-over list of the wide classes.
  const unsigned int n_wide = nabc[2] + nabc[3] + nabc[5];
  bool pg_wide = c->pl_enabled && c->K.pg_wide != 0 && dense && n_wide > 0 && (uint64_t)n_wide * (uint64_t)c->K.pg_wide_frac > (uint64_t)U;
  if (pg_wide) {
    VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream4, c->ev[2], 0));
    vgs_status sb = vgs_pairlists_build(c, c->stream2, nullptr, nullptr, nullptr, 0, true, LP.ctab, LP.ctab_scale, LP.d2_stop, 0, LcGate{nullptr, 0u}, true);
    if (sb != VGS_OK) return sb;
    pg_wide = c->pl_enabled;   // (the pool did not fit: the shell classes below)
  }
  if (pg_wide) {
    const PairLists PLw = {(const uint2*)c->pl_state.p, c->pl_ent.p, c->pl_state.p + (size_t)c->V * 9, c->pl_w_ring};
    const PgGeom G = {c->vox_code.p, c->P.voxel_size, (float)c->box.min[0], (float)c->box.min[1], (float)c->box.min[2], c->adj_r2};
    uint32_t* const wbits = c->cb_enabled ? c->conn_bits.p : (uint32_t*)nullptr;
    // neighbourhoods above 1024 voxels: queued by the instantiation below them for the extra-large one (whole balls of up to ten voxels:
    // 4189 offsets), in a list of its own behind the samples' 
    uint32_t* const ids_xl_pg = c->work_ids.p + (10 + LW_HO_BINS) * U;
    unsigned int* const d_nxl = (unsigned int*)(c->counters.p + 56);
    // The extra-large instantiation finds a vertex through a cube of 21^3 slots around the voxel: every offset of the ball must lie within
    // ten voxels per axis.  cb_R is the ball's largest |offset| per axis (adj_R is the bound of the loop that enumerates the ball, one or
    // two more: at graph 0.5 / voxel 0.05 it is 11 and the ball's reach is 10).  A wider ball: nobody is queued for it -- what class D's
    // instantiation cannot hold goes to the hand-over list, the dense kernel and the general kernels behind it, as before round 5.
    // ... and it is launched only when a neighbourhood above class D's 1024 voxels exists (counted with the classes; round 6): its 154 KB
    // workgroups need a CU's whole LDS, so an EMPTY launch of it waits for class C0's workgroups beside it to drain -- 2.0 ms on config 2,
    // where nobody is that large -- and holds back what is queued behind it on the builder's stream
    const bool pg_xl = c->cb_R <= 10 && !c->K.no_pg_xl && n_above_d != 0u;
    c->lc_tail.pg_xl_queued = pg_xl && nabc[3] > 0;
    VGS_HIP_TRY(c, hipEventRecord(c->ev_ho2, c->stream2));   // the rows are built
    VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream4, c->ev_ho2, 0));
    // the largest neighbourhoods first on the builder's stream, the bulk of them (up to 320 voxels) beside them on the other
    if (nabc[3] > 0)
      hipLaunchKernelGGL((k_localcut_pg<PG_D>), dim3(vgs_xcd_grid(nabc[3])), dim3(512), 0, c->stream2, ids_d, 0, 1, (const unsigned int*)nullptr, nabc[3], 1, c->adj_key.p,
                         c->adj_cnt.p, c->adj_stride, c->adj_off.p, c->node.p, LP, PLw, G, c->conn.p, cnt, ids_g, d_ng, c->evals.p, c->lc_pending.p, wbits, c->cb_R,
                         c->cb_words, LcGate{nullptr, 0u}, pg_xl ? ids_xl_pg : (uint32_t*)nullptr, pg_xl ? d_nxl : (unsigned int*)nullptr);
    if (nabc[3] > 0 && pg_xl)   // what was too big for it: whole balls of up to ten voxels (the length of the list is on the device)
      hipLaunchKernelGGL((k_localcut_pg<PG_XL>), dim3(256), dim3(64 * PG_XL_NW), 0, c->stream2, ids_xl_pg, 0, 1, d_nxl, 0u, 0, c->adj_key.p,
                         c->adj_cnt.p, c->adj_stride, c->adj_off.p, c->node.p, LP, PLw, G, c->conn.p, cnt, ids_g, d_ng, c->evals.p, c->lc_pending.p, wbits, c->cb_R,
                         c->cb_words, LcGate{nullptr, 0u}, (uint32_t*)nullptr, (unsigned int*)nullptr);
    if (nabc[2] > 0)
      hipLaunchKernelGGL((k_localcut_pg<PG_C>), dim3(vgs_xcd_grid(nabc[2])), dim3(256), 0, c->stream2, ids_c, 0, 1, (const unsigned int*)nullptr, nabc[2], 1, c->adj_key.p,
                         c->adj_cnt.p, c->adj_stride, c->adj_off.p, c->node.p, LP, PLw, G, c->conn.p, cnt, ids_g, d_ng, c->evals.p, c->lc_pending.p, wbits, c->cb_R,
                         c->cb_words, LcGate{nullptr, 0u}, (uint32_t*)nullptr, (unsigned int*)nullptr);
    if (nabc[5] > 0)
      hipLaunchKernelGGL((k_localcut_pg<PG_C0>), dim3(vgs_xcd_grid(nabc[5])), dim3(256), 0, c->stream4, ids_c0, 0, 1, (const unsigned int*)nullptr, nabc[5], 1, c->adj_key.p,
                         c->adj_cnt.p, c->adj_stride, c->adj_off.p, c->node.p, LP, PLw, G, c->conn.p, cnt, ids_g, d_ng, c->evals.p, c->lc_pending.p, wbits, c->cb_R,
                         c->cb_words, LcGate{nullptr, 0u}, (uint32_t*)nullptr, (unsigned int*)nullptr);
    VGS_HIP_TRY(c, hipEventRecord(c->ev[12], c->stream4));
    VGS_HIP_TRY(c, hipEventRecord(c->ev[5], c->stream2));
    VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev[12], 0));
    // what they could not take: the dense kernel of the wide classes, as for the shell classes' hand-overs
    hipLaunchKernelGGL((k_localcut_dense<DN_LARGE>), dim3(n_wide < 4 * GRID_G ? n_wide : 4 * GRID_G), dim3(512), 0, c->stream2, ids_g, 0, 1, d_ng, c->adj_key.p,
                       c->adj_cnt.p, c->adj_stride, c->node.p, LP, c->conn.p, cnt, ids_g2, d_ng2, c->evals.p, LcGate{nullptr, 0u}, (uint32_t*)nullptr, (unsigned int*)nullptr);
  } else {
    // class D (more than 512 neighbours) on its own stream: eight wavefronts per voxel up to 1024 neighbours; beyond
    // that the kernel hands the voxel over (list g) to the workgroup kernel with its histogram rounds
    if (nabc[3] + nabc[5] > 0) {
      VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream4, c->ev[2], 0));
      // class C0 shares class D's side stream (D is a handful of voxels): it runs beside class C, its hand-overs join C's list
      if (nabc[5] > 0)
        hipLaunchKernelGGL((k_localcut_wave<WAVE_C0, LCAP_C0, NW_C>), dim3(((nabc[5] + 7) / 8) * 8), dim3(64 * NW_C), 0, c->stream4, (const uint32_t*)nullptr, 0,
                           ids_c0, (int)nabc[5], (const unsigned int*)nullptr, c->adj_key.p, c->adj_cnt.p, c->adj_stride, c->node.p, WP, c->conn.p, cnt,
                           ids_g, d_ng, c->evals.p, dbg_buf, c->adj_have_off ? c->adj_off.p : (const uint16_t*)nullptr);
      if (nabc[3] > 0)
      hipLaunchKernelGGL((k_localcut_wave<WAVE_D, LCAP_D, NW_D>), dim3(((nabc[3] + 7) / 8) * 8), dim3(64 * NW_D), 0, c->stream4, (const uint32_t*)nullptr, 0,
                         ids_d, (int)nabc[3], (const unsigned int*)nullptr, c->adj_key.p, c->adj_cnt.p, c->adj_stride, c->node.p, WP, c->conn.p, cnt,
                         ids_g, d_ng, c->evals.p, dbg_buf, c->adj_have_off ? c->adj_off.p : (const uint16_t*)nullptr);
      VGS_HIP_TRY(c, hipEventRecord(c->ev[12], c->stream4));
    }
    if (nabc[2] > 0)
      hipLaunchKernelGGL((k_localcut_wave<WAVE_C, LCAP_C, NW_C>), dim3(((nabc[2] + 7) / 8) * 8), dim3(64 * NW_C), 0, c->stream2, (const uint32_t*)nullptr, 0, ids_c, (int)nabc[2], (const unsigned int*)nullptr,
                         c->adj_key.p, c->adj_cnt.p, c->adj_stride, c->node.p, WP, c->conn.p, cnt, ids_g, d_ng, c->evals.p, dbg_buf, c->adj_have_off ? c->adj_off.p : (const uint16_t*)nullptr);
    VGS_HIP_TRY(c, hipEventRecord(c->ev[5], c->stream2));   // class C done (its hand-overs follow)
    if (nabc[3] + nabc[5] > 0) VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev[12], 0));   // the hand-over launch below follows C, C0 and D
    // class C / D hand-overs: fixed grid, length read on the device (no host round trip)
    vgs_status st = VGS_OK;
    const unsigned int ncd = nabc[2] + nabc[3] + nabc[5];
    if (ncd > 0) {
      if (!dense) st = launch_block(c->stream2, ids_g, ncd < GRID_G ? ncd : GRID_G, false, d_ng, 0);
      else   // the grid strides over the list: two voxels per CU at a time, a few rounds of them
        hipLaunchKernelGGL((k_localcut_dense<DN_LARGE>), dim3(ncd < 4 * GRID_G ? ncd : 4 * GRID_G), dim3(512), 0, c->stream2, ids_g, 0, 1, d_ng, c->adj_key.p,
                           c->adj_cnt.p, c->adj_stride, c->node.p, LP, c->conn.p, cnt, ids_g2, d_ng2, c->evals.p, LcGate{nullptr, 0u}, (uint32_t*)nullptr, (unsigned int*)nullptr);
    }
    if (st != VGS_OK) return st;
  }
  VGS_HIP_TRY(c, hipEventRecord(c->ev[3], c->stream2));
  // the samples of the one-wavefront classes (LwParams::ho_bins) first: the others read what they found
  // the one-wavefront classes sort one-word keys (regsort::sort_desc32) when every weight of phase A fits their window: thr0 >= 0.5, i.e.
  // cut <= 0.5 -- both task files; another instantiation with the 64-bit network otherwise (VGS_NO_SORT32: always)
  const bool sort32 = LW_SORT32 != 0 && !c->K.no_sort32 && vm_cut_threshold(1.0f, LP.cut, 1) >= 0.5f;
#define LW_LAUNCH_1W(MAXM_, LCAP_, SAMPLED_, GRID, STRM, ...)                                                                                        \
  do {                                                                                                                                             \
    if (sort32) hipLaunchKernelGGL((k_localcut_wave<MAXM_, LCAP_, 1, SAMPLED_, true>), GRID, dim3(64), 0, STRM, __VA_ARGS__);                        \
    else hipLaunchKernelGGL((k_localcut_wave<MAXM_, LCAP_, 1, SAMPLED_, false>), GRID, dim3(64), 0, STRM, __VA_ARGS__);                             \
  } while (0)
  if (nabc[6] > 0)
    LW_LAUNCH_1W(WAVE_B, LCAP_B, true, dim3(vgs_xcd_grid(nabc[6])), c->stream3, (const uint32_t*)nullptr, 0,
                       ids_s, (int)nabc[6], (const unsigned int*)nullptr, c->adj_key.p, c->adj_cnt.p, c->adj_stride, c->node.p, WP, c->conn.p, cnt, ids_f, d_nf, c->evals.p, dbg_buf, c->adj_have_off ? c->adj_off.p : (const uint16_t*)nullptr);
  // classes A and B have the same LDS footprint, so their workgroups interleave freely; B (heavier) goes first
  if (nabc[1] > 0)
    LW_LAUNCH_1W(WAVE_B, LCAP_B, false, dim3(vgs_xcd_grid(nabc[1])), c->stream3, (const uint32_t*)nullptr, 0,
                       ids_b, (int)nabc[1], (const unsigned int*)nullptr, c->adj_key.p, c->adj_cnt.p, c->adj_stride, c->node.p, WP, c->conn.p, cnt, ids_f, d_nf, c->evals.p, dbg_buf, c->adj_have_off ? c->adj_off.p : (const uint16_t*)nullptr);
  VGS_HIP_TRY(c, hipEventRecord(c->ev[8], c->stream3));
  VGS_HIP_TRY(c, hipEventRecord(c->ev[10], c->stream));
  // class A1 (the voxels the lazy schedule is likely to work on for long, or give up on) is the launch's first list: the
  // long-running wavefronts start first, the light ones fill the tail.  (Running A1 as a launch of its own with its
  // hand-overs on a side stream was measured: the 34 KB workgroups of the hand-over kernel starve beside the bulk.  So was
  // a 1024-edge list for A1: 40 % fewer hand-overs, but the step gets slower.)
  if (n_bulk > 0) {
    LwParams WPA = WP;
    unsigned int grid_a = vgs_xcd_grid(nabc[4]) + vgs_xcd_grid(nabc[0]);
    if (early_sizes) {   // the two list lengths are on the device: roundup8(a) + roundup8(b) <= roundup8(a + b) + 8
      WPA.n_first_dev = d_nabc + 4; WPA.n_main_dev = d_nabc + 0;
      grid_a = vgs_xcd_grid(n_bulk) + 8u;
    }
    LW_LAUNCH_1W(WAVE_A, LCAP_A, false, dim3(grid_a), c->stream, ids_a1, (int)nabc[4],
                       ids_a, (int)nabc[0], (const unsigned int*)nullptr, c->adj_key.p, c->adj_cnt.p, c->adj_stride, c->node.p, WPA, c->conn.p, cnt, ids_f, d_nf, c->evals.p, dbg_buf, c->adj_have_off ? c->adj_off.p : (const uint16_t*)nullptr);
#undef LW_LAUNCH_1W
  }
  VGS_HIP_TRY(c, hipEventRecord(c->ev[11], c->stream));
  // Hand-overs of classes A/B go to the workgroup kernel: fixed grid, list length read on the device (no host round
  // trip before the launch); the host checks the length afterwards (vgs_localcut_finish).  The kernel runs on the side
  // stream of class B, behind the bulk, and is NOT waited for here: the merge stage starts crossValidation on the rows that
  // touch no handed-over voxel meanwhile (its small workgroups leave the LDS to the 34 KB workgroups of this kernel).
  // (A second pass through the wave kernel with a 1024-edge list and 16 rounds was measured: it costs as much as the
  // workgroup kernel and still hands half of them over.)
  const unsigned int nab = nabc[0] + nabc[1] + nabc[4] + nabc[6];
  // about 1.4 % of the A/B voxels are handed over on the urban scenes; idle workgroups of this kernel are not free
  unsigned int grid_f = std::min<unsigned int>(nab, std::min<unsigned int>(GRID_F, nab / 32 + 256));
  // Not more workgroups than the device holds at once (LD_SMALL_WG_PER_CU per CU; the pair-list kernel's fit as many): they are all resident
  // when the merge stage's first kernels start on the main stream a few microseconds later, and keep their CUs' LDS and registers until the
  // lists are done.  With a workgroup per row the one-wavefront workgroups of k_cross and k_union_mutual take every slot a finished
  // workgroup leaves -- a four-wavefront workgroup with 30 KB needs a slot on every SIMD at once and never finds one, whatever its stream's
  // priority: on URB10M the kernel took 1.05 ms beside them and 0.41 ms alone (round 6 timeline).
  if (c->K.ho_grid != 0) {
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
    const unsigned int fit = c->K.ho_grid > 0 ? (unsigned int)c->K.ho_grid : (unsigned int)n_cu * LD_SMALL_WG_PER_CU;
    grid_f = std::min(grid_f, fit);
  }
  {
    VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream3, c->ev[11], 0));
    if (dense && nab > 0)
      hipLaunchKernelGGL(k_ho_lists, dim3((unsigned)((U + 1023) / 1024)), dim3(256), 0, c->stream3, c->lc_pending.p, U, ids_f, U, d_nf, cnt, d_gate,
                         (c->pl_enabled && c->K.pg_min_frac > 0) ? (unsigned int)(U / c->K.pg_min_frac) : 0xffffffffu);
    if (dense && nab > 0) VGS_HIP_TRY(c, hipEventRecord(c->ev_ho, c->stream3));   // the hand-over lists are built, the word of LcGate is written
    vgs_status st = VGS_OK;
    if (!dense) {
      st = launch_block(c->stream3, ids_f, grid_f, true, d_nf, 0);
    } else if (grid_f > 0 && c->pl_enabled) {
      c->pl_enabled_at_launch = true;
      // Hand-overs of the one-wavefront classes, two ways, decided ON THE DEVICE by how many there are (the launches are queued before
      // anybody knows: LcGate).  Few (a smooth scene's clutter: 1 % of the voxels): the dense kernel evaluates their pairs itself --
      // building the pair lists of every row their neighbourhoods touch would evaluate more pairs than that.  Many (range noise, vegetation:
      // every neighbourhood wants all of its heavy pairs): the rows are built, every heavy pair ONCE instead of once per neighbourhood,
      // and one workgroup per voxel reads them in bands of descending weight (localcut_pg.hpp).  What either cannot take goes on to the
      // general kernel through the same list.
      const LcGate g_few = {d_gate, LC_FEW}, g_many = {d_gate, LC_MANY};
      const uint32_t* ids4[LW_HO_BINS]; const unsigned int* nd4[LW_HO_BINS];
      for (int k = 0; k < LW_HO_BINS; ++k) { ids4[k] = ids_f + (size_t)k * U; nd4[k] = d_nf + k; }
      // (the pool is sized by vgs_pairlists_begin, once per run)
      auto pair_lists = [&]() { return PairLists{(const uint2*)c->pl_state.p, c->pl_ent.p, c->pl_state.p + (size_t)c->V * 9, c->pl_w_ring}; };
      const PgGeom G = {c->vox_code.p, c->P.voxel_size, (float)c->box.min[0], (float)c->box.min[1], (float)c->box.min[2], c->adj_r2};
      // (Few hand-overs through the pair lists as well -- mark the rows their neighbourhoods touch, build, read -- was measured in round 5:
      // 8.0 against 6.9 ms on URB10M.  Its 5 k hand-overs sit in tree crowns whose 100 k voxels are all somebody's neighbour: twenty rows
      // built per voxel cut.)
      {
      // (Round 5 let the dense kernel queue the neighbourhoods with more heavy edges than its list holds for the pair-list kernel, their rows
      // marked and built behind it -- VGS_DENSE_TO_PG: measured on URB10M, no gain, and its build was not ordered behind the all-rows build on
      // the other stream: ADVICE r5.  Removed; those neighbourhoods take the dense kernel's own bands.)
      hipLaunchKernelGGL((k_localcut_dense<DN_SMALL>), dim3(grid_f), dim3(256), 0, c->stream3, ids_f, (int)U, LW_HO_BINS, d_nf, c->adj_key.p, c->adj_cnt.p,
                         c->adj_stride, c->node.p, LP, c->conn.p, cnt, ids_f2, d_nf2, c->evals.p, g_few, (uint32_t*)nullptr, (unsigned int*)nullptr);
      // (the pair-list chain on a stream of its own: when it is not wanted its four empty launches end beside the dense kernel, not behind it)
      VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream4, c->ev_ho, 0));
      // (many hand-overs: their neighbourhoods cover practically every row, so every row is built -- marking the wanted ones is 15 M
      // scattered byte stores, a millisecond on the noisy surface)
      st = vgs_pairlists_build(c, c->stream4, ids4, nd4, nullptr, LW_HO_BINS, true, LP.ctab, LP.ctab_scale, LP.d2_stop, 1, g_many, false);
      if (st == VGS_OK && c->pl_enabled) {
        hipLaunchKernelGGL((k_localcut_pg<PG_SMALL>), dim3(grid_f), dim3(256), 0, c->stream4, ids_f, (int)U, LW_HO_BINS, d_nf, 0u, 0, c->adj_key.p, c->adj_cnt.p,
                           c->adj_stride, c->adj_off.p, c->node.p, LP, pair_lists(), G, c->conn.p, cnt, ids_f2, d_nf2, c->evals.p, (uint8_t*)nullptr,
                           (uint32_t*)nullptr, 0, 0, g_many, (uint32_t*)nullptr, (unsigned int*)nullptr);
      } else if (st == VGS_OK) {   // (the pool did not fit: the dense kernel takes them all)
        hipLaunchKernelGGL((k_localcut_dense<DN_SMALL>), dim3(grid_f), dim3(256), 0, c->stream4, ids_f, (int)U, LW_HO_BINS, d_nf, c->adj_key.p, c->adj_cnt.p,
                           c->adj_stride, c->node.p, LP, c->conn.p, cnt, ids_f2, d_nf2, c->evals.p, g_many, (uint32_t*)nullptr, (unsigned int*)nullptr);
      }
      }
    } else if (grid_f > 0) {
      hipLaunchKernelGGL((k_localcut_dense<DN_SMALL>), dim3(grid_f), dim3(256), 0, c->stream3, ids_f, (int)U, LW_HO_BINS, d_nf, c->adj_key.p, c->adj_cnt.p,
                         c->adj_stride, c->node.p, LP, c->conn.p, cnt, ids_f2, d_nf2, c->evals.p, LcGate{nullptr, 0u}, (uint32_t*)nullptr, (unsigned int*)nullptr);
    }
    if (st != VGS_OK) return st;
    if (grid_f > 0 && dense && c->pl_enabled_at_launch) {   // stream3 ends when both chains have
      VGS_HIP_TRY(c, hipEventRecord(c->ev_ho2, c->stream4));
      VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream3, c->ev_ho2, 0));
    }
    VGS_HIP_TRY(c, hipEventRecord(c->ev[4], c->stream3));
  }
  // the main stream goes on once every class has produced its rows or marked them pending
  VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev[8], 0));
  VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev[5], 0));
  if (nabc[3] + nabc[5] > 0) VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev[12], 0));
  if (c->K.no_overlap) {   // diagnostics: the merge stage starts behind the hand-over kernels
    VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev[4], 0));
    VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev[3], 0));
    if (c->cb_enabled)   // (the pending marks are about to go: the handed-over voxels' connect bits now)
      hipLaunchKernelGGL(k_conn_bits, dim3((unsigned)U), dim3(64), 0, c->stream, c->lc_pending.p, U, c->adj_cnt.p, c->adj_stride, c->conn.p, c->adj_off.p,
                         c->cb_R, c->cb_words, c->conn_bits.p, CbLists{}, 0);
    VGS_HIP_TRY(c, hipMemsetAsync(c->lc_pending.p, 0, (size_t)U, c->stream));
  }
  // crossValidation's first pass looks at the word k_ho_lists writes (merge.hip) WITHOUT waiting for it: a workgroup that finds it still
  // undecided goes ahead as for LC_FEW, and should the word turn out LC_MANY the full pass that follows redoes whatever it did
  if (c->pl_enabled_at_launch) c->lc_tail.gated = true;
  c->lc_tail.grid_f = grid_f;
  for (int k = 0; k < 5; ++k) c->lc_tail.nabc[k] = nabc[k];
  c->lc_tail.nabc[2] += nabc[5];
  VGS_HIP_TRY(c, hipEventRecord(c->ev[13], c->stream));   // end of the stage's main-stream work (vgs_localcut_finish measures the tail behind it)
  c->counts[13] = nabc[0] + nabc[4]; c->counts[14] = nabc[1] + nabc[2] + nabc[5] + nabc[6]; c->counts[15] = nabc[3];  // bulk launch (A1 + A), the other wave classes, class D
  c->counts[VGS_N_PAIRS] = -1;  // per-voxel counts stay in c->evals; vgs_get_counts sums them when asked
  VGS_HIP_TRY(c, hipGetLastError());
  return VGS_OK;
}

// Second half of the stage, called by the merge stage once it has nothing left to do beside the hand-over kernels: waits
// for them, reads the stage's flags and list lengths back (one copy), finishes lists longer than their fixed grids, and
// returns the number of rows crossValidation has put off.
vgs_status vgs_localcut_finish(vgs_ctx* c, unsigned int* n_deferred) {
  if (n_deferred) *n_deferred = 0;
  if (!c->lc_tail.open) return VGS_OK;
  c->lc_tail.open = false;
  const int64_t U = c->U;
  unsigned long long* cnt = (unsigned long long*)c->counters.p;
  const unsigned int grid_f = c->lc_tail.grid_f, GRID_G = c->lc_tail.grid_g;
  uint32_t* ids_f = c->work_ids.p + 8 * U;
  uint32_t* ids_g = c->work_ids.p + 5 * U;
  LcParams LP;
  LP.W = make_weight_params(c->P);
  LP.cut = c->P.cut_thred;
  LP.prune_unused = 0;
  LP.xl_from = c->K.dbg_xl_from;
  LP.dbg_max_m = c->K.dbg_max_m;
  {
    const float reach = 2.0f * c->P.graph_size + 4.0f * c->P.voxel_size;
    LP.d2_stop = lc_d2_stop(LP.W, LP.cut, reach * reach * 1.01f);
    LP.ctab = c->lc_ctab.p; LP.ctab_scale = c->lc_ctab_scale;   // built by vgs_stage_localcut for these parameters
  }
  constexpr int SMALL_M = 128, SMALL_CAP = LC_SMALL_CAP;
  constexpr int LARGE_M = 2048, LARGE_CAP = 8192;
  // Neighbourhoods above LARGE_M used voxels (the reference sizes its matrix to any n, VS:1815-1818): the same kernel with the
  // per-vertex state of a whole search ball in LDS (the adjacency rows end at 8192 ball offsets) and half the edge list --
  // 144 KB, one workgroup per CU.  The large instantiation queues them (its overflow list reuses class D's work list, which
  // is consumed by now).
  constexpr int XL_M = 8192, XL_CAP = 4096;
  uint32_t* ids_xl = c->work_ids.p + 3 * U;
  auto launch_rest = [&](const uint32_t* ids, unsigned int nw, bool mid) -> vgs_status {
    if (nw == 0) return VGS_OK;
    if (mid) {
      auto kern = k_localcut<SMALL_M, SMALL_CAP, false>;
      const size_t sm = lc_smem_bytes<SMALL_M, SMALL_CAP, false>();
      VGS_HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
      hipLaunchKernelGGL(kern, dim3(nw), dim3(LC_TB), sm, c->stream, ids, (int)nw, (const unsigned int*)nullptr, c->adj_key.p, c->adj_cnt.p,
                         c->adj_stride, c->node.p, LP, c->conn.p, cnt, c->evals.p, (uint32_t*)nullptr);
    } else {
      auto kern = k_localcut<LARGE_M, LARGE_CAP, false>;
      const size_t sm = lc_smem_bytes<LARGE_M, LARGE_CAP, false>();
      VGS_HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
      hipLaunchKernelGGL(kern, dim3(nw), dim3(LC_TB), sm, c->stream, ids, (int)nw, (const unsigned int*)nullptr, c->adj_key.p, c->adj_cnt.p,
                         c->adj_stride, c->node.p, LP, c->conn.p, cnt, c->evals.p, ids_xl);
    }
    return VGS_OK;
  };
  VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev[4], 0));
  VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev[3], 0));
  VGS_HIP_TRY(c, hipEventRecord(c->ev[7], c->stream));
  // one read-back: the kernels' flags (words 0-7), the lengths of the hand-over lists (word 11), the rows put off (word 13)
  unsigned long long hc[64] = {0};   // (words 58-63: the pair lists' pool and the pair-list kernel's counts)
  VGS_READBACK(c, hc, cnt, sizeof(hc));
  const unsigned long long lc_why[4] = {hc[3], hc[4], hc[5], hc[6]};   // (the first words of hc are read again behind the kernels launched below)
  // neighbourhoods queued for the extra-large pair-list instantiation although it was not launched: their rows would never be written
  if (!c->lc_tail.pg_xl_queued && (unsigned int)(hc[56] & 0xffffffffull) != 0u) { c->err = "local cut: voxels queued for a kernel that was not launched"; return VGS_E_STATE; }
  c->lc_tail.many = c->lc_tail.gated && (unsigned int)(hc[57] & 0xffffffffull) == LC_MANY;
  c->lc_diag[9] = (int64_t)hc[63]; c->lc_diag[10] = (int64_t)(hc[58] & 0xffffffffull); c->lc_diag[11] = (int64_t)hc[62]; c->lc_diag[12] = (int64_t)hc[60];
  const unsigned long long* h = hc;
  // hand-overs of the one-wavefront classes (all size lists together), of classes C/D
  const unsigned int nfg[2] = {(unsigned int)((hc[14] & 0xffffffffull) + (hc[14] >> 32) + (hc[15] & 0xffffffffull) + (hc[15] >> 32)),
                               (unsigned int)(hc[11] >> 32)};
  const unsigned int nf = nfg[0] + nfg[1];
  if (n_deferred) *n_deferred = (unsigned int)(hc[13] & 0xffffffffull);
  const bool dense = c->lc_tail.dense;   // as the launch half of the stage saw it
  const unsigned int nf2 = (unsigned int)(hc[12] & 0xffffffffull), ng2 = (unsigned int)(hc[12] >> 32);   // sent on by the dense kernels
  {
    vgs_status st = VGS_OK;
    bool more = false;
    if (dense) {
      // the dense kernels stride over their whole lists; what they could not hold (a list overflowed, more than 512
      // neighbours) is finished by the general kernel here
      if (nf2 > 0) { st = launch_rest(c->work_ids.p + 7 * U, nf2, true); more = true; }
      if (st == VGS_OK && ng2 > 0) { st = launch_rest(c->work_ids.p + 4 * U, ng2, false); more = true; }
    } else {
      // diagnostics path: lists longer than the general kernel's fixed grids
      if (nfg[0] > grid_f) { st = launch_rest(ids_f + grid_f, nfg[0] - grid_f, true); more = true; }
      if (st == VGS_OK && nfg[1] > GRID_G) { st = launch_rest(ids_g + GRID_G, nfg[1] - GRID_G, false); more = true; }
    }
    if (st != VGS_OK) return st;
    if (more) {
      VGS_HIP_TRY(c, hipEventRecord(c->ev[7], c->stream));
      VGS_HIP_TRY(c, hipMemcpyAsync(hc, cnt, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
      VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
  }
  c->lc_diag[8] = 0;
  if (hc[1] > 0) {
    // the extra-large class: everything the 2048-vertex instantiation queued
    static_assert(XL_M <= 65535 && XL_M >= 8192, "pair ids are 2 x 16 bits; a row holds at most 8192 ball offsets");
    const unsigned int n_xl = (unsigned int)hc[1];
    if ((int64_t)n_xl > U) { c->err = "local cut: corrupt overflow count"; return VGS_E_HIP; }
    auto kern = k_localcut<XL_M, XL_CAP, false>;
    const size_t sm = lc_smem_bytes<XL_M, XL_CAP, false>();
    VGS_HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    VGS_HIP_TRY(c, hipMemsetAsync(cnt + 1, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(kern, dim3(n_xl), dim3(LC_TB), sm, c->stream, ids_xl, (int)n_xl, (const unsigned int*)nullptr, c->adj_key.p, c->adj_cnt.p,
                       c->adj_stride, c->node.p, LP, c->conn.p, cnt, c->evals.p, (uint32_t*)nullptr);
    VGS_HIP_TRY(c, hipGetLastError());
    VGS_HIP_TRY(c, hipEventRecord(c->ev[7], c->stream));
    VGS_HIP_TRY(c, hipMemcpyAsync(hc, cnt, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->lc_diag[8] = (int64_t)n_xl;
  }
  if (c->cb_enabled && nf > 0) {
    // the hand-over kernels are through: their voxels' connect bits, one workgroup per entry of the hand-over lists (whatever was
    // sent on to the general or the extra-large kernel is in these lists too)
    CbLists L;
    const unsigned int nb4[4] = {(unsigned int)(hc[14] & 0xffffffffull), (unsigned int)(hc[14] >> 32), (unsigned int)(hc[15] & 0xffffffffull), (unsigned int)(hc[15] >> 32)};
    unsigned int at = 0;
    for (int k = 0; k < 4; ++k) { L.ids[k] = ids_f + (size_t)k * U; at += nb4[k]; L.end[k] = at; }
    L.ids[4] = ids_g; at += nfg[1]; L.end[4] = at;
    hipLaunchKernelGGL(k_conn_bits, dim3(at), dim3(64), 0, c->stream, c->lc_pending.p, U, c->adj_cnt.p, c->adj_stride, c->conn.p, c->adj_off.p,
                       c->cb_R, c->cb_words, c->conn_bits.p, L, 1);
    VGS_HIP_TRY(c, hipGetLastError());
  }
  c->counts[VGS_N_REATTACHED + 2] = nf;  // diagnostics: voxels handed over by the wave kernels
  c->lc_diag[0] = (int64_t)lc_why[0]; c->lc_diag[1] = (int64_t)(lc_why[1] + lc_why[2] + lc_why[3]); c->lc_diag[2] = (int64_t)nfg[0]; c->lc_diag[3] = (int64_t)nf2 + (int64_t)ng2;
  c->lc_diag[4] = (int64_t)nfg[1]; c->lc_diag[5] = (int64_t)h[1]; c->lc_diag[6] = (int64_t)(hc[13] & 0xffffffffull); c->lc_diag[7] = (int64_t)h[0];
  VGS_HIP_TRY(c, hipGetLastError());
  float kms = 0.f;
  VGS_HIP_TRY(c, hipEventElapsedTime(&kms, c->ev[6], c->ev[7]));
  c->times[VGS_T_LOCALCUT_KERNEL] = kms;
  VGS_HIP_TRY(c, hipEventElapsedTime(&kms, c->ev[10], c->ev[11]));
  c->times[VGS_T_LOCALCUT_BULK] = kms;
  if (hipEventElapsedTime(&kms, c->ev[13], c->ev[7]) == hipSuccess && kms > 0.f) c->lc_tail.tail_ms = kms;
  if (c->K.debug && nfg[0] > 0) {
    std::vector<uint32_t> idf(nfg[0]), ev((size_t)U), ac((size_t)U);
    {
      const unsigned int nb4[4] = {(unsigned int)(hc[14] & 0xffffffffull), (unsigned int)(hc[14] >> 32), (unsigned int)(hc[15] & 0xffffffffull), (unsigned int)(hc[15] >> 32)};
      size_t at = 0;
      for (int k = 0; k < 4; ++k) {
        if (nb4[k]) VGS_HIP_TRY(c, hipMemcpy(idf.data() + at, ids_f + (size_t)k * U, (size_t)nb4[k] * 4, hipMemcpyDeviceToHost));
        at += nb4[k];
      }
    }
    VGS_HIP_TRY(c, hipMemcpy(ev.data(), c->evals.p, (size_t)U * 4, hipMemcpyDeviceToHost));
    VGS_HIP_TRY(c, hipMemcpy(ac.data(), c->adj_cnt.p, (size_t)U * 4, hipMemcpyDeviceToHost));
    double sm = 0, se = 0, sp = 0; uint32_t mn = ~0u, mx = 0;
    for (uint32_t u : idf) { sm += ac[u]; se += ev[u]; sp += 0.5 * ac[u] * (ac[u] - 1.0); mn = ac[u] < mn ? ac[u] : mn; mx = ac[u] > mx ? ac[u] : mx; }
    fprintf(stderr, "[vgs] handed-over voxels: %u, m avg %.1f min %u max %u, evaluations avg %.0f of %.0f pairs, slow-path %llu\n", nfg[0], sm / nfg[0], mn, mx, se / nfg[0], sp / nfg[0], h[7]);
    if (c->nl_enabled) {
      // how well does a voxel's own near-list length predict a hand-over?
      std::vector<uint32_t> nc((size_t)c->V);
      std::vector<uint32_t> uid((size_t)U);
      VGS_HIP_TRY(c, hipMemcpy(nc.data(), c->nl_tot.p, nc.size() * 4, hipMemcpyDeviceToHost));
      VGS_HIP_TRY(c, hipMemcpy(uid.data(), c->used_ids.p, uid.size() * 4, hipMemcpyDeviceToHost));
      long long hall[34] = {0}, hho[34] = {0};
      for (size_t u2 = 0; u2 < (size_t)U; ++u2) { int k = (int)nc[uid[u2]]; hall[k > 32 ? 33 : k]++; }
      for (uint32_t u2 : idf) { int k = (int)nc[uid[u2]]; hho[k > 32 ? 33 : k]++; }
      fprintf(stderr, "[vgs] own near-list length: all / handed over:");
      for (int k = 0; k < 34; ++k) if (hall[k]) fprintf(stderr, " %d:%lld/%lld", k, hall[k], hho[k]);
      fprintf(stderr, "\n");
    }
  }
  if (c->K.debug) fprintf(stderr, "[vgs] localcut classes a=%lld b=%lld c=%lld fallback=%lld bail(shrink)=%llu bail(full)=%llu bail(collapse)=%llu bail(phaseB)=%llu\n", (long long)c->counts[13], (long long)c->counts[14], (long long)c->counts[15], (long long)c->counts[12], lc_why[0], lc_why[1], lc_why[2], lc_why[3]);
#ifdef VGS_PROF
  {
    unsigned long long lp[2][16];
    VGS_HIP_TRY(c, hipMemcpyFromSymbol(lp, HIP_SYMBOL(g_lc_prof), sizeof(lp)));
    const char* nm[16] = {"rows<<32", "incident", "screen", "eval", "sortA", "mergeA", "passB", "sortB", "mergeB", "rounds", "sum_nlA", "sum_nq", "nA", "sum_nlB", "nB", ""};
    for (int k = 0; k < 2; ++k) {
      fprintf(stderr, "[vgs-prof] k_localcut %s:", k ? "large" : "small");
      for (int j = 0; j < 15; ++j) fprintf(stderr, " %s=%.4g", nm[j], j == 0 ? (double)(lp[k][0] >> 32) : (double)lp[k][j]);
      fprintf(stderr, "\n");
    }
    unsigned long long z[2][16] = {{0}};
    VGS_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_lc_prof), z, sizeof(z)));
    unsigned long long dp[16];
    VGS_HIP_TRY(c, hipMemcpyFromSymbol(dp, HIP_SYMBOL(g_dn_prof), sizeof(dp)));
    const char* dn[16] = {"rows", "incident", "phaseA", "sortA", "mergeA", "passB", "sortB", "mergeB", "unreachable_B(+band cycles)", "max_wall_10ns", "sum_nlA", "nA", "sum_nlB", "nB", "max_cycles", "n>400k"};
    fprintf(stderr, "[vgs-prof] k_localcut_dense:");
    for (int j = 0; j < 16; ++j) if (dn[j][0]) fprintf(stderr, " %s=%.4g", dn[j], j == 8 ? (double)(dp[j] & 0xffffffffull) : (double)dp[j]);
    fprintf(stderr, "\n");
    VGS_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_dn_prof), z, sizeof(dp)));
    if (getenv("VGS_DN_TRACE")) {
      std::vector<unsigned long long> tr(4 * 16384 + 1);
      VGS_HIP_TRY(c, hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_dn_trace), tr.size() * 8));
      const size_t n = std::min<size_t>(tr[4 * 16384], 16384);
      FILE* f = fopen(getenv("VGS_DN_TRACE"), "w");
      if (f) { for (size_t k = 0; k < n; ++k) fprintf(f, "%llu %llu %llx %llu %llu\n", tr[4*k], tr[4*k+1], tr[4*k+2], tr[4*k+3] & 0xffffffffull, tr[4*k+3] >> 32); fclose(f); }
      const unsigned long long z1 = 0;
      VGS_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_dn_trace), &z1, 8, (4 * 16384) * 8));
    }
    unsigned long long pg[24];
    VGS_HIP_TRY(c, hipMemcpyFromSymbol(pg, HIP_SYMBOL(g_pg_prof), sizeof(pg)));
    const char* pn[24] = {"stage", "read", "sort", "merge", "carry", "ring", "phaseB", "result", "", "", "sum_edges", "ring_edges", "bands", "nB", "voxels", "rounds", "alive_at_round", "star_at_round", "chain_members", "cyc_walk", "cyc_claim", "cyc_commit", "cyc_flatten", "cyc_merge_wave0"};
    fprintf(stderr, "[vgs-prof] k_localcut_pg:");
    for (int j = 0; j < 24; ++j) if (pn[j][0]) fprintf(stderr, " %s=%.4g", pn[j], (double)pg[j]);
    fprintf(stderr, "\n");
    unsigned long long z24[24] = {0};
    VGS_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_pg_prof), z24, sizeof(pg)));
  }
  {
    std::vector<uint32_t> dbg(4 * (size_t)U);
    VGS_HIP_TRY(c, hipMemcpy(dbg.data(), s_dbg.p, dbg.size() * 4, hipMemcpyDeviceToHost));
    // histogram of per-voxel cost by m bucket
    double cyc[9] = {0}, cntb[9] = {0}, rnd[9] = {0}, evl[9] = {0}; double maxc = 0; size_t maxu = 0;
    for (size_t q = 0; q < (size_t)U; ++q) { uint32_t mm = dbg[4*q]; if (!mm) continue; int b = mm <= 32 ? 0 : mm <= 64 ? 1 : mm <= 96 ? 2 : mm <= 128 ? 3 : mm <= 160 ? 4 : mm <= 192 ? 5 : mm <= 224 ? 6 : 7; double cy = 16.0 * dbg[4*q+2]; cyc[b] += cy; cntb[b] += 1; rnd[b] += dbg[4*q+1]; evl[b] += dbg[4*q+3]; if (cy > maxc) { maxc = cy; maxu = q; } }
    for (int b = 0; b < 8; ++b) if (cntb[b] > 0) fprintf(stderr, "[vgs-prof] m-bucket %d: voxels=%.0f avg_cycles=%.0f avg_rounds=%.2f avg_evals=%.0f total_cycles=%.3g\n", b, cntb[b], cyc[b]/cntb[b], rnd[b]/cntb[b], evl[b]/cntb[b], cyc[b]);
    fprintf(stderr, "[vgs-prof] slowest voxel u=%zu m=%u rounds=%u cycles=%.0f evals=%u\n", maxu, dbg[4*maxu], dbg[4*maxu+1], maxc, dbg[4*maxu+3]);
  }
  {
    unsigned long long pr[16];
    VGS_HIP_TRY(c, hipMemcpy(pr, cnt + 16, sizeof(pr), hipMemcpyDeviceToHost));
    const char* nm[16] = {"gather", "enumerate_near", "evaluate", "sort", "merge", "carry", "phaseB", "merge_iterations", "rounds", "sorted_keys", "general_rounds", "voxels", "merges", "sum_m", "enumerate_general", "merge_steps"};
    fprintf(stderr, "[vgs-prof]");
    for (int k = 0; k < 16; ++k) fprintf(stderr, " %s=%.3g", nm[k], (double)pr[k]);
    fprintf(stderr, "\n");
  }
#endif
  if (h[1]) { c->err = "a voxel has more than 8192 used neighbours (beyond the adjacency rows' own limit)"; return VGS_E_UNSUPPORTED; }
  if (h[2]) { c->err = "degenerate neighbourhood: more than 8192 pair weights inside one 2^-22 interval"; return VGS_E_UNSUPPORTED; }
  return VGS_OK;
}
