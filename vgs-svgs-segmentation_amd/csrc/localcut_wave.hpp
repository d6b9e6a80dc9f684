// localcut_wave.hpp -- the fast path of the local graph cut: ONE WAVEFRONT PER VOXEL, lazy evaluation.
// Included by localcut.hip (needs LcParams and wave_sync from there).
//
// Same sequential semantics as k_localcut, different schedule.  The reference evaluates all
// n^2 weights and sorts them, yet on a surface the cut is decided by the few hundred heaviest edges: the
// descending scan merges the neighbourhood along its nearest-neighbour edges and then nothing can merge any more.
// The weight is bounded by proximity alone:  D >= dist_space / sig_p  =>  w <= ub(d) = exp(-0.5 * d / sig_p / sig_w^2)
// (VS:1736-1737; the float evaluation is monotone step by step), so pairs are evaluated in
// shells of increasing centroid distance and the scan is advanced only down to the level ub(shell radius):
// every edge not yet evaluated is provably lighter than that level.  Each round:
//   1. enumerate the candidate pairs (different segments) whose squared centroid distance falls in the shell
//      [cut_lo, cut_hi) -- 6 LDS reads and ~10 VALU per pair, no transcendental;
//   2. evaluate the full weight only for those (one pair per lane, all lanes busy);
//   3. bitonic-sort the edge list (new + carried over) in LDS: <= 512 keys instead of n^2;
//   4. merge sequentially down to the level, 64 edges per step, with the segment ids / thresholds / sizes of
//      the 64 edges held in registers: after a merge every lane patches its own copy (a handful of VALU ops,
//      no LDS round trip), the vertex->segment table is fixed up once per step through a representative chain;
//   5. edges lighter than the level are carried to the next round; stop when < 2 segments can still merge.
#ifndef LOCALCUT_WAVE_HPP_
#define LOCALCUT_WAVE_HPP_

struct LwParams {
  LcParams lc;
  float r2_graph;   // graph_size^2: scale of the first shell
  float d2_all;     // squared distance no pair of one neighbourhood can reach: last shell is open ended
  float shell0;     // first shell = shell0 * r2_graph / m
  float grow;       // shell growth factor (in squared distance)
};

__device__ __forceinline__ float lw_readlane_f(float x, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}

// upper bound of the weight of any pair whose squared centroid distance is >= d2 (same float pipeline as
// vm_distance_weight with all other distances dropped, then a 1e-6 relative margin for vm_exp's <= 1 ulp error)
__device__ __forceinline__ float lw_level(float d2, const VgsWeightParams& W) {
  const float d = vm_sqrt(d2);
  float D;
  if (!W.svgs) { const float s = d * W.inv_sig_p; D = vm_sqrt(s * s); }
  else D = vm_sqrt(d * d * W.inv_sig_p);
  return vm_exp((-0.5f * D) * W.inv_sig_w2) * 1.000001f;
}

template <int MAXM, int LCAP>
__global__ __launch_bounds__(64) void k_localcut_wave(const uint32_t* __restrict__ work, int n_work,
                                                      const uint64_t* __restrict__ adj_key, const uint32_t* __restrict__ adj_cnt,
                                                      int adj_stride, const NodeRec* __restrict__ node, LwParams P,
                                                      uint8_t* __restrict__ conn, unsigned long long* __restrict__ counters,
                                                      uint32_t* __restrict__ fallback, unsigned int* __restrict__ n_fallback) {
  __shared__ __attribute__((aligned(16))) NodeRec rec[MAXM];
  __shared__ __attribute__((aligned(16))) uint64_t list[LCAP];
  __shared__ float thr[MAXM];
  __shared__ uint16_t seg[MAXM], rep[MAXM], ssz[MAXM], loc[MAXM];
  __shared__ uint16_t alist[MAXM];  // vertices whose segment can still merge (ascending), see step 5

  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= n_work) return;
  const uint32_t u = work[blockIdx.x];
  const int n = (int)adj_cnt[u];
  const uint64_t* row = adj_key + (int64_t)u * adj_stride;
  uint8_t* crow = conn + (int64_t)u * adj_stride;
  const float cut = P.lc.cut;
  const VgsWeightParams& W = P.lc.W;

  // ---- gather the used neighbours in adjacency order; their global ids go through the list area ----
  uint32_t* gid = (uint32_t*)list;
  for (int k = lane; k < n; k += 64) crow[k] = 0;
  int m = 0;
  for (int base = 0; base < n; base += 64) {
    const int k = base + lane;
    bool keep = false;
    uint32_t t = 0;
    if (k < n) {
      t = (uint32_t)row[k];
      keep = P.lc.prune_unused ? ((node[t].flags & VGS_F_EIG) != 0) : true;
    }
    const unsigned long long mk = __ballot(keep);
    const int pos = m + __popcll(mk & ((1ull << lane) - 1ull));
    if (keep && pos < MAXM) { gid[pos] = t; loc[pos] = (uint16_t)k; }
    m += __popcll(mk);
  }
  if (m > MAXM) {  // classification guarantees this does not happen; hand over to the general kernel anyway
    if (lane == 0) fallback[atomicAdd(n_fallback, 1u)] = u;
    return;
  }
  wave_sync();
  {
    const float4* src = (const float4*)node;
    float4* dst = (float4*)rec;
    for (int e = lane; e < m * 4; e += 64) dst[e] = src[(size_t)gid[e >> 2] * 4 + (e & 3)];
  }
  const float thr0 = vm_cut_threshold(1.0f, cut, 1);
  for (int c = lane; c < m; c += 64) { seg[c] = (uint16_t)c; rep[c] = (uint16_t)c; ssz[c] = 1; thr[c] = thr0; alist[c] = (uint16_t)c; }
  wave_sync();

  unsigned long long n_evals = 0;
  int n_list = 0;      // edges carried in the list (sorted, all lighter than the level of the previous round)
  int merges = 0;
  bool bail = false;
  if (m >= 2) {
    int n_act = m;       // vertices still able to merge; pairs are enumerated among them only
    float cut_lo = 0.0f;
    float cut_hi = P.shell0 * P.r2_graph / (float)m;
    int shrink = 0;
    for (int round = 0; round < 4096; ++round) {
      bool final_round = !(cut_hi < P.d2_all);
      // ---- 1. enumerate the pairs of this shell ----
      const int free_slots = LCAP - n_list;
      const int Pact = n_act * (n_act - 1) / 2;
      int count = 0;
      {
        int ia = 0, qq = lane;
        for (int base = 0; base < Pact; base += 64) {
          while (ia < n_act - 1 && qq >= n_act - 1 - ia) { qq -= (n_act - 1 - ia); ++ia; }
          const bool valid = ia < n_act - 1;
          bool inr = false;
          uint32_t pid = 0;
          if (valid) {
            const int a = alist[ia], b = alist[ia + 1 + qq];  // a < b: alist is ascending
            if (merges == 0 || seg[a] != seg[b]) {
              float d2 = 1.0e4f;  // dist_space stays 100 when a centroid has a zero component (VS:1829)
              if ((rec[a].flags & VGS_F_POS) && (rec[b].flags & VGS_F_POS)) {
                const float dx = rec[a].c[0] - rec[b].c[0], dy = rec[a].c[1] - rec[b].c[1], dz = rec[a].c[2] - rec[b].c[2];
                d2 = (dx * dx + dy * dy) + dz * dz;
              }
              inr = (d2 >= cut_lo) && (final_round || d2 < cut_hi);
              pid = ((uint32_t)a << 16) | (uint32_t)b;
            }
          }
          const unsigned long long mk = __ballot(inr);
          if (inr) {
            const int pos = n_list + count + __popcll(mk & ((1ull << lane) - 1ull));
            if (pos < LCAP) list[pos] = (uint64_t)pid;
          }
          count += __popcll(mk);
          qq += 64;
        }
      }
      if (count > free_slots) {
        // the shell holds more pairs than the list: shrink it (assume uniform density in d2) and redo
        if (++shrink > 24 || free_slots < 32) { if (lane == 0) atomicAdd(&counters[free_slots < 32 ? 4 : 3], 1ull); bail = true; break; }
        const float hi = final_round ? P.d2_all : cut_hi;
        cut_hi = cut_lo + (hi - cut_lo) * (0.75f * (float)free_slots / (float)count);
        if (!(cut_hi > cut_lo)) { if (lane == 0) atomicAdd(&counters[5], 1ull); bail = true; break; }
        continue;
      }
      shrink = 0;
      wave_sync();
      // ---- 2. full weight of the shell's pairs ----
      int dropped = 0;
      for (int base = n_list; base < n_list + count; base += 64) {
        const int e = base + lane;
        bool nanw = false;
        if (e < n_list + count) {
          const uint32_t pid = (uint32_t)list[e];
          const float w = vm_pair_weight(rec[pid >> 16], rec[pid & 0xffffu], W);
          nanw = (w != w);
          list[e] = nanw ? 0ull : (((uint64_t)vm_bits(w) << 32) | (uint64_t)(0xffffffffu - pid));
        }
        dropped += __popcll(__ballot(nanw));
      }
      n_evals += (unsigned long long)count;
      n_list += count;
      // ---- 3. sort descending (weight, then ascending (a, b)); NaN edges (key 0) fall off the end ----
      int np = 64;
      while (np < n_list) np <<= 1;
      for (int k = n_list + lane; k < np; k += 64) list[k] = 0ull;
      wave_sync();
      for (int size = 2; size <= np; size <<= 1) {
        for (int strd = size >> 1; strd > 0; strd >>= 1) {
          for (int t = lane; t < (np >> 1); t += 64) {
            const int lo = ((t / strd) * (strd << 1)) + (t % strd);
            const int hi = lo + strd;
            const bool dn = ((lo & size) == 0);
            const uint64_t x = list[lo], y = list[hi];
            if ((x < y) == dn) { list[lo] = y; list[hi] = x; }
          }
          wave_sync();
        }
      }
      n_list -= dropped;
      // ---- 4. merge down to the level ----
      const float level = final_round ? -1.0f : lw_level(cut_hi, W);
      int pos = 0;
      bool reached_level = false;
      while (pos < n_list && !reached_level) {
        const int e = pos + lane;
        float w = 0.f;
        int sa = 0, sb = 0;
        bool proc = false;
        if (e < n_list) {
          const uint64_t key = list[e];
          w = vm_from_bits((uint32_t)(key >> 32));
          proc = w > level;
          const uint32_t pid = 0xffffffffu - (uint32_t)key;
          sa = seg[pid >> 16];
          sb = seg[pid & 0xffffu];
          while (rep[sa] != sa) sa = rep[sa];
          while (rep[sb] != sb) sb = rep[sb];
        }
        const int nproc = __popcll(__ballot(proc));  // sorted: the processable edges are a prefix of the step
        float ta = thr[sa], tb = thr[sb];
        int za = ssz[sa], zb = ssz[sb];
        bool alive = proc;
        while (true) {
          const bool pass = alive && (sa != sb) && (w > ta) && (w > tb);
          const unsigned long long mk = __ballot(pass);
          if (mk == 0ull) break;
          const int f = __builtin_amdgcn_readfirstlane(__ffsll((long long)mk) - 1);
          const float wf = lw_readlane_f(w, f);
          const int s1 = __builtin_amdgcn_readlane(sa, f), s2 = __builtin_amdgcn_readlane(sb, f);
          const float t1 = lw_readlane_f(ta, f), t2 = lw_readlane_f(tb, f);
          const int z1 = __builtin_amdgcn_readlane(za, f), z2 = __builtin_amdgcn_readlane(zb, f);
          const int keep = (t1 >= t2) ? s1 : s2;   // VS:1972-1983
          const int gone = (t1 >= t2) ? s2 : s1;
          const int nsz = z1 + z2;
          const float nthr = vm_cut_threshold(wf, cut, nsz);  // seg_int = w (VS:1988)
          if (sa == gone) sa = keep;
          if (sb == gone) sb = keep;
          if (sa == keep) { ta = nthr; za = nsz; }
          if (sb == keep) { tb = nthr; zb = nsz; }
          alive = alive && (lane > f);
          if (lane == 0) { rep[gone] = (uint16_t)keep; thr[keep] = nthr; ssz[keep] = (uint16_t)nsz; ssz[gone] = 0; }
          ++merges;
        }
        wave_sync();
        if (nproc < 64) { pos += nproc; reached_level = true; } else pos += 64;
        if (merges >= m - 1) break;  // one segment left
      }
      // vertex -> live representative
      for (int c = lane; c < m; c += 64) {
        int s = seg[c];
        while (rep[s] != s) s = rep[s];
        seg[c] = (uint16_t)s;
      }
      wave_sync();
      if (merges >= m - 1 || final_round) break;
      // ---- 5. freeze and carry ----
      // Every edge still to come (carried or not yet evaluated) weighs <= level.  A segment whose threshold is
      // >= level can therefore never merge again (its threshold only moves when it merges): its vertices leave the
      // pair enumeration and its edges are dropped.  Exact, and it is what keeps plane/plane borders cheap.
      int n_new = 0;
      for (int base = 0; base < n_act; base += 64) {
        const int ia = base + lane;
        bool act = false;
        int v = 0;
        if (ia < n_act) { v = alist[ia]; act = thr[seg[v]] < level; }
        const unsigned long long mk = __ballot(act);
        wave_sync();
        if (act) alist[n_new + __popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)v;
        n_new += __popcll(mk);
        wave_sync();
      }
      n_act = n_new;
      int active_segs = 0;
      for (int base = 0; base < m; base += 64) {
        const int c = base + lane;
        active_segs += __popcll(__ballot((c < m) && (ssz[c] != 0) && (thr[c] < level)));
      }
      if (active_segs < 2) break;
      int kept = 0;
      for (int base = pos; base < n_list; base += 64) {
        const int e = base + lane;
        bool keep_e = false;
        uint64_t key = 0;
        if (e < n_list) {
          key = list[e];
          const uint32_t pid = 0xffffffffu - (uint32_t)key;
          const int sa = seg[pid >> 16], sb = seg[pid & 0xffffu];
          keep_e = (sa != sb) && (thr[sa] < level) && (thr[sb] < level);
        }
        const unsigned long long mk = __ballot(keep_e);
        wave_sync();  // all lanes have read their entry before anyone overwrites the front of the list
        if (keep_e) list[kept + __popcll(mk & ((1ull << lane) - 1ull))] = key;
        kept += __popcll(mk);
        wave_sync();
      }
      n_list = kept;
      cut_lo = cut_hi;
      cut_hi = cut_hi * P.grow;
    }
  }
  if (bail) {
    if (lane == 0) fallback[atomicAdd(n_fallback, 1u)] = u;
    return;
  }
  // ---- result: the segment of vertex 0 (the voxel itself) ----
  {
    const uint16_t s0 = seg[0];
    for (int c = lane; c < m; c += 64)
      if (seg[c] == s0) crow[loc[c]] = 1;
  }
  if (lane == 0 && n_evals) atomicAdd(&counters[0], n_evals);
}

#endif
