// localcut_pg.hpp -- the local graph cut from the PAIR LISTS (pairlist.hpp): one workgroup per voxel, edges taken in exact bands
// of descending weight, no weight evaluated inside the ball.  Included by localcut.hip behind localcut_dense.hpp.
//
// Reference semantics (voxel_segmentation.h:1913-2029) kept exactly, as in the other classes: edges in descending
// (weight, then ascending pair) order, an edge merges two segments iff w > max(seg_int - cut / size), the result is the segment of
// vertex 0.  What this class changes once more is where the edges come from.  The one-wavefront classes walk shells of increasing
// centroid DISTANCE and need a bound between distance and weight (fact U of localcut_wave.hpp) to know how far down the scan may
// go; the dense hand-over class evaluates every pair of the neighbourhood.  Here every vertex a of the neighbourhood brings the
// list of its heavy pairs sorted by WEIGHT, so "all edges heavier than L" is a prefix of every list:
//
//   band:  every still-active vertex offers its next q entries; L = the heaviest entry NOT offered (>= the floor, see below);
//          the offered entries heavier than L whose partner is a vertex of this neighbourhood (hash: lattice offset -> vertex),
//          active and in another segment, are this band's edges -- complete: every unread entry weighs <= L;
//          sort (regsort.hpp), merge down to L with the claim scheme of localcut_wave.hpp, freeze (fact F), next band.
//
// The scan ends as soon as the voxel's own segment is frozen (thr >= L), one segment is left, or the lists are exhausted.
// Floors: the lists hold the pairs inside each other's ball only (pairlist.hpp); pairs of the neighbourhood outside it weigh at
// most w_ring, so bands stay above w_ring until the "ring" pairs have been evaluated -- once, by this kernel, among the vertices
// still active then -- and joined the carried edges; after that the floor is 1 - cut (fact S), below which phase B finishes on
// direct evaluations as in localcut_dense.hpp.
// A neighbourhood this kernel cannot take (a vertex without a list, more vertices than MAXM, a list that overflows even at one
// entry per vertex) goes to `fallback`: the classes of round 4 are still behind it.
#ifndef LOCALCUT_PG_HPP_
#define LOCALCUT_PG_HPP_

#include "pairlist.hpp"

#ifndef PG_PF
#define PG_PF 2   // (4: five spilled registers at 96)
#endif
struct PgGeom {   // what the in-ball test of the ring stage needs: the voxel lattice as adjacency.hip sees it
  const uint64_t* vox_code;
  float res_f, min_x, min_y, min_z;
  float r2;       // float(graph_size^2): the FLANN predicate's right-hand side
};

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store and load of the
// wavefront (vmcnt(0)): behind the row's stores or the list entries' loads that is a memory round trip per barrier, and this
// kernel's barriers all publish LDS data (what comes from memory is waited for where it is used).
__device__ __forceinline__ void pg_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#ifndef PG_CHAIN
#define PG_CHAIN 1
#endif

// inclusive prefix sum over the 64 lanes of a wavefront (all lanes active): four DPP shifts inside each row of 16, then the rows' totals
__device__ __forceinline__ int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return x;
}

#ifdef VGS_PROF
__device__ unsigned long long g_pg_prof[24];
#define PGP_T0() long long _pt0 = clock64()
#define PGP_ACC(slot) do { long long _pt1 = clock64(); if (tid == 0) atomicAdd(&g_pg_prof[slot], (unsigned long long)(_pt1 - _pt0)); _pt0 = _pt1; } while (0)
#define PGP_CNT(slot, v) do { if (tid == 0) atomicAdd(&g_pg_prof[slot], (unsigned long long)(v)); } while (0)
#define PGP_CNTL(slot, v) do { if (lane == 0) atomicAdd(&g_pg_prof[slot], (unsigned long long)(v)); } while (0)
#else
#define PGP_T0() do {} while (0)
#define PGP_ACC(slot) do {} while (0)
#define PGP_CNT(slot, v) do {} while (0)
#define PGP_CNTL(slot, v) do {} while (0)
#endif

// MAXM vertices, LCAP edges in flight (<= 512 * NW: the sort's block size), NW wavefronts, OCC workgroups per CU the registers allow;
// RING_LDS: centroids and normals of the vertices staged in LDS for the ring stage (not for the extra-large instantiation)
// DMAP: 0 = the vertex at a lattice offset is found through a hash; otherwise the side of a cube of 16-bit slots indexed by the offset
// itself (the extra-large instantiation: a hash for 4 k vertices would take 32 KB, the 21^3 cube of a ball of ten voxels 18 KB)
template <int MAXM, int LCAP, int NW, int OCC, bool RING_LDS, int DMAP = 0>
__global__ __launch_bounds__(64 * NW, OCC) void k_localcut_pg(const uint32_t* __restrict__ work, int work_stride, int n_lists,
                                                             const unsigned int* __restrict__ n_work_dev, unsigned int n_work_host, int xcd_order,
                                                             const uint64_t* __restrict__ adj_key, const uint32_t* __restrict__ adj_cnt, int adj_stride,
                                                             const uint16_t* __restrict__ adj_off, const NodeRec* __restrict__ node, LcParams P,
                                                             PairLists PL, PgGeom G, uint8_t* __restrict__ conn,
                                                             unsigned long long* __restrict__ counters, uint32_t* __restrict__ fallback,
                                                             unsigned int* __restrict__ n_fallback, uint32_t* __restrict__ evals_out,
                                                             uint8_t* __restrict__ pending_mark, uint32_t* __restrict__ cbits, int cb_R, int cb_words, LcGate gate,
                                                             uint32_t* __restrict__ too_big, unsigned int* __restrict__ n_too_big) {
  constexpr int TB = 64 * NW;
  constexpr int PSH = 16;
  constexpr uint32_t PMASK = 0xffffu, PCOMP = 0xffffffffu;
  constexpr int HCAP = DMAP ? 1 : (MAXM <= 128 ? 256 : (MAXM <= 512 ? 1024 : (MAXM <= 1024 ? 2048 : (MAXM <= 4096 ? 8192 : 16384))));
  static_assert(DMAP != 0 || HCAP >= 2 * MAXM, "hash load <= 1/2");
  constexpr int DCELLS = DMAP ? DMAP * DMAP * DMAP : 1, DR = (DMAP - 1) / 2;
  __shared__ uint16_t dmap[DCELLS + (DCELLS & 1)];   // DMAP: vertex at every offset of the cube, 0xffff = none
  static_assert(LCAP <= 512 * NW, "one block of 512 keys per wavefront (regsort.hpp)");
  __shared__ uint64_t lk[LCAP];               // weight bits << 32 | ~pair id
  __shared__ uint32_t htab[HCAP];             // (packed offset + 1) << 16 | vertex, 0 = empty
  __shared__ uint32_t lpos[MAXM], lend[MAXM]; // the unread part of every vertex's pair list (pool indices)
  __shared__ float thr[MAXM];
  __shared__ uint32_t claim[MAXM];
  __shared__ uint16_t hlat[MAXM], lcons[MAXM], seg[MAXM], rep[MAXM], ssz[MAXM], alist[MAXM];
  __shared__ float rc[RING_LDS ? 7 : 1][RING_LDS ? MAXM : 1];   // cx, cy, cz, nx, ny, nz, flags of every vertex (ring stage)
  __shared__ float ctab3[3][32];              // voxel centres along each axis, offsets -16 .. 15 from the voxel (adjacency.hip's table)
  __shared__ int s_i[8];
  __shared__ unsigned int s_L;
  enum { S_CNT = 0, S_NACT, S_BAD, S_POS, S_MERGES, S_NQ, S_FLAG, S_PAIRS };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (!lc_gate_open(gate)) return;
  const VgsWeightParams& W = P.W;
  const float cut = P.cut;
  const float thr0 = vm_cut_threshold(1.0f, cut, 1);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  unsigned int n_cut = 0;   // voxels this workgroup cut (counters[63], one atomic per workgroup)

  auto h_slot = [&](uint32_t key15) -> uint32_t { return (key15 * 2654435761u) >> 12; };
  auto d_cell = [&](uint32_t key15) -> int {   // cell of a packed offset in the cube, -1 outside
    const int dx = (int)(key15 & 31u) - 16 + DR, dy = (int)((key15 >> 5) & 31u) - 16 + DR, dz = (int)((key15 >> 10) & 31u) - 16 + DR;
    if ((unsigned)dx >= (unsigned)DMAP || (unsigned)dy >= (unsigned)DMAP || (unsigned)dz >= (unsigned)DMAP) return -1;
    return (dz * DMAP + dy) * DMAP + dx;
  };
  auto h_find = [&](uint32_t key15) -> int {
    if constexpr (DMAP != 0) {
      const int cell = d_cell(key15);
      if (cell < 0) return -1;
      const int v = (int)dmap[cell];
      return v == 0xffff ? -1 : v;
    } else {
      uint32_t h = h_slot(key15);
      while (true) {
        const uint32_t e = htab[h & (HCAP - 1)];
        if (e == 0u) return -1;
        if ((e >> 16) == key15 + 1u) return (int)(e & 0xffffu);
        ++h;
      }
    }
  };

  // returns whether this kernel cut the voxel (wrote its row); a voxel passed on (too_big, fallback) is somebody else's
  auto process = [&](const uint32_t u) -> bool {
    const int m = __builtin_amdgcn_readfirstlane((int)adj_cnt[u]);
    const uint64_t* row = adj_key + (int64_t)u * adj_stride;
    const uint16_t* orow = adj_off + (int64_t)u * adj_stride;
    uint8_t* crow = conn + (int64_t)u * adj_stride;
    PGP_T0();
    auto R = [&](int v) -> const NodeRec& { return node[(uint32_t)row[v]]; };
    auto hand_on = [&]() {   // all threads; the row is written by whoever takes the voxel from the fallback list
      if (tid == 0) { fallback[atomicAdd(n_fallback, 1u)] = u; atomicAdd(&counters[7], 1ull); if (pending_mark) pending_mark[u] = 0xff; }
    };
    if (m > MAXM && too_big != nullptr && orow[0] != 0xffffu) {   // the next larger instantiation takes it
      if (tid == 0) too_big[atomicAdd(n_too_big, 1u)] = u;
      return false;
    }
    if (m > MAXM || orow[0] == 0xffffu || (P.dbg_max_m > 0 && m > P.dbg_max_m)) { hand_on(); return false; }
    static_assert(sizeof(uint64_t) * LCAP >= 4 * 1024, "the bit row (<= 31^3 offsets) fits the edge list");
    // ---- the neighbourhood: lists, offsets, hash, segment state ----
    if constexpr (DMAP != 0) { for (int k = tid; k < (DCELLS + 1) / 2; k += TB) ((uint32_t*)dmap)[k] = 0xffffffffu; }
    else { for (int k = tid; k < HCAP; k += TB) htab[k] = 0u; }
    if (tid == 0) { s_i[S_BAD] = 0; s_i[S_MERGES] = 0; s_i[S_PAIRS] = 0; }
    pg_barrier();
    {
      bool bad = false;
      for (int c = tid; c < m; c += TB) {
        const uint32_t t = (uint32_t)row[c];
        const uint2 ix = PL.idx[t];
        bad = bad || ix.y >= PL_UNUSABLE;
        lpos[c] = ix.x; lend[c] = ix.x + (ix.y >= PL_UNUSABLE ? 0u : ix.y);
        const uint32_t key15 = orow[c];
        hlat[c] = (uint16_t)key15;
        if constexpr (DMAP != 0) {
          const int cell = d_cell(key15);
          if (cell >= 0) dmap[cell] = (uint16_t)c; else bad = true;   // (an offset outside the cube: not this instantiation's ball)
        } else {
          uint32_t h = h_slot(key15);
          while (atomicCAS(&htab[h & (HCAP - 1)], 0u, ((key15 + 1u) << 16) | (uint32_t)c) != 0u) ++h;
        }
        seg[c] = (uint16_t)c; rep[c] = (uint16_t)c; ssz[c] = 1; thr[c] = thr0; claim[c] = 0xffffffffu;
        if constexpr (RING_LDS) {
          const NodeRec& rcd = node[t];
          rc[0][c] = rcd.c[0]; rc[1][c] = rcd.c[1]; rc[2][c] = rcd.c[2]; rc[3][c] = rcd.n[0]; rc[4][c] = rcd.n[1]; rc[5][c] = rcd.n[2];
          rc[6][c] = __uint_as_float(rcd.flags);
        }
      }
      if (bad) s_i[S_BAD] = 1;
    }
    if (tid < 96) {
      const uint64_t code = G.vox_code[(uint32_t)row[0]];   // the voxel itself is the first entry of its row
      const int a = tid >> 5, k = tid & 31;
      const uint32_t key = vm_compact21(code >> (2 - a)) + (uint32_t)(k - 16);
      ctab3[a][k] = vm_voxel_center(key, G.res_f, a == 0 ? G.min_x : (a == 1 ? G.min_y : G.min_z));
    }
    pg_barrier();
    if (s_i[S_BAD]) { hand_on(); return false; }
    unsigned int my_pairs = 0;
    bool done = m < 2 || PL.any[(uint32_t)row[0]] != 1;   // no heavy pair holds the voxel: it stays alone (uniform)
    bool handed = false;
    PGP_ACC(0);

    // Merge of lk[0, cnt) (sorted) down to (not including) weights <= level, on wavefront 0: the claim scheme of localcut_wave.hpp
    // with the chain walk of localcut_dense.hpp (sizes and thresholds travel with the walk).  Leaves the position of the first
    // unprocessed edge in s_i[S_POS], the merges so far in s_i[S_MERGES], seg[] flattened.  Ends with a workgroup barrier.
    auto merge_list = [&](int cnt, float level) {
      if (wave == 0) {
        int merges = s_i[S_MERGES];
        int pos = 0;
        bool reached = false;
#ifdef VGS_PROF
        int pr_rounds = 0, pr_alive = 0, pr_star = 0, pr_members = 0;
        long long pr_walk = 0, pr_claim = 0, pr_commit = 0, pr_t0 = clock64(), pr_flat = 0;
#endif
        while (pos < cnt && !reached) {
          const int e = pos + lane;
          float w = 0.f;
          int sa = 0, sb = 0;
          bool alive = false;
          if (e < cnt) {
            const uint64_t key = lk[e];
            w = vm_from_bits((uint32_t)(key >> 32));
            alive = w > level;
            const uint32_t pid = PCOMP - (uint32_t)key;
            sa = seg[pid >> PSH];
            sb = seg[pid & PMASK];
          }
          const int nproc = __popcll(__ballot(alive));   // sorted: the processable edges are a prefix of the step
          if constexpr (PG_CHAIN != 0 && !RING_LDS) {
          // One iteration decides (a) every edge no earlier undecided edge of the step touches on either side -- the rule of
          // localcut_wave.hpp -- (b) every edge that is the first at ONE of its segments and too light for that segment (its
          // state is the one the edge will find; a rejected edge changes nothing), and (c) a whole CHAIN of edges into one
          // segment X (the larger side of the first undecided edge), which rule (a) would take one iteration each.  Chain members
          // are the edges X--Y that are the first to touch their Y and heavy enough for it, up to the first edge whose order
          // against the chain the claims cannot show (an X--Y edge whose Y an edge outside the chain touched first; an edge
          // outside the chain touching a member's Y: once Y is inside X it is an edge of X).  A later X--Y edge with a member's Y
          // is skipped: it lies inside X once the member has merged, and a member that does not merge ends the chain (below).
          // What a merge leaves behind depends on its own weight and the size only (thr = w - cut / size), so with every earlier
          // member presumed to merge each member's test against X is a prefix sum away; the first member that fails it is
          // rejected for good (its presumption held) and so is every member behind it.
          while (true) {
            float ta = 0.f, tb = 0.f;
            int za = 0, zb = 0;
#ifdef VGS_PROF
            long long q0 = clock64();
#endif
            if (alive) {
              int r = rep[sa], rb = rep[sb];
              za = (int)ssz[sa]; zb = (int)ssz[sb]; ta = thr[sa]; tb = thr[sb];
              while (r != sa || rb != sb) { sa = r; sb = rb; r = rep[sa]; rb = rep[sb]; za = (int)ssz[sa]; zb = (int)ssz[sb]; ta = thr[sa]; tb = thr[sb]; }
              alive = sa != sb;
            }
            const unsigned long long am = __ballot(alive);
#ifdef VGS_PROF
            { const long long q1 = clock64(); pr_walk += q1 - q0; q0 = q1; }
#endif
            if (am == 0ull) break;
            const int f = __builtin_amdgcn_readfirstlane(__ffsll((long long)am) - 1);
            const int X = __builtin_amdgcn_readlane(za >= zb ? sa : sb, f);
            const bool xa = sa == X, xb = sb == X;
            const bool star = alive && (xa | xb);
            const unsigned long long xm = __ballot(star);
#ifdef VGS_PROF
            ++pr_rounds; pr_alive += __popcll(am); pr_star += __popcll(xm);
#endif
            if (alive) { if (!xa) atomicMin(&claim[sa], (uint32_t)lane); if (!xb) atomicMin(&claim[sb], (uint32_t)lane); }
            wave_sync();
            uint32_t ca = (uint32_t)lane, cb = (uint32_t)lane;
            if (alive) { if (!xa) ca = claim[sa]; if (!xb) cb = claim[sb]; }
            wave_sync();
            if (alive) { if (!xa) claim[sa] = 0xffffffffu; if (!xb) claim[sb] = 0xffffffffu; }
            const bool fa = ca == (uint32_t)lane, fb = cb == (uint32_t)lane;   // (the side that is X counts as first here)
#ifdef VGS_PROF
            { const long long q1 = clock64(); pr_claim += q1 - q0; q0 = q1; }
#endif
            const bool ca_star = ((xm >> (ca & 63u)) & 1ull) != 0ull, cb_star = ((xm >> (cb & 63u)) & 1ull) != 0ull;
            const bool stop = alive && ((!fa && ca_star != star) || (!fb && cb_star != star));
            const unsigned long long sm = __ballot(stop);
            const unsigned long long open = sm == 0ull ? ~0ull : ((1ull << (__ffsll((long long)sm) - 1)) - 1ull);   // the lanes in front of the first stop
            // first at a segment other than X, that segment's state is the one the edge will find: rejected there, it is rejected
            const bool rej = alive && ((fa && !xa && !(w > ta)) || (fb && !xb && !(w > tb)));
            const bool first = alive && fa && fb && !rej;
            bool member = first && star && ((open >> lane) & 1ull) != 0ull;
            unsigned long long P = __ballot(member);   // the members presumed to merge
            const bool chain = (P & (P - 1ull)) != 0ull;   // (no two of them: the first undecided edge is decided like the others)
            if (!chain) { P = 0ull; member = false; }
            const bool decided = first && (!star || (!chain && lane == f));
            // ---- the chain ----
#ifdef VGS_PROF
            pr_members += __popcll(P);
#endif
            if (chain) {
              const float tX0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xa ? ta : tb), f));   // (lane f holds X's state whether it is a member or not)
              const int zX0 = __builtin_amdgcn_readlane(xa ? za : zb, f);
              const int zz = member ? (xa ? zb : za) : 0;
              const int incl = wave_incl_scan(zz);
              const unsigned long long before = P & lt_mask;
              const int pl = before == 0ull ? lane : 63 - __clzll((long long)before);
              const float w_prev = __shfl(w, pl, 64);
              const float tXb = before == 0ull ? tX0 : vm_cut_threshold(w_prev, cut, zX0 + incl - zz);
              // (a member's own Y has accepted it above.)  Rejected by X, a member shows X frozen: the weights only fall from here
              // and X's threshold changes with a merge only, so every member behind it is rejected too
              const unsigned long long bad = __ballot(member && !(w > tXb));
              if (bad != 0ull) P &= (1ull << (__ffsll((long long)bad) - 1)) - 1ull;
              if (((P >> lane) & 1ull) != 0ull) {
                const int Y = xa ? sb : sa;
                rep[Y] = (uint16_t)X;
                ssz[Y] = 0;
                if ((P >> lane) >> 1 == 0ull) { thr[X] = vm_cut_threshold(w, cut, zX0 + incl); ssz[X] = (uint16_t)(zX0 + incl); }   // the last one leaves X's state
              }
            }
            // ---- the edges decided on their own ----
            const bool pass = decided && (w > ta) && (w > tb);
            if (pass) {
              const int keep = (ta >= tb) ? sa : sb;   // VS:1972-1983: the segment with the larger threshold survives
              const int gone = sa ^ sb ^ keep;
              const int nsz = za + zb;
              rep[gone] = (uint16_t)keep;
              thr[keep] = vm_cut_threshold(w, cut, nsz);   // seg_int = w (VS:1988)
              ssz[keep] = (uint16_t)nsz;
              ssz[gone] = 0;
            }
            merges += __popcll(P) + __popcll(__ballot(pass));
            alive = alive && !decided && !member && !rej;
            wave_sync();
#ifdef VGS_PROF
            { const long long q1 = clock64(); pr_commit += q1 - q0; q0 = q1; }
#endif
          }
          } else {
          while (true) {
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
#ifdef VGS_PROF
            {
              const unsigned long long am = __ballot(alive);
              const int f = __ffsll((long long)am) - 1;
              const int fa = __shfl(sa, f, 64), fb = __shfl(sb, f, 64);
              const int X = ssz[fa] >= ssz[fb] ? fa : fb;
              const unsigned long long xm = __ballot(alive && (sa == X || sb == X));
              ++pr_rounds; pr_alive += __popcll(am); pr_star += __popcll(xm);
            }
#endif
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
              const int gone = sa ^ sb ^ keep;
              rep[gone] = (uint16_t)keep;
              thr[keep] = vm_cut_threshold(w, cut, nsz);   // seg_int = w (VS:1988)
              ssz[keep] = (uint16_t)nsz;
              ssz[gone] = 0;
            }
            merges += __popcll(__ballot(pass));
            alive = alive && !decided;
            wave_sync();
          }
          }
          if (nproc < 64) { pos += nproc; reached = true; } else pos += 64;
          if (merges >= m - 1) break;
          if (!reached && pos < cnt) {
            // every edge from here on weighs at most wn: the voxel's own segment frozen above that ends the scan (fact F)
            const float wn = vm_from_bits((uint32_t)(lk[pos] >> 32));
            int s0 = 0, r;
            while ((r = rep[s0]) != s0) s0 = r;
            if (!(thr[s0] < wn)) { pos = cnt; break; }   // (what is left of the list is dead for the voxel: nothing to carry)
          }
        }
#ifdef VGS_PROF
        pr_flat = clock64();
#endif
        for (int c = lane; c < m; c += 64) {
          int s = seg[c];
          while (rep[s] != s) s = rep[s];
          seg[c] = (uint16_t)s;
        }
        if (lane == 0) { s_i[S_POS] = pos; s_i[S_MERGES] = merges; }
#ifdef VGS_PROF
        { const long long q1 = clock64(); PGP_CNTL(22, q1 - pr_flat); PGP_CNTL(23, q1 - pr_t0); }
        PGP_CNTL(19, pr_walk); PGP_CNTL(20, pr_claim); PGP_CNTL(21, pr_commit);
        PGP_CNTL(15, pr_rounds); PGP_CNTL(16, pr_alive); PGP_CNTL(17, pr_star); PGP_CNTL(18, pr_members);
#endif
      }
      pg_barrier();
    };
    auto sort_list = [&](int cnt) __attribute__((always_inline)) {
      regsort::sort_desc_block<NW>(lk, cnt, wave, lane, [&]() { pg_barrier(); });
    };
    // pair p of the row-major triangle over nv vertices (a < b), decoded from the end (localcut_wave.hpp: enum_section)
    auto decode = [&](uint32_t p, int nv, uint32_t Pn, int& a, int& b) {
      const uint32_t qq = Pn - 1u - p;
      uint32_t r = (uint32_t)((__builtin_amdgcn_sqrtf((float)(8u * qq + 1u)) - 1.0f) * 0.5f);
      r += (((r + 1u) * (r + 2u)) >> 1) <= qq ? 1u : 0u;
      r -= ((r * (r + 1u)) >> 1) > qq ? 1u : 0u;
      a = nv - 2 - (int)r;
      b = nv - 1 - (int)(qq - ((r * (r + 1u)) >> 1));
    };

    bool phase_a_complete = false;
    if (!done) {
      const float floor_ring = PL.w_ring > thr0 ? PL.w_ring : thr0;
      bool ring_done = !(PL.w_ring > thr0);   // ring pairs that cannot be heavy are no edges of phase A
      float lev = __builtin_huge_valf();      // every edge heavier than this has been processed
      int n_carry = 0;
      int bands = 0;
      int min_lq = 0, lq_hint = 0;
      while (true) {
        ++bands;
        // ---- the vertices that still have unread entries and can still merge (fact F) ----
        if (tid == 0) { s_i[S_NACT] = 0; }
        pg_barrier();
        for (int base = 0; base < m; base += TB) {
          const int v = base + tid;
          bool act = false;
          if (v < m) act = lpos[v] < lend[v] && thr[seg[v]] < lev;
          const unsigned long long mk = __ballot(act);
          int b = 0;
          if (mk != 0ull && lane == 0) b = atomicAdd(&s_i[S_NACT], __popcll(mk));
          b = __builtin_amdgcn_readfirstlane(b);
          if (act) alist[b + __popcll(mk & lt_mask)] = (uint16_t)v;
        }
        pg_barrier();
        const int n_act = s_i[S_NACT];
        const float floor = ring_done ? thr0 : floor_ring;
        int free_slots = LCAP - n_carry;
        // log2 of the entries a vertex offers per band: as many as fit the list if every offered entry became an edge -- or one step
        // more than last time when the last band left three quarters of the list empty (most offers are no edges: partners outside
        // the neighbourhood, ends already in one segment, frozen ends); a band that overflows the list is offered again, halved
        int lq = 5;
        while (lq > 1 && (n_act << lq) > free_slots) --lq;
        if (lq < lq_hint) lq = lq_hint < 5 ? lq_hint : 5;
        if (lq < min_lq) lq = min_lq;   // (a band that consumed nothing: equal keys filled the offer -- the next offer is wider)
        float L = floor;
        int cnt = n_carry;
        while (true) {   // (repeated with a smaller offer when the band overflows the list)
          const int q = 1 << lq;
          if (tid == 0) { s_L = vm_bits(floor); s_i[S_CNT] = 0; }
          pg_barrier();
          // -- 1. L: the heaviest entry behind the offers --
          {
            uint32_t best = 0u;
            for (int it = tid; it < n_act; it += TB) {
              const int v = alist[it];
              const uint32_t g = lpos[v] + (uint32_t)q;
              if (g < lend[v]) { const float4 E = PL.ent[g]; const float k = E.x > E.y ? E.x : E.y; const uint32_t kb = vm_bits(k); best = kb > best ? kb : best; }
            }
            for (int o = 32; o > 0; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)best, o, 64); best = x > best ? x : best; }
            if (lane == 0 && best != 0u) atomicMax(&s_L, best);
          }
          pg_barrier();
          L = vm_from_bits(s_L);
          // -- 2. the offered entries heavier than L --
          const int items = n_act << lq;
          // PG_PF trips' entries are requested together: the loop waits for memory once per PG_PF trips, not once per trip
          for (int base = 0; base < items; base += TB * PG_PF) {
            float4 E4[PG_PF];
            int v4[PG_PF];
            bool has4[PG_PF];
#pragma unroll
            for (int k = 0; k < PG_PF; ++k) {
              const int item = base + k * TB + tid;
              has4[k] = item < items;
              v4[k] = has4[k] ? (int)alist[item >> lq] : 0;
              const uint32_t g = lpos[v4[k]] + (uint32_t)(item & (q - 1));
              has4[k] = has4[k] && g < lend[v4[k]];
              E4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
              if (has4[k]) E4[k] = PL.ent[g];
            }
#pragma unroll
            for (int k = 0; k < PG_PF; ++k) {
              if (base + k * TB >= items) break;   // uniform
              const int item = base + k * TB + tid;
              const int v = v4[k];
              const float4 E = E4[k];
              bool take = false, inr = false;
              uint32_t pid = 0;
              float w = 0.0f;
              if (has4[k]) {
                take = (E.x > E.y ? E.x : E.y) > L;
                if (take) {
                  const uint32_t la = hlat[v], of = __float_as_uint(E.z);
                  const int bx = (int)(la & 31u) + (int)(of & 31u) - 16, by = (int)((la >> 5) & 31u) + (int)((of >> 5) & 31u) - 16,
                            bz = (int)((la >> 10) & 31u) + (int)((of >> 10) & 31u) - 16;
                  if ((unsigned)bx < 32u && (unsigned)by < 32u && (unsigned)bz < 32u) {
                    const int vb = h_find((uint32_t)bx | ((uint32_t)by << 5) | ((uint32_t)bz << 10));
                    if (vb >= 0) {
                      w = v < vb ? E.x : E.y;   // the row's order decides which end is the weight's first argument
                      const int sa = seg[v], sb = seg[vb];
                      const float t1 = thr[sa], t2 = thr[sb];
                      // fact S: an edge at or below thr0 belongs to phase B; fact F: both ends must still be able to merge
                      inr = (w > thr0) & (sa != sb) & (t1 < lev) & (t2 < lev);
                      const int lo = v < vb ? v : vb, hi = v ^ vb ^ lo;
                      pid = ((uint32_t)lo << PSH) | (uint32_t)hi;
                    }
                  }
                }
              }
              // an entry is consumed when it is heavier than L: a prefix of the vertex's offer (the offers of one vertex sit in one wavefront)
              const unsigned long long tk = __ballot(take);
              if (item < items && (item & (q - 1)) == 0) lcons[v] = (uint16_t)__popcll(tk & (((q == 64 ? 0ull : (1ull << q)) - 1ull) << (lane & 63)));
              const unsigned long long mk = __ballot(inr);
              int b = 0;
              if (mk != 0ull && lane == 0) b = atomicAdd(&s_i[S_CNT], __popcll(mk));
              b = __builtin_amdgcn_readfirstlane(b);
              if (inr) { const int at = n_carry + b + __popcll(mk & lt_mask); if (at < LCAP) lk[at] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(PCOMP - pid); }
            }
          }
          pg_barrier();
          cnt = n_carry + s_i[S_CNT];
          if (cnt <= LCAP) break;
          if (lq == 0 || lq <= min_lq) { handed = true; break; }   // more edges than the list holds at one entry per vertex
          --lq;
          lq_hint = lq;
          pg_barrier();
        }
        if (handed) break;
        {
          bool any_c = false;
          for (int it = tid; it < n_act; it += TB) { const int v = alist[it]; const uint32_t cn = (uint32_t)lcons[v]; lpos[v] += cn; any_c = any_c || cn != 0u; }
          if (tid == 0) s_i[S_FLAG] = 0;
          pg_barrier();
          if (any_c) s_i[S_FLAG] = 1;
          pg_barrier();
          if (s_i[S_FLAG] == 0 && L > floor && n_act > 0) {   // uniform: nothing consumed although entries are left
            min_lq = lq + 1;
            if (min_lq > 6) { handed = true; break; }   // more than 64 equal keys in one list: degenerate ties
          } else {
            min_lq = 0;
          }
        }
        lq_hint = (cnt - n_carry) * 4 < free_slots ? lq + 1 : lq;
        PGP_ACC(1);
        PGP_CNT(10, cnt);
        // -- 3. sort, merge down to L --
        if (cnt > 0) {
          sort_list(cnt);
          PGP_ACC(2);
          merge_list(cnt, L);
          PGP_ACC(3);
        } else {
          if (tid == 0) s_i[S_POS] = 0;
          pg_barrier();
        }
        const int pos = s_i[S_POS];
        lev = L;
        {
          const int s0 = seg[0];
          if (s_i[S_MERGES] >= m - 1 || !(thr[s0] < L)) break;   // one segment left, or the voxel's own segment is frozen (fact F)
        }
        // -- 4. carry: the edges at or below L (one orientation of a pair above the band, the other inside) and ring edges --
        {
          // in place and ascending: a write never passes the read position; one wavefront (the tail is a handful of edges)
          if (wave == 0) {
            int kept = 0;
            for (int base = pos; base < cnt; base += 64) {
              const int e = base + lane;
              bool keep_e = false;
              uint64_t kk = 0;
              if (e < cnt) {
                kk = lk[e];
                const uint32_t pid = PCOMP - (uint32_t)kk;
                const int sa = seg[pid >> PSH], sb = seg[pid & PMASK];
                const float t1 = thr[sa], t2 = thr[sb];
                keep_e = (sa != sb) & (t1 < L) & (t2 < L);
              }
              const unsigned long long mk = __ballot(keep_e);
              wave_sync();
              if (keep_e) lk[kept + __popcll(mk & lt_mask)] = kk;
              kept += __popcll(mk);
              wave_sync();
            }
            if (lane == 0) s_i[S_POS] = kept;
          }
          pg_barrier();
          n_carry = s_i[S_POS];
        }
        PGP_ACC(4);
        if (L > floor) continue;
        if (ring_done) { phase_a_complete = true; break; }   // every list is read to its end (all keys are above thr0)
        // ---- the ring stage: pairs of still-active vertices that are not in each other's ball, evaluated here, once ----
        ring_done = true;
        {
          if (tid == 0) { s_i[S_NACT] = 0; s_i[S_NQ] = 0; s_i[S_CNT] = 0; }
          pg_barrier();
          for (int base = 0; base < m; base += TB) {
            const int v = base + tid;
            const bool act = v < m && thr[seg[v]] < lev;
            const unsigned long long mk = __ballot(act);
            int b = 0;
            if (mk != 0ull && lane == 0) b = atomicAdd(&s_i[S_NACT], __popcll(mk));
            b = __builtin_amdgcn_readfirstlane(b);
            if (act) alist[b + __popcll(mk & lt_mask)] = (uint16_t)v;
          }
          pg_barrier();
          const int na = s_i[S_NACT];
          const uint32_t Pn = (uint32_t)(na * (na - 1) / 2);
          // the queue of screened pairs sits in the upper half of the edge list; edges grow behind the carried ones in the lower half
          uint32_t* const queue = (uint32_t*)(lk + LCAP / 2);
          constexpr int QCAP = LCAP;   // 32-bit slots
          bool over = n_carry > LCAP / 4;
          for (uint32_t base = 0; base < Pn && !over; base += QCAP) {
            for (uint32_t p = base + (uint32_t)tid; p < base + QCAP; p += TB) {   // same trip count for every thread
              bool keep = false;
              int a = 0, b = 0;
              if (p < Pn) {
                int ia, ib;
                decode(p, na, Pn, ia, ib);
                const int xa = alist[ia], xb = alist[ib];
                a = xa < xb ? xa : xb; b = xa ^ xb ^ a;
                if (seg[a] != seg[b]) {
                  const uint32_t la = hlat[a], lb = hlat[b];
                  const float tx = ctab3[0][la & 31u] - ctab3[0][lb & 31u], ty = ctab3[1][(la >> 5) & 31u] - ctab3[1][(lb >> 5) & 31u],
                              tz = ctab3[2][(la >> 10) & 31u] - ctab3[2][(lb >> 10) & 31u];
                  const float d2c = (tx * tx + ty * ty) + tz * tz;
                  if (!(d2c < G.r2)) {   // not in each other's row (adjacency.hip's predicate): no list holds this pair
                    ++my_pairs;
                    NodeRec A, B;
                    if constexpr (RING_LDS) {
                      A.c[0] = rc[0][a]; A.c[1] = rc[1][a]; A.c[2] = rc[2][a]; A.n[0] = rc[3][a]; A.n[1] = rc[4][a]; A.n[2] = rc[5][a]; A.flags = __float_as_uint(rc[6][a]);
                      B.c[0] = rc[0][b]; B.c[1] = rc[1][b]; B.c[2] = rc[2][b]; B.n[0] = rc[3][b]; B.n[1] = rc[4][b]; B.n[2] = rc[5][b]; B.flags = __float_as_uint(rc[6][b]);
                    } else {
                      const NodeRec& ra = R(a); const NodeRec& rb = R(b);
                      A.c[0] = ra.c[0]; A.c[1] = ra.c[1]; A.c[2] = ra.c[2]; A.n[0] = ra.n[0]; A.n[1] = ra.n[1]; A.n[2] = ra.n[2]; A.flags = ra.flags;
                      B.c[0] = rb.c[0]; B.c[1] = rb.c[1]; B.c[2] = rb.c[2]; B.n[0] = rb.n[0]; B.n[1] = rb.n[1]; B.n[2] = rb.n[2]; B.flags = rb.flags;
                    }
                    const float dx = A.c[0] - B.c[0], dy = A.c[1] - B.c[1], dz = A.c[2] - B.c[2];
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    const uint32_t both = A.flags & B.flags;
                    if ((both & VGS_F_POS) != 0u && d2 >= P.d2_stop) {
                      keep = false;
                    } else if ((both & (VGS_F_POS | VGS_F_NRM)) == (VGS_F_POS | VGS_F_NRM) && d2 > 0.0f) {
                      int kb = (int)(d2 * P.ctab_scale);
                      kb = kb > LC_TBINS - 1 ? LC_TBINS - 1 : kb;
                      const float dot = vm_dot3(A.n, B.n);
                      keep = !(dot <= P.ctab[kb] && dot >= -1.0f);
                    } else {
                      keep = !(vm_weight_bound_da(A, B, W) <= thr0);
                    }
                  }
                }
              }
              const unsigned long long mk = __ballot(keep);
              if (mk != 0ull) {
                int qb = 0;
                if (lane == 0) qb = atomicAdd(&s_i[S_NQ], __popcll(mk));
                qb = __builtin_amdgcn_readfirstlane(qb);
                if (keep) { const int at = qb + __popcll(mk & lt_mask); if (at < QCAP) queue[at] = ((uint32_t)a << PSH) | (uint32_t)b; }
              }
            }
            pg_barrier();
            const int nq = s_i[S_NQ];
            if (nq > QCAP) { over = true; break; }
            for (int e = tid; e < nq; e += TB) {
              const uint32_t pid = queue[e];
              const float w = vm_pair_weight(R((int)(pid >> PSH)), R((int)(pid & PMASK)), W);
              if (w > thr0) {
                const int at = n_carry + atomicAdd(&s_i[S_CNT], 1);
                if (at < LCAP / 2) lk[at] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(PCOMP - pid);
              }
            }
            pg_barrier();
            if (tid == 0) s_i[S_NQ] = 0;
            if (n_carry + s_i[S_CNT] > LCAP / 2) over = true;   // uniform: written before the barrier above
            pg_barrier();
          }
          if (over) { handed = true; break; }
          n_carry += s_i[S_CNT];
          PGP_CNT(11, s_i[S_CNT]);
          pg_barrier();
        }
        PGP_ACC(5);
      }
      PGP_CNT(12, bands); (void)bands;
    }
    if (handed) { hand_on(); return false; }

    // =========================== phase B: edges at or below a singleton's threshold ===========================
    if (phase_a_complete && s_i[S_MERGES] < m - 1) {
      const int s0 = seg[0];
      const bool s0_active = (ssz[s0] >= 2) && (thr[s0] < thr0);
      if (s0_active) {   // uniform
        if (tid == 0) { s_i[S_CNT] = 0; s_i[S_FLAG] = 0; }
        if (wave == 0) {
          // the vertices of the active segments, the voxel's own segment first
          int n0 = 0;
          for (int base = 0; base < m; base += 64) {
            const int v = base + lane;
            const bool own = v < m && seg[v] == s0;
            const unsigned long long mk = __ballot(own);
            if (own) alist[n0 + __popcll(mk & lt_mask)] = (uint16_t)v;
            n0 += __popcll(mk);
          }
          int nb = n0;
          for (int base = 0; base < m; base += 64) {
            const int v = base + lane;
            bool act = false;
            if (v < m) { const int sv = seg[v]; act = sv != s0 && (ssz[sv] >= 2) && (thr[sv] < thr0); }
            const unsigned long long mk = __ballot(act);
            if (act) alist[nb + __popcll(mk & lt_mask)] = (uint16_t)v;
            nb += __popcll(mk);
          }
          if (lane == 0) { s_i[S_NACT] = nb; s_i[S_NQ] = n0; }
        }
        pg_barrier();
        const int nb = s_i[S_NACT], n0 = s_i[S_NQ];
        {
          // the voxel's segment changes only through an edge of its own heavier than its threshold (localcut_dense.hpp)
          const float Lb = thr[s0];
          const int no = nb - n0;
          bool hit = false;
          for (int i2 = tid; i2 < n0 * no; i2 += TB) {
            const int x = alist[i2 / no], y = alist[n0 + i2 % no];
            const int a = x < y ? x : y, b = x < y ? y : x;
            ++my_pairs;
            if (!(vm_weight_bound_da(R(a), R(b), W) <= Lb)) {
              const float w = vm_pair_weight(R(a), R(b), W);
              hit = hit || (w > Lb && w <= thr0);
            }
          }
          if (hit) s_i[S_FLAG] = 2;
          pg_barrier();
        }
        if (s_i[S_FLAG] == 2) {
          const uint32_t Pb = (uint32_t)(nb * (nb - 1) / 2);
          for (uint32_t p = (uint32_t)tid; p < Pb; p += TB) {
            int ia, ib;
            decode(p, nb, Pb, ia, ib);
            const int xa = alist[ia], xb = alist[ib];
            const int a = xa < xb ? xa : xb, b = xa < xb ? xb : xa;   // the list is not in vertex order
            if (seg[a] != seg[b]) {
              const float w = vm_pair_weight(R(a), R(b), W);
              ++my_pairs;
              if (w <= thr0) {   // heavier edges were examined in phase A; NaN compares false
                const int at = atomicAdd(&s_i[S_CNT], 1);
                if (at < LCAP) lk[at] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(PCOMP - (((uint32_t)a << PSH) | (uint32_t)b));
              }
            }
          }
          pg_barrier();
          const int nlB = s_i[S_CNT];
          if (nlB <= LCAP) {
            sort_list(nlB);
            merge_list(nlB, -1.0f);
          } else {
            // More such pairs than the list holds (a solid volume: thousands of vertices whose segments all stay below thr0).  Round 5 handed
            // the voxel on -- for a neighbourhood above 2048 vertices that ends in the extra-large instantiation of the general kernel, 4 s a
            // voxel on the r = 10 block.  Round 6: bands of descending weight as in localcut_dense.hpp -- one pass histograms the weights
            // (PG_NBIN bins over [0, thr0], in the lists' position words: every list is read to its end by now), then every band -- the
            // heaviest whole bins that fit the list -- is collected by a pass of its own, sorted and merged; pairs merged away meanwhile only
            // make later bands shorter; the bands end when the voxel's own segment is frozen below the next one (fact F).
            constexpr int PG_NBIN = MAXM >= 1024 ? 1024 : (MAXM >= 512 ? 512 : (MAXM >= 256 ? 256 : 128));
            uint32_t* const hist = lpos;
            const float scale = (float)PG_NBIN / thr0;
            auto bin_of = [&](float w) -> int { const int bb = (int)(w * scale); return bb < 0 ? 0 : (bb > PG_NBIN - 1 ? PG_NBIN - 1 : bb); };
            for (int k = tid; k < PG_NBIN; k += TB) hist[k] = 0u;
            pg_barrier();
            for (uint32_t p = (uint32_t)tid; p < Pb; p += TB) {
              int ia, ib;
              decode(p, nb, Pb, ia, ib);
              const int xa = alist[ia], xb = alist[ib];
              const int a = xa < xb ? xa : xb, b = xa < xb ? xb : xa;
              if (seg[a] != seg[b]) {
                const float w = vm_pair_weight(R(a), R(b), W);
                ++my_pairs;
                if (w <= thr0) atomicAdd(&hist[bin_of(w)], 1u);
              }
            }
            pg_barrier();
            int top = PG_NBIN;   // bins [top, PG_NBIN) are done
            while (true) {
              if (tid == 0) {
                unsigned int acc = 0;
                int lo = top;
                while (lo > 0 && acc + hist[lo - 1] <= (unsigned int)LCAP) { --lo; acc += hist[lo]; }
                s_i[S_NQ] = lo;
                s_i[S_CNT] = 0;
              }
              pg_barrier();
              const int lo = s_i[S_NQ];
              if (lo == top) { hand_on(); return false; }   // one bin alone overflows the list: degenerate ties
              for (uint32_t p = (uint32_t)tid; p < Pb; p += TB) {
                int ia, ib;
                decode(p, nb, Pb, ia, ib);
                const int xa = alist[ia], xb = alist[ib];
                const int a = xa < xb ? xa : xb, b = xa < xb ? xb : xa;
                if (seg[a] != seg[b]) {
                  const float w = vm_pair_weight(R(a), R(b), W);
                  ++my_pairs;
                  if (w <= thr0) {
                    const int bb = bin_of(w);
                    if (bb >= lo && bb < top) {
                      const int at = atomicAdd(&s_i[S_CNT], 1);
                      if (at < LCAP) lk[at] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(PCOMP - (((uint32_t)a << PSH) | (uint32_t)b));
                    }
                  }
                }
              }
              pg_barrier();
              const int nband = s_i[S_CNT] < LCAP ? s_i[S_CNT] : LCAP;   // <= the histogram's count of these bins
              sort_list(nband);
              merge_list(nband, -1.0f);
              if (lo == 0) break;
              top = lo;
              // every pair left has a bin below `top`, so it weighs less than this
              const float wub = (float)top / scale * 1.0001f;
              const int r0 = seg[0];   // flattened by merge_list
              if (!(thr[r0] < wub) || s_i[S_MERGES] >= m - 1) break;
              pg_barrier();   // (the counters are rewritten at the top)
            }
            if (tid == 0) atomicAdd(&counters[0], 1ull);   // "banded"
          }
          PGP_CNT(13, 1);
        }
        PGP_ACC(6);
      }
    }
    // ---- result: the segment of the voxel itself, the whole row (nobody zeroes the table first) ----
    {
      const int s0 = seg[0];
      for (int c = tid; c < m; c += TB) crow[c] = (seg[c] == s0) ? 1 : 0;
      if (cbits) {
        // ... and as one bit per ball offset for crossValidation's lattice lookup (localcut_wave.hpp writes the same row): the edge
        // list's LDS is free now
        uint32_t* const cb = (uint32_t*)lk;
        pg_barrier();
        for (int k = tid; k < cb_words; k += TB) cb[k] = 0u;
        pg_barrier();
        for (int c = tid; c < m; c += TB)
          if (seg[c] == s0) { const uint32_t bi = vgs_cb_index(orow[c], cb_R); atomicOr(&cb[bi >> 5], 1u << (bi & 31u)); }
        pg_barrier();
        uint32_t* const outb = cbits + (size_t)u * (size_t)cb_words;
        for (int k = tid; k < cb_words; k += TB) outb[k] = cb[k];
      }
    }
    for (int o = 32; o > 0; o >>= 1) my_pairs += __shfl_xor(my_pairs, o, 64);
    if (lane == 0 && my_pairs) atomicAdd(&s_i[S_PAIRS], (int)my_pairs);
    pg_barrier();
    if (tid == 0) evals_out[u] = (uint32_t)s_i[S_PAIRS];
    PGP_CNT(14, 1);
    PGP_ACC(7);
    return true;
  };   // process

  // Either hand-over lists (largest neighbourhoods first, work_stride apart, their lengths on the device: a fixed grid strides over
  // them) or a class of its own (one list, its length known to the host; workgroup b runs on XCD b % 8 and every XCD takes one
  // contiguous eighth of the list, so that neighbouring voxels share their L2)
  const unsigned int xcd_total = (((n_work_host + 7u) >> 3) << 3), per_xcd = (n_work_host + 7u) >> 3;
  for (unsigned int wi = blockIdx.x;; wi += gridDim.x) {
    uint32_t u = 0;
    if (n_work_dev) {
      unsigned int wpos = wi;
      int wbin = 0;
      while (wbin < n_lists && wpos >= n_work_dev[wbin]) { wpos -= n_work_dev[wbin]; ++wbin; }
      if (wbin == n_lists) break;
      u = work[(size_t)wbin * work_stride + wpos];
    } else {
      if (wi >= xcd_total) break;
      const unsigned int it = xcd_order ? (wi & 7u) * per_xcd + (wi >> 3) : wi;
      if (it >= n_work_host || (xcd_order && (wi >> 3) >= per_xcd)) continue;
      u = work[it];
    }
    if (process((uint32_t)__builtin_amdgcn_readfirstlane((int)u))) ++n_cut;
    pg_barrier();   // the next voxel reuses every array
  }
  if (tid == 0 && n_cut) atomicAdd(&counters[63], (unsigned long long)n_cut);
}

#endif
