// localcut_wave.hpp -- the fast path of the local graph cut: ONE WAVEFRONT PER VOXEL, lazy evaluation.
// Included by localcut.hip (needs LcParams and wave_sync from there).
//
// Same sequential semantics as k_localcut, different schedule.  The reference evaluates all
// n^2 weights and sorts them, yet on a surface the cut is decided by the few hundred heaviest edges: the
// descending scan merges the neighbourhood along its nearest-neighbour edges and then nothing can merge any more.
//
// Three facts make a lazy schedule exact:
//  (U) proximity bounds the weight:  D >= dist_space / sig_p  =>  w <= ub(d) = exp(-0.5 * d / sig_p / sig_w^2)
//      (VS:1736-1737; every float step of the evaluation is monotone).  Pairs are therefore evaluated
//      in shells of increasing centroid distance and the scan only advances down to level = ub(shell radius):
//      every edge not yet evaluated is provably lighter than the level.
//  (F) a segment's threshold seg_int - cut/size moves only when the segment merges, and a merge needs an edge
//      heavier than the threshold.  Once the scan is at `level`, a segment with threshold >= level is frozen for
//      ever: its vertices leave the pair enumeration and its edges are dropped.
//  (S) a singleton's threshold is the constant 1 - cut, so an edge with w <= 1 - cut can only ever join two
//      NON-singleton segments.  Phase A handles all edges heavier than 1 - cut (shell by shell) and does not even
//      store lighter ones; afterwards every singleton is frozen, and phase B re-evaluates the (few) pairs between
//      the non-singleton segments that are still below their thresholds and finishes the scan on them.
//
// Per shell: (1) enumerate candidate pairs by squared centroid distance (2 LDS reads, ~10 VALU per pair, no
// transcendental); (2) full weight only for the shell's pairs, one per lane; (3) LDS bitonic sort of <= 512 keys
// instead of n^2; (4) sequential merge, 64 edges per step, segment ids / thresholds / sizes of the step held in
// registers and patched in place after every merge (no LDS round trip on the critical path).
#ifndef LOCALCUT_WAVE_HPP_
#define LOCALCUT_WAVE_HPP_

#include <type_traits>

#include "nearlist.hpp"
#include "regsort.hpp"

// VGS_PROF=1 builds accumulate per-phase shader cycles (s_memtime) into counters[16..31] (diagnostics only)
#ifdef VGS_PROF
#define LW_T0() long long _t0 = clock64()
#define LW_ACC(slot) do { long long _t1 = clock64(); if (lane == 0) prof[slot] += (unsigned long long)(_t1 - _t0); _t0 = _t1; } while (0)
#define LW_CNT(slot, v) do { if (lane == 0) prof[slot] += (unsigned long long)(v); } while (0)
#else
#define LW_T0() do {} while (0)
#define LW_ACC(slot) do {} while (0)
#define LW_CNT(slot, v) do {} while (0)
#endif

// wavefronts per SIMD the compiler must leave room for (register budget 512 / LW_WAVES).  6 and 7 run equally fast on
// URB10M; at 7 (72 registers) the kernel spills 48 bytes per lane to scratch (+1.1 GB of HBM writes per launch, measured
// with WRITE_SIZE), at 6 (80 registers) two registers, once.
#ifndef LW_WAVES
#define LW_WAVES 6
#endif
// pairs per lane and trip of the pair enumeration (their LDS reads are issued together); 4 spills 6 registers at 72
#ifndef LW_TRIP
#define LW_TRIP 3
#endif
// (A bucket sort of the bulk class's edge lists by weight -- LDS histogram, scan, rank inside the bucket; a third of the LDS
// network's instructions -- was measured in round 3: exact, but slower, 4.7 against 4.2 ms in the same build: its atomics, the scan
// and the rank loops are chains of dependent LDS round trips.  What did pay is regsort.hpp: no LDS round trips at all.)
// phase A of the one-wavefront classes sorts its edge list in registers (regsort.hpp) instead of through LDS
#ifndef LW_REG_SORT
#define LW_REG_SORT 4
#endif
// ... on one-word keys (regsort::sort_desc32; 0: the 64-bit network, for A/B builds)
#ifndef LW_SORT32
#define LW_SORT32 1
#endif
// lanes per vertex when the first shell is read from the near-pair lists (8 or 16; see near_enum)
#ifndef LW_CUT_OVER_ONE
#define LW_CUT_OVER_ONE 1
#endif
#ifndef LW_NEAR_LPV1
#define LW_NEAR_LPV1 5
#endif
#ifndef LW_NEAR_LPV2
#define LW_NEAR_LPV2 8
#endif
// groups of vertices whose near-pair list entries are requested together (first shells)
#ifndef LW_NEAR_GROUPS
#define LW_NEAR_GROUPS 2
#endif
// class C0 (MAXM 320, 26 KB of LDS): six workgroups fit a CU only if its wavefronts keep to the 80 registers of six per SIMD (at four per
// SIMD the compiler takes 91 and the CU holds five workgroups, as for class C): config 2 10.5 -> 10.1 ms, no spilled vector register
#ifndef LW_WAVES_C0
#define LW_WAVES_C0 6
#endif
#ifndef LW_MASTER_ROT
#define LW_MASTER_ROT 0
#endif

// hand-over marks of the one-wavefront classes: pending[u] = 1 + list + LW_HO_BINS * why; k_ho_lists counts the reasons into these words
#define LW_PENDING_LISTED 0xff   // pending[u] of a voxel that is already in a hand-over list (k_ho_lists leaves it alone)
#define LW_HO_VOTE 0x100
#define LW_VOTE_BASE 64    // counter words 64 .. 127
#ifndef LW_VOTE_WORDS
#define LW_VOTE_WORDS 64u  // words in use (a power of two <= 64).  NOT fewer: every wavefront of the bulk launch reads one of them past the scalar
                           // cache, and with 16 words (two cache lines) those reads pile up on two L2 channels -- the bulk kernel took 5.9 instead of 2.7 ms
#endif
enum { LW_WHY_GAVE_UP = 0, LW_WHY_VOTED, LW_WHY_SIZE, LW_N_WHY };   // (one reason for every way the lazy schedule gives a voxel up: the kernel has no register to spare for more)
struct LwParams {
  LcParams lc;
  float r2_graph;   // graph_size^2: scale of the first shell
  float d2_all;     // squared distance no pair of one neighbourhood can reach: last shell is open ended
  float shell0;     // first shell = shell0 * r2_graph / m
  float grow;       // shell growth factor (in squared distance)
  float cap_frac;   // the first shell is sized for at most this fraction of the list
  int dbg_stop;     // diagnostics: leave the first round after step N (0 = run normally)
  int max_rounds;   // shells a wavefront works through before it hands the voxel over (classes A/B)
  int dbg_max_m;    // tests: hand over neighbourhoods larger than this (0 = the kernel's own limit)
  NearLists near;   // per-voxel lists of the heavy pairs within two lattice steps (nearlist.hpp): the first shell walks them
  uint8_t* pending; // per used voxel: set when the voxel is handed over (its connect row is final only after the hand-over kernel)
  int near_min_own; // multi-wavefront classes: first shells from the near-pair lists only if the vertices' lists hold this many entries on average
  // One-wavefront classes, bit LW_HO_VOTE of ho_bins: is the lazy schedule worth trying on this scene at all?  Every sixteenth voxel of
  // these classes is a SAMPLE: it runs in an instantiation of its own (SAMPLED, launched beside the others), always tries, and books how it
  // ended -- finished / gave the voxel up -- in one of 64 counter words (LW_VOTE_BASE; low / high half).  The others read one of the words
  // when they start and, once seven of eight samples gave up, hand their voxel over without trying.  Under centimetres of range noise
  // 98 % of the voxels work through their shells only to be handed over; the hand-over kernel reads pair lists and does not care.
  // Scheduling only: both paths give the same connect list.  (The bookkeeping is in an instantiation of its own because the kernel has
  // no register to spare: two more scalars alive to its end cost the bulk launch four spilled vector registers.)
  int ho_bins;      // one-wavefront classes: hand-over lists by neighbourhood size (LW_HO_BINS, largest first) or 1; | LW_HO_VOTE
  int ho_stride;    // distance between those lists in the hand-over array (= the number of used voxels)
                    // (one-wavefront classes with ho_bins > 1: a hand-over only marks pending[u] = 1 + list + LW_HO_BINS * why and k_ho_lists builds the lists)
  // the connect list once more as a bit per ball offset, for crossValidation's lattice lookup (vgs_context.hpp: conn_bits); null: off
  uint32_t* cbits;
  int cb_R;
  int cb_words;
  // bulk launch: lengths of the first and the main work list read on the device (the host launches with a grid for their sum); null: n_first / n_work
  const unsigned int* n_first_dev;
  const unsigned int* n_main_dev;
};

// The hand-over kernel's cost grows with the square of the neighbourhood size and its launch ends with the slowest voxel:
// the one-wavefront classes sort their hand-overs into lists by size, and the launch takes the list of the largest first.
#define LW_HO_BINS 4
__device__ __forceinline__ int lw_ho_bin(int m) { return m > 112 ? 0 : (m > 96 ? 1 : (m > 64 ? 2 : 3)); }

__device__ __forceinline__ float lw_readlane_f(float x, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}
// LDS diet.  The kernel is bound by dependent LDS/L2 latency, not by issue slots (measured: time ~ 2.9 ms + 99 ms /
// (wavefronts per CU) on URB10M), so the footprint is cut to 5 KB per wavefront = the 32-wavefronts-per-CU cap:
// centroids SoA without a flag word (an unusable position is a NaN x), one-byte vertex/segment indices up to 256
// vertices, and the edge list as two arrays (weight bits, complemented pair id) instead of one 8-byte key.
//
// NW > 1: NW wavefronts per voxel share the LDS arrays (large neighbourhoods: 34 KB per voxel would otherwise leave one
// wavefront per SIMD).  Wavefront 0 runs the algorithm; the others wait at a workgroup barrier for the three heavy,
// order-free sections -- pair enumeration, weight evaluation, sort -- take their share, and wait again.  The merge and
// the bookkeeping between shells stay on wavefront 0.  Candidates are then appended through an LDS counter, so their
// order in the list depends on timing; the sort that follows removes that (keys are unique).
template <int MAXM, int LCAP, int NW = 1, bool SAMPLED = false, bool SORT32 = false>
__global__ __launch_bounds__(64 * NW, MAXM <= 255 ? LW_WAVES : (NW > 1 ? (MAXM == 320 ? LW_WAVES_C0 : 4) : 2)) void k_localcut_wave(const uint32_t* __restrict__ work_first, int n_first,
                                                      const uint32_t* __restrict__ work, int n_work,
                                                      const unsigned int* __restrict__ n_work_dev,
                                                      const uint64_t* __restrict__ adj_key, const uint32_t* __restrict__ adj_cnt,
                                                      int adj_stride, const NodeRec* __restrict__ node, LwParams P,
                                                      uint8_t* __restrict__ conn, unsigned long long* __restrict__ counters,
                                                      uint32_t* __restrict__ fallback, unsigned int* __restrict__ n_fallback,
                                                      uint32_t* __restrict__ evals_out, uint32_t* __restrict__ dbg_out,
                                                      const uint16_t* __restrict__ adj_off) {
  constexpr bool SMALL = MAXM <= 255;  // indices and segment sizes fit a byte
  typedef typename std::conditional<SMALL, uint8_t, uint16_t>::type idx_t;   // vertex / segment index, segment size
  // pair id = (a << PSH) | b with a < b
  constexpr int PSH = SMALL ? 8 : 16;
  constexpr uint32_t PMASK = (1u << PSH) - 1u;
  constexpr uint32_t PCOMP = SMALL ? 0xffffu : 0xffffffffu;  // stored complemented: descending (w, ~pid) = w desc, pid asc
  // edge list: 64-bit keys, weight bits above the complemented pair id -- one compare orders (w desc, pid asc), one LDS
  // access moves an edge.  A dropped entry is 0, below every real edge; a candidate that has no weight yet is its plain pid.
  __shared__ uint64_t lk[LCAP];
  // centroids (cx = NaN when the position is unusable, VS:1829) -- or, while the first shell is read from the near-pair
  // lists, the lattice offset of every vertex from the voxel and the vertex sitting at every offset of the 11^3 ball
  // (0xff = none).  The two never live at the same time: the general enumeration stages the centroids when it first runs.
  constexpr bool NEAR = SMALL && NW == 1;
  constexpr int NMAP_DIM = 2 * NL_BALL + 1;
  constexpr int NMAP_BYTES = (NMAP_DIM * NMAP_DIM * NMAP_DIM + 3) / 4 * 4;
  constexpr int HCAP_ = (!SMALL || NW > 1) ? (MAXM <= 512 ? 1024 : 2048) : 1;   // (the hash of the multi-wavefront classes, see below)
  constexpr int CBUF_BYTES = (NEAR && NMAP_BYTES + 2 * MAXM > 12 * MAXM) ? NMAP_BYTES + 2 * MAXM
                           : (((!SMALL || NW > 1) && HCAP_ * 4 + 2 * MAXM > 12 * MAXM) ? HCAP_ * 4 + 2 * MAXM : 12 * MAXM);
  __shared__ __attribute__((aligned(16))) unsigned char cbuf[CBUF_BYTES];
  float* const cx = (float*)cbuf;
  float* const cy = cx + MAXM;
  float* const cz = cy + MAXM;
  uint32_t* const nmap4 = (uint32_t*)cbuf;
  uint8_t* const nmap = cbuf;
  uint16_t* const nlat = (uint16_t*)(cbuf + NMAP_BYTES);   // (dx+5) | (dy+5) << 4 | (dz+5) << 8
  __shared__ float thr[MAXM];
  __shared__ uint32_t claim[MAXM];                // merge: first undecided edge of the step touching a segment (all ones between uses)
  constexpr bool GID_LDS = false;  // ids are read from the adjacency row (L2) when a record is needed: the LDS copy bought nothing and costs list slots
  __shared__ uint32_t gid[GID_LDS ? MAXM : 1];    // global voxel ids: the few pairs that get a full evaluation read their records through L2
  __shared__ idx_t seg[MAXM], rep[MAXM], ssz[MAXM];
  __shared__ idx_t alist[MAXM];  // vertices whose segment can still merge (ascending)
  __shared__ idx_t minor[MAXM];  // active vertices outside the largest active segment (ascending)
  __shared__ int sh_i[16];       // NW > 1: command and arguments of the current section, its results
  __shared__ float sh_f[4];
  enum { SH_CMD = 0, SH_NLIST, SH_NACT, SH_NMIN, SH_BIG, SH_MINOR, SH_FINAL, SH_MERGED, SH_PACT, SH_COUNT, SH_DROPPED, SH_CNT, SH_NEARBAD, SH_NEARSUM };
  enum { CMD_QUIT = 0, CMD_ENUM, CMD_EVAL, CMD_SORT, CMD_NEAR, CMD_STAGE };
  // Multi-wavefront classes (NEARH): the near-pair lists are read through a HASH of "lattice offset from the voxel -> vertex"
  // (search balls up to 15 voxels: the direct 31^3 map of the one-wavefront classes would not fit, a 21^3 one was measured
  // and cost two of five workgroups per CU).  Entry = (packed offset + 1) << 16 | vertex, 0 = empty; linear probing at load
  // <= 1/2; hlat[v] = v's packed offset, 5 bits per axis (offset + 16).
  constexpr bool NEARH = !SMALL || NW > 1;
  // Both live in the centroids' bytes (as the one-wavefront classes' map does): the general enumeration stages the
  // centroids when it first runs, and the lists are not read after that -- with arrays of its own the class-C workgroup
  // grew from 31.8 to 36.9 KB, four per CU instead of five, and every scene got 18 % slower.
  constexpr int HCAP = NEARH ? HCAP_ : 1;
  static_assert(!NEARH || HCAP * 4 + 2 * MAXM <= CBUF_BYTES, "hash and offsets must fit the centroid buffer");
  uint32_t* const htab = (uint32_t*)cbuf;
  uint16_t* const hlat = (uint16_t*)(cbuf + (size_t)HCAP * 4);
  auto h_slot = [&](uint32_t key15) -> uint32_t { return (key15 * 2654435761u) >> (32 - (HCAP == 1024 ? 10 : 11)); };
  auto h_find = [&](uint32_t key15) -> int {   // vertex at that offset, -1 if none
    uint32_t h = h_slot(key15);
    while (true) {
      const uint32_t e = htab[h & (HCAP - 1)];
      if (e == 0u) return -1;
      if ((e >> 16) == key15 + 1u) return (int)(e & 0xffffu);
      ++h;
    }
  };

  int lane = threadIdx.x & 63;   // (not const: see the top of the shell loop)
  // NW > 1: the wavefront that runs the sequential parts ("wavefront 0" below) is a different hardware wavefront from workgroup to
  // workgroup (LW_MASTER_ROT): the workgroups of one CU then keep their busy wavefronts on different SIMDs instead of all on the
  // SIMD that hosts every workgroup's first wavefront -- one wavefront alone issues a VALU instruction every 5.6 cycles, a SIMD one
  // every 2.2 (tools/valu_roof.hip), so three such wavefronts on one SIMD already wait for each other.
#if LW_MASTER_ROT
  const int wave = NW > 1 ? (int)(((threadIdx.x >> 6) + blockIdx.x) & (unsigned)(NW - 1)) : 0;
#else
  const int wave = threadIdx.x >> 6;
#endif
  auto blk_sync = [&]() { if constexpr (NW > 1) __syncthreads(); else wave_sync(); };
  // workgroup b runs on XCD b % 8 (observed; used for speed only): give every XCD one contiguous eighth of the
  // Morton-ordered work list so that neighbouring voxels share their L2
  // An optional first list (the heavier voxels of the launch) is dealt out the same way before the main list, so
  // that the long-running wavefronts start first and the light ones fill the tail.
  // A hand-over list is launched with a fixed grid while its length is still on the device (n_work_dev).
  // It is taken in plain order, so that a list longer than the grid leaves a suffix for the caller.
  if (n_work_dev) n_work = (int)*n_work_dev;
  if (P.n_first_dev) { n_first = (int)*P.n_first_dev; n_work = (int)*P.n_main_dev; }
  const unsigned int nb_first = ((unsigned int)n_first + 7u) & ~7u;
  const bool first = blockIdx.x < nb_first;
  const unsigned int bidx = first ? blockIdx.x : blockIdx.x - nb_first;
  const int n_mine = first ? n_first : n_work;
  const int per_xcd = (n_mine + 7) >> 3;
  const int widx = n_work_dev ? (int)bidx : (int)(bidx & 7u) * per_xcd + (int)(bidx >> 3);
  if (widx >= n_mine) return;
  if (!n_work_dev && (int)(bidx >> 3) >= per_xcd) return;   // a grid sized for an upper bound of the list (lengths read on the device): surplus workgroups must not wrap into another XCD's share
  const uint32_t u = (first ? work_first : work)[widx];
  const int n = (int)adj_cnt[u];
  const uint64_t* row = adj_key + (int64_t)u * adj_stride;
  uint8_t* crow = conn + (int64_t)u * adj_stride;
  const float cut = P.lc.cut;
  const VgsWeightParams& W = P.lc.W;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  auto diag = [&](int word) {   // lane 0 calls
    if constexpr (SMALL && NW == 1) { if ((P.ho_bins & 0xff) > 1) return; }   // (k_ho_lists counts the marks: no same-address atomics from a hundred thousand voxels)
    atomicAdd(&counters[word], 1ull);
  };
#ifdef VGS_PROF
  unsigned long long prof[16] = {0};
#endif
  LW_T0();
#ifdef VGS_PROF
  const long long t_start = clock64();
#endif

  // ---- the neighbourhood in adjacency order (the adjacency stage already dropped inert unused voxels) ----
  auto R = [&](int v) -> const NodeRec& { return node[GID_LDS ? gid[v] : (uint32_t)row[v]]; };
  const int m = n;
  if (m > MAXM || (P.dbg_max_m > 0 && m > P.dbg_max_m)) {  // beyond this kernel's arrays: hand over to the general kernel
    if (threadIdx.x == 0) {
      const int bin = (SMALL && NW == 1 && (P.ho_bins & 0xff) > 1) ? lw_ho_bin(m) : 0;
      if (SMALL && NW == 1 && (P.ho_bins & 0xff) > 1) P.pending[u] = (uint8_t)(1 + bin + LW_HO_BINS * LW_WHY_SIZE);
      else { fallback[(size_t)bin * P.ho_stride + atomicAdd(n_fallback + bin, 1u)] = u; P.pending[u] = LW_PENDING_LISTED; }
    }
    return;
  }
  if constexpr (SMALL && NW == 1) {
    if (!SAMPLED && (P.ho_bins & LW_HO_VOTE) != 0 && m >= 2) {
      // the scene's samples (see LwParams::ho_bins): a scalar load that bypasses the scalar cache, which does not see the samples'
      // atomics -- no vector register is held for it, and the wavefront that hands over leaves here
      unsigned long long vote_word;
      const unsigned long long* vp = counters + LW_VOTE_BASE + (blockIdx.x & (LW_VOTE_WORDS - 1u));
      asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(vote_word) : "s"(vp) : "memory");
      const uint32_t fin = (uint32_t)vote_word, gave = (uint32_t)(vote_word >> 32);
      if ((fin + gave >= 8u) && (gave * 8u >= (fin + gave) * 7u)) {
        if (lane == 0) P.pending[u] = (uint8_t)(1 + lw_ho_bin(m) + LW_HO_BINS * LW_WHY_VOTED);
        return;
      }
    }
  }
  const float thr0 = vm_cut_threshold(1.0f, cut, 1);  // a singleton's threshold: seg_int = 1 (VS:1918)
  // cut / size, looked up across lanes.  (A table in LDS instead -- one read where the cross-lane lookup costs two ds_bpermute -- was
  // measured in round 3: its 384 bytes take the workgroup from 6368 to 6752 bytes of LDS, 22 instead of 24 wavefronts per CU, and
  // the launch went from 3.9 to 4.2 ms.  The kernel follows its occupancy, not its instruction count.)
  const float cut_tab0 = cut / (float)(lane + 1), cut_tab1 = cut / (float)(lane + 65);
  bool near_ok = false;   // wave-uniform: the first shell can be read from the near-pair lists
  bool cen_ready = true;  // centroids are staged
  // The lattice offset of every vertex from the voxel comes with the adjacency row (adj_off, written by the adjacency stage):
  // one coalesced 2-byte read per vertex instead of a gather of the neighbours' records.  orow[0] = 0xffff: this row has none.
  const uint16_t* orow = adj_off ? adj_off + (int64_t)u * adj_stride : nullptr;
  if constexpr (NEAR) {
    near_ok = P.near.enabled != 0 && P.near.direct != 0 && orow != nullptr;
    if (near_ok) {
      for (int k = lane; k < NMAP_BYTES / 4; k += 64) nmap4[k] = 0xffffffffu;
      wave_sync();
    }
  }
  if constexpr (NEARH) {
    near_ok = P.near.enabled != 0 && orow != nullptr;
    if (near_ok) {
      for (int k = (int)threadIdx.x; k < HCAP; k += 64 * NW) htab[k] = 0u;
      if (threadIdx.x == 0) { sh_i[SH_NEARBAD] = 0; sh_i[SH_NEARSUM] = 0; }
      blk_sync();
    }
  }
  auto stage_centroids = [&]() {
    for (int c = (int)threadIdx.x; c < m; c += 64 * NW) {
      const NodeRec& rc = node[(uint32_t)row[c]];
      cx[c] = (rc.flags & VGS_F_POS) ? rc.c[0] : vm_nan();
      cy[c] = rc.c[1];
      cz[c] = rc.c[2];
    }
  };
  bool near_bad = false;
  int near_sum = 0;   // multi-wavefront classes: sum of the vertices' near-list lengths
  uint32_t tid_reg0 = 0u, tid_reg1 = 0u;   // one-wavefront classes: global ids of vertices lane and lane + 64 (MAXM <= 128)
  for (int c = (int)threadIdx.x; c < m; c += 64 * NW) {
    const uint32_t t = (uint32_t)row[c];
    if constexpr (NEAR) { if (c < 64) tid_reg0 = t; else tid_reg1 = t; }
    if (GID_LDS) gid[c] = t;
    seg[c] = (idx_t)c; rep[c] = (idx_t)c; ssz[c] = 1; thr[c] = thr0; alist[c] = (idx_t)c; claim[c] = 0xffffffffu;
    if constexpr (NEAR) {
      if (near_ok) {
        const uint32_t pk = orow[c];
        const int ox = (int)(pk & 31u) - 16 + NL_BALL, oy = (int)((pk >> 5) & 31u) - 16 + NL_BALL, oz = (int)((pk >> 10) & 31u) - 16 + NL_BALL;
        const bool inb = pk != 0xffffu && (unsigned)ox < (unsigned)NMAP_DIM && (unsigned)oy < (unsigned)NMAP_DIM && (unsigned)oz < (unsigned)NMAP_DIM;
        if (inb) {
          nlat[c] = (uint16_t)(ox | (oy << 4) | (oz << 8));
          nmap[(oz * NMAP_DIM + oy) * NMAP_DIM + ox] = (uint8_t)c;
        }
        // a vertex without a list of its own (NL_NONE) shows when its list is first read: entry 0 holds a NaN distance (near_enum)
        near_bad = near_bad || !inb;
      }
    }
    if constexpr (NEARH) {
      if (near_ok) {
        const uint32_t pk = orow[c];
        const bool inb = pk != 0xffffu;   // balls end at 15 voxels: every offset fits 5 bits per axis
        if (inb) {
          const uint32_t key15 = pk;
          hlat[c] = (uint16_t)key15;
          uint32_t h = h_slot(key15);
          while (atomicCAS(&htab[h & (HCAP - 1)], 0u, ((key15 + 1u) << 16) | (uint32_t)c) != 0u) ++h;
        }
        if (!inb || P.near.cnt[t] == NL_NONE) sh_i[SH_NEARBAD] = 1;
        near_sum += (int)P.near.tot[t];
      }
    }
  }
  if constexpr (NEARH) {
    if (near_ok) {   // uniform
      for (int o = 32; o > 0; o >>= 1) near_sum += __shfl_xor(near_sum, o, 64);
      if (lane == 0) atomicAdd(&sh_i[SH_NEARSUM], near_sum);
    }
  }
  if constexpr (NEAR) {
    if (near_ok) {
      near_ok = __ballot(near_bad) == 0ull;
      cen_ready = false;
    }
  }
  if constexpr (NEARH) {
    blk_sync();
    // not in clutter: where the vertices have few heavy near pairs the first shells are nearly empty in the lists -- rounds that
    // cost their bookkeeping and bring no merges (measured on noisy surfaces: +9 %)
    if (near_ok) near_ok = sh_i[SH_NEARBAD] == 0 && sh_i[SH_NEARSUM] >= P.near_min_own * m;
    cen_ready = !near_ok;
    if (!near_ok) { blk_sync(); stage_centroids(); }   // all wavefronts are still here
    blk_sync();
  } else {
    if (!near_ok) { if (!cen_ready) wave_sync(); stage_centroids(); cen_ready = true; }
    blk_sync();
  }

  LW_ACC(0);  // gather
  if (P.shell0 < 0.0f) return;  // diagnostics: gather-only run
  unsigned int n_evals = 0;   // pair evaluations of this voxel (wave-uniform)
  int merges = 0;
  bool bail = false;
  // Descending sort of the edge list [0, cnt): the bitonic network in its one-direction form (each merge starts with a
  // mirror step, every comparator puts the larger key at the lower index).  Slots >= cnt then act as keys below every
  // real one that never move, so cnt need not be a power of two and nothing is padded.
  auto sort_section = [&](int cnt) __attribute__((always_inline)) {   // all wavefronts of the workgroup
    int np = 64;
    while (np < cnt) np <<= 1;
    auto cmpx = [&](int lo, int hi) {
      if (hi >= cnt) return;
      const uint64_t x = lk[lo], y = lk[hi];
      if (x < y) { lk[lo] = y; lk[hi] = x; }
    };
    blk_sync();
    if constexpr (NW > 1 && LW_REG_SORT != 0) {   // 512 keys per wavefront in registers, the widest strides through LDS
      regsort::sort_desc_block<NW>(lk, cnt, wave, lane, [&]() { __syncthreads(); });
      return;
    }
    for (int size = 2, sbit = 1; size <= np; size <<= 1, ++sbit) {
      for (int t = (int)threadIdx.x; t < (np >> 1); t += 64 * NW) {
        const int blk = t >> (sbit - 1), i = t & ((size >> 1) - 1);
        cmpx((blk << sbit) + i, (blk << sbit) + size - 1 - i);
      }
      blk_sync();
      for (int sl = sbit - 2; sl >= 0; --sl) {
        const int strd = 1 << sl;
        for (int t = (int)threadIdx.x; t < (np >> 1); t += 64 * NW) {
          const int lo = ((t >> sl) << (sl + 1)) | (t & (strd - 1));
          cmpx(lo, lo + strd);
        }
        blk_sync();
      }
    }
  };

  // ---- the two other shared sections ----
  // candidates of one shell: pids appended behind the n_list carried edges; returns the count (NW == 1) or adds to sh_i[SH_COUNT]
  auto enum_section = [&](int n_list, int n_act, int n_min, int big, bool use_minor, bool final_round, bool merged, int Pact,
                          float cut_lo, float cut_hi) -> int {
    int count = 0;
    auto append = [&](bool inr, uint32_t pid) {
      const unsigned long long mk = __ballot(inr);
      int base;
      if constexpr (NW == 1) {
        base = count;
        count += __popcll(mk);
      } else {
        int b = 0;
        if (mk != 0ull && lane == 0) b = atomicAdd(&sh_i[SH_COUNT], __popcll(mk));
        base = __builtin_amdgcn_readfirstlane(b);
      }
      if (inr) {
        const int pos = n_list + base + __popcll(mk & lt_mask);
        if (pos < LCAP) lk[pos] = (uint64_t)pid;
      }
    };
    if (use_minor) {
      // pairs (x, y): x outside the largest segment, y any active vertex of another segment; a pair of two
      // outside vertices is taken once (x < y)
      for (int ix = wave; ix < n_min; ix += NW) {
        const int x = minor[ix];
        const float pxx = cx[x], pxy = cy[x], pxz = cz[x];
        const int sx = seg[x];
        for (int base = 0; base < n_act; base += 64) {
          const int iy = base + lane;
          bool inr = false;
          uint32_t pid = 0;
          if (iy < n_act) {
            const int y = alist[iy];
            const int sy = seg[y];
            if (sy != sx && (sy == big || x < y)) {
              const float dx = pxx - cx[y], dy = pxy - cy[y], dz = pxz - cz[y];  // (a-b)^2 == (b-a)^2: order-free
              float d2 = (dx * dx + dy * dy) + dz * dz;
              d2 = (d2 == d2) ? d2 : 1.0e4f;  // dist_space stays 100 when a centroid has a zero component (VS:1829)
              inr = (d2 >= cut_lo) && (final_round || d2 < cut_hi);
              pid = (x < y) ? (((uint32_t)x << PSH) | (uint32_t)y) : (((uint32_t)y << PSH) | (uint32_t)x);
            }
          }
          append(inr, pid);
        }
      }
    } else {
      // each lane walks pairs p = lane, lane + 64, ... in row-major (ia, ib) order; LW_TRIP pairs per trip so that the
      // LDS reads of a trip are issued together
      // pair p (row-major over ia < ib) is decoded arithmetically, counting from the END of the triangle:
      // q = P-1-p lies in row r = floor((sqrt(8q+1)-1)/2) from the end (8q+1 < 2^24: exact in float at the row starts)
      const bool ident = (n_act == m);  // alist is still the identity
      for (int base = 64 * LW_TRIP * wave; base < Pact; base += 64 * LW_TRIP * NW) {
        int va[LW_TRIP], vb[LW_TRIP];
        bool ok[LW_TRIP];
#pragma unroll
        for (int k = 0; k < LW_TRIP; ++k) {
          const uint32_t p = (uint32_t)(base + 64 * k + lane);
          ok[k] = p < (uint32_t)Pact;
          const uint32_t q = ok[k] ? ((uint32_t)Pact - 1u - p) : 0u;
          // raw v_sqrt_f32 (1 ulp) is enough: the two compares below repair an off-by-one row
          uint32_t r = (uint32_t)((__builtin_amdgcn_sqrtf((float)(8u * q + 1u)) - 1.0f) * 0.5f);
          r += (((r + 1u) * (r + 2u)) >> 1) <= q ? 1u : 0u;
          r -= ((r * (r + 1u)) >> 1) > q ? 1u : 0u;
          const int ia = n_act - 2 - (int)r;
          const int ib = n_act - 1 - (int)(q - ((r * (r + 1u)) >> 1));
          va[k] = ident ? ia : (int)alist[ia];
          vb[k] = ident ? ib : (int)alist[ib];   // va < vb: alist is ascending
        }
        float ax[LW_TRIP], ay[LW_TRIP], az[LW_TRIP], bx[LW_TRIP], by[LW_TRIP], bz[LW_TRIP];
        bool diff[LW_TRIP];
#pragma unroll
        for (int k = 0; k < LW_TRIP; ++k) {
          ax[k] = cx[va[k]]; ay[k] = cy[va[k]]; az[k] = cz[va[k]];
          bx[k] = cx[vb[k]]; by[k] = cy[vb[k]]; bz[k] = cz[vb[k]];
          diff[k] = ok[k] && (!merged || seg[va[k]] != seg[vb[k]]);
        }
        // the trip's candidates are appended together: one LDS counter round trip per trip instead of one per pair of the lane
        // (multi-wavefront classes; same list order as pair-by-pair appends)
        bool inr[LW_TRIP];
        unsigned long long mks[LW_TRIP];
        int tot = 0;
#pragma unroll
        for (int k = 0; k < LW_TRIP; ++k) {
          const float dx = ax[k] - bx[k], dy = ay[k] - by[k], dz = az[k] - bz[k];
          float d2 = (dx * dx + dy * dy) + dz * dz;
          d2 = (d2 == d2) ? d2 : 1.0e4f;  // dist_space stays 100 when a centroid has a zero component (VS:1829)
          inr[k] = diff[k] && (d2 >= cut_lo) && (final_round || d2 < cut_hi);
          mks[k] = __ballot(inr[k]);
          tot += __popcll(mks[k]);
        }
        int base_t;
        if constexpr (NW == 1) {
          base_t = count;
          count += tot;
        } else {
          int b0 = 0;
          if (tot != 0 && lane == 0) b0 = atomicAdd(&sh_i[SH_COUNT], tot);
          base_t = __builtin_amdgcn_readfirstlane(b0);
        }
#pragma unroll
        for (int k = 0; k < LW_TRIP; ++k) {
          if (inr[k]) {
            const int pos = n_list + base_t + __popcll(mks[k] & lt_mask);
            if (pos < LCAP) lk[pos] = (uint64_t)(((uint32_t)va[k] << PSH) | (uint32_t)vb[k]);
          }
          base_t += __popcll(mks[k]);
        }
      }
    }
    return count;
  };
  // full weight of the candidates [n_list, n_list + count); NaN (Q3) and weights <= thr0 (fact S) become dropped
  // entries; returns their number (NW == 1) or adds it to sh_i[SH_DROPPED]
  auto eval_section = [&](int n_list, int count) -> int {
    int dropped = 0;
    // In clutter, proximity + normal angle alone often prove w <= thr0 and save the full evaluation; on smooth surfaces
    // four of five candidates survive and the bound is pure overhead.  It is switched on for the rest of the shell as
    // soon as a batch loses more than half of its candidates.
    bool screen = false;
    for (int base = n_list + 64 * wave; base < n_list + count; base += 64 * NW) {
      const int e = base + lane;
      bool drop = false;
      if (e < n_list + count) {
        const uint32_t pid = (uint32_t)lk[e];
        const NodeRec& A = R(pid >> PSH);
        const NodeRec& B = R(pid & PMASK);
        float w = 0.0f;
        if (!screen || !(vm_weight_bound_da(A, B, W) <= thr0)) w = vm_pair_weight(A, B, W);
        drop = !(w > thr0);
        lk[e] = drop ? 0ull : (((uint64_t)vm_bits(w) << 32) | (uint64_t)(PCOMP - pid));
      }
      const int nd = __popcll(__ballot(drop));
      dropped += nd;
      screen = screen || (2 * nd > 64);
    }
    if constexpr (NW > 1) { if (lane == 0 && dropped) atomicAdd(&sh_i[SH_DROPPED], dropped); }
    return dropped;
  };

  // Shells inside the reach of the near-pair lists: every pair (va < vb) of the neighbourhood with squared centroid
  // distance in [cut_lo, cut_hi) (cut_hi <= the lists' reach) and a weight above the singleton threshold, straight into the
  // edge list with its stored weight -- what enum_section + eval_section produce for that shell, without the n^2/2 distance
  // tests and without a single weight evaluation.  After the first shell only pairs of two still-active vertices
  // (threshold of their segment below the level the scan has reached: fact F) in different segments count.  Vertex va contributes the entries of ITS list that point at a later vertex: the stored
  // weight has the list's owner as first argument, which is the orientation the cut wants.
  // Four vertices per step, sixteen lanes each (lists are sorted by distance: on a surface the shell ends before entry 16);
  // a vertex whose sixteenth entry is still inside the shell gets a step of its own for the rest.
  // LPV lanes per vertex.  Every pair is in ONE list (nearlist.hpp): a vertex on a surface owns four of its eight first-shell
  // pairs, so five lanes per vertex read the first shell (twelve vertices per trip of 64 lanes; nine lanes and seven vertices
  // while every pair was listed twice, sixteen and four at first) and eight the later, wider ones.
  // Later shells (LATER): only the vertices that can still merge -- alist[0, n_act), kept by the freeze step -- own entries
  // that count (a pair needs both ends active, so its owner is among them).
  auto near_enum = [&](auto lpv_c, auto later_c, int n_act, int n_list, float cut_lo, float cut_hi, bool merged, float act_level) -> int {
    constexpr int LPV = decltype(lpv_c)::value, ROWS = 64 / LPV;
    constexpr bool LATER = decltype(later_c)::value;
    const int nv = LATER ? n_act : m;
    int count = 0;
    auto vertex_id = [&](int va) -> uint32_t {   // global id of vertex va from the registers of lane va & 63
      const uint32_t lo = (uint32_t)__shfl((int)tid_reg0, va & 63, 64), hi = (uint32_t)__shfl((int)tid_reg1, va & 63, 64);
      return va < 64 ? lo : hi;
    };
    // entry E = (d2, w(va, b), w(b, va), offset of b from va) of vertex va's list; all lanes call
    // (lat = nlat[va], read by the caller while the entry is still on its way: one LDS round trip less behind the memory one)
    auto take = [&](int va, uint32_t lat, bool act, float4 E) -> bool {
      bool inr = false;
      uint32_t pid = 0;
      float w = 0.0f;
      const bool inside = act && E.x < cut_hi;
      if (inside && E.x >= cut_lo) {
        // nibble-wise (a + 5) + (s + 2): the partner's offset from the voxel, + 7
        const uint32_t q = lat + __float_as_uint(E.w);
        const int bx = (int)(q & 15u) - NL_REACH, by = (int)((q >> 4) & 15u) - NL_REACH, bz = (int)((q >> 8) & 15u) - NL_REACH;
        if ((unsigned)bx < (unsigned)NMAP_DIM && (unsigned)by < (unsigned)NMAP_DIM && (unsigned)bz < (unsigned)NMAP_DIM) {
          const int vb = nmap[(bz * NMAP_DIM + by) * NMAP_DIM + bx];
          if (vb != 0xff) {
            // the row's order decides which end is the weight's first argument (fact S: an edge at or below thr0 is not stored)
            w = va < vb ? E.y : E.z;
            if (w > thr0) {
              inr = true;
              if (merged) { const int sa = seg[va], sb = seg[vb]; const float t1 = thr[sa], t2 = thr[sb]; inr = (sa != sb) & (t1 < act_level) & (t2 < act_level); }   // (both loads before either compare)
              const int lo = va < vb ? va : vb, hi = va ^ vb ^ lo;
              pid = ((uint32_t)lo << PSH) | (uint32_t)hi;
            }
          }
        }
      }
      const unsigned long long mk = __ballot(inr);
      if (inr) {
        const int pos = n_list + count + __popcll(mk & lt_mask);
        if (pos < LCAP) lk[pos] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(PCOMP - pid);
      }
      count += __popcll(mk);
      return inside;
    };
    const int row = lane / LPV, j = lane - row * LPV;   // LPV need not divide 64: the last lanes then idle
    bool none = false;   // a vertex whose voxel has no list: NaN distance in entry 0 (nearlist.hip)
    // LW_NEAR_GROUPS groups of ROWS vertices per trip: their list entries are requested together (the ids come out of
    // registers, so the only memory round trip of a trip is the entries themselves)
    for (int base = 0; base < nv; base += ROWS * LW_NEAR_GROUPS) {
      float4 E[LW_NEAR_GROUPS];
      int vas[LW_NEAR_GROUPS];
      uint32_t lats[LW_NEAR_GROUPS];
      bool act[LW_NEAR_GROUPS], in[LW_NEAR_GROUPS];
#pragma unroll
      for (int g = 0; g < LW_NEAR_GROUPS; ++g) {
        const int iv = base + ROWS * g + row;
        act[g] = iv < nv && row < ROWS;
        vas[g] = LATER ? (act[g] ? (int)alist[iv] : 0) : iv;
        const size_t o = (size_t)vertex_id(vas[g]) * NL_S + (size_t)j;
        E[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act[g]) E[g] = P.near.ent[o];
        lats[g] = (uint32_t)nlat[vas[g]];
        none = none || (E[g].x != E[g].x);
      }
#pragma unroll
      for (int g = 0; g < LW_NEAR_GROUPS; ++g) {
        in[g] = false;
        if (g == 0 || base + ROWS * g < nv) in[g] = take(vas[g], lats[g], act[g], E[g]);   // (uniform) the last trip may hold one group only
      }
      // the shell may go on behind entry LPV - 1 of a vertex: the rest of such lists is taken MV vertices per step, MW lanes each
      constexpr int MW = (NL_S - LPV <= 16) ? 16 : 32, MV = 64 / MW;
      static_assert(NL_S - LPV <= MW, "one step reads the rest of a list");
#pragma unroll
      for (int g = 0; g < LW_NEAR_GROUPS; ++g) {
        unsigned long long more = __ballot(in[g] && j == LPV - 1);
        while (more) {
          int va = 0;
          bool a2 = false;
#pragma unroll
          for (int h = 0; h < MV; ++h) {
            if (more) {
              const int l0 = __ffsll((long long)more) - 1;
              more &= more - 1ull;
              const int vh = __builtin_amdgcn_readlane(vas[g], l0);
              if (lane / MW == h) { va = vh; a2 = (lane & (MW - 1)) < NL_S - LPV; }
            }
          }
          const size_t o2 = (size_t)vertex_id(va) * NL_S + LPV + (size_t)(lane & (MW - 1));   // all lanes: a cross-lane read returns 0 from a lane that is switched off
          float4 E2 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a2) E2 = P.near.ent[o2];
          take(va, (uint32_t)nlat[va], a2, E2);
        }
      }
    }
    if (__ballot(none) != 0ull) return -1;   // the general enumeration takes this neighbourhood
    return count;
  };

  // The same for the multi-wavefront classes (a section: all wavefronts, candidates appended through the LDS counter):
  // four vertices per trip and wavefront, partner vertices found through the hash.
  auto near_enum_h = [&](int n_list, float cut_lo, float cut_hi, bool merged, float act_level) {
    auto take = [&](int va, bool act, float4 E) {
      bool inr = false;
      uint32_t pid = 0;
      float w = 0.0f;
      if (act && E.x < cut_hi && E.x >= cut_lo) {
        const uint32_t la = hlat[va], sl = __float_as_uint(E.w);
        const int bx = (int)(la & 31u) + (int)(sl & 15u) - NL_REACH, by = (int)((la >> 5) & 31u) + (int)((sl >> 4) & 15u) - NL_REACH,
                  bz = (int)((la >> 10) & 31u) + (int)((sl >> 8) & 15u) - NL_REACH;
        if ((unsigned)bx < 32u && (unsigned)by < 32u && (unsigned)bz < 32u) {
          const int vb = h_find((uint32_t)bx | ((uint32_t)by << 5) | ((uint32_t)bz << 10));
          if (vb >= 0) {
            w = va < vb ? E.y : E.z;   // the row's order decides which end is the weight's first argument
            if (w > thr0) {
              inr = true;
              if (merged) { const int sa = seg[va], sb = seg[vb]; const float t1 = thr[sa], t2 = thr[sb]; inr = (sa != sb) & (t1 < act_level) & (t2 < act_level); }   // (both loads before either compare)
              const int lo = va < vb ? va : vb, hi = va ^ vb ^ lo;
              pid = ((uint32_t)lo << PSH) | (uint32_t)hi;
            }
          }
        }
      }
      const unsigned long long mk = __ballot(inr);
      int b = 0;
      if (mk != 0ull && lane == 0) b = atomicAdd(&sh_i[SH_COUNT], __popcll(mk));
      b = __builtin_amdgcn_readfirstlane(b);
      if (inr) {
        const int pos = n_list + b + __popcll(mk & lt_mask);
        if (pos < LCAP) lk[pos] = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(PCOMP - pid);
      }
    };
    const int j = lane & 15;
    static_assert(NL_S <= 16, "one trip reads a whole list");
    // Two dependent global loads per trip (the vertex's voxel id out of the row, then its list) used to be waited for trip by
    // trip -- 19 trips of ~2100 cycles for 305 vertices.  Now the ids run two trips ahead and the entries one (round 4).
    constexpr int STEP = NW * 4;
    const int v0 = wave * 4 + (lane >> 4);
    const float4 none4 = make_float4(__builtin_huge_valf(), 0.f, 0.f, 0.f);
    uint32_t id_next = (v0 + STEP < m) ? (uint32_t)row[v0 + STEP] : 0u;                    // ids of trip 1
    float4 E_next = none4;                                                                 // entries of trip 0
    if (v0 < m && j < NL_S) E_next = P.near.ent[(size_t)(uint32_t)row[v0] * NL_S + (size_t)j];
    for (int base = wave * 4; base < m; base += STEP) {
      const int va = base + (lane >> 4);
      const bool act = va < m;
      const float4 E = E_next;
      E_next = none4;
      if (va + STEP < m && j < NL_S) E_next = P.near.ent[(size_t)id_next * NL_S + (size_t)j];
      id_next = (va + 2 * STEP < m) ? (uint32_t)row[va + 2 * STEP] : 0u;
      take(va, act, E);
    }
  };

  // ---- wavefronts 1 .. NW-1: serve the sections until wavefront 0 is done ----
  if constexpr (NW > 1) {
    if (wave != 0) {
      while (true) {
        __syncthreads();
        const int cmd = sh_i[SH_CMD];
        if (cmd == CMD_QUIT) return;
        if (cmd == CMD_ENUM)
          enum_section(sh_i[SH_NLIST], sh_i[SH_NACT], sh_i[SH_NMIN], sh_i[SH_BIG], sh_i[SH_MINOR] != 0, sh_i[SH_FINAL] != 0, sh_i[SH_MERGED] != 0,
                       sh_i[SH_PACT], sh_f[0], sh_f[1]);
        else if (cmd == CMD_EVAL)
          eval_section(sh_i[SH_NLIST], sh_i[SH_CNT]);
        else if (cmd == CMD_NEAR)
          near_enum_h(sh_i[SH_NLIST], sh_f[0], sh_f[1], sh_i[SH_MERGED] != 0, sh_f[2]);
        else if (cmd == CMD_STAGE)
          stage_centroids();
        else
          sort_section(sh_i[SH_CNT]);
        __syncthreads();
      }
    }
  }
  // ---- wavefront 0: calls into the sections (alone when NW == 1) ----
  auto run_enum = [&](int n_list, int n_act, int n_min, int big, bool use_minor, bool final_round, bool merged, int Pact, float cut_lo,
                      float cut_hi) -> int {
    if constexpr (NW == 1) {
      return enum_section(n_list, n_act, n_min, big, use_minor, final_round, merged, Pact, cut_lo, cut_hi);
    } else {
      if (lane == 0) {
        sh_i[SH_CMD] = CMD_ENUM; sh_i[SH_NLIST] = n_list; sh_i[SH_NACT] = n_act; sh_i[SH_NMIN] = n_min; sh_i[SH_BIG] = big;
        sh_i[SH_MINOR] = use_minor; sh_i[SH_FINAL] = final_round; sh_i[SH_MERGED] = merged; sh_i[SH_PACT] = Pact; sh_i[SH_COUNT] = 0;
        sh_f[0] = cut_lo; sh_f[1] = cut_hi;
      }
      __syncthreads();
      enum_section(n_list, n_act, n_min, big, use_minor, final_round, merged, Pact, cut_lo, cut_hi);
      __syncthreads();
      return sh_i[SH_COUNT];
    }
  };
  auto run_near = [&](int n_list, float cut_lo, float cut_hi, bool merged, float act_level) -> int {   // NW > 1
    if (lane == 0) {
      sh_i[SH_CMD] = CMD_NEAR; sh_i[SH_NLIST] = n_list; sh_i[SH_MERGED] = merged; sh_i[SH_COUNT] = 0;
      sh_f[0] = cut_lo; sh_f[1] = cut_hi; sh_f[2] = act_level;
    }
    __syncthreads();
    near_enum_h(n_list, cut_lo, cut_hi, merged, act_level);
    __syncthreads();
    return sh_i[SH_COUNT];
  };
  auto run_stage = [&]() {   // NW > 1: the hash gives way to the centroids
    if (lane == 0) sh_i[SH_CMD] = CMD_STAGE;
    __syncthreads();
    stage_centroids();
    __syncthreads();
  };
  auto run_eval = [&](int n_list, int count) -> int {
    if constexpr (NW == 1) {
      return eval_section(n_list, count);
    } else {
      if (lane == 0) { sh_i[SH_CMD] = CMD_EVAL; sh_i[SH_NLIST] = n_list; sh_i[SH_CNT] = count; sh_i[SH_DROPPED] = 0; }
      __syncthreads();
      eval_section(n_list, count);
      __syncthreads();
      return sh_i[SH_DROPPED];
    }
  };
  auto sort_list = [&](int cnt) {
    if constexpr (NW == 1) {
      sort_section(cnt);
    } else {
      if (lane == 0) { sh_i[SH_CMD] = CMD_SORT; sh_i[SH_CNT] = cnt; }
      __syncthreads();
      sort_section(cnt);
      __syncthreads();
    }
  };
  auto master = [&]() {

  // Merge of the sorted edge list [0, cnt) down to (not including) weights <= level, in the reference's sequential
  // order; returns the position of the first unprocessed edge; afterwards seg[] maps every vertex to its live
  // representative.  64 edges per step, one per lane.  An edge is DECIDED when no earlier undecided edge of the step
  // touches either of its segments (found with one LDS min-claim per segment): the state it sees is then the state
  // the sequential scan would show it, so it merges or is rejected for good, and all decided edges of an iteration
  // act at once on disjoint segments.  Segment state (representative, threshold, size) lives in LDS only.
  auto cut_over = [&](int nsz) -> float {  // cut / nsz, as vm_cut_threshold divides (called by all lanes)
    if constexpr (MAXM <= 128) {
      const int a = ((nsz - 1) & 63) << 2;
      const float lo = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(cut_tab0)));
#if LW_CUT_OVER_ONE
      if (__ballot(nsz > 64) == 0ull) return lo;   // (wave-uniform) sizes above 64 come late: most iterations need one lookup
#endif
      const float hi = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(cut_tab1)));
      return nsz > 64 ? hi : lo;
    } else {
      return cut / (float)(nsz > 0 ? nsz : 1);
    }
  };
  auto merge_list = [&](int cnt, float level) -> int {
    int pos = 0;
    bool reached = false;
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
      const int nproc = __popcll(__ballot(alive));  // sorted: the processable edges are a prefix of the step
      while (true) {
        if (alive) {
          int r;
          // both chains at once: one LDS round trip per step of the longer one (they are 1.08 steps long on average; the kernel
          // waits for LDS round trips here, not for LDS issue slots).  (Loading sizes and thresholds along with every step, so
          // that the cross-lane lookup of cut / size can travel with the claims, saved another trip and no time.)
          int rb = rep[sb];
          r = rep[sa];
          while (r != sa || rb != sb) { sa = r; sb = rb; r = rep[sa]; rb = rep[sb]; }
          alive = sa != sb;  // inside one segment: skipped now and for ever
        }
#ifdef VGS_PROF
        LW_CNT(7, 1);
#endif
        if (__ballot(alive) == 0ull) break;
        if (alive) { atomicMin(&claim[sa], (uint32_t)lane); atomicMin(&claim[sb], (uint32_t)lane); }
        wave_sync();
        bool decided = false;
        float ta = 0.f, tb = 0.f;
        int nsz = 1;
        if (alive) {
          const uint32_t ca = claim[sa], cb = claim[sb];   // both loads before either compare
          decided = (ca == (uint32_t)lane) & (cb == (uint32_t)lane);
          ta = thr[sa]; tb = thr[sb];
          nsz = (int)ssz[sa] + (int)ssz[sb];
        }
        wave_sync();
        if (alive) { claim[sa] = 0xffffffffu; claim[sb] = 0xffffffffu; }
        const float co = cut_over(nsz);
        const bool pass = decided && (w > ta) && (w > tb);
        if (pass) {
          const int keep = (ta >= tb) ? sa : sb;   // VS:1972-1983: the segment with the larger threshold survives
          const int gone = sa ^ sb ^ keep;          // (two selects on one condition issue four times slower than a select and two xors: tools/vcc_rate.hip)
          rep[gone] = (idx_t)keep;
          thr[keep] = w - co;                      // = vm_cut_threshold(w, cut, nsz): seg_int = w (VS:1988)
          ssz[keep] = (idx_t)nsz;
          ssz[gone] = 0;
        }
        merges += __popcll(__ballot(pass));
        alive = alive && !decided;
        wave_sync();
      }
      LW_CNT(15, 1);
      if (nproc < 64) { pos += nproc; reached = true; } else pos += 64;
      if (merges >= m - 1) break;  // one segment left
    }
    for (int c = lane; c < m; c += 64) {
      int s = seg[c];
      while (rep[s] != s) s = rep[s];
      seg[c] = (idx_t)s;
    }
    wave_sync();
    return pos;
  };
  // ---- can the voxel itself ever merge?  While it is a singleton it needs an incident edge heavier than thr0.  On
  // surfaces vertex 0 joins a segment in the first shell, so the test is only made when it has not (after shell one):
  // the cheap bound first, the full weight for the pairs it lets through -- at most m - 1 evaluations that save a voxel
  // in clutter from working through every shell (and from being handed over) only to end up alone.
  auto never_merges = [&]() -> bool {
    bool any = false;
    for (int base = 1; base < m; base += 64) {
      const int x = base + lane;
      bool heavy = false;
      if (x < m && !(vm_weight_bound_da(R(0), R(x), W) <= thr0)) heavy = vm_pair_weight(R(0), R(x), W) > thr0;
      any = any || heavy;
    }
    return __ballot(any) == 0ull;
  };
  // In-place compaction of the entries [from, from + count) that are not dropped (0); returns the new end of the list.
  auto close_gaps = [&](int from, int count) -> int {
    int kept = from;
    for (int base = from; base < from + count; base += 64) {
      const int e = base + lane;
      uint64_t kk = 0;
      if (e < from + count) kk = lk[e];
      const bool live = kk != 0ull;   // a dropped entry is 0
      const unsigned long long mk = __ballot(live);
      wave_sync();
      if (live) { const int d = kept + __popcll(mk & lt_mask); lk[d] = kk; }
      kept += __popcll(mk);
      wave_sync();
    }
    return kept;
  };
  if (m >= 2) {
    // =========================== phase A: edges heavier than a singleton's threshold ===========================
    int n_list = 0;      // edges carried in the list (sorted, all lighter than the previous level, heavier than thr0)
    int n_act = m;       // vertices still able to merge; pairs are enumerated among them only
    int n_min = 0, big = -1;  // later rounds: candidate pairs all involve a vertex outside the largest active segment
    float cut_lo = 0.0f;
    // first shell: about shell0/2 pairs per vertex on a surface (pairs within d of each other ~ m^2 d^2 / (2 R^2)),
    // but never more than cap_frac of the list
    float cut_hi = P.shell0 * P.r2_graph / (float)m;
    {
      const float cap = 2.0f * (P.cap_frac * (float)LCAP) * P.r2_graph / ((float)m * (float)m);
      cut_hi = cut_hi < cap ? cut_hi : cap;
      if (near_ok) cut_hi = cut_hi < P.near.d2max ? cut_hi : P.near.d2max;   // the lists reach this far
    }
    int shrink = 0;
    bool phase_a_complete = false;
    int rounds = 0;
    float act_level = __builtin_huge_valf();   // vertices whose segment's threshold is below this are still active
    while (true) {
      // what the shell loop derives from the lane id alone (LDS addresses of its arrays, the row pointer) is loop invariant;
      // hoisted, it outlives the register budget and was spilled -- six dwords per lane written to scratch by every wavefront,
      // 0.47 GB of HBM writes per launch -- although each is one instruction away from the lane id.  Opaque here, it is
      // recomputed where it is used.
      if constexpr (NW == 1) asm volatile("" : "+v"(lane));
      // a neighbourhood that keeps hundreds of edges waiting above thr0 makes slow progress here: after a few
      // passes hand it to the workgroup-per-voxel kernel, which holds 8192 edges and evaluates every pair once
      // (examining all remaining pairs in this kernel and keeping the survivors only was tried: the neighbourhoods that
      // get here hold more edges above thr0 than the list, or overflow it in phase B, and are handed over anyway)
      if (++rounds > (MAXM > 128 ? 14 : P.max_rounds)) { if (lane == 0) diag(3); bail = true; break; }
      // ---- 1. enumerate the pairs of this shell ----
      const int free_slots = LCAP - n_list;
      const bool use_minor = (big >= 0) && (2 * n_min < n_act);
      const int Pact = use_minor ? n_min * n_act : n_act * (n_act - 1) / 2;   // (upper bound of) candidate pairs
      // once every pair between the still-active vertices fits in the list there is no point in further shells
      const bool final_round = !(cut_hi < P.d2_all) || (merges > 0 && Pact <= free_slots);
      bool near_round = (NEAR || NEARH) && near_ok && !final_round && !(cut_hi > P.near.d2max);   // a shell inside the lists' reach
      int count;
      if constexpr (NEARH && NW > 1) {
        if (near_round) count = run_near(n_list, cut_lo, cut_hi, merges > 0, act_level);
        else {
          if (!cen_ready) { run_stage(); cen_ready = true; near_ok = false; }
          count = run_enum(n_list, n_act, n_min, big, use_minor, final_round, merges > 0, Pact, cut_lo, cut_hi);
        }
      } else if constexpr (NEAR) {
        count = 0;
        if (near_round) {
          count = rounds == 1 ? near_enum(std::integral_constant<int, LW_NEAR_LPV1>{}, std::false_type{}, m, n_list, cut_lo, cut_hi, merges > 0, act_level)
                              : near_enum(std::integral_constant<int, LW_NEAR_LPV2>{}, std::true_type{}, n_act, n_list, cut_lo, cut_hi, merges > 0, act_level);
          if (count < 0) { near_ok = false; near_round = false; }   // some vertex has no list: no shell of this voxel comes from the lists
        }
        if (!near_round) {
          if (!cen_ready) { wave_sync(); stage_centroids(); cen_ready = true; wave_sync(); }   // the offset map is not needed any more
          count = run_enum(n_list, n_act, n_min, big, use_minor, final_round, merges > 0, Pact, cut_lo, cut_hi);
        }
      } else {
        count = run_enum(n_list, n_act, n_min, big, use_minor, final_round, merges > 0, Pact, cut_lo, cut_hi);
      }
#ifdef VGS_PROF
      if (near_round) { LW_ACC(1); } else { LW_ACC(14); LW_CNT(10, 1); }   // enumerate: shells read from the near-pair lists / general enumeration (slot 10 counts the latter)
#endif
      if (P.dbg_stop == 1) return;
      LW_CNT(8, 1);  // rounds
      if (count > free_slots) {
        // the shell holds more pairs than the list: shrink it (assume uniform density in d2) and redo
        if (++shrink > 24 || free_slots < 32) { if (lane == 0) diag(free_slots < 32 ? 4 : 3); bail = true; break; }
        const float hi = final_round ? P.d2_all : cut_hi;
        cut_hi = cut_lo + (hi - cut_lo) * (0.75f * (float)free_slots / (float)count);
        if (!(cut_hi > cut_lo)) { if (lane == 0) diag(5); bail = true; break; }
        continue;
      }
      shrink = 0;
      wave_sync();
      // ---- 2. full weight of the shell's pairs; NaN (Q3) and weights <= thr0 (fact S) are not stored ----
      const int dropped = near_round ? 0 : run_eval(n_list, count);   // the lists hold weights above thr0 only
      n_evals += (unsigned int)count;
      auto sort_len = [](int c) { int np = 64; while (np < c) np <<= 1; return np; };
      LW_ACC(2);  // evaluate
      int behind = 0;   // dropped entries the sort leaves behind the real ones
      if (dropped > 0 && sort_len(n_list + count - dropped) < sort_len(n_list + count)) {
        // close the gaps before sorting when that halves the sort network (its length is the next power of two):
        // ascending and in place, a write never passes the read position
        wave_sync();
        const int kept = close_gaps(n_list, count);
        n_list = kept;
      } else {
        n_list += count;
        behind = dropped;
      }
      if constexpr (NW == 1 && LW_REG_SORT != 0) {
        // the one call of the register network (regsort.hpp; its sizes are 2000 instructions: phase B keeps the LDS network)
        wave_sync();
        if constexpr (SORT32 && LCAP <= 512) {
          // phase A's weights lie in (thr0, 1]: one-word keys (weight above thr0, slot), full-rate comparators (regsort.hpp, round 6).  The
          // host launches this instantiation when thr0 >= 0.5 (cut <= 0.5: 2^23 float values between thr0 and 1); a key outside that window
          // cannot occur then, and should one (a weight above 1) the voxel is handed over rather than sorted wrongly.  (Keeping the 64-bit
          // network in the same kernel as the fall-back cost it six spilled registers.)
          const uint32_t wbase = vm_bits(thr0) + 1u;
          bool sorted;
          if (n_list <= 64) sorted = regsort::sort_desc32<1>(lk, n_list, lane, wbase);
          else if (n_list <= 128) sorted = regsort::sort_desc32<2>(lk, n_list, lane, wbase);
          else if (n_list <= 256) sorted = regsort::sort_desc32<4>(lk, n_list, lane, wbase);
          else sorted = regsort::sort_desc32<8>(lk, n_list, lane, wbase);
          if (!sorted) { if (lane == 0) diag(3); bail = true; break; }
        } else {
          if (n_list <= 64) regsort::sort_desc<1>(lk, n_list, lane);
          else if (n_list <= 128) regsort::sort_desc<2>(lk, n_list, lane);
          else if (n_list <= 256) regsort::sort_desc<4>(lk, n_list, lane);
          else if constexpr (LW_REG_SORT >= 8) regsort::sort_desc<8>(lk, n_list, lane);
          else regsort::sort_desc_two_halves<4>(lk, n_list, lane);
        }
        wave_sync();
      } else {
        sort_list(n_list);
      }
      n_list -= behind;
      if (P.dbg_stop == 2) return;
      // ---- 3. (sorted above) 4. merge down to the level ----
      LW_CNT(9, n_list);
      LW_ACC(3);  // sort
      if (P.dbg_stop == 3) return;
      const float level = final_round ? -1.0f : vm_weight_bound_d(cut_hi, W);
      const int pos = merge_list(n_list, level);
      LW_ACC(4);  // merge
      if (P.dbg_stop == 4) return;
      if (merges >= m - 1) break;
      if (final_round || !(level > thr0)) { phase_a_complete = true; break; }
      if (rounds == 1 && ssz[seg[0]] == 1 && never_merges()) break;  // the voxel stays alone: its connect list is itself
      // only the segment of vertex 0 (the voxel itself) is reported: once it is frozen its membership is final -- the usual
      // end on a surface, so it is tested before the bookkeeping of the next shell, not after
      if (!(thr[seg[0]] < level)) break;
      // ---- 5. freeze (fact F) and carry ----
      int n_new = 0;
      for (int base = 0; base < n_act; base += 64) {
        const int ia = base + lane;
        bool act = false;
        int v = 0;
        if (ia < n_act) { v = alist[ia]; act = thr[seg[v]] < level; }
        const unsigned long long mk = __ballot(act);
        wave_sync();
        if (act) alist[n_new + __popcll(mk & lt_mask)] = (idx_t)v;
        n_new += __popcll(mk);
        wave_sync();
      }
      n_act = n_new;
      {
        // largest active segment and the active vertices outside it
        int best = -1;
        for (int base = 0; base < m; base += 64) {
          const int c = base + lane;
          if (c < m && ((ssz[c] != 0) & (thr[c] < level))) { const int key = ((int)ssz[c] << 16) | c; best = key > best ? key : best; }
        }
        for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(best, o, 64); best = other > best ? other : best; }
        big = best >= 0 ? (best & 0xffff) : -1;
        n_min = 0;
        for (int base = 0; base < n_act; base += 64) {
          const int ia = base + lane;
          bool out = false;
          int v = 0;
          if (ia < n_act) { v = alist[ia]; out = (int)seg[v] != big; }
          const unsigned long long mk = __ballot(out);
          if (out) minor[n_min + __popcll(mk & lt_mask)] = (idx_t)v;
          n_min += __popcll(mk);
        }
        wave_sync();
      }
      int active_segs = 0;
      for (int base = 0; base < m; base += 64) {
        const int c = base + lane;
        active_segs += __popcll(__ballot((c < m) && ((ssz[c] != 0) & (thr[c] < level))));
      }
      if (active_segs < 2) break;  // nothing can merge at any weight <= level
      int kept = 0;
      for (int base = pos; base < n_list; base += 64) {
        const int e = base + lane;
        bool keep_e = false;
        uint64_t kk = 0;
        if (e < n_list) {
          kk = lk[e];
          const uint32_t pid = PCOMP - (uint32_t)kk;
          const int sa = seg[pid >> PSH], sb = seg[pid & PMASK];
          const float t1 = thr[sa], t2 = thr[sb];   // both loads before either compare
          keep_e = (sa != sb) & (t1 < level) & (t2 < level);
        }
        const unsigned long long mk = __ballot(keep_e);
        wave_sync();  // all lanes have read their entry before anyone overwrites the front of the list
        if (keep_e) { const int d = kept + __popcll(mk & lt_mask); lk[d] = kk; }
        kept += __popcll(mk);
        wave_sync();
      }
      n_list = kept;
      LW_ACC(5);  // freeze + carry
      act_level = level;
      cut_lo = cut_hi;
      cut_hi = cut_hi * P.grow;
      if (near_ok && cut_lo < P.near.d2max && cut_hi > P.near.d2max) cut_hi = P.near.d2max;   // one more shell from the lists
    }
    // =========================== phase B: edges at or below a singleton's threshold ===========================
    if (phase_a_complete && !bail && merges < m - 1) {
      // every edge heavier than thr0 has been examined in order; singletons are frozen.  What can still merge:
      // non-singleton segments whose threshold is below thr0, through edges with w <= thr0 between them.
      int nb = 0;
      for (int base = 0; base < m; base += 64) {
        const int v = base + lane;
        bool act = false;
        if (v < m) { const int s = seg[v]; act = (ssz[s] >= 2) && (thr[s] < thr0); }
        const unsigned long long mk = __ballot(act);
        if (act) alist[nb + __popcll(mk & lt_mask)] = (idx_t)v;
        nb += __popcll(mk);
      }
      int active_segs = 0;
      for (int base = 0; base < m; base += 64) {
        const int c = base + lane;
        active_segs += __popcll(__ballot((c < m) && (ssz[c] >= 2) && (thr[c] < thr0)));
      }
      wave_sync();
#ifdef VGS_PROF
      _t0 = clock64();
#endif
      {
        const int s0 = seg[0];
        if (!((ssz[s0] >= 2) && (thr[s0] < thr0))) active_segs = 0;  // vertex 0's segment is frozen: nothing to report changes
      }
      if (active_segs >= 2) {
        const int Pb = nb * (nb - 1) / 2;
        int count = 0;
        int ia = 0, qq = lane;
        for (int base = 0; base < Pb; base += 64) {
          while (ia < nb - 1 && qq >= nb - 1 - ia) { qq -= (nb - 1 - ia); ++ia; }
          bool inr = false, evaluated = false;
          uint64_t kk = 0;
          if (ia < nb - 1) {
            const int a = alist[ia], b = alist[ia + 1 + qq];
            if (seg[a] != seg[b]) {
              const float w = vm_pair_weight(R(a), R(b), W);
              evaluated = true;
              inr = (w <= thr0);  // heavier edges were examined in phase A; NaN compares false
              kk = ((uint64_t)vm_bits(w) << 32) | (uint64_t)(PCOMP - (((uint32_t)a << PSH) | (uint32_t)b));
            }
          }
          n_evals += (unsigned int)__popcll(__ballot(evaluated));
          const unsigned long long mk = __ballot(inr);
          if (inr) {
            const int pos = count + __popcll(mk & lt_mask);
            if (pos < LCAP) lk[pos] = kk;
          }
          count += __popcll(mk);
          qq += 64;
        }
        if (count > LCAP) {
          if (lane == 0) diag(6);
          bail = true;
        } else {
          wave_sync();
          sort_list(count);
          merge_list(count, -1.0f);
        }
        LW_ACC(6);  // phase B
      }
    }
  }
  if (bail) {
    if (lane == 0) {
      const int bin = (SMALL && NW == 1 && (P.ho_bins & 0xff) > 1) ? lw_ho_bin(m) : 0;
      // One-wavefront classes: only the mark; k_ho_lists builds the hand-over lists from the marks afterwards.  (Appending here is one
      // RETURNING atomic on one of four addresses per handed-over voxel: on a noisy scene 130 k of them serialise at ~30 ns each and
      // the launch cannot end before they have -- 4 ms, whatever the wavefronts do.)
      if (SMALL && NW == 1 && (P.ho_bins & 0xff) > 1) P.pending[u] = (uint8_t)(1 + bin);
      else { fallback[(size_t)bin * P.ho_stride + atomicAdd(n_fallback + bin, 1u)] = u; P.pending[u] = LW_PENDING_LISTED; }
      if constexpr (SAMPLED) atomicAdd(&counters[LW_VOTE_BASE + (blockIdx.x & (LW_VOTE_WORDS - 1u))], 1ull << 32);   // a sample that gave up
    }
    return;
  }
  // ---- result: the segment of vertex 0 (the voxel itself) ----
  {
    const int s0 = seg[0];
    for (int c = lane; c < m; c += 64) crow[c] = ((int)seg[c] == s0) ? 1 : 0;   // the whole row: nobody zeroes the table first
    if (P.cbits) {
      // ... and as bits by ball offset (the edge list's LDS is free now): all words of the row are written, zeros when the row
      // carries no lattice offsets (the centre bit -- the voxel itself, always a member -- then says "no bits": the reader searches instead)
      uint32_t* const cb = (uint32_t*)lk;
      static_assert(sizeof(uint64_t) * LCAP >= 4 * VGS_CB_MAX_WORDS, "the bit row fits the edge list");
      wave_sync();
      for (int k = lane; k < P.cb_words; k += 64) cb[k] = 0u;
      wave_sync();
      // (Building the row from the LDS copy of the lattice offsets where the near-pair lists served every shell -- no second read of
      // the row's offsets -- was measured in round 4: it costs the bulk kernel four spilled registers, 2.67 -> 2.70 ms.)
      if (orow != nullptr && orow[0] != 0xffffu)
        for (int c = lane; c < m; c += 64)
          if ((int)seg[c] == s0) { const uint32_t idx = vgs_cb_index(orow[c], P.cb_R); atomicOr(&cb[idx >> 5], 1u << (idx & 31u)); }
      wave_sync();
      uint32_t* const out = P.cbits + (size_t)u * (size_t)P.cb_words;
      for (int k = lane; k < P.cb_words; k += 64) out[k] = cb[k];
    }
  }
  if (lane == 0) evals_out[u] = n_evals;  // summed on the host on request: no same-address atomics on the hot path
  if constexpr (SAMPLED) { if (lane == 0 && m >= 2) atomicAdd(&counters[LW_VOTE_BASE + (blockIdx.x & (LW_VOTE_WORDS - 1u))], 1ull); }   // a sample that finished
#ifdef VGS_PROF
  if (lane == 0 && dbg_out) { long long tnow = clock64(); dbg_out[4 * (size_t)u + 0] = (uint32_t)m; dbg_out[4 * (size_t)u + 1] = (uint32_t)prof[8]; dbg_out[4 * (size_t)u + 2] = (uint32_t)((tnow - t_start) >> 4); dbg_out[4 * (size_t)u + 3] = (uint32_t)n_evals; }
  if (lane == 0) { prof[11] = 1; prof[12] = (unsigned long long)merges; prof[13] = (unsigned long long)m; for (int k = 0; k < 16; ++k) if (prof[k]) atomicAdd(&counters[16 + k], prof[k]); }
#endif
  };  // master
  master();
  if constexpr (NW > 1) {
    if (lane == 0) sh_i[SH_CMD] = CMD_QUIT;
    __syncthreads();
  }
}

#endif
