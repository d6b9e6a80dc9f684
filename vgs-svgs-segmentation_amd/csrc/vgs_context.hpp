// vgs_context.hpp -- internal state of one engine context (device buffers, stream, stage state).
// Data model in HBM (SoA, grow-only buffers reused across runs; sized for 288 GB):
//   points      xyz_in (caller stride) -> sort keys code[N] (u64), perm[N] (u32)        voxelize
//   voxels      vox_code[V] (u64 Morton, x-major), vox_start[V+1], node[V] (64 B record) features
//   hash        hkey[H] (u64), hval[H] (u32), H = pow2 >= 2V                             adjacency
//   adjacency   adj_key[U * stride] (u64: float bits of d2 << 32 | voxel id), adj_cnt[U] adjacency
//   connect     conn[U * stride] (u8 bit0 = in L0(i), bit1 = mutual)                     local cut / merge
//   components  parent[V], vox_label[V], point label[N]                                  merge / labels
#ifndef VGS_CONTEXT_HPP_
#define VGS_CONTEXT_HPP_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/vgs.h"
#include "vgs_math.h"
#include "pairlist.hpp"

typedef VgsNode NodeRec;
static_assert(sizeof(NodeRec) == 64, "NodeRec must be 64 bytes");

// One-workgroup-per-item kernels: workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only), so
// item = (b % 8) * ceil(n/8) + b / 8 hands every XCD one contiguous eighth of the Morton-ordered item list and
// spatial neighbours share an L2.  Launch vgs_xcd_grid(n) workgroups and drop items >= n.
#ifdef __HIPCC__
__device__ __forceinline__ int64_t vgs_xcd_item(unsigned int b, int64_t n) {
  const int64_t per = (n + 7) >> 3;
  return (int64_t)(b & 7u) * per + (int64_t)(b >> 3);
}
#endif
static inline unsigned int vgs_xcd_grid(int64_t n) { return (unsigned int)(((n + 7) >> 3) << 3); }

// grow-only device buffer
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// PCL OctreePointCloud bounding-box growth: product-side restatement used by the host
// part of the voxelize stage (the oracle has its own copy; the two never share code).
struct OctreeBox {
  double min[3] = {0, 0, 0}, max[3] = {0, 0, 0};
  double res = 0;
  int depth = 0;
  bool defined = false;
  uint64_t shift[3] = {0, 0, 0};
  bool contains(const float* p) const;
  void adopt(const float* p);  // grows until p fits
};

struct Epoch {  // keys of points with index >= first use this box
  int64_t first;
  double min[3];
  uint64_t shift[3];
};

#define VGS_MAX_EPOCHS 96

enum Stage { ST_NONE = 0, ST_POINTS = 1, ST_VOXELS = 2, ST_FEATURES = 3, ST_ADJACENCY = 4, ST_SEGMENTED = 5 };

// Diagnostics / schedule-tuning knobs (none changes a result).  Read from the environment ONCE, when the context is created
// (vgs_read_env_knobs): a stage never calls getenv, so a host that changes its environment between two calls cannot make the
// two halves of a stage disagree.
struct VgsKnobs {
  int a1_max = 4;            // VGS_A1MAX: own near-list length up to which a class-A voxel goes first
  float shell0 = 8.0f;       // VGS_SHELL0
  float cap_frac = 0.7f;     // VGS_CAPFRAC
  int dbg_stop = 0;          // VGS_DBG_STOP
  int max_rounds = 6;        // VGS_ROUNDS
  int dbg_max_m = 0;         // VGS_DBG_MAXM
  int dbg_xl_from = 0;       // VGS_DBG_XL_FROM: tests -- the 2048-vertex general kernel passes neighbourhoods above N on to the extra-large one
  int near_min_own = 7;      // VGS_NEARMINOWN
  int fv_blocks = 512;       // VGS_FV_BLOCKS
  int only_class = -1;       // VGS_ONLY_CLASS (-DVGS_PROF builds)
  bool no_dense = false;     // VGS_NO_DENSE
  bool no_overlap = false;   // VGS_NO_OVERLAP
  bool no_near = false;      // VGS_NO_NEAR
  bool no_adjmasks = false;  // VGS_NO_ADJMASKS
  bool no_packed_sort = false;   // VGS_NO_PACKED_SORT: (code, index) pairs through the voxelize sort instead of one packed key
  bool no_early_union = false;   // VGS_NO_EARLY_UNION: the union-find runs behind closestCheck as in rounds 1-3
  int vote_period = 64;          // VGS_VOTE_PERIOD (power of two): one voxel in so many of the one-wavefront classes runs as a sample
  bool no_tile_early = false;    // VGS_NO_TILE_EARLY: a tile's first hooks and unions wait behind closestCheck (rounds 3-4)
  int cross_lds_kb = 0;          // VGS_CROSS_LDS: KB of (unused) LDS per wavefront of crossValidation's FIRST pass and its unions -- caps how many
                                 // of them a CU holds beside the hand-over kernel's workgroups, which need four wave slots at once
  bool no_vccs_tiles = false;    // VGS_NO_VCCS_TILES: the supervoxel expansion rounds gather their 26 labels through the neighbour table
  bool no_c0 = false;            // VGS_NO_C0: no separate class for neighbourhoods of 129..320 voxels
  bool vccs_pingpong = false;    // VGS_VCCS_PINGPONG: (vccs_mode 1) the live-flag sweeps alternate between two arrays (A/B twin of round 6's in-place sweeps)
  bool no_adj_wide = false;      // VGS_NO_ADJ_WIDE: rows above 2048 used neighbours keep the one-wavefront general kernel (A/B twin of round 6's workgroup per row)
  bool no_grow_prefix = false;   // VGS_NO_GROW_PREFIX: the octree box grows by one scan launch and one adopt launch per step from the first point on
  bool no_sort32 = false;        // VGS_NO_SORT32: the one-wavefront classes of the local cut keep the 64-bit sort network (A/B twin of round 6's one-word keys)
  bool no_pg_xl = false;         // VGS_NO_PG_XL: no extra-large pair-list instantiation (neighbourhoods above 1024 voxels take the hand-over path)
  bool no_connbits = false;  // VGS_NO_CONNBITS: crossValidation searches the neighbour's row (the path of rounds 1-3)
  bool no_pairlists = false; // VGS_NO_PAIRLISTS: no pair lists (pairlist.hpp); hand-overs and wide classes take the kernels of round 4
  bool no_vote = false;      // VGS_NO_VOTE: every one-wavefront voxel tries the lazy schedule (LwParams::vote off)
  int vote_force = 0;        // VGS_VOTE_FORCE (diagnostics): every one-wavefront voxel that is not a sample is handed over
  bool vccs_nbr_normals = false; // VGS_VCCS_NBR_NORMALS: vccs_mode 1's two-ring normals from the [26][V] neighbour table, as until round 6
  int ho_grid = -1;          // VGS_HO_GRID: workgroups of the hand-over kernels of the one-wavefront classes (-1: as many as the device holds at once; 0: one per row up to 16384)
  int pg_min_frac = 8;       // VGS_PG_MINFRAC: hand-overs go through the pair lists when they are more than 1/N of the used voxels (0: never)
  int pg_wide = 1;           // VGS_PG_WIDE: neighbourhoods above 128 voxels are cut from the pair lists (0: the multi-wavefront shell classes)
  int pg_wide_frac = 8;      // VGS_PG_WIDEFRAC: ... when they are more than 1/N of the used voxels
  bool debug = false;        // VGS_DEBUG
};

struct vgs_ctx {
  vgs_params P;
  VgsKnobs K;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // side streams: the heavy local-cut classes overlap the light one
  hipStream_t stream3 = nullptr;
  hipStream_t stream4 = nullptr;
  std::string err;
  int stage = ST_NONE;
  void* pin = nullptr;   // 4 KB of pinned host memory: small read-backs land here (a pageable destination makes the copy blocking and slower)

  // input
  const float* xyz = nullptr;  // device pointer (one of xyz_buf or the caller's)
  // two input buffers: while the stages run on one cloud, vgs_stage_points copies the next one into the other on the
  // upload stream (a sequence of clouds; the points are read by the voxelize stage only)
  DevBuf<float> xyz_buf[2];
  int xyz_cur = 0;
  hipStream_t s_h2d = nullptr, s_d2h = nullptr;   // copy streams: uploads of the next cloud, downloads of the last labels
  hipEvent_t ev_h2d = nullptr, ev_d2h = nullptr;
  hipEvent_t ev_rb = nullptr;    // split read-backs (vgs_readback_begin / _end)
  bool gathered = false;         // voxelize: the leaf-order gather was queued behind the count read-back
  bool bricks_ready = false;     // the features stage has already queued the brick table of this voxel table (behind its read-back)
  bool staged = false;         // a cloud is on its way into xyz_buf[1 - xyz_cur]
  int64_t staged_n = 0;
  int staged_stride = 12;
  bool pt_labels_pending = false;    // a tile's merge stage did not scatter the per-point labels (vgs_ensure_point_labels does, on request)
  bool labels_event_valid = false;   // ev[15] has been recorded behind the last kernel that wrote pt_label
  bool d2h_open = false;       // vgs_get_point_labels_async has a copy in flight ...
  const int32_t* d2h_src = nullptr;   // ... out of this buffer
  int64_t N = 0;
  int stride_f = 3;

  // octree
  OctreeBox box;
  bool grid_pinned = false;
  bool grid_covers = false;   // ... and the caller vouches that it covers every finite point of the current cloud (vgs_set_grid_covering): no scan
  int n_epochs = 0;   // growth epochs of the last voxelize (the epochs themselves stay in grow_state on the device)

  // voxelize
  DevBuf<uint64_t> code_a, code_b;
  DevBuf<uint32_t> perm_a, perm_b;
  DevBuf<uint8_t> sort_tmp;
  DevBuf<uint64_t> grow_state;         // GrowState of the box growth (voxelize.hip)
  DevBuf<uint32_t> head_flag, pt_vox;  // per sorted position
  DevBuf<uint64_t> vox_code;
  DevBuf<uint32_t> vox_start;
  DevBuf<float> xs, ys, zs;  // points in sorted order (SoA)
  int64_t Nf = 0;            // finite points
  int64_t V = 0;
  int code_bits = 0;

  // features
  DevBuf<NodeRec> node;
  DevBuf<uint32_t> used_ids, used_rank;
  int64_t U = 0;

  // adjacency
  DevBuf<uint64_t> hkey;
  DevBuf<uint32_t> hval;
  uint32_t hbits = 0;
  DevBuf<int32_t> offsets;  // packed dx,dy,dz
  DevBuf<uint64_t> adj_masks;  // ball cells per (voxel position in its brick, brick offset): k_adjacency_masks
  int adj_mask_nb = 0;         // bricks per axis the ball can touch (0 = no mask table)
  bool adj_tab_valid = false;  // offsets / masks / length tables are on the device for (adj_tab_graph, adj_tab_voxel)
  float adj_tab_graph = 0.f, adj_tab_voxel = 0.f;
  // per adjacency row: start position of every group of equal integer offset length (crossValidation searches only
  // inside the group of the wanted distance); adj_nvals = the distinct lengths, adj_nrank[length] = its index
  DevBuf<uint16_t> adj_gtab;
  DevBuf<int32_t> adj_nvals;
  DevBuf<uint8_t> adj_nrank;
  int adj_ngroups = 0, adj_gstride = 0;
  bool adj_have_gtab = false;
  int n_off = 0;
  int adj_R = 0;   // largest |offset| per axis of the ball table
  int adj_stride = 0;
  DevBuf<uint64_t> adj_key;
  DevBuf<uint16_t> adj_off;   // per row entry: packed lattice offset from the row's voxel, (dx+16) | (dy+16) << 5 | (dz+16) << 10; row[0] = 0xffff: none
  bool adj_have_off = false;
  // crossValidation through the lattice (round 4): every cut also leaves its connect list as a BIT per lattice offset (conn_bits: U
  // rows of cb_words words; bit = vgs_cb_index(offset) in the cube of side 2 cb_R + 1 around the voxel, cb_R = the ball's largest
  // offset per axis; the centre bit -- the voxel itself, always a member -- says "this row has bits"), so "is i in L0(k)?" is one
  // bit of k's row, at the index of the negated offset = cube size - 1 - index, instead of a search in k's 8-byte keys.
  DevBuf<uint32_t> conn_bits;
  int cb_words = 0, cb_R = 0;
  bool cb_enabled = false;     // the cuts of this run wrote conn_bits
  DevBuf<uint32_t> adj_cnt, adj_mused;  // per used voxel: stored row length, number of ALL neighbours
  bool adj_pruned = false;              // rows hold used neighbours only
  float adj_r2 = 0.f;

  // near-pair lists (nearlist.hip): per voxel id, the heavy pairs with a used voxel at most two lattice steps away
  DevBuf<uint8_t> nl_cnt;
  DevBuf<uint32_t> nl_tot;
  DevBuf<float4> nl_ent;
  bool nl_enabled = false;
  int nl_reach_steps = 0;     // lattice steps up to which the lists are complete (nearlist.hip)
  bool nl_direct = false;     // the ball fits the direct offset map of the one-wavefront classes

  // pair lists (pairlist.hip): every heavy pair inside a voxel's ball, built on demand for the rows k_localcut_pg reads
  DevBuf<uint8_t> pl_state;    // [V] index (uint2: first entry, count) | [V] wanted marks | [V] "part of a heavy pair" flags: one fill resets all three
  DevBuf<float4> pl_ent;       // the pool of entries
  DevBuf<uint32_t> pl_work;    // work list of rows, redo list
  bool pl_enabled = false;
  bool pl_enabled_at_launch = false;   // this run queued the pair-list chain for its hand-overs (stream4 joins stream3 before the stage's last event)
  hipEvent_t ev_ho = nullptr, ev_ho2 = nullptr;   // hand-over lists built / pair-list chain done
  float pl_w_ring = 0.0f;      // PairLists::w_ring of this run

  // local cut / merge
  DevBuf<uint8_t> conn;
  DevBuf<uint32_t> evals;      // per used voxel: pair evaluations of its local cut (diagnostics, summed on request)
  // hand-overs of the local cut run on a side stream while the merge stage already cross-validates the rows they cannot
  // touch: per used voxel "handed over, connect row not final yet", the rows put off, and what vgs_localcut_finish needs
  DevBuf<uint8_t> lc_pending;
  DevBuf<uint8_t> lc_defer_flag;   // per row: the first pass of crossValidation put it off (written by every row of that pass)
  DevBuf<uint32_t> lc_defer;
  struct { bool open = false; bool dense = true; bool gated = false; /* merge's first pass looks at LcGate's word */ bool many = false; /* ... and found LC_MANY */ unsigned int grid_f = 0, grid_g = 0, nabc[5] = {0, 0, 0, 0, 0}; float tail_ms = 0.f; bool pg_xl_queued = false; /* class D's pair-list kernel queued for the extra-large one, which was launched */ } lc_tail;
  int64_t lc_diag[16] = {0};   // vgs_get_schedule_counters[_ex]
  DevBuf<float> lc_ctab;      // screening table of the dense hand-over kernels (localcut.hip: lc_screen_table)
  float lc_ctab_key[8] = {0}, lc_ctab_scale = 0.0f;
  bool lc_ctab_valid = false;
  DevBuf<uint32_t> csize;      // per voxel: list length after crossValidation (0 for unused)
  DevBuf<int32_t> attach;      // per voxel: closestCheck target or -1
  DevBuf<uint8_t> cc_flags;    // per voxel: bit0 candidate, bit1 success
  DevBuf<uint32_t> parent;
  DevBuf<uint32_t> csz, kept_rank;
  DevBuf<int32_t> vox_label;
  DevBuf<int32_t> pt_label;      // labels of the current cloud
  DevBuf<int32_t> pt_label_alt;  // the buffer the next run writes while an asynchronous download still reads pt_label
  // getClusterIdx on the device (clusters.hip): offsets and point indices of the kept clusters, default order; rebuilt after a run on request
  DevBuf<int64_t> cl_off;
  DevBuf<int32_t> cl_idx;
  bool cl_valid = false;
  DevBuf<uint64_t> counters;   // device-side counters (pairs, flags)
  DevBuf<uint32_t> work_ids;   // scratch index lists

  int64_t counts[VGS_N_COUNTS] = {0};
  double times[VGS_T_COUNT] = {0};
  hipEvent_t ev[16] = {nullptr};   // 14, 15: the label tail of the merge stage (VGS_T_LABELS); 15 also orders label downloads behind the label kernel
  hipEvent_t tev[VGS_T_COUNT][2] = {{nullptr}};   // stage timers: begin / end of every stage, read lazily (capi.hip: timed)
  bool tev_pending[VGS_T_COUNT] = {false};

  // SVGS
  DevBuf<int32_t> sv_label;     // per point: supervoxel label (0 = unassigned), what getLabeledCloud returns (SS:283)
  int32_t sv_max_label = 0;
  bool sv_have_labels = false;
  bool sv_labels_external = false;  // supplied through svgs_set_supervoxel_labels (independent of the VCCS parameters)
  int64_t sv_label_n = -1;          // number of points the labelling was made for
  DevBuf<uint32_t> sv_key_a, sv_key_b;
  DevBuf<uint64_t> cell_code_a, cell_code_b;
  DevBuf<uint32_t> cell_id_a, cell_id_b, cell_start;
  // VCCS-style supervoxel stage
  DevBuf<float> vc_cen, vc_nrm, vc_dist, vc_state;
  DevBuf<int32_t> vc_nbr, vc_label;
  DevBuf<int32_t> vc_nbr4;    // vc_nbr once more as [7][V] int4 (28 entries per voxel, two unused): four neighbours to a load for the expansion rounds
  DevBuf<uint32_t> vc_tile_start;   // expansion over tiles (vccs.hip): start of every 8^3 tile's run of voxels
  DevBuf<uint32_t> vc_tile_of, vc_tchg;        // vccs_mode 1: tile of every voxel; per tile "a live flag changed in that sweep" (two sweeps' worth)
  DevBuf<int32_t> vc_nbr_tiles;                // vccs_mode 1: the tiles on the 26 sides of a tile ([NT][27], -1: none)
  DevBuf<int32_t> vc_plive;                    // vccs_mode 1 over tiles: owner << 1 | live, two sweeps' worth
  DevBuf<uint16_t> vc_cell;                    // a voxel's cell in its tile's 10^3 label array
  DevBuf<unsigned int> vc_ring;   // (vccs_mode 1) the sweeps' change counters of a pass
  DevBuf<unsigned int> vc_dbg;    // (vccs_mode 1, VGS_DEBUG) lanes that moved a voxel, per pass and round
  DevBuf<uint64_t> vc_halo, vc_tile_meta, vc_pool;   // (voxel, cell) of the voxels in the tiles' shells; (offset, length) per tile; entries handed out
  DevBuf<uint64_t> vc_seedkey;
  DevBuf<long long> vc_sums;
  DevBuf<uint32_t> vc_count;
  DevBuf<float> vc_accu;        // vccs_mode 1: 1-ring covariance accumulators (10 floats per voxel)
  DevBuf<uint8_t> vc_live, vc_alive;   // vccs_mode 1: leaf is expanded from at its owner's turn (two sweeps' worth); supervoxel still exists

  // multi-GPU
  bool have_region = false;
  double own_lo[2] = {0, 0}, own_hi[2] = {0, 0};
  DevBuf<uint8_t> owned;        // per voxel: 1 = centre inside this rank's region
  int64_t own_first = 0, n_own = -1;   // points [own_first, own_first + n_own) of the cloud were loaded by this rank itself, the rest came with other ranks' strips (-1: not told)
  DevBuf<uint8_t> mixsrc;       // per voxel: holds own-loaded points / holds points from other ranks' strips
  DevBuf<uint8_t> straddle;     // per voxel: 1 = its cube crosses the border of the region (it may hold points of two ranks)
  DevBuf<uint64_t> bnd_code;    // boundary records
  DevBuf<int32_t> bnd_root;
  DevBuf<int32_t> root_label;   // per voxel id: label of the component rooted there
  DevBuf<uint64_t> bnd_code2;   // compact protocol: unique boundary voxels (code, root, owned count of the root)
  DevBuf<int32_t> bnd_root2, bnd_cnt;
  DevBuf<uint8_t> broot;        // per voxel id: 1 = root named by a boundary record
  int64_t bnd_unique = -1, bnd_kept_local = 0;  // results of the last vgs_get_boundary_roots (-1 = not computed)
};

#define VGS_HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) {                                                                          \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                                \
      if (_e == hipErrorOutOfMemory) { (void)hipGetLastError(); return VGS_E_NOMEM; }                \
      return VGS_E_HIP;                                                                              \
    }                                                                                                \
  } while (0)

// small device -> host read-back through the context's pinned scratch, followed by a wait for the stream
static inline vgs_status vgs_readback(vgs_ctx* c, void* dst, const void* src_dev, size_t bytes) {
  if (bytes > 4096 || !c->pin) {
    VGS_HIP_TRY(c, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VGS_OK;
  }
  VGS_HIP_TRY(c, hipMemcpyAsync(c->pin, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  memcpy(dst, c->pin, bytes);
  return VGS_OK;
}
// Longest row of connect bits the kernels hold in LDS: 624 words = a cube of 27 cells a side = balls of up to 13 voxels (the edge list of
// the smallest one-wavefront class, 312 keys of 8 bytes, is where its row is put together).  Wider balls (up to the engine's own limit
// of graph_size / voxel_size = 12 ... 15 by rounding) keep crossValidation's search in the neighbour's row.  (Round 4: 256 words, 9 voxels --
// config 2's ball of ten fell back silently.)
#define VGS_CB_MAX_WORDS 624
// bit of the lattice offset p (packed (dx+16) | (dy+16) << 5 | (dz+16) << 10, as the rows' adj_off holds it) in a row of connect bits
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t vgs_cb_index(uint32_t p, int R) {
  const int D = 2 * R + 1;
  const int dx = (int)(p & 31u) - 16 + R, dy = (int)((p >> 5) & 31u) - 16 + R, dz = (int)((p >> 10) & 31u) - 16 + R;
  return (uint32_t)((dz * D + dy) * D + dx);
}
#endif
#define VGS_READBACK(ctx, dst, src, bytes) do { vgs_status _s = vgs_readback((ctx), (dst), (src), (bytes)); if (_s != VGS_OK) return _s; } while (0)

// The same in two halves (round 4): the copy is queued NOW, the host collects it LATER.  Kernels launched in between that do not need
// the number run while the host makes its round trip (20-30 us of an idle GPU per read-back otherwise: profiles/r04_step_timeline.txt).
// Uses the upper half of the pinned scratch, so that a plain vgs_readback in between does not overwrite it.
static inline vgs_status vgs_readback_begin(vgs_ctx* c, const void* src_dev, size_t bytes) {
  if (bytes > 2048 || !c->pin || !c->ev_rb) return VGS_E_ARG;
  VGS_HIP_TRY(c, hipMemcpyAsync((char*)c->pin + 2048, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
  VGS_HIP_TRY(c, hipEventRecord(c->ev_rb, c->stream));
  return VGS_OK;
}
static inline vgs_status vgs_readback_end(vgs_ctx* c, void* dst, size_t bytes) {
  VGS_HIP_TRY(c, hipEventSynchronize(c->ev_rb));
  memcpy(dst, (char*)c->pin + 2048, bytes);
  return VGS_OK;
}
static inline bool vgs_can_split_readback(const vgs_ctx* c) { return c->pin != nullptr && c->ev_rb != nullptr; }

vgs_status vgs_cut_order(vgs_ctx* c, std::vector<uint16_t>& ord_host, std::vector<uint32_t>& k_host, bool want_lists = false,
                         const uint8_t* list_flag = nullptr, std::vector<uint32_t>* list_cnt = nullptr, std::vector<int32_t>* list_ids = nullptr);   // cutorder.hip
void vgs_read_env_knobs(vgs_ctx* c);   // capi.hip; called by vgs_create only

// stage implementations (one .hip file each)
vgs_status vgs_stage_voxelize(vgs_ctx* c);
vgs_status vgs_stage_features(vgs_ctx* c);
vgs_status vgs_stage_adjacency(vgs_ctx* c);
vgs_status vgs_run_adjacency(vgs_ctx* c, bool full, uint64_t* out_key, uint32_t* out_cnt, uint32_t* out_nall, float r2, const uint32_t* ids = nullptr,
                             int64_t n_ids = 0);
bool vgs_unused_are_inert(const vgs_params& p);
vgs_status vgs_stage_nearlists(vgs_ctx* c);   // part of the local-cut stage
vgs_status vgs_pairlists_begin(vgs_ctx* c, hipStream_t strm);   // pairlist.hip, part of the local-cut stage
vgs_status vgs_pairlists_build(vgs_ctx* c, hipStream_t strm, const uint32_t* const* ids, const unsigned int* const* n_dev, const unsigned int* n_host,
                               int n_lists, bool all_rows, const float* ctab, float ctab_scale, float d2_stop, int slot, const LcGate& gate,
                               bool big_rows_too);
vgs_status vgs_stage_localcut(vgs_ctx* c);   // launches everything; its hand-over kernels may still run when it returns
vgs_status vgs_ensure_point_labels(vgs_ctx* c);   // merge.hip
vgs_status vgs_localcut_finish(vgs_ctx* c, unsigned int* n_deferred);  // waits for them, checks the stage's flags (called by the merge stage)
vgs_status vgs_stage_merge(vgs_ctx* c);
vgs_status vgs_clusters_on_device(vgs_ctx* c);   // clusters.hip
vgs_status vgs_stage_vccs(vgs_ctx* c);
vgs_status vgs_stage_svgs_group(vgs_ctx* c);
vgs_status vgs_stage_svgs_neighbours(vgs_ctx* c);
vgs_status vgs_grow_box_from(vgs_ctx* c, OctreeBox& box, bool record_epochs);
vgs_status vgs_compute_owned(vgs_ctx* c);

#endif
