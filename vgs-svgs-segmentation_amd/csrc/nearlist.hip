// nearlist.hip -- builder of the near-pair lists (see nearlist.hpp); runs at the head of the local-cut stage.
// One wavefront per used voxel a: the first entries of its adjacency row (sorted by lattice distance) are its
// candidates; a lane takes one candidate, gathers its record, and evaluates the weight if the pair qualifies.
#include <cstdlib>

#include "vgs_context.hpp"

#include "nearlist.hpp"

// NLB_G voxels per wavefront: the candidates of all of them that pass the cheap tests (lattice reach, centroid distance)
// are queued in LDS, so that the expensive part -- the weight -- runs on full wavefronts (a voxel alone fills a third of one).
#ifndef NLB_G
#define NLB_G 8
#endif
#ifndef NLB_QCAP
#define NLB_QCAP 128
#endif
#ifndef NL_WAVES
#define NL_WAVES 8   // 64 registers, no spill: the 32 wavefronts per CU the 2.4 KB queue allows (at the default budget the compiler took 69: 28)
#endif
__global__ __launch_bounds__(64, NL_WAVES) void k_near_lists(const uint32_t* __restrict__ used_ids, int64_t U, const uint64_t* __restrict__ adj_key,
                                                   const uint32_t* __restrict__ adj_cnt, int adj_stride, const NodeRec* __restrict__ node,
                                                   const uint16_t* __restrict__ gtab, int gstride, VgsWeightParams W, float thr0,
                                                   float lat_d2_lim, float d2max, const uint64_t* __restrict__ vox_code,
                                                   float res_f, float min_x, float min_y, float min_z, float cube_tol,
                                                   uint8_t* __restrict__ out_cnt, uint32_t* __restrict__ out_tot, float4* __restrict__ out_ent) {
  __shared__ uint32_t q_t[NLB_QCAP];      // partner voxel id
  __shared__ float q_d2[NLB_QCAP], q_w[NLB_QCAP], q_w2[NLB_QCAP];   // centroid distance^2; w(a, b), NaN = pair not kept; w(b, a)
  __shared__ uint16_t q_slot[NLB_QCAP];
  __shared__ uint8_t q_g[NLB_QCAP];       // which voxel of the group
  __shared__ uint32_t s_vid[NLB_G];
  __shared__ int s_start[NLB_G + 1], s_kept[NLB_G];
  const int lane = threadIdx.x;
  const int64_t ngroups = (U + NLB_G - 1) / NLB_G;
  const int64_t grp = vgs_xcd_item(blockIdx.x, ngroups);
  if (grp >= ngroups) return;
  const int64_t u0 = grp * NLB_G;
  const int ng = (int)((U - u0 < NLB_G) ? (U - u0) : NLB_G);
  const unsigned long long lt = (1ull << lane) - 1ull;
  // The reach argument (nearlist.hpp) wants the centroid inside the voxel's cube; float sums of large coordinates can leave
  // it outside by more than the slack built into d2max: such a voxel gets no list.  Lane g checks voxel g of the group, all
  // of them at once (one round trip instead of one per voxel).
  bool cube_lane_ok = true;   // balloted where the lists are closed: by then the loads have long arrived
  {
    bool ok = true;
    if (lane < ng) {
      const uint32_t vi = used_ids[u0 + lane];
      const NodeRec& r = node[vi];
      const uint64_t code = vox_code[vi];
      const float fx = vm_voxel_center(vm_compact21(code >> 2), res_f, min_x), fy = vm_voxel_center(vm_compact21(code >> 1), res_f, min_y),
                  fz = vm_voxel_center(vm_compact21(code), res_f, min_z);
      const float lim = 0.5f * res_f + cube_tol;
      const float ulp = 1.2e-7f;   // the float centre itself is off by up to half a spacing of its magnitude
      const bool inside = fabsf(r.c[0] - fx) + fabsf(fx) * ulp <= lim && fabsf(r.c[1] - fy) + fabsf(fy) * ulp <= lim &&
                          fabsf(r.c[2] - fz) + fabsf(fz) * ulp <= lim;
      ok = inside || !(r.flags & VGS_F_POS);
    }
    cube_lane_ok = ok;
  }
  int g = 0;
  while (g < ng) {
    // ---- 1. queue the candidates of as many voxels of the group as fit ----
    const int g_first = g;
    int nq = 0;
    for (; g < ng; ++g) {
      // half of the (2 * NL_REACH + 1)^3 - 1 cells can qualify at most (positive offsets)
      if (nq + ((2 * NL_REACH + 1) * (2 * NL_REACH + 1) * (2 * NL_REACH + 1) - 1) / 2 > NLB_QCAP) break;
      const int64_t u = u0 + g;
      const uint32_t i = used_ids[u];
      if (lane == 0) { s_vid[g] = i; s_start[g] = nq; }
      // a row whose centre distances are not within half a lattice step^2 of their offsets' integer lengths (coordinates so
      // large that float rounding rivals the voxel size; adjacency.hip marks it) gives no safe candidate prefix: no list
      if (gtab[u * gstride] == 0xffffu) { if (lane == 0) s_kept[g] = -1; continue; }
      const NodeRec& me = node[i];
      if (lane == 0) s_kept[g] = 0;
      const int n = (int)adj_cnt[u];
      const uint64_t* row = adj_key + u * adj_stride;
      const uint32_t mpad = me.pad;
      const float ax = (me.flags & VGS_F_POS) ? me.c[0] : vm_nan();   // as the cut stages centroids: an unusable position is a NaN x
      const float ay = me.c[1], az = me.c[2];
      for (int base = 1; base < n; base += 64) {   // entry 0 is the voxel itself
        const int k = base + lane;
        bool cand = false;
        uint32_t t = 0;
        if (k < n) {
          const uint64_t kk = row[k];
          cand = vm_from_bits((uint32_t)(kk >> 32)) <= lat_d2_lim;   // centre distance: beyond sqrt(12) lattice steps no offset fits the reach
          t = (uint32_t)kk;
        }
        if (__ballot(cand) == 0ull) break;   // the row is sorted by centre distance
        bool ok = false;
        float d2 = 0.f;
        uint32_t slot = 0;
        if (cand) {
          const NodeRec& nb = node[t];
          const uint32_t npad = nb.pad;
          const int dx = nl_diff10(npad & 1023u, mpad & 1023u), dy = nl_diff10((npad >> 10) & 1023u, (mpad >> 10) & 1023u),
                    dz = nl_diff10((npad >> 20) & 1023u, (mpad >> 20) & 1023u);
          const bool positive = dz > 0 || (dz == 0 && (dy > 0 || (dy == 0 && dx > 0)));   // the pair lives in this voxel's list
          if (positive && dx >= -NL_REACH && dx <= NL_REACH && dy >= -NL_REACH && dy <= NL_REACH && dz >= -NL_REACH && dz <= NL_REACH) {
            const float bx = (nb.flags & VGS_F_POS) ? nb.c[0] : vm_nan();
            const float ex = ax - bx, ey = ay - nb.c[1], ez = az - nb.c[2];   // the cut's own expression (order-free: squares)
            d2 = (ex * ex + ey * ey) + ez * ez;
            d2 = (d2 == d2) ? d2 : 1.0e4f;
            ok = d2 < d2max;
            slot = (uint32_t)(dx + NL_REACH) | ((uint32_t)(dy + NL_REACH) << 4) | ((uint32_t)(dz + NL_REACH) << 8);
          }
        }
        const unsigned long long mk = __ballot(ok);
        if (ok) {
          const int pos = nq + __popcll(mk & lt);   // < NLB_QCAP: checked before the voxel was started
          q_t[pos] = t; q_d2[pos] = d2; q_slot[pos] = (uint16_t)slot; q_g[pos] = (uint8_t)g;
        }
        nq += __popcll(mk);
      }
    }
    if (lane == 0) s_start[g] = nq;
    __syncthreads();
    // ---- 2. weights, on full wavefronts: both orientations of every queued pair ----
    for (int e = lane; e < 2 * nq; e += 64) {
      const int q = e >> 1;
      const NodeRec& A = node[s_vid[q_g[q]]];
      const NodeRec& B = node[q_t[q]];
      const float w = (e & 1) ? vm_pair_weight(B, A, W) : vm_pair_weight(A, B, W);
      if (e & 1) q_w2[q] = w; else q_w[q] = w;
    }
    __syncthreads();
    for (int q = lane; q < nq; q += 64) {
      const float w1 = q_w[q], w2 = q_w2[q];
      if (!((w1 > thr0) || (w2 > thr0))) q_w[q] = vm_nan();   // NaN compares false: a pair is kept when either orientation is heavy
      else if (!(w1 == w1)) q_w[q] = 0.0f;                     // (kept for w(b, a): its own weight must not look like the mark)
    }
    __syncthreads();
    // ---- 3. every kept entry finds its place in its voxel's list: ascending (d2, slot) among the kept ones ----
    for (int e = lane; e < nq; e += 64) {
      const float mw = q_w[e];
      if (mw == mw) {
        const int gg = q_g[e];
        const int s0 = s_start[gg], s1 = s_start[gg + 1];
        const float md = q_d2[e];
        const uint32_t ms = q_slot[e];
        int r = 0;
        for (int q = s0; q < s1; ++q) {
          const float qw = q_w[q], qd = q_d2[q];
          r += (qw == qw && (qd < md || (qd == md && (uint32_t)q_slot[q] < ms))) ? 1 : 0;
        }
        if (r < NL_S) {
          const size_t o = (size_t)s_vid[gg] * NL_S + (size_t)r;
          out_ent[o] = make_float4(md, mw, q_w2[e], __uint_as_float(ms));
        }
        if (s_kept[gg] >= 0) atomicAdd(&s_kept[gg], 1);
        atomicAdd(&out_tot[s_vid[gg]], 1u); atomicAdd(&out_tot[q_t[e]], 1u);
      }
    }
    __syncthreads();
    // ---- 4. counts and end-of-list marks ----
    const unsigned long long cube_ok = __ballot(cube_lane_ok);
    for (int x = lane; x < (g - g_first) * NL_S; x += 64) {
      const int gg = g_first + x / NL_S, j = x % NL_S;
      const int kept = s_kept[gg];
      if (kept >= 0 && kept <= NL_S && j >= kept) out_ent[(size_t)s_vid[gg] * NL_S + (size_t)j] = make_float4(__builtin_huge_valf(), 0.0f, 0.0f, 0.0f);
      if (j == 0) {
        const bool none = kept < 0 || kept > NL_S || !((cube_ok >> gg) & 1ull);
        out_cnt[s_vid[gg]] = none ? (uint8_t)NL_NONE : (uint8_t)kept;
        // a reader that does not look at the count (the one-wavefront classes of the cut) finds "no list" as a NaN distance in entry 0
        if (none) out_ent[(size_t)s_vid[gg] * NL_S] = make_float4(vm_nan(), 0.0f, 0.0f, 0.0f);
      }
    }
    __syncthreads();
  }
}

// the lists exist when voxels are on a lattice (VGS), unused voxels are out of the rows (they would need entries too),
// and the neighbourhood ball fits the consumer's offset map
vgs_status vgs_stage_nearlists(vgs_ctx* c) {
  c->nl_enabled = false;
  if (c->P.method != 2 || !c->adj_pruned || !c->adj_have_gtab || c->U == 0 || c->K.no_near) return VGS_OK;
  const double rr = (double)c->P.graph_size / (double)c->P.voxel_size;
  // some offset reaches NL_BALL + 1 (adjacency.hip: lim2): the one-wavefront classes cannot use the lists then (their offset
  // map ends at NL_BALL), the multi-wavefront classes look partners up in a hash and still can
  c->nl_direct = !(rr * rr * (1.0 + 1e-4) + 1e-3 >= (double)((NL_BALL + 1) * (NL_BALL + 1)));
  // The builder takes a voxel's candidates from its adjacency row, so the lists hold the partners INSIDE the search ball
  // only.  They are complete for pairs up to R lattice steps apart (Chebyshev) if every such offset lies strictly inside the
  // ball: 3 R^2 < (graph_size / voxel_size)^2, with a margin for the float predicate that decides the ball's rim.  (Found by
  // tools/fuzz_parity.py at graph_size = 2 voxels: the offset (2,1,0) is outside the ball, yet centroids that far apart on
  // the lattice can be one voxel apart.)  The reference's own defaults (0.5 / 0.15 = 3.33 voxels) give R = 1.
  int reach_steps = 0;
  while (reach_steps < NL_REACH && 3.0 * (reach_steps + 1) * (reach_steps + 1) < rr * rr * (1.0 - 1e-3)) ++reach_steps;
  c->nl_reach_steps = reach_steps;
  if (reach_steps == 0) return VGS_OK;
  const int64_t V = c->V, U = c->U;
  VGS_HIP_TRY(c, c->nl_cnt.ensure(V)); VGS_HIP_TRY(c, c->nl_tot.ensure(V));
  VGS_HIP_TRY(c, c->nl_ent.ensure((size_t)V * NL_S));
  VGS_HIP_TRY(c, hipMemsetAsync(c->nl_tot.p, 0, (size_t)V * sizeof(uint32_t), c->stream));
  VgsWeightParams W;
  W.inv_sig_p = 1.0f / c->P.sig_p; W.inv_sig_n = 1.0f / c->P.sig_n; W.inv_sig_o = 1.0f / c->P.sig_o;
  W.inv_sig_e = 1.0f / c->P.sig_e; W.inv_sig_c = 1.0f / c->P.sig_c;
  W.inv_sig_w2 = 1.0f / (c->P.sig_w * c->P.sig_w);
  W.svgs = 0;
  const float res = c->P.voxel_size;
  const float thr0 = vm_cut_threshold(1.0f, c->P.cut_thred, 1);
  const float reach = (float)NL_REACH * res;
  // a pair inside the reach has a lattice offset of at most 3 * NL_REACH^2 squared steps; half a step^2 of slack for the float centres
  const float lat_lim = ((float)(3 * NL_REACH * NL_REACH) + 0.5f) * res * res;
  hipLaunchKernelGGL(k_near_lists, dim3(vgs_xcd_grid((U + NLB_G - 1) / NLB_G)), dim3(64), 0, c->stream, c->used_ids.p, U, c->adj_key.p, c->adj_cnt.p, c->adj_stride,
                     c->node.p, c->adj_gtab.p, c->adj_gstride, W, thr0, lat_lim, reach * reach, c->vox_code.p, res, (float)c->box.min[0],
                     (float)c->box.min[1], (float)c->box.min[2], NL_CUBE_TOL * res, c->nl_cnt.p, c->nl_tot.p, c->nl_ent.p);
  VGS_HIP_TRY(c, hipGetLastError());
  c->nl_enabled = true;
  return VGS_OK;
}
