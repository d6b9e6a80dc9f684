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
#ifndef NLB_PF
#define NLB_PF 4   // rows whose heads are requested together
#endif
#ifndef NL_WAVES
#define NL_WAVES 8   // 64 registers, no spill: the 32 wavefronts per CU the 2.4 KB queue allows (at the default budget the compiler took 69: 28)
#endif
__global__ __launch_bounds__(64, NL_WAVES) void k_near_lists(const uint32_t* __restrict__ used_ids, int64_t U, const uint64_t* __restrict__ adj_key,
                                                   const uint32_t* __restrict__ adj_cnt, int adj_stride, const NodeRec* __restrict__ node,
                                                   const uint16_t* __restrict__ gtab, int gstride, VgsWeightParams W, float thr0,
                                                   float lat_d2_lim, float d2max, const uint64_t* __restrict__ vox_code,
                                                   float res_f, float min_x, float min_y, float min_z, float cube_tol,
                                                   uint8_t* __restrict__ out_cnt, uint32_t* __restrict__ out_tot, float4* __restrict__ out_ent,
                                                   const uint16_t* __restrict__ adj_off) {
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
    // Round 5: the rows of NLB_PF voxels are requested TOGETHER -- head of the row (entries 1..64: on a surface every entry within the
    // reach is among them), its packed lattice offsets (adj_off: which candidates lie at a positive offset within the reach is decided
    // without touching their records), the voxel's own record -- and the records of the candidates that qualify are gathered for all
    // of them before the first is used.  (Was: voxel after voxel, each with its row -> records -> queue chain of dependent round trips;
    // this step took half of the kernel's 0.43 ms, three quarters of it parked.)
    bool full = false;
    while (g < ng && !full) {
      uint64_t hk[NLB_PF];
      uint32_t ho[NLB_PF], hslot[NLB_PF];
      float hbx[NLB_PF], hby[NLB_PF], hbz[NLB_PF];
      bool hcand[NLB_PF], hneed[NLB_PF];
      // (wave-uniform, requested with the rows: the voxel's id, its row's length and marks; its record follows with the candidates' records)
      uint32_t hi[NLB_PF], hgt[NLB_PF], ho0[NLB_PF];
      int hn[NLB_PF];
      float hax[NLB_PF], hay[NLB_PF], haz[NLB_PF];
#pragma unroll
      for (int j = 0; j < NLB_PF; ++j) {
        hk[j] = ~0ull; ho[j] = 0u; hi[j] = 0u; hgt[j] = 0xffffu; ho0[j] = 0xffffu; hn[j] = 0;
        if (g + j < ng) {
          const int64_t u = u0 + g + j;
          hi[j] = used_ids[u]; hn[j] = (int)adj_cnt[u]; hgt[j] = gtab[u * gstride];
          if (adj_off != nullptr) {
            const int k = 1 + lane;
            ho0[j] = adj_off[u * adj_stride];
            if (k < adj_stride) { hk[j] = adj_key[u * adj_stride + k]; ho[j] = adj_off[u * adj_stride + k]; }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NLB_PF; ++j) {
        hcand[j] = false; hneed[j] = false; hslot[j] = 0u; hbx[j] = 0.f; hby[j] = 0.f; hbz[j] = 0.f; hax[j] = 0.f; hay[j] = 0.f; haz[j] = 0.f;
        if (g + j < ng) {
          const NodeRec& me = node[hi[j]];
          hax[j] = (me.flags & VGS_F_POS) ? me.c[0] : vm_nan();   // as the cut stages centroids: an unusable position is a NaN x
          hay[j] = me.c[1]; haz[j] = me.c[2];
        }
        if (g + j < ng && adj_off != nullptr) {
          const int n = hn[j];
          hcand[j] = (1 + lane < n) && vm_from_bits((uint32_t)(hk[j] >> 32)) <= lat_d2_lim;
          if (hcand[j]) {
            const int dx = (int)(ho[j] & 31u) - 16, dy = (int)((ho[j] >> 5) & 31u) - 16, dz = (int)((ho[j] >> 10) & 31u) - 16;
            const bool positive = dz > 0 || (dz == 0 && (dy > 0 || (dy == 0 && dx > 0)));
            if (positive && dx >= -NL_REACH && dx <= NL_REACH && dy >= -NL_REACH && dy <= NL_REACH && dz >= -NL_REACH && dz <= NL_REACH) {
              const NodeRec& nb = node[(uint32_t)hk[j]];
              hneed[j] = true;
              hbx[j] = (nb.flags & VGS_F_POS) ? nb.c[0] : vm_nan();
              hby[j] = nb.c[1]; hbz[j] = nb.c[2];
              hslot[j] = (uint32_t)(dx + NL_REACH) | ((uint32_t)(dy + NL_REACH) << 4) | ((uint32_t)(dz + NL_REACH) << 8);
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NLB_PF; ++j) {
        if (g >= ng || full) break;
        // half of the (2 * NL_REACH + 1)^3 - 1 cells can qualify at most (positive offsets)
        if (nq + ((2 * NL_REACH + 1) * (2 * NL_REACH + 1) * (2 * NL_REACH + 1) - 1) / 2 > NLB_QCAP) { full = true; break; }
        const int64_t u = u0 + g;
        const uint32_t i = hi[j];
        if (lane == 0) { s_vid[g] = i; s_start[g] = nq; }
        // a row whose centre distances are not within half a lattice step^2 of their offsets' integer lengths (coordinates so
        // large that float rounding rivals the voxel size; adjacency.hip marks it) gives no safe candidate prefix: no list
        if (hgt[j] == 0xffffu) { if (lane == 0) s_kept[g] = -1; ++g; continue; }
        if (lane == 0) s_kept[g] = 0;
        const int n = hn[j];
        const uint64_t* row = adj_key + u * adj_stride;
        const float ax = hax[j], ay = hay[j], az = haz[j];
        int base = 1;
        if (ho0[j] != 0xffffu) {
          // the head, from what was requested above
          bool ok = false;
          float d2 = 0.f;
          if (hneed[j]) {
            const float ex = ax - hbx[j], ey = ay - hby[j], ez = az - hbz[j];   // the cut's own expression (order-free: squares)
            d2 = (ex * ex + ey * ey) + ez * ez;
            d2 = (d2 == d2) ? d2 : 1.0e4f;
            ok = d2 < d2max;
          }
          const unsigned long long mk = __ballot(ok);
          if (ok) {
            const int pos = nq + __popcll(mk & lt);   // < NLB_QCAP: checked before the voxel was started
            q_t[pos] = (uint32_t)hk[j]; q_d2[pos] = d2; q_slot[pos] = (uint16_t)hslot[j]; q_g[pos] = (uint8_t)g;
          }
          nq += __popcll(mk);
          // the row is sorted by centre distance: it goes on behind the head only if the head's last entry was still a candidate
          base = ((__ballot(hcand[j]) >> 63) & 1ull) ? 65 : n;
        }
        for (; base < n; base += 64) {   // entry 0 is the voxel itself
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
            const uint32_t npad = nb.pad, mpad = node[i].pad;
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
        ++g;
      }
    }
    if (lane == 0) s_start[g] = nq;
    __syncthreads();
#if defined(NLB_STOP) && NLB_STOP == 1
    if (lane == 0) out_cnt[s_vid[g_first]] = (uint8_t)nq; continue;
#endif
    // ---- 2. weights, on full wavefronts: both orientations of every queued pair ----
    // (round 5: one lane per pair computes both -- only the convexity distance depends on the order of the two voxels, vgs_math.h:
    // vm_pair_weight_both -- where two lanes each evaluated the whole weight)
    for (int q = lane; q < nq; q += 64) {
      const NodeRec& A = node[s_vid[q_g[q]]];
      const NodeRec& B = node[q_t[q]];
      float w12, w21;
      vm_pair_weight_both(A, B, W, &w12, &w21);
      q_w[q] = w12; q_w2[q] = w21;
    }
    __syncthreads();
    for (int q = lane; q < nq; q += 64) {
      const float w1 = q_w[q], w2 = q_w2[q];
      if (!((w1 > thr0) || (w2 > thr0))) q_w[q] = vm_nan();   // NaN compares false: a pair is kept when either orientation is heavy
      else if (!(w1 == w1)) q_w[q] = 0.0f;                     // (kept for w(b, a): its own weight must not look like the mark)
    }
    __syncthreads();
#if defined(NLB_STOP) && NLB_STOP == 2
    if (lane == 0) out_cnt[s_vid[g_first]] = (uint8_t)(q_w[0] > 0.f); continue;
#endif
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
                     (float)c->box.min[1], (float)c->box.min[2], NL_CUBE_TOL * res, c->nl_cnt.p, c->nl_tot.p, c->nl_ent.p,
                     c->adj_have_off ? c->adj_off.p : (const uint16_t*)nullptr);
  VGS_HIP_TRY(c, hipGetLastError());
  c->nl_enabled = true;
  return VGS_OK;
}
