// nearlist.hip -- builder of the near-pair lists (see nearlist.hpp); runs at the head of the local-cut stage.
// One wavefront per used voxel a: the first entries of its adjacency row (sorted by lattice distance) are its
// candidates; a lane takes one candidate, gathers its record, and evaluates the weight if the pair qualifies.
#include <cstdlib>

#include "vgs_context.hpp"

#include "nearlist.hpp"

__global__ __launch_bounds__(64) void k_near_lists(const uint32_t* __restrict__ used_ids, int64_t U, const uint64_t* __restrict__ adj_key,
                                                   const uint32_t* __restrict__ adj_cnt, int adj_stride, const NodeRec* __restrict__ node,
                                                   const uint16_t* __restrict__ gtab, int gstride, VgsWeightParams W, float thr0,
                                                   float lat_d2_lim, float d2max,
                                                   uint8_t* __restrict__ out_cnt, uint16_t* __restrict__ out_slot, float2* __restrict__ out_dw) {
  __shared__ float s_d2[NL_S], s_w[NL_S];
  __shared__ uint16_t s_slot[NL_S];
  const int64_t u = vgs_xcd_item(blockIdx.x, U);
  if (u >= U) return;
  const int lane = threadIdx.x;
  const uint32_t i = used_ids[u];
  const int n = (int)adj_cnt[u];
  const uint64_t* row = adj_key + u * adj_stride;
  // a row whose centre distances are not within half a lattice step^2 of their offsets' integer lengths (coordinates so
  // large that float rounding rivals the voxel size; adjacency.hip marks it) gives no safe candidate prefix: no list
  if (gtab[u * gstride] == 0xffffu) { if (lane == 0) out_cnt[i] = NL_NONE; return; }
  const NodeRec me = node[i];
  const float ax = (me.flags & VGS_F_POS) ? me.c[0] : vm_nan();   // as the cut stages centroids: an unusable position is a NaN x
  int total = 0;
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
    bool valid = false;
    float d2 = 0.f, w = 0.f;
    uint32_t slot = 0;
    if (cand) {
      const NodeRec nb = node[t];
      const int dx = nl_diff10(nb.pad & 1023u, me.pad & 1023u), dy = nl_diff10((nb.pad >> 10) & 1023u, (me.pad >> 10) & 1023u),
                dz = nl_diff10((nb.pad >> 20) & 1023u, (me.pad >> 20) & 1023u);
      if (dx >= -NL_REACH && dx <= NL_REACH && dy >= -NL_REACH && dy <= NL_REACH && dz >= -NL_REACH && dz <= NL_REACH) {
        const float bx = (nb.flags & VGS_F_POS) ? nb.c[0] : vm_nan();
        const float ex = ax - bx, ey = me.c[1] - nb.c[1], ez = me.c[2] - nb.c[2];   // the cut's own expression (order-free: squares)
        d2 = (ex * ex + ey * ey) + ez * ez;
        d2 = (d2 == d2) ? d2 : 1.0e4f;
        if (d2 < d2max) {
          w = vm_pair_weight(me, nb, W);
          valid = w > thr0;   // NaN compares false
          slot = (uint32_t)(dx + NL_REACH) | ((uint32_t)(dy + NL_REACH) << 4) | ((uint32_t)(dz + NL_REACH) << 8);
        }
      }
    }
    const unsigned long long mk = __ballot(valid);
    if (valid) {
      const int pos = total + __popcll(mk & ((1ull << lane) - 1ull));
      if (pos < NL_S) { s_d2[pos] = d2; s_w[pos] = w; s_slot[pos] = (uint16_t)slot; }
    }
    total += __popcll(mk);
  }
  if (total > NL_S) { if (lane == 0) out_cnt[i] = NL_NONE; return; }
  __syncthreads();
  if (lane < total) {
    // ascending (d2, slot): every entry counts the smaller ones
    const float md = s_d2[lane];
    const uint32_t ms = s_slot[lane];
    int r = 0;
    for (int q = 0; q < total; ++q) {
      const float qd = s_d2[q];
      r += (qd < md || (qd == md && (uint32_t)s_slot[q] < ms)) ? 1 : 0;
    }
    const size_t o = (size_t)i * NL_S + (size_t)r;
    out_dw[o] = make_float2(md, s_w[lane]); out_slot[o] = (uint16_t)ms;
  } else if (lane < NL_S) {
    out_dw[(size_t)i * NL_S + (size_t)lane] = make_float2(__builtin_huge_valf(), 0.0f);   // end of list for readers
  }
  if (lane == 0) out_cnt[i] = (uint8_t)total;
}

// the lists exist when voxels are on a lattice (VGS), unused voxels are out of the rows (they would need entries too),
// and the neighbourhood ball fits the consumer's offset map
vgs_status vgs_stage_nearlists(vgs_ctx* c) {
  c->nl_enabled = false;
  if (c->P.method != 2 || !c->adj_pruned || !c->adj_have_gtab || c->U == 0 || getenv("VGS_NO_NEAR")) return VGS_OK;
  const double rr = (double)c->P.graph_size / (double)c->P.voxel_size;
  if (rr * rr * (1.0 + 1e-4) + 1e-3 >= (double)((NL_BALL + 1) * (NL_BALL + 1))) return VGS_OK;   // some offset reaches NL_BALL + 1 (adjacency.hip: lim2)
  const int64_t V = c->V, U = c->U;
  VGS_HIP_TRY(c, c->nl_cnt.ensure(V)); VGS_HIP_TRY(c, c->nl_slot.ensure((size_t)V * NL_S));
  VGS_HIP_TRY(c, c->nl_dw.ensure((size_t)V * NL_S));
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
  hipLaunchKernelGGL(k_near_lists, dim3(vgs_xcd_grid(U)), dim3(64), 0, c->stream, c->used_ids.p, U, c->adj_key.p, c->adj_cnt.p, c->adj_stride,
                     c->node.p, c->adj_gtab.p, c->adj_gstride, W, thr0, lat_lim, reach * reach, c->nl_cnt.p, c->nl_slot.p, c->nl_dw.p);
  VGS_HIP_TRY(c, hipGetLastError());
  c->nl_enabled = true;
  return VGS_OK;
}
