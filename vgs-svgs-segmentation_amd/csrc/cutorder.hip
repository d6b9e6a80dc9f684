// cutorder.hip -- on request only (output formatting, not the hot path): the ORDER of every connect list.
//
// cutGraphSegmentation returns the vertex list of the segment that holds the centre voxel, and that list is in merge-history
// order: every merge appends the absorbed segment's vertices behind the survivor's (voxel_segmentation.h:1986-1998,
// 2003-2026).  clusteringVoxels' depth-first walk follows the lists in that order (VS:2032-2053), so the element order of
// getClusterIdx() depends on it.  The hot path keeps a flag per adjacency slot -- membership, which is all the merge and
// the labels need.  This file recomputes the order for the lists that were found:
//
//   the merges of the centre's segment S0 only ever join sub-segments of S0 (whatever merged with them is in S0), and
//   rejected edges change nothing, so the merge history of S0 is that of the sequential scan over the pairs INSIDE S0.
//
// Per used voxel: all pair weights inside S0 (k (k - 1) / 2 of them, first argument = the earlier row position), one
// segmented radix sort of (weight desc, pair id asc) keys over all voxels of a chunk (rocPRIM), then one wavefront per
// voxel replays the scan with linked vertex lists.  Tie rule and orientation as the oracle's lean flavour (unique pairs
// a < b, keep = segment of a on equal thresholds): the reference's own n x n std::sort leaves the order of the two
// orientations of a pair unspecified.
#include <cstring>
#include <string.h>

#include <algorithm>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "vgs_context.hpp"

#define CO_MAXK 2048   // rows end at 2048 entries (the general local-cut kernel's limit)

// k = |S0| of every used voxel (the set flags of its connect row)
__global__ void k_co_count(const uint8_t* __restrict__ conn, const uint32_t* __restrict__ adj_cnt, int adj_stride, int64_t U, uint32_t* __restrict__ kout) {
  const int64_t u = (int64_t)blockIdx.x;
  if (u >= U) return;
  const int n = (int)adj_cnt[u];
  const uint8_t* row = conn + u * adj_stride;
  int k = 0;
  for (int c = threadIdx.x; c < n; c += 64) k += row[c] ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) k += __shfl_xor(k, o, 64);
  if (threadIdx.x == 0) kout[u] = (uint32_t)k;
}

// keys of the pairs inside S0 of voxel u0 + blockIdx.x: weight bits above the complemented pair id (row positions); NaN -> 0
__global__ __launch_bounds__(256) void k_co_eval(int64_t u0, const uint64_t* __restrict__ adj_key, const uint32_t* __restrict__ adj_cnt, int adj_stride,
                                                 const uint8_t* __restrict__ conn, const NodeRec* __restrict__ node, VgsWeightParams W,
                                                 const uint64_t* __restrict__ offs, uint64_t* __restrict__ keys) {
  __shared__ uint16_t pos[CO_MAXK];
  __shared__ int s_k;
  const int64_t u = u0 + (int64_t)blockIdx.x;
  const int n = (int)adj_cnt[u];
  const uint64_t* row = adj_key + u * adj_stride;
  const uint8_t* crow = conn + u * adj_stride;
  if (threadIdx.x == 0) {   // S0 in row order (a few hundred entries at most; sequential is fine for an on-request pass)
    int k = 0;
    for (int c = 0; c < n; ++c) if (crow[c]) pos[k++] = (uint16_t)c;
    s_k = k;
  }
  __syncthreads();
  const int k = s_k;
  const int64_t np = (int64_t)k * (k - 1) / 2;
  uint64_t* out = keys + offs[blockIdx.x];
  for (int64_t p = threadIdx.x; p < np; p += blockDim.x) {
    // pair p in row-major order over i < j
    int i = (int)(((double)(2 * k - 1) - sqrt((double)(2 * k - 1) * (double)(2 * k - 1) - 8.0 * (double)p)) * 0.5);
    while (i > 0 && (int64_t)i * (2 * k - i - 1) / 2 > p) --i;
    while ((int64_t)(i + 1) * (2 * k - i - 2) / 2 <= p) ++i;
    const int j = i + 1 + (int)(p - (int64_t)i * (2 * k - i - 1) / 2);
    const int va = pos[i], vb = pos[j];
    const float w = vm_pair_weight(node[(uint32_t)row[va]], node[(uint32_t)row[vb]], W);
    const uint32_t pid = ((uint32_t)va << 16) | (uint32_t)vb;
    out[p] = (w != w) ? 0ull : (((uint64_t)vm_bits(w) << 32) | (uint64_t)(0xffffffffu - pid));
  }
}

// replay of the sequential scan (VS:1955-2001) over the sorted pairs of S0; ord[u * stride + r] = row position of the r-th
// vertex of the returned list
__global__ __launch_bounds__(64) void k_co_merge(int64_t u0, const uint32_t* __restrict__ adj_cnt, int adj_stride, const uint8_t* __restrict__ conn,
                                                 float cut, const uint64_t* __restrict__ offs, const uint64_t* __restrict__ keys,
                                                 uint16_t* __restrict__ ord, unsigned int* __restrict__ bad) {
  __shared__ uint16_t pos[CO_MAXK], lidx[CO_MAXK];      // S0 in row order; row position -> index in S0
  __shared__ uint16_t par[CO_MAXK], head[CO_MAXK], tail[CO_MAXK], nxt[CO_MAXK], ssz[CO_MAXK];
  __shared__ float thr[CO_MAXK];
  const int64_t u = u0 + (int64_t)blockIdx.x;
  const int n = (int)adj_cnt[u];
  const uint8_t* crow = conn + u * adj_stride;
  const int lane = threadIdx.x;
  int k = 0;
  if (lane == 0) {
    for (int c = 0; c < n; ++c) if (crow[c]) { pos[k] = (uint16_t)c; lidx[c] = (uint16_t)k; ++k; }
  }
  k = __shfl(k, 0, 64);
  const float thr0 = vm_cut_threshold(1.0f, cut, 1);
  __syncthreads();
  for (int i = lane; i < k; i += 64) { par[i] = (uint16_t)i; head[i] = (uint16_t)i; tail[i] = (uint16_t)i; nxt[i] = 0xffffu; ssz[i] = 1; thr[i] = thr0; }
  __syncthreads();
  uint16_t* orow = ord + u * adj_stride;
  if (k == 0) return;
  const int64_t np = (int64_t)k * (k - 1) / 2;
  const uint64_t* in = keys + offs[blockIdx.x];
  int merges = 0;
  // every lane runs the same scalar replay on the same values (the compiler keeps it uniform); 64 keys are fetched at a time
  for (int64_t base = 0; base < np && merges < k - 1; base += 64) {
    const uint64_t mine = (base + lane < np) ? in[base + lane] : 0ull;
    const int cnt = (int)((np - base) < 64 ? (np - base) : 64);
    for (int q = 0; q < cnt && merges < k - 1; ++q) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, q);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), q);
      if (lo == 0u && hi == 0u) { base = np; break; }   // NaN pairs sort last: nothing behind them merges (VS:1998: w > thr is false)
      const float w = vm_from_bits(hi);
      const uint32_t pid = 0xffffffffu - lo;
      int a = lidx[pid >> 16], b = lidx[pid & 0xffffu];
      while (par[a] != a) a = par[a];
      while (par[b] != b) b = par[b];
      if (a == b) continue;
      const float ta = thr[a], tb = thr[b];
      if (!((w > ta) && (w > tb))) continue;
      const int keep = (ta >= tb) ? a : b, gone = (ta >= tb) ? b : a;   // VS:1972-1983
      if (lane == 0) {
        const int nsz = (int)ssz[keep] + (int)ssz[gone];
        par[gone] = (uint16_t)keep;
        nxt[tail[keep]] = head[gone];    // seg_ver_idx[keep] gets the absorbed segment's vertices appended (VS:1990-1993)
        tail[keep] = tail[gone];
        ssz[keep] = (uint16_t)nsz;
        thr[keep] = vm_cut_threshold(w, cut, nsz);
      }
      ++merges;
      __syncthreads();
    }
  }
  __syncthreads();
  if (lane == 0) {
    int r0 = lidx[0];   // row position 0 is the voxel itself; it is in S0
    if (!crow[0]) { atomicAdd(bad, 1u); return; }
    while (par[r0] != r0) r0 = par[r0];
    int cntv = 0;
    for (int v = head[r0]; v != 0xffff; v = nxt[v]) { orow[cntv++] = pos[v]; if (cntv > k) break; }
    if (cntv != k) atomicAdd(bad, 1u);   // the replay must find exactly the set the hot path found
  }
}

// The lists themselves, compact (round 5): what the host walk of getClusterIdx's reference order needs is, per used voxel, the voxel ids
// of its list in that order -- after crossValidation only the entries whose mutual flag is set (it filters in list order, VS:2119-2152).
// Counted, scanned, written here; the host downloads these few ids instead of the whole tables of keys, flags and positions
// (2.4 GB at the 10 M-point config, the ids 85 MB).
__global__ __launch_bounds__(64) void k_co_list_count(const uint16_t* __restrict__ ord, const uint32_t* __restrict__ kk, const uint8_t* __restrict__ flag,
                                                      int adj_stride, int64_t U, uint32_t* __restrict__ cnt) {
  const int64_t u = (int64_t)blockIdx.x;
  if (u >= U) return;
  const int k = (int)kk[u];
  int n = 0;
  for (int r = threadIdx.x; r < k; r += 64) n += (flag == nullptr || flag[u * adj_stride + ord[u * adj_stride + r]]) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if (threadIdx.x == 0) cnt[u] = (uint32_t)n;
}
__global__ __launch_bounds__(64) void k_co_list_fill(const uint16_t* __restrict__ ord, const uint32_t* __restrict__ kk, const uint8_t* __restrict__ flag,
                                                     const uint64_t* __restrict__ adj_key, int adj_stride, int64_t U, const uint64_t* __restrict__ start,
                                                     int32_t* __restrict__ ids) {
  const int64_t u = (int64_t)blockIdx.x;
  if (u >= U) return;
  const int k = (int)kk[u];
  const int lane = threadIdx.x;
  uint64_t at = start[u];
  for (int base = 0; base < k; base += 64) {   // in list order: a ballot keeps it
    const int r = base + lane;
    bool keep = false;
    uint32_t t = 0;
    if (r < k) {
      const int64_t s = u * adj_stride + ord[u * adj_stride + r];
      keep = flag == nullptr || flag[s] != 0;
      t = (uint32_t)adj_key[s];
    }
    const unsigned long long mk = __ballot(keep);
    if (keep) ids[at + (uint64_t)__popcll(mk & ((1ull << lane) - 1ull))] = (int32_t)t;
    at += (uint64_t)__popcll(mk);
  }
}

static VgsWeightParams co_weight_params(const vgs_params& p) {
  VgsWeightParams W;
  W.inv_sig_p = 1.0f / p.sig_p; W.inv_sig_n = 1.0f / p.sig_n; W.inv_sig_o = 1.0f / p.sig_o;
  W.inv_sig_e = 1.0f / p.sig_e; W.inv_sig_c = 1.0f / p.sig_c;
  W.inv_sig_w2 = 1.0f / (p.sig_w * p.sig_w);
  W.svgs = (p.method == 3) ? 1 : 0;
  return W;
}

// host: ord_host[u * stride + r] for r < k_host[u]
// list_flag / list_cnt / list_ids (optional): instead of the positions, the compact lists -- list_flag null: every member of the cut's
// list (connect_cut), else the flags that filter it (the mutual flags of crossValidation)
vgs_status vgs_cut_order(vgs_ctx* c, std::vector<uint16_t>& ord_host, std::vector<uint32_t>& k_host, bool want_lists, const uint8_t* list_flag,
                         std::vector<uint32_t>* list_cnt, std::vector<int32_t>* list_ids) {
  const int64_t U = c->U;
  ord_host.clear(); k_host.assign((size_t)U, 0);
  if (list_cnt) list_cnt->assign((size_t)U, 0);
  if (list_ids) list_ids->clear();
  if (U == 0) return VGS_OK;
  // rows hold at most CO_MAXK entries: a longer one fails the local cut itself (VGS_E_UNSUPPORTED there); the row STRIDE (all
  // lattice offsets of the ball) may well be larger
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  DevBuf<uint32_t> d_k; DevBuf<uint16_t> d_ord; DevBuf<uint64_t> d_offs, d_keys_a, d_keys_b; DevBuf<uint8_t> d_tmp; DevBuf<unsigned int> d_bad;
  auto release = [&]() { d_k.release(); d_ord.release(); d_offs.release(); d_keys_a.release(); d_keys_b.release(); d_tmp.release(); d_bad.release(); };
  vgs_status st = VGS_OK;
  auto fail = [&](hipError_t e, const char* what) { c->err = std::string("vgs_cut_order: ") + what + ": " + hipGetErrorString(e); st = VGS_E_HIP; };
  do {
    hipError_t e;
    if ((e = d_k.ensure(U)) != hipSuccess || (e = d_ord.ensure((size_t)U * c->adj_stride)) != hipSuccess || (e = d_bad.ensure(1)) != hipSuccess) { fail(e, "alloc"); break; }
    if ((e = hipMemsetAsync(d_bad.p, 0, 4, c->stream)) != hipSuccess) { fail(e, "memset"); break; }
    hipLaunchKernelGGL(k_co_count, dim3((unsigned)U), dim3(64), 0, c->stream, c->conn.p, c->adj_cnt.p, c->adj_stride, U, d_k.p);
    if ((e = hipMemcpyAsync(k_host.data(), d_k.p, (size_t)U * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) { fail(e, "copy"); break; }
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) { fail(e, "count"); break; }
    // chunks of consecutive voxels with at most ~2^29 pairs (4 GB of keys, twice)
    const uint64_t budget = 1ull << 29;
    std::vector<uint64_t> offs;
    const VgsWeightParams W = co_weight_params(c->P);
    for (int64_t u0 = 0; u0 < U && st == VGS_OK;) {
      offs.assign(1, 0ull);
      int64_t u1 = u0;
      while (u1 < U) {
        const uint64_t k = k_host[(size_t)u1], np = k * (k - (k > 0 ? 1 : 0)) / 2;
        if (u1 > u0 && offs.back() + np > budget) break;
        offs.push_back(offs.back() + np);
        ++u1;
      }
      const int64_t m = u1 - u0;
      const uint64_t total = offs.back();
      if (total >= (1ull << 32)) { c->err = "vgs_cut_order: a single neighbourhood with more than 2^32 pairs"; st = VGS_E_UNSUPPORTED; break; }
      if ((e = d_offs.ensure((size_t)m + 1)) != hipSuccess || (e = d_keys_a.ensure((size_t)total + 1)) != hipSuccess || (e = d_keys_b.ensure((size_t)total + 1)) != hipSuccess) { fail(e, "alloc"); break; }
      if ((e = hipMemcpyAsync(d_offs.p, offs.data(), ((size_t)m + 1) * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess) { fail(e, "copy"); break; }
      hipLaunchKernelGGL(k_co_eval, dim3((unsigned)m), dim3(256), 0, c->stream, u0, c->adj_key.p, c->adj_cnt.p, c->adj_stride, c->conn.p, c->node.p, W, d_offs.p, d_keys_a.p);
      const uint64_t* sorted = d_keys_a.p;
      if (total > 0) {
        size_t tb = 0;
        if ((e = rocprim::segmented_radix_sort_keys_desc(nullptr, tb, d_keys_a.p, d_keys_b.p, (unsigned int)total, (unsigned int)m, d_offs.p, d_offs.p + 1, 0, 64, c->stream)) != hipSuccess) { fail(e, "sort size"); break; }
        if ((e = d_tmp.ensure(tb + 16)) != hipSuccess) { fail(e, "alloc"); break; }
        if ((e = rocprim::segmented_radix_sort_keys_desc(d_tmp.p, tb, d_keys_a.p, d_keys_b.p, (unsigned int)total, (unsigned int)m, d_offs.p, d_offs.p + 1, 0, 64, c->stream)) != hipSuccess) { fail(e, "sort"); break; }
        sorted = d_keys_b.p;
      }
      hipLaunchKernelGGL(k_co_merge, dim3((unsigned)m), dim3(64), 0, c->stream, u0, c->adj_cnt.p, c->adj_stride, c->conn.p, c->P.cut_thred, d_offs.p, sorted, d_ord.p, d_bad.p);
      if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) { fail(e, "chunk"); break; }   // offs (host vector) is reused by the next chunk
      u0 = u1;
    }
    if (st != VGS_OK) break;
    unsigned int bad = 0;
    if ((e = hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost)) != hipSuccess) { fail(e, "copy"); break; }
    if (bad) { c->err = "vgs_cut_order: the replay of " + std::to_string(bad) + " local cuts does not end in the connect set the hot path found"; st = VGS_E_STATE; break; }
    if (want_lists && list_cnt && list_ids) {
      DevBuf<uint32_t> d_cnt; DevBuf<uint64_t> d_start; DevBuf<int32_t> d_ids;
      do {
        if ((e = d_cnt.ensure(U)) != hipSuccess || (e = d_start.ensure(U + 1)) != hipSuccess) { fail(e, "alloc"); break; }
        hipLaunchKernelGGL(k_co_list_count, dim3((unsigned)U), dim3(64), 0, c->stream, d_ord.p, d_k.p, list_flag, c->adj_stride, U, d_cnt.p);
        if ((e = hipMemcpyAsync(list_cnt->data(), d_cnt.p, (size_t)U * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) { fail(e, "copy"); break; }
        if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) { fail(e, "count lists"); break; }
        std::vector<uint64_t> start((size_t)U + 1, 0ull);
        for (int64_t u = 0; u < U; ++u) start[(size_t)u + 1] = start[(size_t)u] + (*list_cnt)[(size_t)u];
        const uint64_t total = start[(size_t)U];
        list_ids->resize((size_t)total);
        if (total == 0) break;
        if ((e = d_ids.ensure((size_t)total)) != hipSuccess) { fail(e, "alloc"); break; }
        if ((e = hipMemcpyAsync(d_start.p, start.data(), ((size_t)U + 1) * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess) { fail(e, "copy"); break; }
        hipLaunchKernelGGL(k_co_list_fill, dim3((unsigned)U), dim3(64), 0, c->stream, d_ord.p, d_k.p, list_flag, c->adj_key.p, c->adj_stride, U, d_start.p, d_ids.p);
        if ((e = hipMemcpyAsync(list_ids->data(), d_ids.p, (size_t)total * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) { fail(e, "copy"); break; }
        if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) { fail(e, "fill lists"); break; }
      } while (false);
      d_cnt.release(); d_start.release(); d_ids.release();
      break;
    }
    ord_host.resize((size_t)U * c->adj_stride);
    if ((e = hipMemcpy(ord_host.data(), d_ord.p, ord_host.size() * 2, hipMemcpyDeviceToHost)) != hipSuccess) { fail(e, "copy"); break; }
  } while (false);
  release();
  return st;
}
