// clusters.hip -- getClusterIdx() on the device (voxel_segmentation.h:117, filled by drawColorMapofPointsinClusters VS:963-1009;
// supervoxel_segmentation.h:2109-2126): the reference's end product, per kept cluster the indices of its points.
//
// Default order (VGS_ORDER_VOXEL_ID: clusters by ascending smallest voxel id = the labels' order, inside a cluster ascending voxel id,
// inside a voxel ascending point index): the points are already in voxel order with ascending indices inside a voxel (perm, the
// stable sort of the voxelize stage), so the lists are ONE STABLE SORT of that order by the label of the point's voxel -- one or two
// radix passes over 4-byte keys (the kept labels need ceil(log2(K + 1)) bits) -- and the offsets a binary search per label in the
// sorted keys.  Nothing leaves the device until the caller asks for it; round 4 downloaded perm, the point -> voxel map and the
// labels and walked them on the host (never timed).
// The reference's own element order (VGS_ORDER_REFERENCE: the depth-first walk of recursionSearch, seed last) stays a host walk
// over the connect lists (capi.hip); what it downloads is compacted on the device first (k_final_lists below).
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "vgs_context.hpp"

// key of sorted position j: the kept label of its voxel, K for points of dropped clusters (they sort behind every kept cluster)
__global__ void k_cluster_keys(const uint32_t* __restrict__ pt_vox, const int32_t* __restrict__ vox_label, int64_t nf, uint32_t K,
                               uint32_t* __restrict__ key) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nf) return;
  const uint32_t v = pt_vox[j];
  const int32_t l = v == 0xffffffffu ? -1 : vox_label[v];
  key[j] = l < 0 ? K : (uint32_t)l;
}
// offsets[k] = first position of a key >= k in the sorted keys (k = 0 .. K: offsets[K] = number of points in kept clusters)
__global__ void k_cluster_offsets(const uint32_t* __restrict__ key, int64_t nf, uint32_t K, int64_t* __restrict__ off) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > K) return;
  int64_t lo = 0, hi = nf;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (key[mid] < k) lo = mid + 1; else hi = mid; }
  off[k] = lo;
}

// The lists in HBM: cl_off[K + 1] (int64), cl_idx[offsets[K]] (int32).  Valid until the next run of the stages.
vgs_status vgs_clusters_on_device(vgs_ctx* c) {
  if (c->cl_valid) return VGS_OK;
  const int64_t K = c->counts[VGS_N_KEPT], nf = c->Nf;
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  VGS_HIP_TRY(c, c->cl_off.ensure((size_t)K + 2));
  VGS_HIP_TRY(c, c->cl_idx.ensure((size_t)(nf > 0 ? nf : 1)));
  if (nf == 0 || K == 0) {
    VGS_HIP_TRY(c, hipMemsetAsync(c->cl_off.p, 0, ((size_t)K + 1) * sizeof(int64_t), c->stream));
    VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->cl_valid = true;
    return VGS_OK;
  }
  // scratch: the voxelize stage's code buffers are free once the voxel table exists (8 N bytes each: two 32-bit arrays)
  VGS_HIP_TRY(c, c->code_a.ensure((size_t)nf)); VGS_HIP_TRY(c, c->code_b.ensure((size_t)nf));
  uint32_t* key_in = (uint32_t*)c->code_a.p;
  uint32_t* key_out = key_in + nf;
  uint32_t* val_out = (uint32_t*)c->cl_idx.p;
  const int TB = 256;
  const unsigned nb = (unsigned)((nf + TB - 1) / TB);
  hipLaunchKernelGGL(k_cluster_keys, dim3(nb), dim3(TB), 0, c->stream, c->pt_vox.p, c->vox_label.p, nf, (uint32_t)K, key_in);
  unsigned bits = 1;
  while (bits < 32 && (1ull << bits) <= (unsigned long long)K) ++bits;   // keys 0 .. K
  size_t tmp = 0;
  VGS_HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, tmp, key_in, key_out, c->perm_b.p, val_out, (size_t)nf, 0, bits, c->stream));
  VGS_HIP_TRY(c, c->sort_tmp.ensure(tmp));
  VGS_HIP_TRY(c, rocprim::radix_sort_pairs(c->sort_tmp.p, tmp, key_in, key_out, c->perm_b.p, val_out, (size_t)nf, 0, bits, c->stream));
  hipLaunchKernelGGL(k_cluster_offsets, dim3((unsigned)((K + 1 + TB - 1) / TB)), dim3(TB), 0, c->stream, key_out, nf, (uint32_t)K, c->cl_off.p);
  VGS_HIP_TRY(c, hipGetLastError());
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->cl_valid = true;
  return VGS_OK;
}

extern "C" vgs_status vgs_get_clusters_device(vgs_ctx* c, const int64_t** offsets_dev, const int32_t** point_idx_dev) {
  if (!c || !offsets_dev || !point_idx_dev) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED) { c->err = "vgs_get_clusters_device: segment first (drawColorMapofPointsinClusters precedes getClusterIdx, VS:1006)"; return VGS_E_STATE; }
  vgs_status s = vgs_clusters_on_device(c);
  if (s != VGS_OK) return s;
  *offsets_dev = c->cl_off.p;
  *point_idx_dev = c->cl_idx.p;
  return VGS_OK;
}
