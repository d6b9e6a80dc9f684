// voxelize.hip -- stage a1/a2: points -> occupied voxels in PCL leaf order.
// Replaces OctreePointCloud::addPointsFromInputCloud + setVoxelCenters + getVoxelNum
// (reference test:54-62, voxel_segmentation.h:146-189; octree semantics SURVEY.md B.1).
//
// HBM-bound integer work: one 12/16-byte read per point for the code, one stable LSD radix sort
// of (code, index) pairs, one run-length pass, one gather of the points into leaf order (SoA) so that
// the PCA stage reads each voxel's points as one contiguous, ascending-index run.
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <limits>

#include "vgs_context.hpp"

// ---------------------------------------------------------------------------------------------
// OctreePointCloud bounding box growth (host side; only record-setting points reach it)
// ---------------------------------------------------------------------------------------------
bool OctreeBox::contains(const float* p) const {
  if (!defined) return false;
  for (int a = 0; a < 3; ++a)
    if ((double)p[a] < min[a] || (double)p[a] >= max[a]) return false;
  return true;
}

void OctreeBox::adopt(const float* p) {
  const double eps = (double)std::numeric_limits<float>::epsilon();
  while (true) {
    if (!defined) {
      // first point: box of one voxel around it, padded symmetrically to depth 1 (getKeyBitSize)
      for (int a = 0; a < 3; ++a) { min[a] = (double)p[a] - res / 2; max[a] = (double)p[a] + res / 2; }
      unsigned mk = 2;
      for (int a = 0; a < 3; ++a) mk = std::max(mk, (unsigned)((max[a] - min[a]) / res));
      depth = (int)std::min(32u, (unsigned)std::ceil(std::log2((double)mk) - eps));
      double side = (double)(1u << depth) * res - eps;
      for (int a = 0; a < 3; ++a) {
        double over = (side - (max[a] - min[a])) / 2.0;
        min[a] -= over;
        max[a] += over;
      }
      defined = true;
      continue;
    }
    bool hi[3], any = false;
    for (int a = 0; a < 3; ++a) {
      hi[a] = (double)p[a] >= max[a];
      any = any || hi[a] || ((double)p[a] < min[a]);
    }
    if (!any) return;
    double side = (double)(1u << depth) * res;
    for (int a = 0; a < 3; ++a)
      if (!hi[a]) { min[a] -= side; shift[a] += (1ull << depth); }  // old root becomes the upper child
    depth++;
    side = (double)(1u << depth) * res - eps;
    for (int a = 0; a < 3; ++a) max[a] = min[a] + side;
  }
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool finite3(float x, float y, float z) {
  return (vm_bits(x) & 0x7f800000u) != 0x7f800000u && (vm_bits(y) & 0x7f800000u) != 0x7f800000u &&
         (vm_bits(z) & 0x7f800000u) != 0x7f800000u;
}

// The growth of the octree box runs on the device: a scan finds the smallest index of a finite point outside the box,
// one thread replays PCL's growth rule (OctreeBox::adopt, same double arithmetic) for that point and records the
// epoch, the next scan starts behind it.  The host queues a batch of (scan, adopt) pairs and reads the state once;
// pairs queued after the last growth return at once.
struct GrowState {
  double min[3], max[3], res;
  unsigned long long shift[3];
  unsigned long long found;      // search word of the running scan (smallest violating index, ~0 = none)
  long long start;               // next scan starts here
  int depth, defined, done, n_epochs, record, overflow;
  Epoch epochs[VGS_MAX_EPOCHS];
};

struct BoxD { double min[3], max[3]; int defined; };
__device__ __forceinline__ bool fv_outside(float x, float y, float z, const BoxD& g) {
  if (!finite3(x, y, z)) return false;
  return !g.defined || (double)x < g.min[0] || (double)x >= g.max[0] || (double)y < g.min[1] ||
         (double)y >= g.max[1] || (double)z < g.min[2] || (double)z >= g.max[2];
}

// Packed xyz (12-byte points) is read as three 16-byte loads per four points; the running answer is polled once per trip.
__global__ void k_first_violation(const float* __restrict__ xyz, int stride_f, int64_t n, GrowState* __restrict__ gs) {
  if (gs->done) return;
  BoxD box;   // a private copy: the search word below lives in the same structure
  for (int a = 0; a < 3; ++a) { box.min[a] = gs->min[a]; box.max[a] = gs->max[a]; }
  box.defined = gs->defined;
  unsigned long long* result = &gs->found;
  const int64_t start = gs->start;
  unsigned long long best = ~0ull;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // an undefined box is violated by every finite point: the first one is near `start`, a few workgroups are enough
  const int64_t nthreads = box.defined ? (int64_t)gridDim.x * blockDim.x : (int64_t)8 * blockDim.x;
  if (tid >= nthreads) return;
  if (stride_f == 3 && (((uintptr_t)xyz) & 15u) == 0) {
    // groups of four points = 48 bytes = three float4; group g holds points 4g .. 4g+3
    const float4* q = (const float4*)xyz;
    for (int64_t g = (start >> 2) + tid; 4 * g < n; g += nthreads) {
      if ((unsigned long long)(4 * g) > __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      float v[12];
      if (4 * g + 4 <= n) {
        const float4 a = q[3 * g], b = q[3 * g + 1], c4 = q[3 * g + 2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        v[8] = c4.x; v[9] = c4.y; v[10] = c4.z; v[11] = c4.w;
      } else {
        for (int k = 0; k < 12; ++k) v[k] = (4 * g + k / 3 < n) ? xyz[12 * g + k] : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t i = 4 * g + k;
        if (i >= start && i < n && best == ~0ull && fv_outside(v[3 * k], v[3 * k + 1], v[3 * k + 2], box)) best = (unsigned long long)i;
      }
      if (best != ~0ull) break;
    }
  } else {
    for (int64_t i = start + tid; i < n; i += nthreads) {
      if ((unsigned long long)i > __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      const float* p = xyz + i * stride_f;
      if (fv_outside(p[0], p[1], p[2], box)) { best = (unsigned long long)i; break; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long other = __shfl_down(best, o, 64);
    best = other < best ? other : best;
  }
  // (same-address atomics serialise: only a wavefront that improves the answer issues one)
  if ((threadIdx.x & 63) == 0 && best < __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(result, best);
}

// one thread: OctreeBox::adopt for the point the scan found (the same statements in the same order, double arithmetic)
__device__ void grow_adopt(const float* __restrict__ xyz, int stride_f, GrowState* g, unsigned long long idx) {
  const float* pp = xyz + (int64_t)idx * stride_f;
  const float p[3] = {pp[0], pp[1], pp[2]};
  const double eps = 1.1920928955078125e-07;   // std::numeric_limits<float>::epsilon()
  const double res = g->res;
  while (true) {
    if (!g->defined) {
      // first point: box of one voxel around it; (max - min) / res is 1 (or 0 after rounding), so getKeyBitSize gives depth 1
      for (int a = 0; a < 3; ++a) { g->min[a] = (double)p[a] - res / 2; g->max[a] = (double)p[a] + res / 2; }
      g->depth = 1;
      const double side = (double)(1u << g->depth) * res - eps;
      for (int a = 0; a < 3; ++a) {
        const double over = (side - (g->max[a] - g->min[a])) / 2.0;
        g->min[a] -= over;
        g->max[a] += over;
      }
      g->defined = 1;
      continue;
    }
    bool hi[3], any = false;
    for (int a = 0; a < 3; ++a) {
      hi[a] = (double)p[a] >= g->max[a];
      any = any || hi[a] || ((double)p[a] < g->min[a]);
    }
    if (!any) break;
    double side = (double)(1u << g->depth) * res;
    for (int a = 0; a < 3; ++a)
      if (!hi[a]) { g->min[a] -= side; g->shift[a] += (1ull << g->depth); }  // old root becomes the upper child
    g->depth++;
    side = (double)(1u << g->depth) * res - eps;
    for (int a = 0; a < 3; ++a) g->max[a] = g->min[a] + side;
    if (g->depth > 30) break;   // far beyond the supported depth of 21: the host reports it
  }
  if (g->record) {
    if (g->n_epochs >= VGS_MAX_EPOCHS) { g->overflow = 1; g->done = 1; return; }
    Epoch& e = g->epochs[g->n_epochs++];
    e.first = (int64_t)idx;
    for (int a = 0; a < 3; ++a) { e.min[a] = g->min[a]; e.shift[a] = g->shift[a]; }
  }
  g->start = (long long)idx + 1;
}
__global__ void k_adopt(const float* __restrict__ xyz, int stride_f, GrowState* __restrict__ g, int pinned) {
  if (g->done) return;
  const unsigned long long idx = g->found;
  g->found = ~0ull;
  if (idx == ~0ull) { g->done = 1; return; }
  if (pinned) { g->done = 2; return; }   // a point outside a pinned grid: the host reports it
  grow_adopt(xyz, stride_f, g, idx);
}

// The growth over the first n_prefix points in ONE launch (round 6): a cloud in random order has grown its box for the last time within its
// first few dozen points -- every growth step doubles the box, and a point outside a box that already covers most of the scene comes soon
// -- yet every step used to be a scan launch and an adopt launch (sixteen launches and 0.17 ms in front of the first key).  One workgroup
// keeps the state in LDS, finds the first point of the prefix outside the box, adopts it, and starts again behind it until the prefix
// holds none; the scans over the whole cloud (k_first_violation) continue from there, and normally find nothing.
__global__ __launch_bounds__(1024) void k_grow_prefix(const float* __restrict__ xyz, int stride_f, int64_t n_prefix, GrowState* __restrict__ gs, int pinned) {
  __shared__ GrowState S;
  __shared__ unsigned long long s_best[16];
  const int tid = threadIdx.x;
  static_assert(sizeof(GrowState) % 8 == 0, "copied as 64-bit words");
  for (int k = tid; k < (int)(sizeof(GrowState) / 8); k += 1024) ((unsigned long long*)&S)[k] = ((const unsigned long long*)gs)[k];
  __syncthreads();
  if (S.done) return;
  while (true) {
    BoxD box;
    for (int a = 0; a < 3; ++a) { box.min[a] = S.min[a]; box.max[a] = S.max[a]; }
    box.defined = S.defined;
    const int64_t start = S.start;
    unsigned long long best = ~0ull;
    for (int64_t i = start + tid; i < n_prefix; i += 1024) {
      const float* p = xyz + i * stride_f;
      if (fv_outside(p[0], p[1], p[2], box)) { best = (unsigned long long)i; break; }
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_down(best, o, 64); best = other < best ? other : best; }
    __syncthreads();   // (everybody has read the state)
    if ((tid & 63) == 0) s_best[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
      unsigned long long b = ~0ull;
      for (int w = 0; w < 16; ++w) b = s_best[w] < b ? s_best[w] : b;
      if (b == ~0ull) { if (S.start < n_prefix) S.start = n_prefix; S.found = ~0ull; s_best[0] = 0ull; }
      else if (pinned) { S.done = 2; s_best[0] = 0ull; }
      else { grow_adopt(xyz, stride_f, &S, b); s_best[0] = (S.done || S.overflow) ? 0ull : 1ull; }
    }
    __syncthreads();
    if (s_best[0] == 0ull) break;
    __syncthreads();   // (s_best is rewritten at the top)
  }
  for (int k = tid; k < (int)(sizeof(GrowState) / 8); k += 1024) ((unsigned long long*)gs)[k] = ((const unsigned long long*)&S)[k];
}

// code = valid bit | Morton(key), key generated with the box of the point's insertion epoch
// The epochs are read where the growth left them (GrowState on the device): no table goes through the host.
// KeyT: uint32_t when the valid bit + 3 * depth bits fit 32 (scenes up to 1024 voxels across: the 10 M-point bench scene has
// depth 10) -- the radix sort then moves 8 instead of 12 bytes per point and pass -- uint64_t otherwise.
template <typename KeyT>
__global__ void k_make_codes(const float* __restrict__ xyz, int stride_f, int64_t n, const GrowState* __restrict__ gs,
                             double res, int code_bits, KeyT* __restrict__ code, uint32_t* __restrict__ perm, int pack_shift) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = xyz + i * stride_f;
  float x = p[0], y = p[1], z = p[2];
  uint64_t c = 0;
  if (finite3(x, y, z)) {
    int e = gs->n_epochs - 1;
    while (e > 0 && i < gs->epochs[e].first) --e;   // epoch 0 also covers what lies in front of its first point (nothing finite)
    const Epoch& ep = gs->epochs[e];
    // keys are generated with the box of the insertion epoch and shifted by what the box has grown downwards since
    uint32_t kx = vm_axis_key(x, ep.min[0], res) + (uint32_t)(gs->shift[0] - ep.shift[0]);
    uint32_t ky = vm_axis_key(y, ep.min[1], res) + (uint32_t)(gs->shift[1] - ep.shift[1]);
    uint32_t kz = vm_axis_key(z, ep.min[2], res) + (uint32_t)(gs->shift[2] - ep.shift[2]);
    c = (1ull << code_bits) | vm_morton(kx, ky, kz);
  }
  // pack_shift > 0 (64-bit keys with room below the code): the point index travels in the key's low bits and the sort moves keys
  // only -- 8 instead of 12 bytes per point and pass; the sort looks at the bits from pack_shift up, so equal codes keep their
  // ascending index order exactly as the stable pair sort keeps it
  if (pack_shift > 0) { code[i] = (KeyT)((c << pack_shift) | (uint64_t)i); return; }
  code[i] = (KeyT)c;
  perm[i] = (uint32_t)i;
}

// run heads in the sorted code array (invalid codes == 0 sit at the tail of the descending order)
template <typename KeyT>
__global__ void k_heads(const KeyT* __restrict__ code, int64_t n, uint32_t* __restrict__ head, int pack_shift, uint32_t* __restrict__ perm_out) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) {
    const uint64_t raw = (uint64_t)code[j];
    const uint64_t c = raw >> pack_shift;
    head[j] = (c != 0 && (j == 0 || ((uint64_t)code[j - 1] >> pack_shift) != c)) ? 1u : 0u;
    if (pack_shift > 0) perm_out[j] = (uint32_t)(raw & ((1ull << pack_shift) - 1ull));   // the sorted order, unpacked for everybody downstream
  }
}

// number of valid (non-zero) codes = index of the first zero in the descending array
template <typename KeyT>
__global__ void k_count_valid(const KeyT* __restrict__ code, int64_t n, unsigned long long* __restrict__ n_valid, int pack_shift) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (((uint64_t)code[mid] >> pack_shift) != 0) lo = mid + 1; else hi = mid; }
  *n_valid = (unsigned long long)lo;
}
__global__ void k_copy_last(const uint32_t* __restrict__ scan, int64_t n, unsigned long long* __restrict__ out) { *out = (unsigned long long)scan[n - 1]; }

template <typename KeyT>
__global__ void k_voxel_table(const KeyT* __restrict__ code, const uint32_t* __restrict__ head,
                              const uint32_t* __restrict__ scan, int64_t n, uint64_t mask, uint32_t* __restrict__ pt_vox,
                              uint64_t* __restrict__ vox_code, uint32_t* __restrict__ vox_start, int pack_shift) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint64_t c = (uint64_t)code[j] >> pack_shift;
  if (c == 0) { pt_vox[j] = 0xffffffffu; return; }
  uint32_t v = scan[j] - 1u;
  pt_vox[j] = v;
  if (head[j]) { vox_code[v] = c & mask; vox_start[v] = (uint32_t)j; }
}

__global__ void k_gather_points(const float* __restrict__ xyz, int stride_f, const uint32_t* __restrict__ perm, int64_t nf,
                                float* __restrict__ xs, float* __restrict__ ys, float* __restrict__ zs) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nf) return;
  const float* p = xyz + (int64_t)perm[j] * stride_f;
  xs[j] = p[0]; ys[j] = p[1]; zs[j] = p[2];
}

__global__ void k_set_u32(uint32_t* p, uint32_t v) { *p = v; }

// ---------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------
// Sequential semantics, parallel execution: only points that fall outside the current box change it, so the
// device finds the next such point and the host replays PCL's growth rule for it.
vgs_status vgs_grow_box_from(vgs_ctx* c, OctreeBox& box, bool record_epochs) {
  static_assert(sizeof(GrowState) % 8 == 0, "GrowState is copied as 64-bit words");
  VGS_HIP_TRY(c, c->counters.ensure(64));
  VGS_HIP_TRY(c, c->grow_state.ensure(sizeof(GrowState) / 8));
  GrowState* d_g = (GrowState*)c->grow_state.p;
  static thread_local GrowState h;   // 7 KB: not on the stack
  std::memset(&h, 0, sizeof(h));
  for (int a = 0; a < 3; ++a) { h.min[a] = box.min[a]; h.max[a] = box.max[a]; h.shift[a] = box.shift[a]; }
  h.res = box.res; h.depth = box.depth; h.defined = box.defined ? 1 : 0;
  h.found = ~0ull; h.start = 0; h.record = record_epochs ? 1 : 0;
  const size_t hb_bytes = offsetof(GrowState, epochs) + sizeof(Epoch);
  // The caller pinned a grid it has itself replayed over this very cloud or over its bounding box (vgs_set_grid_covering: the tiled
  // driver, every step): nothing can grow and nothing needs checking -- the sixteen launches and the read-back of the scan below are
  // 0.2 ms of a tile's step -- so the device gets the final state as it is: one epoch from the first point on, growth done.
  const bool covering = c->grid_pinned && c->grid_covers && record_epochs && box.defined;
  if (covering) h.done = 1;
  if (record_epochs && box.defined) {
    Epoch& e = h.epochs[h.n_epochs++];
    e.first = 0;
    for (int a = 0; a < 3; ++a) { e.min[a] = box.min[a]; e.shift[a] = box.shift[a]; }
  }
  {
    // header + the epoch a pinned grid starts with; through the pinned scratch when there is one (a copy out of pageable memory is
    // staged by the runtime and holds the stream's first kernel back by tens of microseconds)
    const size_t hb = hb_bytes;
    static_assert(offsetof(GrowState, epochs) + sizeof(Epoch) <= 1024, "the growth header fits its slot of the pinned scratch");
    const void* src = &h;
    if (c->pin) { memcpy((char*)c->pin + 1024, &h, hb); src = (char*)c->pin + 1024; }
    VGS_HIP_TRY(c, hipMemcpyAsync(d_g, src, hb, hipMemcpyHostToDevice, c->stream));
  }
  if (covering) {   // (the state just uploaded is final)
    c->n_epochs = h.n_epochs;
    return VGS_OK;
  }
  // (Round 4 measured the scan and the one-thread growth kernel as ONE launch -- the last workgroup through its scan replays the
  // growth: eight launches instead of sixteen, but the double-precision growth code inflates the scan kernel's registers and the
  // scans take twice as long, 240 against 125 us per cloud.  Two kernels it stays.)
  const int pinned = (c->grid_pinned && record_epochs) ? 1 : 0;
  // few enough threads that the scan moves through the cloud front to back (a growth step is found within the first
  // trip or two: the points come in random order), enough to keep the HBM pipes full on the one scan that reads everything
  const int blocks = (int)std::max<int64_t>(8, std::min<int64_t>((c->N / 4 + 255) / 256 + 1, (int64_t)c->K.fv_blocks));
  // the first points in one launch (k_grow_prefix); the first batch of whole-cloud scans behind it is then two pairs, not eight
  const bool prefix = !c->K.no_grow_prefix;
  if (prefix)
    hipLaunchKernelGGL(k_grow_prefix, dim3(1), dim3(1024), 0, c->stream, c->xyz, c->stride_f, std::min<int64_t>(c->N, 8192), d_g, pinned);
  for (int batch = 0; batch < 64; ++batch) {
    // a scene grows its box about log2(extent / voxel) times; pairs queued after the last growth return at once
    for (int k = 0; k < ((prefix && batch == 0) ? 2 : 8); ++k) {
      hipLaunchKernelGGL(k_first_violation, dim3(blocks), dim3(256), 0, c->stream, c->xyz, c->stride_f, c->N, d_g);
      hipLaunchKernelGGL(k_adopt, dim3(1), dim3(1), 0, c->stream, c->xyz, c->stride_f, d_g, pinned);
    }
    VGS_READBACK(c, &h, d_g, offsetof(GrowState, epochs));   // the epochs stay on the device
    if (h.done) break;
  }
  if (h.done == 2) { c->err = "a point lies outside the pinned grid"; return VGS_E_ARG; }
  if (h.overflow || !h.done) { c->err = "octree grew more than VGS_MAX_EPOCHS times"; return VGS_E_UNSUPPORTED; }
  for (int a = 0; a < 3; ++a) { box.min[a] = h.min[a]; box.max[a] = h.max[a]; box.shift[a] = h.shift[a]; }
  box.depth = h.depth; box.defined = h.defined != 0;
  if (record_epochs) c->n_epochs = h.n_epochs;
  return VGS_OK;
}

static vgs_status grow_box(vgs_ctx* c) {
  if (!c->grid_pinned) {
    c->box = OctreeBox();
    c->box.res = (double)c->P.voxel_size;
  }
  return vgs_grow_box_from(c, c->box, true);
}

// codes -> stable descending sort -> run heads -> voxel table (KeyT as k_make_codes)
template <typename KeyT>
static vgs_status voxelize_sorted_table(vgs_ctx* c) {
  const int64_t N = c->N;
  const int TB = 256;
  const unsigned nb = (unsigned)((N + TB - 1) / TB);
  KeyT* code_a = (KeyT*)c->code_a.p;   // the 64-bit buffers hold either key width
  KeyT* code_b = (KeyT*)c->code_b.p;
  // 64-bit keys with room below the code (34 key bits + 24 index bits at 10 M points): index packed into the key, keys-only sort
  const unsigned key_bits = (unsigned)(c->code_bits + 1);
  int idx_bits = 1;
  while (idx_bits < 32 && ((int64_t)1 << idx_bits) < N) ++idx_bits;
  const int pack_shift = (sizeof(KeyT) == 8 && (int)key_bits + idx_bits <= 64 && !c->K.no_packed_sort) ? idx_bits : 0;
  hipLaunchKernelGGL((k_make_codes<KeyT>), dim3(nb), dim3(TB), 0, c->stream, c->xyz, c->stride_f, N, (const GrowState*)c->grow_state.p, c->box.res,
                     c->code_bits, code_a, c->perm_a.p, pack_shift);

  // stable LSD radix sort, descending code (= PCL LeafNodeIterator order: children visited 7 -> 0),
  // ascending point index inside a leaf (stability)
  // The library's passes take 8 bits each; 9 bits per pass (512 bins: measured 0.44 against 0.48-0.53 ms for 10 M keys of 34 bits)
  // are used when they save a pass -- 33 to 36 key bits: four passes instead of five; 25 to 27: three instead of four.
  using nine_bits = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                               rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 6>, rocprim::kernel_config<1024, 6>, 9,
                                                                                   rocprim::block_radix_rank_algorithm::match>>;
  const bool use9 = (key_bits + 8) / 9 < (key_bits + 7) / 8;
  auto sort_pairs = [&](void* tmp, size_t& bytes) -> hipError_t {
    if (pack_shift > 0)
      return use9 ? rocprim::radix_sort_keys_desc<nine_bits>(tmp, bytes, code_a, code_b, (size_t)N, (unsigned)pack_shift, (unsigned)pack_shift + key_bits, c->stream)
                  : rocprim::radix_sort_keys_desc(tmp, bytes, code_a, code_b, (size_t)N, (unsigned)pack_shift, (unsigned)pack_shift + key_bits, c->stream);
    return use9 ? rocprim::radix_sort_pairs_desc<nine_bits>(tmp, bytes, code_a, code_b, c->perm_a.p, c->perm_b.p, (size_t)N, 0, key_bits, c->stream)
                : rocprim::radix_sort_pairs_desc(tmp, bytes, code_a, code_b, c->perm_a.p, c->perm_b.p, (size_t)N, 0, key_bits, c->stream);
  };
  size_t tmp_bytes = 0;
  VGS_HIP_TRY(c, sort_pairs(nullptr, tmp_bytes));
  size_t scan_bytes = 0;
  VGS_HIP_TRY(c, rocprim::inclusive_scan(nullptr, scan_bytes, c->head_flag.p, c->perm_a.p, (size_t)N, rocprim::plus<uint32_t>(), c->stream));
  VGS_HIP_TRY(c, c->sort_tmp.ensure(std::max(tmp_bytes, scan_bytes)));
  VGS_HIP_TRY(c, sort_pairs(c->sort_tmp.p, tmp_bytes));
  // sorted: code_b, perm_b
  unsigned long long* d_cnt = (unsigned long long*)c->counters.p;
  VGS_HIP_TRY(c, hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL((k_heads<KeyT>), dim3(nb), dim3(TB), 0, c->stream, code_b, N, c->head_flag.p, pack_shift, c->perm_b.p);
  hipLaunchKernelGGL((k_count_valid<KeyT>), dim3(1), dim3(1), 0, c->stream, code_b, N, d_cnt, pack_shift);
  uint32_t* scan = c->perm_a.p;  // perm_a is free after the sort
  VGS_HIP_TRY(c, rocprim::inclusive_scan(c->sort_tmp.p, scan_bytes, c->head_flag.p, scan, (size_t)N, rocprim::plus<uint32_t>(), c->stream));
  hipLaunchKernelGGL(k_copy_last, dim3(1), dim3(1), 0, c->stream, scan, N, d_cnt + 1);   // number of voxels next to the number of finite points
  unsigned long long h2[2] = {0, 0};
  if (vgs_can_split_readback(c)) {
    // the leaf-order gather of the points needs the sorted order only, not the counts: it runs while the host fetches them.  All N
    // positions are gathered (the non-finite points sit at the tail of the order; their slots are never read).
    { vgs_status sb = vgs_readback_begin(c, d_cnt, sizeof(h2)); if (sb != VGS_OK) return sb; }
    VGS_HIP_TRY(c, c->xs.ensure(N + 1)); VGS_HIP_TRY(c, c->ys.ensure(N + 1)); VGS_HIP_TRY(c, c->zs.ensure(N + 1));
    hipLaunchKernelGGL(k_gather_points, dim3(nb), dim3(TB), 0, c->stream, c->xyz, c->stride_f, c->perm_b.p, N, c->xs.p, c->ys.p, c->zs.p);
    c->gathered = true;
    { vgs_status se = vgs_readback_end(c, h2, sizeof(h2)); if (se != VGS_OK) return se; }
  } else {
    VGS_READBACK(c, h2, d_cnt, sizeof(h2));
  }
  const unsigned long long nf = h2[0];
  const uint32_t v_total = (uint32_t)h2[1];
  c->Nf = (int64_t)nf;
  c->V = (int64_t)v_total;
  VGS_HIP_TRY(c, c->vox_code.ensure(c->V + 1)); VGS_HIP_TRY(c, c->vox_start.ensure(c->V + 1));
  const uint64_t mask = (c->code_bits >= 64) ? ~0ull : ((1ull << c->code_bits) - 1ull);
  hipLaunchKernelGGL((k_voxel_table<KeyT>), dim3(nb), dim3(TB), 0, c->stream, code_b, c->head_flag.p, scan, N, mask, c->pt_vox.p,
                     c->vox_code.p, c->vox_start.p, pack_shift);
  hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, c->stream, c->vox_start.p + c->V, (uint32_t)c->Nf);
  return VGS_OK;
}

vgs_status vgs_stage_voxelize(vgs_ctx* c) {
  const int64_t N = c->N;
  c->V = 0; c->Nf = 0; c->U = 0;
  vgs_status st = grow_box(c);
  if (st != VGS_OK) return st;
  if (c->n_epochs == 0) {  // no finite point at all
    c->counts[VGS_N_FINITE] = 0; c->counts[VGS_N_VOXELS] = 0; c->counts[VGS_N_DEPTH] = 0;
    return VGS_OK;
  }
  if (c->box.depth > 21) { c->err = "octree depth > 21 (64-bit voxel codes exhausted)"; return VGS_E_UNSUPPORTED; }
  c->code_bits = 3 * c->box.depth;

  VGS_HIP_TRY(c, c->code_a.ensure(N)); VGS_HIP_TRY(c, c->code_b.ensure(N));
  VGS_HIP_TRY(c, c->perm_a.ensure(N)); VGS_HIP_TRY(c, c->perm_b.ensure(N));
  VGS_HIP_TRY(c, c->head_flag.ensure(N)); VGS_HIP_TRY(c, c->pt_vox.ensure(N));
  const int TB = 256;
  c->gathered = false;
  st = (c->code_bits + 1 <= 32) ? voxelize_sorted_table<uint32_t>(c) : voxelize_sorted_table<uint64_t>(c);
  if (st != VGS_OK) return st;
  if (!c->gathered) { VGS_HIP_TRY(c, c->xs.ensure(c->Nf + 1)); VGS_HIP_TRY(c, c->ys.ensure(c->Nf + 1)); VGS_HIP_TRY(c, c->zs.ensure(c->Nf + 1)); }
  if (c->Nf > 0 && !c->gathered) {
    const unsigned nbf = (unsigned)((c->Nf + TB - 1) / TB);
    hipLaunchKernelGGL(k_gather_points, dim3(nbf), dim3(TB), 0, c->stream, c->xyz, c->stride_f, c->perm_b.p, c->Nf, c->xs.p, c->ys.p,
                       c->zs.p);
  }
  VGS_HIP_TRY(c, hipGetLastError());
  c->counts[VGS_N_FINITE] = c->Nf;
  c->counts[VGS_N_VOXELS] = c->V;
  c->counts[VGS_N_DEPTH] = c->box.depth;
  return VGS_OK;
}
