// tools/regsort_test.hip -- csrc/regsort.hpp against the LDS bitonic network of the local cut, outside the kernel: same lists,
// same LDS footprint and register budget as the bulk class (one wavefront per workgroup, 6368 bytes of LDS, 6 wavefronts per
// SIMD), results compared key for key, both timed.   build: hipcc --offload-arch=gfx950 -O3 -o regsort_test regsort_test.hip
// usage: regsort_test [lists] [mean length]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../vgs-svgs-segmentation_amd/csrc/regsort.hpp"

constexpr int LCAP = 448;
constexpr int LDS_BYTES = 6368;

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6)))
void k_sort(const uint64_t* keys, const int* offs, int n_lists, uint64_t* out) {
  __shared__ uint64_t lds[LDS_BYTES / 8];
  uint64_t* lk = lds;
  const int lane = (int)threadIdx.x;
  for (int li = (int)blockIdx.x; li < n_lists; li += (int)gridDim.x) {
    const int o = offs[li], cnt = offs[li + 1] - o;
    for (int e = lane; e < cnt; e += 64) lk[e] = keys[o + e];
    wave_sync();
    if constexpr (MODE == 0) {
      int np = 64;
      while (np < cnt) np <<= 1;
      auto cmpx = [&](int lo, int hi) {
        if (hi >= cnt) return;
        const uint64_t x = lk[lo], y = lk[hi];
        if (x < y) { lk[lo] = y; lk[hi] = x; }
      };
      for (int size = 2, sbit = 1; size <= np; size <<= 1, ++sbit) {
        for (int t = lane; t < (np >> 1); t += 64) {
          const int blk = t >> (sbit - 1), i = t & ((size >> 1) - 1);
          cmpx((blk << sbit) + i, (blk << sbit) + size - 1 - i);
        }
        wave_sync();
        for (int sl = sbit - 2; sl >= 0; --sl) {
          const int strd = 1 << sl;
          for (int t = lane; t < (np >> 1); t += 64) {
            const int lo = ((t >> sl) << (sl + 1)) | (t & (strd - 1));
            cmpx(lo, lo + strd);
          }
          wave_sync();
        }
      }
    } else {
      if (cnt <= 64) regsort::sort_desc<1>(lk, cnt, lane);
      else if (cnt <= 128) regsort::sort_desc<2>(lk, cnt, lane);
      else if (cnt <= 256) regsort::sort_desc<4>(lk, cnt, lane);
      else if constexpr (MODE == 1) regsort::sort_desc<8>(lk, cnt, lane);
      else regsort::sort_desc_two_halves<4>(lk, cnt, lane);
      wave_sync();
    }
    for (int e = lane; e < cnt; e += 64) out[o + e] = lk[e];
    wave_sync();
  }
}

// the one-word form (round 6): weights in (0.7, 1] as phase A of the local cut holds them -- except where the input says otherwise
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6)))
void k_sort32(const uint64_t* keys, const int* offs, int n_lists, uint64_t* out, uint32_t wbase) {
  __shared__ uint64_t lds[LDS_BYTES / 8];
  uint64_t* lk = lds;
  const int lane = (int)threadIdx.x;
  for (int li = (int)blockIdx.x; li < n_lists; li += (int)gridDim.x) {
    const int o = offs[li], cnt = offs[li + 1] - o;
    for (int e = lane; e < cnt; e += 64) lk[e] = keys[o + e];
    wave_sync();
    {
      bool sorted;
      if (cnt <= 64) sorted = regsort::sort_desc32<1>(lk, cnt, lane, wbase);
      else if (cnt <= 128) sorted = regsort::sort_desc32<2>(lk, cnt, lane, wbase);
      else if (cnt <= 256) sorted = regsort::sort_desc32<4>(lk, cnt, lane, wbase);
      else sorted = regsort::sort_desc32<8>(lk, cnt, lane, wbase);
      if (!sorted) {   // a weight outside the window: the 64-bit network, as the local cut does
        if (cnt <= 64) regsort::sort_desc<1>(lk, cnt, lane);
        else if (cnt <= 128) regsort::sort_desc<2>(lk, cnt, lane);
        else if (cnt <= 256) regsort::sort_desc<4>(lk, cnt, lane);
        else regsort::sort_desc_two_halves<4>(lk, cnt, lane);
      }
      wave_sync();
    }
    for (int e = lane; e < cnt; e += 64) out[o + e] = lk[e];
    wave_sync();
  }
}

// the workgroup form (multi-wavefront classes, dense hand-over kernels): four wavefronts, lists of up to 2048 keys
__global__ __launch_bounds__(256) void k_sort_block(const uint64_t* keys, const int* offs, int n_lists, int scale, uint64_t* out) {
  __shared__ uint64_t lk[2048];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int li = (int)blockIdx.x; li < n_lists; li += (int)gridDim.x) {
    // list li of the block test = `scale` consecutive lists of the input glued together (up to 2048 keys)
    const int first = li * scale, last = first + scale < n_lists * scale ? first + scale : n_lists * scale;
    const int o = offs[first];
    int cnt = offs[last] - o;
    cnt = cnt < 2048 ? cnt : 2048;
    __syncthreads();
    for (int e = tid; e < cnt; e += 256) lk[e] = keys[o + e];
    regsort::sort_desc_block<4>(lk, cnt, wave, lane, [&]() { __syncthreads(); });
    for (int e = tid; e < cnt; e += 256) out[o + e] = lk[e];
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n_lists = argc > 1 ? std::atoi(argv[1]) : 400000;
  const int mean = argc > 2 ? std::atoi(argv[2]) : 200;
  std::mt19937_64 rng(7);
  std::vector<int> offs(n_lists + 1, 0);
  for (int i = 0; i < n_lists; ++i) {
    int c = (int)(rng() % (unsigned)(2 * mean + 1));
    if (i < 16) c = i == 0 ? 0 : (i == 1 ? 1 : (i == 2 ? 64 : (i == 3 ? 65 : (i == 4 ? 128 : (i == 5 ? 129 : (i == 6 ? 256 : (i == 7 ? 257 : (i == 8 ? 448 : 447))))))));
    c = std::min(c, LCAP);
    offs[i + 1] = offs[i] + c;
  }
  std::vector<uint64_t> keys((size_t)offs[n_lists]);
  for (size_t i = 0; i < keys.size(); ++i) {
    const uint64_t r = rng();
    // weight bits of a float in (0.3, 1], a 16-bit pair id; 3 % dropped entries (0); a few equal weights
    const float w = 0.3f + 0.7f * (float)((r >> 11) % 1000003) / 1000003.0f;
    uint32_t wb; memcpy(&wb, &w, 4);
    if ((r & 63) == 1) wb = 0x3f000000u;
    keys[i] = (r & 31) == 0 ? 0ull : (((uint64_t)wb << 32) | (uint32_t)(0xffffu - ((r >> 40) & 0x3fffu)));
  }
  uint64_t *d_keys, *d_out[3]; int* d_offs;
  CK(hipMalloc(&d_keys, keys.size() * 8)); CK(hipMalloc(&d_out[0], keys.size() * 8)); CK(hipMalloc(&d_out[1], keys.size() * 8)); CK(hipMalloc(&d_out[2], keys.size() * 8));
  CK(hipMalloc(&d_offs, offs.size() * 4));
  CK(hipMemcpy(d_keys, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_offs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 24 * 4;
  float ms[3] = {0, 0, 0};
  for (int rep = 0; rep < 4; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k_sort<0>, dim3(grid), dim3(64), 0, 0, d_keys, d_offs, n_lists, d_out[0]);
      else if (mode == 1) hipLaunchKernelGGL(k_sort<1>, dim3(grid), dim3(64), 0, 0, d_keys, d_offs, n_lists, d_out[1]);
      else hipLaunchKernelGGL(k_sort<2>, dim3(grid), dim3(64), 0, 0, d_keys, d_offs, n_lists, d_out[2]);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (rep > 0) ms[mode] += t / 3.0f;
    }
  std::vector<uint64_t> a(keys.size()), b(keys.size()), c(keys.size());
  CK(hipMemcpy(a.data(), d_out[0], keys.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d_out[1], keys.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c.data(), d_out[2], keys.size() * 8, hipMemcpyDeviceToHost));
  size_t bad_ref = 0, bad = 0, bad2 = 0;
  for (int i = 0; i < n_lists; ++i) {
    std::vector<uint64_t> ref(keys.begin() + offs[i], keys.begin() + offs[i + 1]);
    std::sort(ref.begin(), ref.end(), [](uint64_t x, uint64_t y) { return x > y; });
    for (int e = 0; e < (int)ref.size(); ++e) { bad_ref += a[offs[i] + e] != ref[e]; bad += b[offs[i] + e] != ref[e]; bad2 += c[offs[i] + e] != ref[e]; }
  }
  {   // block form: glue `scale` lists into one of up to 2048 keys
    const int scale = std::max(1, 1400 / std::max(1, mean)), nb = n_lists / scale;
    hipLaunchKernelGGL(k_sort_block, dim3(std::min(nb, 256 * 5)), dim3(256), 0, 0, d_keys, d_offs, nb, scale, d_out[2]);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(c.data(), d_out[2], keys.size() * 8, hipMemcpyDeviceToHost));
    size_t bad3 = 0;
    for (int li = 0; li < nb; ++li) {
      const int o = offs[li * scale];
      const int cnt = std::min(offs[li * scale + scale] - o, 2048);
      std::vector<uint64_t> ref(keys.begin() + o, keys.begin() + o + cnt);
      std::sort(ref.begin(), ref.end(), [](uint64_t x, uint64_t y) { return x > y; });
      for (int e = 0; e < cnt; ++e) bad3 += c[o + e] != ref[e];
    }
    std::printf("block form, %d lists of up to 2048 keys (wrong keys %zu)\n", nb, bad3);
    if (bad3) return 1;
  }
  {   // one-word keys: weights in (0.7, 1] quantised so that equal weights are common (runs of two to six keys), 3 % dropped entries,
      // one list in sixteen holds a weight below the window (the whole list then takes the 64-bit network)
    std::vector<uint64_t> k32(keys.size());
    std::mt19937_64 r2(11);
    for (int i = 0; i < n_lists; ++i)
      for (int e = offs[i]; e < offs[i + 1]; ++e) {
        const uint64_t r = r2();
        float w = 0.7f + 0.3f * (float)(1 + (r >> 11) % ((i & 3) == 0 ? 97u : 1000003u)) / ((i & 3) == 0 ? 97.0f : 1000003.0f);
        if ((i & 15) == 5 && (r & 7) == 0) w = 0.4f;
        uint32_t wb; memcpy(&wb, &w, 4);
        k32[e] = (r & 31) == 0 ? 0ull : (((uint64_t)wb << 32) | (uint32_t)(0xffffu - ((r >> 40) & 0x3fffu)));
      }
    // (unique keys, as the local cut's are: a pair id appears once per list)
    for (int i = 0; i < n_lists; ++i) {
      std::sort(k32.begin() + offs[i], k32.begin() + offs[i + 1]);
      for (int e = offs[i] + 1; e < offs[i + 1]; ++e) if (k32[e] != 0 && k32[e] == k32[e - 1]) k32[e - 1] = 0;
      std::shuffle(k32.begin() + offs[i], k32.begin() + offs[i + 1], r2);
    }
    const float thr0 = 0.7f; uint32_t tb; memcpy(&tb, &thr0, 4);
    CK(hipMemcpy(d_keys, k32.data(), k32.size() * 8, hipMemcpyHostToDevice));
    float ms32 = 0, ms64 = 0;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_sort32, dim3(grid), dim3(64), 0, 0, d_keys, d_offs, n_lists, d_out[0], tb + 1u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); if (rep > 0) ms32 += t / 3.0f;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_sort<2>, dim3(grid), dim3(64), 0, 0, d_keys, d_offs, n_lists, d_out[1]);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t, e0, e1)); if (rep > 0) ms64 += t / 3.0f;
    }
    CK(hipMemcpy(a.data(), d_out[0], keys.size() * 8, hipMemcpyDeviceToHost));
    size_t bad32 = 0, ties = 0;
    for (int i = 0; i < n_lists; ++i) {
      std::vector<uint64_t> ref(k32.begin() + offs[i], k32.begin() + offs[i + 1]);
      std::sort(ref.begin(), ref.end(), [](uint64_t x, uint64_t y) { return x > y; });
      for (int e = 0; e < (int)ref.size(); ++e) { bad32 += a[offs[i] + e] != ref[e]; ties += e > 0 && ref[e] != 0 && (ref[e] >> 32) == (ref[e - 1] >> 32); }
    }
    std::printf("one-word keys %.3f ms against %.3f ms for the 64-bit network on the same lists, %zu keys tie with their neighbour (wrong keys %zu)\n", ms32, ms64, ties, bad32);
    if (bad32) return 1;
    CK(hipMemcpy(d_keys, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
  }
  std::printf("lists %d  keys %zu  lds network %.3f ms (wrong keys %zu)  register network %.3f ms (wrong keys %zu)  with two halves above 256 keys %.3f ms (wrong keys %zu)\n", n_lists, keys.size(), ms[0], bad_ref, ms[1], bad, ms[2], bad2);
  return (bad_ref || bad || bad2) ? 1 : 0;
}
