// tools/vcc_rate.hip -- does a VALU instruction that reads its lane mask from VCC issue slower than one that reads it from another
// SGPR pair?  (tools/valu_rate.hip showed 9.6 ns per v_cndmask_b32 with vcc against 2.0 ns with s[22:23].)
// build: hipcc -O2 --offload-arch=gfx950 vcc_rate.hip -o vcc_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 4096;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
  const float x = seed * 0.5f + 1.0f;
  asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[22:23], 0x3333" ::: "vcc", "s22", "s23");
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (MODE == 0) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
      else if constexpr (MODE == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
      else if constexpr (MODE == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a[i]) : "v"(x));
      else if constexpr (MODE == 3) { asm volatile("v_cmp_gt_f32_e32 vcc, %1, %0\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc"); }
      else if constexpr (MODE == 4) { asm volatile("v_cmp_gt_f32_e64 s[22:23], %1, %0\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a[i]) : "v"(x) : "s22", "s23"); }
      else if constexpr (MODE == 5) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      else if constexpr (MODE == 6) { if (i < 8) asm volatile("v_cmp_gt_f32_e32 vcc, %2, %0\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(a[2 * i]), "+v"(a[2 * i + 1]) : "v"(x) : "vcc"); }
      else if constexpr (MODE == 7) { if (i < 4) asm volatile("v_cmp_gt_f32_e32 vcc, %4, %0\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc" : "+v"(a[4 * i]), "+v"(a[4 * i + 1]), "+v"(a[4 * i + 2]), "+v"(a[4 * i + 3]) : "v"(x) : "vcc"); }
      else if constexpr (MODE == 8) { if (i < 8) asm volatile("v_cndmask_b32_e32 %0, %0, %2, vcc\n v_add_f32_e32 %1, %2, %1" : "+v"(a[2 * i]), "+v"(a[2 * i + 1]) : "v"(x)); }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 123.456f) out[0] = s;
}
int main() {
  float* d; CK(hipMalloc(&d, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* nm[9] = {"v_cndmask_b32_e32 (vcc)", "v_cndmask_b32_e64 vcc", "v_cndmask_b32_e64 s[22:23]", "v_cmp_e32 vcc + v_cndmask_e32 vcc (pairs)", "v_cmp_e64 s[22:23] + v_cndmask_e64 (pairs)", "v_add_f32", "v_cmp + 2 x v_cndmask_e32 (8 triples: 24 instr.)", "v_cmp + 4 x v_cndmask_e32 (4 groups: 20 instr.)", "v_cndmask_e32, v_add_f32 alternating (16 instr.)"};
  for (int mode = 0; mode < 9; ++mode) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      const dim3 g(256 * 4), b(256);
      if (mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, d, 1.0f); else if (mode == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, d, 1.0f);
      else if (mode == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, d, 1.0f); else if (mode == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, d, 1.0f);
      else if (mode == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, d, 1.0f); else if (mode == 5) hipLaunchKernelGGL(k<5>, g, b, 0, 0, d, 1.0f);
      else if (mode == 6) hipLaunchKernelGGL(k<6>, g, b, 0, 0, d, 1.0f); else if (mode == 7) hipLaunchKernelGGL(k<7>, g, b, 0, 0, d, 1.0f); else hipLaunchKernelGGL(k<8>, g, b, 0, 0, d, 1.0f);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    // 4 wavefronts per SIMD: ITER * 16 instructions (pairs: 32) per wavefront
    const double per = (double)ms * 1e6 / ((double)ITER * 16 * 4);
    printf("%-52s %.3f ms  %.2f ns per sixteenth of a trip per SIMD\n", nm[mode], ms, per);
  }
  return 0;
}
