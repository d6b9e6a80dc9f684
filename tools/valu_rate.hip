// tools/valu_rate.hip -- issue-rate microbenchmark for gfx950 (MI355X): how many cycles does one SIMD need per wave64
// instruction?  Settles the roof bench.py prices the VALU-bound local-cut kernel against (VERDICT round 2, item 3:
// "16-lane SIMDs, 4 cycles" vs the guide's "SIMD-32, 2 cycles").
//
// Method: workgroups of 256 threads (one wavefront per SIMD of a CU), W workgroups per CU; every lane runs ITER trips of
// 16 independent chains of one instruction (inline asm, nothing for the compiler to fold).  Shader cycles come from
// s_memtime (clock64) around the loop, the clock rate from wall_clock64 (100 MHz constant counter) and from HIP events.
// cycles per instruction per SIMD = loop cycles x SIMDs busy / wave instructions issued.
//
// Build: hipcc -O2 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate ; run on the GPU box: tools/valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 2048;
constexpr int CHAINS = 16;

enum Op { FMA_F32 = 0, ADD_F32, MUL_F32, ADD_U32, MULLO_U32, MAD_U32_U24, PK_FMA_F32, RCP_F32, SQRT_F32, EXP_F32, CNDMASK, FMA_F64, FMA_SALU_MIX, LDS_READ, BPERMUTE, CMP_U64, CMP_U32, MOV_DPP, CNDMASK_SGPR, PERMLANE32_SWAP, MAX_U32, CMP_U32_DPP, CNDMASK_DPP, BCNT, MBCNT, N_OPS };
static const char* OP_NAME[N_OPS] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_add_u32", "v_mul_lo_u32", "v_mad_u32_u24", "v_pk_fma_f32", "v_rcp_f32",
                                     "v_sqrt_f32", "v_exp_f32", "v_cndmask_b32", "v_fma_f64", "v_fma_f32 + s_add_u32 (1:1)", "ds_read_b32", "ds_bpermute_b32",
                                     "v_cmp_gt_u64 (to sgpr pair)", "v_cmp_gt_u32 (to sgpr pair)", "v_mov_b32_dpp quad_perm", "v_cndmask_b32 (sgpr mask)", "v_permlane32_swap_b32", "v_max_u32", "v_max_u32_dpp", "v_cndmask_b32_dpp", "v_bcnt_u32_b32", "v_mbcnt_lo_u32_b32"};

template <int OP>
__global__ __launch_bounds__(1024) void k_rate(float* out, unsigned long long* cyc, unsigned long long* wall, float seed, unsigned long long* rec) {
  __shared__ float lds[1024];
  float a[CHAINS];
  double d[CHAINS / 2];
  float2 p2[CHAINS / 2];
  for (int i = 0; i < CHAINS; ++i) a[i] = seed + (float)(threadIdx.x + i);
  for (int i = 0; i < CHAINS / 2; ++i) { d[i] = (double)a[i]; p2[i] = make_float2(a[i], a[i + 1]); }
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  const float x = seed * 0.5f + 1.0f, y = seed * 0.25f;
  const unsigned int xu = __float_as_uint(x) | 1u;
  unsigned int su = (unsigned int)blockIdx.x;
  const double xd = (double)x, yd = (double)y;
  const float2 x2 = make_float2(x, x), y2 = make_float2(y, y);
  const unsigned int laddr = (threadIdx.x * 4u) & 4095u;
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(x), "v"(y));
      else if constexpr (OP == ADD_F32) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      else if constexpr (OP == MUL_F32) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      else if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(xu));
      else if constexpr (OP == MULLO_U32) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(xu));
      else if constexpr (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %1, %0, %1" : "+v"(a[i]) : "v"(xu));
      else if constexpr (OP == PK_FMA_F32) { if (i < CHAINS / 2) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p2[i]) : "v"(x2), "v"(y2)); }
      else if constexpr (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      else if constexpr (OP == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
      else if constexpr (OP == EXP_F32) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      else if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : );
      else if constexpr (OP == CMP_U64) asm volatile("v_cmp_gt_u64 s[20:21], %0, %1" : : "v"(d[i & 7]), "v"(xd) : "s20", "s21");
      else if constexpr (OP == CMP_U32) asm volatile("v_cmp_gt_u32 s[20:21], %0, %1" : : "v"(a[i]), "v"(xu) : "s20", "s21");
      else if constexpr (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      else if constexpr (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, s[22:23]" : "+v"(a[i]) : "v"(x) : );
      else if constexpr (OP == PERMLANE32_SWAP) { if (i < CHAINS / 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[2 * i]), "+v"(a[2 * i + 1])); }
      else if constexpr (OP == MAX_U32) asm volatile("v_max_u32 %0, %1, %0" : "+v"(a[i]) : "v"(xu));
      else if constexpr (OP == CMP_U32_DPP) asm volatile("v_max_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(xu));
      else if constexpr (OP == CNDMASK_DPP) asm volatile("v_cndmask_b32_dpp %0, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x) : );
      else if constexpr (OP == BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(xu));
      else if constexpr (OP == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(xu));
      else if constexpr (OP == FMA_F64) { if (i < CHAINS / 2) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(d[i]) : "v"(xd), "v"(yd)); }
      else if constexpr (OP == FMA_SALU_MIX) {
        asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(su) : : "scc");
      } else if constexpr (OP == LDS_READ) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(laddr) : "memory");
      else if constexpr (OP == BPERMUTE) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(laddr) : "memory");
    }
    if constexpr (OP == LDS_READ || OP == BPERMUTE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = clock64();
  const unsigned long long w1 = wall_clock64();
  float s = (float)su;
  for (int i = 0; i < CHAINS; ++i) s += a[i];
  for (int i = 0; i < CHAINS / 2; ++i) s += (float)d[i] + p2[i].x + p2[i].y;
  if (s == 123.456f) out[0] = s;   // never true in practice: keeps the chains alive
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
  if ((threadIdx.x & 63) == 0 && rec) {
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* r = rec + 4 * ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    r[0] = hw; r[1] = xcc; r[2] = w0; r[3] = w1;
  }
}

template <int OP>
static void run(int cus, int wg_per_cu, int threads, float* d_out, unsigned long long* d_cyc, unsigned long long* d_wall, unsigned long long* d_rec) {
  const int grid = cus * wg_per_cu;
  const int wpw = threads / 64;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_rate<OP><<<grid, threads>>>(d_out, d_cyc, d_wall, 1.0f, nullptr);   // warm-up
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_rate<OP><<<grid, threads>>>(d_out, d_cyc, d_wall, 1.0f, d_rec);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> rec((size_t)grid * wpw * 4);
  CK(hipMemcpy(rec.data(), d_rec, rec.size() * 8, hipMemcpyDeviceToHost));
  // placement: waves per (xcc, se, cu, simd) and how many of them ran at the same time
  std::map<unsigned int, std::vector<std::pair<unsigned long long, unsigned long long>>> per_simd;
  std::map<unsigned int, int> per_cu;
  unsigned long long first = ~0ull, last = 0;
  double loop = 0;
  for (size_t w = 0; w < (size_t)grid * wpw; ++w) {
    const unsigned int hw = (unsigned int)rec[4 * w], xcc = (unsigned int)rec[4 * w + 1] & 0xf;
    const unsigned int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned int cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    per_simd[(cu_key << 2) | simd].push_back({rec[4 * w + 2], rec[4 * w + 3]});
    per_cu[cu_key]++;
    first = std::min(first, rec[4 * w + 2]); last = std::max(last, rec[4 * w + 3]);
    loop += (double)(rec[4 * w + 3] - rec[4 * w + 2]);
  }
  loop /= (double)grid * wpw;
  size_t min_w = ~(size_t)0, max_w = 0; int max_conc = 0;
  for (auto& kv : per_simd) {
    min_w = std::min(min_w, kv.second.size()); max_w = std::max(max_w, kv.second.size());
    for (auto& a : kv.second) { int cnt = 0; for (auto& b : kv.second) if (b.first <= a.first && a.first < b.second) ++cnt; max_conc = std::max(max_conc, cnt); }
  }
  const int per_trip = (OP == PK_FMA_F32 || OP == FMA_F64 || OP == PERMLANE32_SWAP) ? CHAINS / 2 : CHAINS;
  const double insts_total = (double)ITER * per_trip * grid * wpw;       // wave instructions of the launch
  const double span_us = (double)(last - first) / 100.0;                  // first loop entry to last loop exit (100 MHz counter)
  const double simds = (double)per_simd.size();
  printf("%-28s wg %4d x %4d thr: CUs used %3zu SIMDs used %4zu waves/SIMD %zu..%zu max concurrent/SIMD %d | loop %.1f us span %.1f us event %.3f ms"
         " | %.3f ns per wave-instr per SIMD = %.2f cycles @2.4GHz\n",
         OP_NAME[OP], grid, threads, per_cu.size(), per_simd.size(), min_w, max_w, max_conc, loop / 100.0, span_us, ms,
         span_us * 1e3 * simds / insts_total, span_us * 1e3 * simds / insts_total * 2.4);
  fflush(stdout);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int OP>
static void sweep(int cus, float* d_out, unsigned long long* d_cyc, unsigned long long* d_wall, unsigned long long* d_rec) {
  run<OP>(cus, 1, 256, d_out, d_cyc, d_wall, d_rec);
  run<OP>(cus, 2, 256, d_out, d_cyc, d_wall, d_rec);
  run<OP>(cus, 4, 256, d_out, d_cyc, d_wall, d_rec);
  run<OP>(cus, 8, 256, d_out, d_cyc, d_wall, d_rec);
  run<OP>(cus, 1, 1024, d_out, d_cyc, d_wall, d_rec);
  run<OP>(cus, 2, 1024, d_out, d_cyc, d_wall, d_rec);
  run<OP>(cus * 8, 1, 64, d_out, d_cyc, d_wall, d_rec);
  run<OP>(cus * 8, 4, 64, d_out, d_cyc, d_wall, d_rec);
}

template <int OP>
static void quick(int cus, float* d_out, unsigned long long* d_cyc, unsigned long long* d_wall, unsigned long long* d_rec) {
  run<OP>(cus, 4, 256, d_out, d_cyc, d_wall, d_rec);
}

int main(int argc, char** argv) {
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s  CUs %d  clockRate %d kHz  wavefront %d\n", pr.name, pr.multiProcessorCount, pr.clockRate, pr.warpSize);
  const int cus = pr.multiProcessorCount;
  float* d_out; unsigned long long *d_cyc, *d_wall, *d_rec;
  const size_t max_wg = (size_t)cus * 32;
  CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_cyc, max_wg * 8)); CK(hipMalloc(&d_wall, max_wg * 8)); CK(hipMalloc(&d_rec, max_wg * 16 * 32));
  if (argc > 1) {   // "sort": the instructions of csrc/regsort.hpp, four wavefronts per SIMD
    quick<ADD_U32>(cus, d_out, d_cyc, d_wall, d_rec);
    quick<CMP_U64>(cus, d_out, d_cyc, d_wall, d_rec);
    quick<CMP_U32>(cus, d_out, d_cyc, d_wall, d_rec);
    quick<MOV_DPP>(cus, d_out, d_cyc, d_wall, d_rec);
    quick<CNDMASK>(cus, d_out, d_cyc, d_wall, d_rec);
    quick<CNDMASK_SGPR>(cus, d_out, d_cyc, d_wall, d_rec);
    quick<PERMLANE32_SWAP>(cus, d_out, d_cyc, d_wall, d_rec);
    quick<MAX_U32>(cus, d_out, d_cyc, d_wall, d_rec);
    quick<CMP_U32_DPP>(cus, d_out, d_cyc, d_wall, d_rec);
    quick<CNDMASK_DPP>(cus, d_out, d_cyc, d_wall, d_rec);
    quick<BCNT>(cus, d_out, d_cyc, d_wall, d_rec);
    quick<MBCNT>(cus, d_out, d_cyc, d_wall, d_rec);
    return 0;
  }
  sweep<FMA_F32>(cus, d_out, d_cyc, d_wall, d_rec);
  sweep<ADD_F32>(cus, d_out, d_cyc, d_wall, d_rec);
  sweep<ADD_U32>(cus, d_out, d_cyc, d_wall, d_rec);
  sweep<MULLO_U32>(cus, d_out, d_cyc, d_wall, d_rec);
  sweep<PK_FMA_F32>(cus, d_out, d_cyc, d_wall, d_rec);
  sweep<RCP_F32>(cus, d_out, d_cyc, d_wall, d_rec);
  sweep<SQRT_F32>(cus, d_out, d_cyc, d_wall, d_rec);
  sweep<CNDMASK>(cus, d_out, d_cyc, d_wall, d_rec);
  sweep<FMA_F64>(cus, d_out, d_cyc, d_wall, d_rec);
  sweep<FMA_SALU_MIX>(cus, d_out, d_cyc, d_wall, d_rec);
  sweep<LDS_READ>(cus, d_out, d_cyc, d_wall, d_rec);
  sweep<BPERMUTE>(cus, d_out, d_cyc, d_wall, d_rec);
  return 0;
}
