#!/usr/bin/env python3
"""Writes tools/valu_roof.hip: the VALU issue-rate question of VERDICT r3 item 4, asked with hand-placed registers.

tools/valu_rate.hip (round 3) let the compiler allocate the registers of its inline-asm chains and measured 2.8 cycles per
v_add_f32 and 4.65 per v_fma_f32 per SIMD, against the 2 cycles of MI355X_MICROARCH.md ("Wave scheduling", 157 TF fp32).  Here every
loop is ONE asm block: 64 VALU instructions per trip on registers chosen by this script (VGPR bank = index mod 4), 3 SALU
instructions of loop control per trip, s_memtime outside.  Variants: all three sources in one bank / in three banks, 16 or 32
independent chains, no dependency at all, an SGPR or an inline constant in place of a VGPR source, VOP2 forms (v_fmac, v_add, v_mul),
packed fp32, raised wave priority, 4 chains (latency)."""
import os

ITER = 4096
CH0 = 16          # first chain register (v0..v15 are left to the compiler: arguments, thread ids)


def body(fmt, dests, reps):
    lines = []
    for _ in range(reps):
        for d in dests:
            a = 80 + ((d + 1) % 4)
            b = 80 + ((d + 2) % 4)
            lines.append(fmt.format(d=d, d1=d + 1, a=a, b=b))
    assert len(lines) == 64, len(lines)
    return lines


ev = list(range(CH0, CH0 + 64, 4))        # 16 chains, all in bank 0
VARIANTS = [
    # name, per-trip instruction list, note
    ("fma_same_bank_16", body("v_fma_f32 v{d}, v{d}, v80, v84", ev, 4), "v_fma_f32 d, d, a, b: d, a, b all in VGPR bank 0; 16 chains"),
    ("fma_3banks_16", body("v_fma_f32 v{d}, v{d}, v81, v82", ev, 4), "v_fma_f32 d, d, a, b: banks 0, 1, 2; 16 chains"),
    ("fma_3banks_32", body("v_fma_f32 v{d}, v{d}, v{a}, v{b}", list(range(CH0, CH0 + 32)), 2), "v_fma_f32, sources in three banks, 32 chains"),
    ("fma_nodep", body("v_fma_f32 v{d}, v81, v82, v83", ev, 4), "v_fma_f32 d, a, b, c: no instruction reads what another wrote"),
    ("fma_sgpr", body("v_fma_f32 v{d}, v{d}, s24, v82", ev, 4), "v_fma_f32 d, d, sgpr, b"),
    ("fma_const", body("v_fma_f32 v{d}, v{d}, 0.5, v82", ev, 4), "v_fma_f32 d, d, 0.5 (inline constant), b"),
    ("fmac_vop2", body("v_fmac_f32 v{d}, v81, v82", ev, 4), "v_fmac_f32 d, a, b (VOP2 encoding of d += a * b)"),
    ("add_vop2", body("v_add_f32 v{d}, v81, v{d}", ev, 4), "v_add_f32 d, a, d (VOP2)"),
    ("mul_vop2", body("v_mul_f32 v{d}, v81, v{d}", ev, 4), "v_mul_f32 d, a, d (VOP2)"),
    ("mov", body("v_mov_b32 v{d}, v81", ev, 4), "v_mov_b32 d, a"),
    ("add_u32", body("v_add_u32 v{d}, v81, v{d}", ev, 4), "v_add_u32 d, a, d (VOP2)"),
    ("pk_fma_vgpr", body("v_pk_fma_f32 v[{d}:{d1}], v[{d}:{d1}], v[82:83], v[86:87]", ev, 4), "v_pk_fma_f32 on register pairs, a and b both in banks 2-3"),
    ("pk_fma_sgpr", body("v_pk_fma_f32 v[{d}:{d1}], v[{d}:{d1}], v[82:83], s[24:25]", ev, 4), "v_pk_fma_f32, b an SGPR pair"),
    ("pk_mul", body("v_pk_mul_f32 v[{d}:{d1}], v[{d}:{d1}], v[82:83]", ev, 4), "v_pk_mul_f32 on register pairs"),
    ("fma_3banks_16_prio3", body("v_fma_f32 v{d}, v{d}, v81, v82", ev, 4), "fma_3banks_16 behind s_setprio 3"),
    ("fma_4chains", body("v_fma_f32 v{d}, v{d}, v81, v82", [16, 20, 24, 28], 16), "v_fma_f32, 4 chains: dependent-issue latency"),
    ("fma_1chain", body("v_fma_f32 v{d}, v{d}, v81, v82", [16], 64), "v_fma_f32, ONE chain: latency of a dependent v_fma"),
    # ---- the instruction kinds the local-cut kernels are made of, sources in different banks ----
    ("cmp_vcc", body("v_cmp_lt_f32 vcc, v81, v{d}", ev, 4), "v_cmp_lt_f32 vcc, a, d (VOPC)"),
    ("cmp_sgpr", body("v_cmp_lt_f32 s[26:27], v81, v{d}", ev, 4), "v_cmp_lt_f32 s[26:27], a, d (VOP3 form)"),
    ("cmp_u32_vcc", body("v_cmp_lt_u32 vcc, v81, v{d}", ev, 4), "v_cmp_lt_u32 vcc, a, d"),
    ("cmp_u64_vcc", body("v_cmp_lt_u64 vcc, v[82:83], v[{d}:{d1}]", ev, 4), "v_cmp_lt_u64 vcc, a2, d2 (the edge-key compare)"),
    ("cndmask_vcc", body("v_cndmask_b32 v{d}, v81, v{d}, vcc", ev, 4), "v_cndmask_b32 d, a, d, vcc (VOP2)"),
    ("cndmask_sgpr", body("v_cndmask_b32 v{d}, v81, v{d}, s[26:27]", ev, 4), "v_cndmask_b32 d, a, d, s[26:27] (VOP3)"),
    ("cmp_cndmask", sum((["v_cmp_lt_f32 vcc, v81, v{d}".format(d=d), "v_cndmask_b32 v{d}, v82, v{d}, vcc".format(d=d)] for d in ev * 2), []), "v_cmp_lt_f32 vcc + v_cndmask_b32 .. vcc, alternating (a select)"),
    ("mov_dpp", body("v_mov_b32_dpp v{d}, v81 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", ev, 4), "v_mov_b32_dpp quad_perm"),
    ("mov_dpp_row_shr", body("v_mov_b32_dpp v{d}, v81 row_shr:1 row_mask:0xf bank_mask:0xf", ev, 4), "v_mov_b32_dpp row_shr:1"),
    ("add_dpp", body("v_add_f32_dpp v{d}, v81, v{d} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", ev, 4), "v_add_f32_dpp (an ALU op with a DPP source)"),
    ("permlane32_swap", body("v_permlane32_swap_b32 v{d}, v{d1}", ev, 4), "v_permlane32_swap_b32 d, d+1"),
    ("readlane", body("v_readlane_b32 s28, v{d}, 3", ev, 4), "v_readlane_b32 s28, d, 3"),
    ("readfirstlane", body("v_readfirstlane_b32 s28, v{d}", ev, 4), "v_readfirstlane_b32"),
    ("and_or", body("v_and_or_b32 v{d}, v{d}, v81, v82", ev, 4), "v_and_or_b32 d, d, a, b (VOP3 integer, three banks)"),
    ("lshlrev", body("v_lshlrev_b32 v{d}, 1, v{d}", ev, 4), "v_lshlrev_b32 d, 1, d (VOP2)"),
    ("and_b32", body("v_and_b32 v{d}, v81, v{d}", ev, 4), "v_and_b32 d, a, d (VOP2)"),
    ("bfe_u32", body("v_bfe_u32 v{d}, v{d}, 3, 5", ev, 4), "v_bfe_u32 d, d, 3, 5 (VOP3, inline constants)"),
    ("mul_lo_u32", body("v_mul_lo_u32 v{d}, v{d}, v81", ev, 4), "v_mul_lo_u32 d, d, a"),
    ("mad_u32_u24", body("v_mad_u32_u24 v{d}, v{d}, v81, v82", ev, 4), "v_mad_u32_u24 d, d, a, b (three banks)"),
    ("mbcnt", body("v_mbcnt_lo_u32_b32 v{d}, v81, v{d}", ev, 4), "v_mbcnt_lo_u32_b32 d, a, d"),
    ("bcnt", body("v_bcnt_u32_b32 v{d}, v81, v{d}", ev, 4), "v_bcnt_u32_b32 d, a, d"),
    ("exp_f32", body("v_exp_f32 v{d}, v81", ev, 4), "v_exp_f32 d, a (transcendental)"),
    ("rcp_f32", body("v_rcp_f32 v{d}, v81", ev, 4), "v_rcp_f32 d, a"),
    ("sqrt_f32", body("v_sqrt_f32 v{d}, v81", ev, 4), "v_sqrt_f32 d, a"),
    ("cvt_f32_u32", body("v_cvt_f32_u32 v{d}, v81", ev, 4), "v_cvt_f32_u32 d, a"),
    ("lds_read_b32", body("ds_read_b32 v{d}, v84", ev, 4) + ["s_waitcnt lgkmcnt(0)"], "ds_read_b32 d, addr (lane * 4), one s_waitcnt per 64"),
    ("lds_read_b64", body("ds_read_b64 v[{d}:{d1}], v85", ev, 4) + ["s_waitcnt lgkmcnt(0)"], "ds_read_b64 d2, addr (lane * 8)"),
    ("lds_write_b32", body("ds_write_b32 v84, v{d}", ev, 4) + ["s_waitcnt lgkmcnt(0)"], "ds_write_b32 addr, d"),
    ("lds_bpermute", body("ds_bpermute_b32 v{d}, v84, v{d1}", ev, 4) + ["s_waitcnt lgkmcnt(0)"], "ds_bpermute_b32 d, addr, d+1"),
    ("lds_min_u32", body("ds_min_u32 v84, v{d}", ev, 4) + ["s_waitcnt lgkmcnt(0)"], "ds_min_u32 addr, d (the merge loop's claim)"),
    ("lds_read_dep", ["ds_read_b32 v16, v16", "s_waitcnt lgkmcnt(0)"] * 64, "ds_read_b32 v16, v16 + s_waitcnt, 64 dependent round trips per trip (address 0 holds 0)"),
]

HEAD = r'''// tools/valu_roof.hip -- GENERATED by tools/gen_valu_roof.py (edit that, not this).  See its docstring.
// Build: hipcc -O2 --offload-arch=gfx950 tools/valu_roof.hip -o tools/valu_roof ; GPU box: tools/valu_roof [variant]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%%s: %%s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = %d;
#define CLOBBERS %s

template <int V> __device__ __forceinline__ void loop_body();
'''

KERNEL = r'''
template <int V>
__global__ __launch_bounds__(256) void k_roof(unsigned long long* rec) {
  __shared__ unsigned int lds[1024];   // the LDS variants address bytes [0, 512) of it; zero: the dependent-read chain stays at address 0
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 0u;
  unsigned int hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  __syncthreads();
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = clock64();
  loop_body<V>();
  const unsigned long long t1 = clock64();
  const unsigned long long w1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* r = rec + 6 * ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    r[0] = hw; r[1] = xcc; r[2] = t0; r[3] = t1; r[4] = w0; r[5] = w1;
  }
}

struct Res { double cyc_per_instr_simd, ghz, ns; size_t simds; int wmin, wmax; };

template <int V>
static Res run(int cus, int wg_per_cu, unsigned long long* d_rec) {
  const int grid = cus * wg_per_cu, wpw = 4;
  k_roof<V><<<grid, 256>>>(d_rec);
  CK(hipDeviceSynchronize());
  k_roof<V><<<grid, 256>>>(d_rec);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> rec((size_t)grid * wpw * 6);
  CK(hipMemcpy(rec.data(), d_rec, rec.size() * 8, hipMemcpyDeviceToHost));
  struct S { unsigned long long t0 = ~0ull, t1 = 0, w0 = ~0ull, w1 = 0; int n = 0; };
  std::map<unsigned int, S> simd;
  for (size_t w = 0; w < (size_t)grid * wpw; ++w) {
    const unsigned long long* r = &rec[6 * w];
    const unsigned int hw = (unsigned int)r[0], xcc = (unsigned int)r[1] & 0xf;
    const unsigned int key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 10) | (((hw >> 8) & 0xf) << 4) | ((hw >> 4) & 3);
    S& s = simd[key];
    s.t0 = std::min(s.t0, r[2]); s.t1 = std::max(s.t1, r[3]); s.w0 = std::min(s.w0, r[4]); s.w1 = std::max(s.w1, r[5]); ++s.n;
  }
  // per SIMD: shader cycles from the first loop entry to the last loop exit over the wave instructions it issued; median over SIMDs
  std::vector<double> cpi, ghz, ns;
  int wmin = 1 << 30, wmax = 0;
  for (auto& kv : simd) {
    const S& s = kv.second;
    const double insts = (double)s.n * ITER * 64.0;
    cpi.push_back((double)(s.t1 - s.t0) / insts);
    ns.push_back((double)(s.w1 - s.w0) * 10.0 / insts);
    ghz.push_back((double)(s.t1 - s.t0) / ((double)(s.w1 - s.w0) * 10.0));
    wmin = std::min(wmin, s.n); wmax = std::max(wmax, s.n);
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  return Res{med(cpi), med(ghz), med(ns), simd.size(), wmin, wmax};
}
'''


def main():
    regs = [f'"v{r}"' for r in range(CH0, 88)] + ['"s20"', '"s24"', '"s25"', '"s26"', '"s27"', '"s28"', '"scc"', '"vcc"', '"memory"']
    out = [HEAD % (ITER, ", ".join(regs))]
    for k, (name, lines, note) in enumerate(VARIANTS):
        init = [f"v_mov_b32 v{r}, 1.0" for r in range(CH0, 80)] + [f"v_mov_b32 v{r}, 0.5" for r in range(80, 88)] + \
               ["s_mov_b32 s24, 0.5", "s_mov_b32 s25, 0.5", f"s_mov_b32 s20, {ITER}", "s_mov_b64 s[26:27], -1"]
        if name.startswith("lds_"):
            init += ["v_mbcnt_lo_u32_b32 v84, -1, 0", "v_mbcnt_hi_u32_b32 v84, -1, v84", "v_lshlrev_b32 v85, 3, v84", "v_lshlrev_b32 v84, 2, v84"]
            if name == "lds_read_dep":
                init += ["v_mov_b32 v16, 0"]
        if "prio3" in name:
            init.append("s_setprio 3")
        loop = ["1:"] + lines + ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b"]
        if "prio3" in name:
            loop.append("s_setprio 0")
        text = "\\n\\t".join(init + loop)
        out.append(f'// {name}: {note}\ntemplate <> __device__ __forceinline__ void loop_body<{k}>() {{\n  asm volatile("{text}" ::: CLOBBERS);\n}}\n')
    out.append(KERNEL)
    names = ", ".join(f'"{v[0]}"' for v in VARIANTS)
    notes = ", ".join('"' + v[2].replace('"', "'") + '"' for v in VARIANTS)
    cases = "\n".join(f"      case {k}: r = run<{k}>(cus, w, d_rec); break;" for k in range(len(VARIANTS)))
    out.append(f'''
int main(int argc, char** argv) {{
  static const char* NAMES[] = {{{names}}};
  static const char* NOTES[] = {{{notes}}};
  const int NV = {len(VARIANTS)};
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned long long* d_rec;
  CK(hipMalloc(&d_rec, (size_t)cus * 8 * 4 * 6 * 8));
  printf("# %s, %d CUs; per SIMD: shader cycles (s_memtime) from first loop entry to last loop exit / wave64 instructions issued there; median over SIMDs\\n", prop.name, cus);
  printf("# one trip = 64 VALU + 3 SALU (s_sub, s_cmp, s_cbranch); %d trips\\n", ITER);
  for (int v = 0; v < NV; ++v) {{
    if (argc > 1 && strcmp(argv[1], NAMES[v]) != 0) continue;
    printf("%-22s %s\\n", NAMES[v], NOTES[v]);
    for (int w : {{1, 2, 4, 8}}) {{
      Res r{{}};
      switch (v) {{
{cases}
      }}
      printf("    waves/SIMD %d (placed %d..%d on %zu SIMDs): %.3f cycles per wave-instruction per SIMD, %.3f ns, clock %.2f GHz\\n", w, r.wmin, r.wmax, r.simds,
             r.cyc_per_instr_simd, r.ns, r.ghz);
    }}
  }}
  return 0;
}}
''')
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_roof.hip")
    open(path, "w").write("".join(out))
    print("wrote", path)


if __name__ == "__main__":
    main()
