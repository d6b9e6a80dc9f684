#!/usr/bin/env python3
"""ONE VALU-issue number for the bulk kernel of the local cut (k_localcut_wave<96,448,1>), instead of the range of round 4.

What the hardware cannot tell: SQ_INSTS_VALU counts wave instructions whatever they cost, SQ_ACTIVE_INST_VALU books one quad-cycle
per instruction whatever it costs, and gfx950's per-type counters split by operation, not by issue rate.  What tools/valu_roof.hip
measured (profiles/r04_valu_roof.txt, registers placed by hand): a SIMD issues a wave64 VALU instruction every
    2.2 cycles  -- VOP1/VOP2 arithmetic and moves, VOP3 with its sources in different VGPR banks (index mod 4) or inline constants
    4.1 cycles  -- every compare, every select (v_cndmask), every DPP form, v_readlane / v_readfirstlane / v_writelane, shifts and
                   bit-field ops, integer multiplies and mads, conversions, packed math, and ANY VOP3 with an SGPR source or two VGPR
                   sources in one bank
    8.1 cycles  -- transcendentals (v_exp / v_log / v_rcp / v_rsq / v_sqrt), v_permlane*_swap.

This script multiplies the two things that exist:
  * STATIC: the kernel's ISA (hipcc -S with line tables), every VALU instruction classified as above and attributed to a phase of
    the kernel by the source line it was generated from (gather / enumerate / evaluate / sort / merge / rest);
  * DYNAMIC: SQ_INSTS_VALU of the kernel leaving after each phase of its first shell (tools/pmc_phases.sh -> gpurun_out/pmc_phases.txt,
    committed as profiles/rNN_pmc_phases.txt): how many wave instructions each phase executes per launch.  Within a phase every
    static instruction is taken as equally often executed (its loops dominate it); what runs behind the first shell (later shells,
    phase B, the result rows: re-executions of the same code) is priced at the mix of enumerate + sort + merge + rest.
  issue cycles = sum over phases of count(phase) * (f_full * 2.2 + f_half * 4.1 + f_quarter * 8.1); valu_issue_frac = issue time on
  1024 SIMDs / kernel time.

usage: tools/valu_mix.py <pmc_phases.txt> <kernel_ms> [out.json] [commit]     (CPU: compiles csrc/localcut.hip device-only to assembly)"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vgs-svgs-segmentation_amd", "csrc")
CYC = {"full": 2.2, "half": 4.1, "quarter": 8.06}
CLOCK_GHZ = 2.39     # under load (profiles/r04_valu_roof.txt)
N_SIMD = 1024

HALF_PREFIX = ("v_cmp", "v_cmpx", "v_cndmask", "v_readlane", "v_readfirstlane", "v_writelane", "v_lshlrev", "v_lshrrev", "v_ashrrev", "v_lshl_",
               "v_bfe", "v_bfi", "v_and_or", "v_or3", "v_xad", "v_add_lshl", "v_mul_lo", "v_mul_hi", "v_mad_u32", "v_mad_i32", "v_mad_u64", "v_mad_i64",
               "v_mbcnt", "v_bcnt", "v_cvt", "v_pk_", "v_alignbit", "v_alignbyte", "v_perm_b32", "v_ffbh", "v_ffbl", "v_bfm", "v_div_", "v_ldexp",
               "v_frexp", "v_trunc", "v_floor", "v_ceil", "v_rndne", "v_fract", "v_med3", "v_min3", "v_max3", "v_mov_b64", "v_fma_f64", "v_add_f64",
               "v_mul_f64", "v_lshl_add", "v_add3", "v_swap", "v_sad", "v_dot")
QUARTER_PREFIX = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_permlane")


def classify(mn, ops):
    if mn.startswith(QUARTER_PREFIX):
        return "quarter"
    if "_dpp" in mn or "dpp" in ops or mn.startswith(HALF_PREFIX):
        return "half"
    vop3 = mn.endswith("_e64") or mn.startswith(("v_fma_", "v_mad_", "v_add3", "v_min3", "v_max3"))
    if vop3:
        srcs = ops.split(",")[1:]
        banks, sgpr = [], False
        for s in srcs:
            s = s.strip()
            m = re.match(r"v\[?(\d+)", s)
            if m:
                banks.append(int(m.group(1)) % 4)
            elif re.match(r"(s\[?\d+|vcc|exec|m0|ttmp)", s):
                sgpr = True
        if sgpr or len(banks) != len(set(banks)):
            return "half"
    return "full"


def phase_of(fname, line, ranges):
    if fname.endswith("regsort.hpp"):
        return "sort"
    if fname.endswith("vgs_math.h"):
        return "evaluate"
    if fname.endswith("localcut_wave.hpp"):
        for name, lo, hi in ranges:
            if lo <= line <= hi:
                return name
        return "rest"
    return "rest"


def source_ranges():
    """phases of localcut_wave.hpp by line, found from its own landmarks"""
    src = open(os.path.join(CSRC, "localcut_wave.hpp")).read().split("\n")
    def find(s, start=0):
        for i in range(start, len(src)):
            if s in src[i]:
                return i + 1
        raise SystemExit("landmark not found: " + s)
    k0 = find("void k_localcut_wave(")
    gather_end = find("LW_ACC(0);  // gather")
    enum0, enum1 = find("auto enum_section = "), find("auto eval_section = ") - 1
    eval0, eval1 = enum1 + 1, find("auto near_enum = ") - 1
    near0, near1 = eval1 + 1, find("// ---- wavefronts 1 .. NW-1") - 1
    merge0, merge1 = find("auto cut_over = "), find("auto never_merges = ") - 1
    return [("gather", k0, gather_end), ("enumerate", enum0, enum1), ("evaluate", eval0, eval1), ("enumerate", near0, near1), ("merge", merge0, merge1)]


def main():
    phases_txt, kernel_ms = sys.argv[1], float(sys.argv[2])
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    commit = sys.argv[4] if len(sys.argv) > 4 else None
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "lc.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-ffp-contract=off",
                               "-fno-fast-math", "-gline-tables-only", "-I", os.path.join(ROOT, "include"), "-S", os.path.join(CSRC, "localcut.hip"), "-o", asm],
                              cwd=CSRC, stderr=subprocess.DEVNULL)
        text = open(asm).read().split("\n")
    files, ranges = {}, source_ranges()
    start = next(i for i, l in enumerate(text) if re.match(r"_Z15k_localcut_waveILi96ELi448ELi1ELb0ELb1EE.*:", l))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    for l in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
    cur = ("", 0)
    static = {}
    for l in text[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), ""), int(m.group(2)))
            continue
        m = re.match(r"\s+(v_[a-z0-9_]+)\s*(.*?)(?:\s*;.*)?$", l)
        if not m:
            continue
        mn, ops = m.group(1), m.group(2)
        ph = phase_of(cur[0], cur[1], ranges)
        static.setdefault(ph, {"full": 0, "half": 0, "quarter": 0})[classify(mn, ops)] += 1
    # dynamic counts
    dyn = {}
    for l in open(phases_txt):
        m = re.match(r"(\w+)\s+(\{.*\})", l.strip())
        if m:
            dyn[m.group(1)] = float(eval(m.group(2))["SQ_INSTS_VALU"])
    d = {"gather": dyn["gather"], "enumerate": dyn["enum"] - dyn["gather"], "sort": dyn["sort"] - dyn["enum"], "merge": dyn["merge"] - dyn["sort"],
         "later": dyn["full"] - dyn["merge"]}
    def mix(names):
        tot = {"full": 0, "half": 0, "quarter": 0}
        for n in names:
            for k, v in static.get(n, {}).items():
                tot[k] += v
        s = max(sum(tot.values()), 1)
        return {k: v / s for k, v in tot.items()}
    # the "sort" measurement of the first shell holds the shell's evaluations too (none where the near-pair lists serve it)
    pm = {"gather": mix(["gather"]), "enumerate": mix(["enumerate"]), "sort": mix(["sort", "evaluate"]) if static.get("evaluate") else mix(["sort"]),
          "merge": mix(["merge"]), "later": mix(["enumerate", "sort", "merge", "rest", "evaluate"])}
    pm["sort"] = mix(["sort"])   # (URB10M: 91 % of the voxels never evaluate a weight; the register network is what runs)
    cycles = sum(d[p] * sum(pm[p][k] * CYC[k] for k in CYC) for p in d)
    total = sum(d.values())
    issue_s = cycles / (CLOCK_GHZ * 1e9) / N_SIMD
    res = {"kernel": "k_localcut_wave<96,448,1,false,true>", "commit": commit, "valu_wave_instructions_per_launch": total, "kernel_ms": kernel_ms,
           "static_instructions": {p: static.get(p) for p in sorted(static)}, "dynamic_per_phase": d, "mix_per_phase": pm,
           "mean_cycles_per_instruction": cycles / total, "valu_issue_frac": issue_s / (kernel_ms * 1e-3),
           "share_half_or_slower_dynamic": sum(d[p] * (pm[p]["half"] + pm[p]["quarter"]) for p in d) / total,
           "method": "static ISA mix per phase (tools/valu_mix.py) x SQ_INSTS_VALU per phase (tools/pmc_phases.sh); issue rates profiles/r04_valu_roof.txt"}
    print(json.dumps(res, indent=1))
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
