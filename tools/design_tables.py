#!/usr/bin/env python3
"""DESIGN.md's current-state tables, generated from profiles/<tag>_* (VERDICT r5 item 9): one block per configuration -- step, stages, the
heaviest kernels, their counters -- every number with the file it comes from.  Writes profiles/<tag>_TABLES.md and replaces the text between
the markers `<!-- BEGIN GENERATED TABLES -->` and `<!-- END GENERATED TABLES -->` of DESIGN.md with it.
usage (container, after tools/finish_round.sh): tools/design_tables.py r06 [--check]      --check: fail if DESIGN.md is not up to date"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def load_line(name):
    p = os.path.join(PROF, name)
    if not os.path.exists(p):
        return None
    txt = [l for l in open(p).read().strip().splitlines() if l.startswith("{")]
    return json.loads(txt[-1]) if txt else None


def load_json(name):
    p = os.path.join(PROF, name)
    return json.load(open(p)) if os.path.exists(p) else None


def load_csv(name):
    p = os.path.join(PROF, name)
    return list(csv.DictReader(open(p))) if os.path.exists(p) else None


def short(name, n=64):
    name = name.replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    name = name.replace("rocprim::ROCPRIM_400200_NS::detail::", "rocprim::")
    name = re.sub(r"trampoline_kernel<rocprim::wrapped_(\w+)_config.*", r"\1 (rocprim)", name)
    return name[:n]


def kernel_table(stats, steps, top=10, pmc=None):
    rows = sorted(stats, key=lambda r: -float(r["TotalDurationNs"]))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    pm = {}
    if pmc:
        for r in pmc:
            pm[r["kernel"].replace("void ", "")[:60]] = r
    out = ["| kernel | launches / step | average µs | ms / step | share of kernel time |" + (" SQ_WAIT_ANY / SQ_WAVE_CYCLES | FETCH MB | WRITE MB |" if pmc else ""),
           "|---|---|---|---|---|" + ("---|---|---|" if pmc else "")]
    for r in rows[:top]:
        calls = float(r["Calls"]) / steps
        line = f"| `{short(r['Name'])}` | {calls:.4g} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {100 * float(r['TotalDurationNs']) / total:.1f} % |"
        if pmc:
            q = pm.get(r["Name"].replace("void ", "")[:60])
            def num(k):
                try:
                    return float(q[k])
                except (TypeError, KeyError, ValueError):
                    return None
            wa, wc, fs, ws = (num("SQ_WAIT_ANY"), num("SQ_WAVE_CYCLES"), num("FETCH_SIZE"), num("WRITE_SIZE")) if q else (None,) * 4
            line += f" {wa / wc:.2f} |" if wa is not None and wc else " |"
            line += f" {fs * 1024 / 1e6:.1f} |" if fs is not None else " |"
            line += f" {ws * 1024 / 1e6:.1f} |" if ws is not None else " |"
        out.append(line)
    out.append(f"| all kernels | | | {total / 1e6 / steps:.3f} | |" + (" | | |" if pmc else ""))
    return "\n".join(out)


def stage_row(d):
    s = d.get("stage_ms") or {}
    keys = [k for k in ("supervoxel", "voxelize", "features", "adjacency", "localcut", "merge", "labels") if s.get(k)]
    return "| " + " | ".join(keys) + " |\n|" + "---|" * len(keys) + "\n| " + " | ".join(f"{s[k]:.3f}" for k in keys) + " |"


def config_block(tag, key, title, steps_profiled):
    line = load_line(f"{tag}_{key}_line.json")
    if line is None:
        return None
    out = [f"#### {title}", ""]
    cfg = line.get("config", {})
    out.append(f"`profiles/{tag}_{key}_line.json`: **{line['ms_per_step']:.3f} ms per step** ({line['value'] / 1e6:.1f} M points/s, {line.get('steps')} steps); "
               f"workload: {cfg.get('workload', '')}; counts: "
               + ", ".join(f"{k} {v}" for k, v in (line.get("counts") or {}).items() if k in ("points", "voxels", "used", "adj", "kept", "supervoxels", "class_a", "class_bc", "class_d")) + ".")
    out += ["", "Stages, ms (device events; the local cut's figure includes its tail beside the merge stage):", "", stage_row(line), ""]
    sc = line.get("schedule")
    if sc:
        out += ["Schedule counters: " + ", ".join(f"{k} {v}" for k, v in sc.items() if v) + ".", ""]
    stats = load_csv(f"{tag}_{key}_kernel_stats.csv")
    if stats:
        pmc = load_csv(f"{tag}_{key}_pmc_per_launch.csv")
        out += [f"Kernels (`profiles/{tag}_{key}_kernel_stats.csv`, rocprofv3 --kernel-trace --stats over {steps_profiled} steps"
                + (f"; counters per launch `profiles/{tag}_{key}_pmc_per_launch.csv`, own --pmc passes" if pmc else "") + "):", "",
                kernel_table(stats, steps_profiled, 10, pmc), ""]
    return "\n".join(out)


def bench_block(tag):
    line = load_line(f"{tag}_bench_line.json")
    if line is None:
        return None
    r = line["roofline"]
    out = [f"#### Config 3 = the bench: URB10M, VGS, voxel 0.1 m (BASELINE configs[2])", ""]
    out.append(f"`profiles/{tag}_bench_line.json` (`python bench.py --steps 20 --warmup 3`): **{line['ms_per_step']:.3f} ms per step = {line['value'] / 1e9:.3f}·10⁹ points/s** device-resident; "
               f"host xyz in → host labels out {line['host_to_host']['ms_per_step'] if 'ms_per_step' in line.get('host_to_host', {}) else line['host_to_host'].get('ms_per_cloud', float('nan')):.2f} ms "
               f"({line['host_to_host']['value'] / 1e9:.3f}·10⁹ points/s); CPU baseline (faithful port, 1 core) {line['cpu_baseline']['value']:.0f} points/s.")
    cfgl = line["config"]
    out += ["", f"Counts: {cfgl.get('points_per_gpu')} points, {cfgl.get('voxels')} voxels, {cfgl.get('used_voxels')} used, {cfgl.get('adjacency_entries')} adjacency entries; "
            f"algorithmic bytes B = 28·N + 48·V + 4·E = {r['algorithmic_bytes_per_step'] / 1e6:.1f} MB per step (SURVEY 8d).", ""]
    out += ["Stages, ms:", "", stage_row(line), ""]
    out += [f"Roofline of the dominant kernel (`roofline` in the same line): `{r['kernel']}` {r['kernel_ms']:.3f} ms per launch (HIP events on its stream), "
            f"algorithmic share {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB → {r['achieved']:.1f} GB/s = **{r['frac']:.4f} of 8 TB/s**; counter traffic "
            f"{(r.get('traffic') or 0) / 1e6:.1f} MB per launch (`{r.get('traffic_source')}`, commit {r.get('traffic_commit')}); VALU issue "
            f"{r.get('valu_issue_frac', float('nan')):.2f} of the kernel's time at {r.get('valu_cycles_per_wave_instr', float('nan')):.2f} cycles per wave instruction "
            f"(`{r.get('valu_mix_source')}`); end to end {r['end_to_end_frac']:.4f} of HBM peak; {r['pair_evals_per_s'] / 1e9:.1f}·10⁹ pair evaluations/s.", ""]
    stats = load_csv(f"{tag}_bench_kernel_stats.csv")
    if stats:
        pmc = load_csv(f"{tag}_bench_pmc_per_launch.csv")
        out += [f"Kernels (`profiles/{tag}_bench_kernel_stats.csv`: `bench.py --steps 5 --warmup 2` = 7 steps + set-up under rocprofv3 --kernel-trace --stats; counters "
                f"`profiles/{tag}_bench_pmc_per_launch.csv`):", "", kernel_table(stats, 8, 14, pmc), ""]
    bw = load_json(f"{tag}_stage_bw.json")
    if bw:
        out += [f"Bandwidth-shaped kernels against the 6.29 TB/s copy ceiling (`profiles/{tag}_stage_bw.json`):", "",
                "| kernel | average µs | algorithmic MB | counter FETCH / WRITE MB | GB/s | of the copy ceiling |", "|---|---|---|---|---|---|"]
        for k in bw["kernels"]:
            out.append(f"| `{k['kernel']}` | {k['avg_us']} | {k['algorithmic_MB']} | {k.get('counter_fetch_MB')} / {k.get('counter_write_MB')} | {k['GBs']} | {k['frac_of_copy_ceiling']} |")
        out.append("")
    mix = load_json(f"{tag}_valu_mix.json")
    if mix:
        out += [f"VALU mix of the bulk kernel (`profiles/{tag}_valu_mix.json`): {mix['valu_wave_instructions_per_launch'] / 1e9:.3f}·10⁹ wave instructions per launch, "
                f"{mix['mean_cycles_per_instruction']:.2f} issue cycles each on average, **{100 * mix['share_half_or_slower_dynamic']:.1f} % of the executed instructions half rate or slower**, "
                f"issue fraction {mix['valu_issue_frac']:.2f}; per phase (dynamic instructions): "
                + ", ".join(f"{p} {v / 1e6:.0f} M" for p, v in mix["dynamic_per_phase"].items()) + ".", ""]
    nat = load_line(f"{tag}_bench_native_line.json")
    if nat:
        out += [f"Native tiled driver with a one-rank world (`profiles/{tag}_bench_native_line.json`, `bench.py --gpus 1 --native`): {nat['ms_per_step']:.3f} ms per step.", ""]
    return "\n".join(out)


def c5_block(tag):
    d = load_json(f"{tag}_c5_onegpu.json")
    if d is None:
        return None
    out = ["#### Config 5 at its real size on one GPU: URB80M, 4 × 2 tiles × 10 M points (BASELINE configs[4])", "",
           f"`profiles/{tag}_c5_onegpu.json` (written by `tests/test_gpu_config5.py`): eight ranks of `libvgs_tiles.so` as threads, eight contexts on one MI355X; "
           f"{d['points']} points, {d['kept_segments_tiled']} segments kept (a single engine over the 80 M points: {d['kept_segments_single_engine']}), partition agreement with the "
           f"single engine {d['partition_agreement_with_single_engine']:.5f}, {d['points_outside_closestcheck_identical']} points identical up to renaming outside closestCheck's candidates, "
           f"{d['voxels_shared_by_ranks']} voxels hold points of several ranks; halo redundancy {d['halo_redundancy_factor']:.4f} (SURVEY 8e estimated 1.07); "
           f"{d['hbm_in_use_all_ranks_gb']:.1f} GB of HBM in use by the eight ranks ({d['hbm_per_rank_gb']:.2f} GB each).", "",
           "| rank | own points | halo points | boundary records | used voxels | exchange: sent / received bytes, collectives | driver ms (2nd run): grid / stages / records / exchange / merge / labels |",
           "|---|---|---|---|---|---|---|"]
    for r in d["per_rank"]:
        t = r["driver_ms_second_run"]
        e = r["exchange"]
        out.append(f"| {r['rank']} | {r['own_points']} | {r['halo_points']} | {r['boundary_records']} | {r['used_voxels']} | {e['bytes_sent']} / {e['bytes_received']}, {e['collectives']} | "
                   f"{t['grid']:.2f} / {t['stages']:.1f} / {t['records']:.2f} / {t['exchange']:.2f} / {t['merge']:.2f} / {t['labels']:.2f} |")
    out += ["", "(Eight ranks share ONE GPU here: the stage times say nothing about an 8-GPU node; record counts, halo sizes, exchange bytes and the host-side phases carry over.)", ""]
    return "\n".join(out)


def main():
    tag = sys.argv[1]
    man = load_json(f"{tag}_MANIFEST.json") or {}
    commits = sorted(set(man.values()))
    blocks = [f"<!-- generated by tools/design_tables.py {tag} from profiles/{tag}_* (commit{'s' if len(commits) != 1 else ''} {', '.join(commits) or 'n/a'} per profiles/{tag}_MANIFEST.json); do not edit -->", ""]
    for b in (bench_block(tag),
              config_block(tag, "c2", "Config 2: PC1M, VGS, voxel 0.05 m, graph 0.5 m (BASELINE configs[1])", 6),
              config_block(tag, "c4", "Config 4: URB10M, SVGS, supervoxels in PCL's order = the default (BASELINE configs[3])", 6),
              config_block(tag, "c4s", "Config 4 with the synchronous supervoxel variant (vccs_mode 0: an approximation, not within P2 of the default)", 6),
              c5_block(tag),
              config_block(tag, "c3n", "Noisy surface `c3n`: 5 M points, 3 cm range noise, VGS, voxel 0.1 m (not a BASELINE config)", 6),
              config_block(tag, "xl", "Solid block `xl`: 500 k points, voxel 0.05 m, graph 0.5 m — neighbourhoods up to 4159 voxels (not a BASELINE config)", 4)):
        if b:
            blocks += [b, ""]
    text = "\n".join(blocks).rstrip() + "\n"
    open(os.path.join(PROF, f"{tag}_TABLES.md"), "w").write(text)
    design = os.path.join(ROOT, "DESIGN.md")
    src = open(design).read()
    a, b = "<!-- BEGIN GENERATED TABLES -->", "<!-- END GENERATED TABLES -->"
    if a in src and b in src:
        new = src[:src.index(a) + len(a)] + "\n" + text + src[src.index(b):]
        if "--check" in sys.argv:
            if new != src:
                raise SystemExit("DESIGN.md's generated tables are not up to date with profiles/: run tools/design_tables.py " + tag)
        else:
            open(design, "w").write(new)
    print(f"profiles/{tag}_TABLES.md: {len(text.splitlines())} lines")


if __name__ == "__main__":
    main()
