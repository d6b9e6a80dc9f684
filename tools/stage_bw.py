#!/usr/bin/env python3
"""Achieved bandwidth of the BANDWIDTH-SHAPED kernels of a step (SURVEY.md 8d's honesty note): algorithmic bytes of
each kernel -- what it must read and write once, from the counts of the run -- over its average duration in a rocprofv3 kernel-stats
csv, against the 6.29 TB/s float4-copy ceiling of MI355X (MI355X_MICROARCH.md) and the 8 TB/s spec.
With a per-launch counter table (tools/pmc_table.py) as fourth argument the HBM bytes the counters saw (FETCH_SIZE / WRITE_SIZE, KB -> MB, raw)
stand beside the algorithmic ones: traffic well above them is wasted re-reads (the gather and the label scatter: whole sectors for 12- and 4-byte accesses).
usage: tools/stage_bw.py <kernel_stats.csv> <bench_line.json> [out.json] [pmc_per_launch.csv]"""
import csv, json, sys

rows = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
line = json.load(open(sys.argv[2]))
cfg = line["config"]
N, V, U, E = cfg["points_per_gpu"], cfg["voxels"], cfg["used_voxels"], cfg["adjacency_entries"]


def find(prefix, nth=0):
    hits = [r for n, r in rows.items() if n.replace("void ", "").startswith(prefix)]
    hits.sort(key=lambda r: -float(r["TotalDurationNs"]))
    return hits[nth] if len(hits) > nth else None


key_b = 8   # 64-bit codes on URB10M (34 key bits)
# kernel, algorithmic bytes per call, what they are
spec = [
    ("k_make_codes", (12 + key_b + 4) * N, "xyz in; code + index out"),
    ("k_heads", (key_b + 4) * N, "sorted codes in; run-head flags out"),
    ("k_voxel_table", (key_b + 4 + 4 + 4) * N, "codes, heads, scan in; point->voxel out (+ V table entries)"),
    ("k_gather_points", (4 + 12 + 12) * N, "order in; xyz gathered (12 B random reads); SoA out"),
    ("k_features", 12 * N + 64 * V, "leaf-order points in; 64-byte voxel records out"),
    ("k_point_labels", (4 + 4 + 4) * N, "order + point->voxel in; labels scattered out"),
    ("k_compress", 8 * V, "parent in / out"),
    ("k_flatten", 12 * V, "parent in; parent + sizes out"),
    ("k_merge_init", 20 * V, "five per-voxel arrays out"),
]
pmc = {}
if len(sys.argv) > 4:
    for r in csv.DictReader(open(sys.argv[4])):
        pmc[r["kernel"].replace("void ", "")] = r


def counters(prefix):
    for n, r in pmc.items():
        if n.startswith(prefix) and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
            return round(float(r["FETCH_SIZE"]) * 1024 / 1e6, 1), round(float(r["WRITE_SIZE"]) * 1024 / 1e6, 1)
    return None, None


out = {"points": N, "voxels": V, "used_voxels": U, "adjacency_entries": E, "copy_ceiling_GBs": 6290.0, "hbm_spec_GBs": 8000.0, "kernels": []}
for name, nbytes, what in spec:
    r = find(name)
    if not r:
        continue
    us = float(r["AverageNs"]) / 1e3
    gbs = nbytes / (us * 1e-6) / 1e9
    f_mb, w_mb = counters(name)
    out["kernels"].append({"kernel": name, "avg_us": round(us, 1), "algorithmic_MB": round(nbytes / 1e6, 1), "counter_fetch_MB": f_mb, "counter_write_MB": w_mb,
                           "GBs": round(gbs), "frac_of_copy_ceiling": round(gbs / 6290.0, 3), "what": what})
# the radix sort: every onesweep pass moves key + index in and out
passes = [r for n, r in rows.items() if "radix_sort_onesweep" in n]
passes.sort(key=lambda r: -float(r["TotalDurationNs"]))
if passes:
    r = passes[0]   # the digit passes (the histogram pass is the other instantiation)
    us = float(r["AverageNs"]) / 1e3
    nbytes = 2 * (key_b + 4) * N
    gbs = nbytes / (us * 1e-6) / 1e9
    out["kernels"].append({"kernel": "rocprim radix_sort_onesweep (one digit pass)", "avg_us": round(us, 1), "algorithmic_MB": round(nbytes / 1e6, 1), "GBs": round(gbs),
                           "frac_of_copy_ceiling": round(gbs / 6290.0, 3), "what": "key + index in and out"})
txt = json.dumps(out, indent=1)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(txt + "\n")
for k in out["kernels"]:
    print(f"{k['kernel'][:46]:46s} {k['avg_us']:8.1f} us {k['algorithmic_MB']:8.1f} MB {k['GBs']:6d} GB/s  {100 * k['frac_of_copy_ceiling']:5.1f} % of 6.29 TB/s")
