#!/usr/bin/env python3
"""Per-kernel averages PER LAUNCH of every counter in a set of rocprofv3 --pmc passes (one directory per pass), as one csv; for the
bench command also <tag>_traffic.json = HBM bytes per launch of the dominant kernel (FETCH_SIZE + WRITE_SIZE, both reported in KB),
stamped with the commit the passes were taken at.
usage: tools/pmc_table.py "<glob of pass directories>" <out.csv> <traffic.json or ""> <commit>"""
import collections
import csv
import glob
import json
import os
import sys

DOMINANT = "void k_localcut_wave<96, 448, 1, false"   # (the SAMPLED instantiation <.., true> runs one voxel in sixteen)


def main():
    pat, out_csv, traffic_json, commit = sys.argv[1:5]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for d in glob.glob(pat):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[(k, r["Counter_Name"])] += 1
    names = sorted({c for k in acc for c in acc[k]})
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches"] + names + ["commit"])
        for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", acc[k].get("SQ_INSTS_VALU", 0))):
            w.writerow([k[:90], max(calls[(k, c)] for c in acc[k])] + [f"{acc[k][c] / calls[(k, c)]:.6g}" if c in acc[k] else "" for c in names] + [commit])
    if traffic_json:
        dom = [k for k in acc if k.startswith(DOMINANT)]
        if dom:
            k = dom[0]
            per = lambda c: acc[k][c] / calls[(k, c)] if (k, c) in calls else None
            fetch_kb, write_kb = per("FETCH_SIZE"), per("WRITE_SIZE")
            json.dump({"kernel": k[:42], "commit": commit, "points": 10000000, "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb,
                       "hbm_bytes_per_launch": (fetch_kb + write_kb) * 1024.0 if fetch_kb is not None and write_kb is not None else None,
                       "valu_wave_instructions_per_launch": per("SQ_INSTS_VALU"),
                       "note": "FETCH_SIZE + WRITE_SIZE (KB) from separate rocprofv3 --pmc passes of the bench command; raw counters, no 2x wide-load correction "
                               "(the kernel gathers 8-byte row entries and 16-byte record quarters, not 16 B/lane streams)"},
                      open(traffic_json, "w"), indent=1)


if __name__ == "__main__":
    main()
