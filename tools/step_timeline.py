#!/usr/bin/env python3
"""Whole-step kernel timeline from a rocprofv3 --kernel-trace csv: the last step = from the last k_first_violation batch
to the last k_point_labels.  Prints start / end (us from step start), duration, gap to the previous kernel's end."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_point_labels")]
i1 = ends[-1]
i0 = ends[-2] + 1
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    name = r["Kernel_Name"].replace("void ", "")[:48]
    flag = "  <-- gap" if gap > 8 else ""
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap:7.1f}  {name:48s} q={r['Queue_Id']}{flag}")
    prev_end = max(prev_end, e)
print("step span us:", (prev_end - t0) / 1e3)
