#!/usr/bin/env python3
"""Print the kernel timeline of the last local-cut stage in a rocprofv3 --kernel-trace csv (start, end in us)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_classify"
count = int(sys.argv[3]) if len(sys.argv) > 3 else 14
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(anchor)]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + count]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} {r['Kernel_Name'][:60]:60s} "
          f"grid={r['Grid_Size_X']} q={r['Queue_Id']} vgpr={r['VGPR_Count']} lds={r['LDS_Block_Size']}")
