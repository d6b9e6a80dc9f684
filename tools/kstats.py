#!/usr/bin/env python3
"""Compact per-kernel table from a rocprofv3 --kernel-trace --stats csv: python3 tools/kstats.py <kernel_stats.csv> [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>10s} {'ms/step':>9s}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    print(f"{r['Name'][:60]:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6/steps:9.3f}")
