#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run (rocpd sqlite output)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'%':>6s}")
for name, calls, tot, avg, pct in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{name[:90]:90s} {calls:6d} {tot/1e3:12.1f} {avg/1e3:12.1f} {pct:6.2f}")
