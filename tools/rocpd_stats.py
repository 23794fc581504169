#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd (sqlite) file: name, launches, total / mean / min / max duration (us), grid."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
q = """select s.kernel_name, count(*), sum(d.end - d.start) / 1e3, avg(d.end - d.start) / 1e3, min(d.end - d.start) / 1e3,
       max(d.end - d.start) / 1e3, max(d.grid_size_x), max(d.grid_size_y) from rocpd_kernel_dispatch d
       join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""
for name, n, tot, avg, mn, mx, gx, gy in cur.execute(q):
    if pat in name:
        print(f"{n:6d} {tot:12.1f} {avg:10.1f} {mn:10.1f} {mx:10.1f}  grid {gx}x{gy}  {name[:110]}")
