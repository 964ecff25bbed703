#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (`--kernel-trace --stats`, default output
format of rocprofv3 in ROCm 7.2) as a small text table: calls / total / avg / min / max per
kernel.  Usage: python tools/rocprof_db_summary.py <results.db> [> profiles/NAME.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
    "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(scratch_size) "
    "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} "
      f"{'grid':>8s} {'wg':>4s} {'lds':>7s} {'vgpr':>5s} {'agpr':>5s} {'scratch':>7s}")
for n, c, s, a, mn, mx, gx, wx, lds, vg, ag, sc in rows:
    print(f"{n[:70]:70s} {c:6d} {s / 1e6:10.3f} {a / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * s / total:6.2f} "
          f"{gx:8d} {wx:4d} {lds:7d} {vg:5d} {ag:5d} {sc:7d}")
