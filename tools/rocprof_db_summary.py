#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (`--kernel-trace --stats`, default output
format of rocprofv3 in ROCm 7.2) as a small text table: calls / total / avg / median / min / max per
kernel.  Usage: python tools/rocprof_db_summary.py <results.db> [> profiles/NAME.txt]
(The MEDIAN is the figure to hold against bench.py's event-bracketed launch time: the average also covers launches that share the
chip -- bench.py's two-stream section runs two launches at a time, each then lasts twice as long -- and the first launches after idle.)"""
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
    "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(scratch_size) "
    "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
med = {n: statistics.median(d for (d,) in db.execute("select duration from kernels where name = ?", (n,))) for n, *_ in rows}
print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'median_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} "
      f"{'grid':>8s} {'wg':>4s} {'lds':>7s} {'vgpr':>5s} {'agpr':>5s} {'scratch':>7s}")
for n, c, s, a, mn, mx, gx, wx, lds, vg, ag, sc in rows:
    print(f"{n[:70]:70s} {c:6d} {s / 1e6:10.3f} {a / 1e3:10.2f} {med[n] / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * s / total:6.2f} "
          f"{gx:8d} {wx:4d} {lds:7d} {vg:5d} {ag:5d} {sc:7d}")
