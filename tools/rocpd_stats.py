#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace: per-kernel calls / total / avg / min / max (us).  Kernels are keyed by
(name, grid, workgroup size) so that two layers sharing one instantiation (the conv5 and conv4 correlations are both
corr_wsplit_kernel<9>, grids 196608 and 327168/…) get a row each; `--by-name` restores the one-row-per-name table."""
import sqlite3, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
db = sqlite3.connect(args[0])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
by_grid = "grid_x" in cols and "--by-name" not in sys.argv
key = "%s, grid_x, grid_y, grid_z, workgroup_x" % name_col if by_grid else name_col
rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by sum(end-start) desc" % (key, key)).fetchall()
tot = sum(r[-4] for r in rows)
print("%-72s %-18s %7s %12s %10s %10s %10s %6s" % ("kernel", "grid/wg", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    n, (c, s, a, mn, mx) = r[0], r[-5:]
    g = ("%dx%dx%d/%d" % tuple(r[1:5])) if by_grid else ""
    print("%-72s %-18s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (n[:72], g, c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
