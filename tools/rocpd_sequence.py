#!/usr/bin/env python3
"""Ordered kernel list of the LAST bench.py step in a rocprofv3 rocpd trace (step delimited by a marker kernel):
one line per launch with start offset, duration, grid and a shortened name -- for mapping library kernels to layers."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "corr_fwd_mfma<3"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
grid = ", grid_x, grid_y, grid_z, workgroup_x" if "grid_x" in cols else ""
rows = db.execute("select name, start, end%s from kernels order by start" % grid).fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[0]]
sel = rows[marks[-2] + 1:marks[-1] + 1]
t0 = sel[0][1]
for r in sel:
    g = (" grid %dx%dx%d wg %d" % r[3:7]) if grid else ""
    print("%9.1f %8.1f us%s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, g, r[0][:110]))
