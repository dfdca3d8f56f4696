#!/usr/bin/env python3
"""Idle time of the GPU inside the LAST step of a rocprofv3 rocpd trace (steps delimited by a marker kernel): the gaps between
the end of everything launched so far and the start of the next kernel, largest first, with the kernels on either side -- where
a step waits for the host (developer tool).   rocpd_gaps.py <db> [marker] [top]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "psroi_pm_det_kernel"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[0]]
sel = rows[marks[-2] + 1:marks[-1] + 1]
t0 = sel[0][1]
busy_end = sel[0][2]
gaps = []
for i in range(1, len(sel)):
    n, s, e = sel[i]
    if s > busy_end:
        gaps.append((s - busy_end, i))
    busy_end = max(busy_end, e)
total = sum(g for g, _ in gaps) / 1e3
print("step: %d launches, span %.1f us, idle %.1f us in %d gaps (%d of them > 10 us = %.1f us)" % (
    len(sel), (sel[-1][2] - t0) / 1e3, total, len(gaps), sum(1 for g, _ in gaps if g > 10e3), sum(g for g, _ in gaps if g > 10e3) / 1e3))
for g, i in sorted(gaps, reverse=True)[:top]:
    print("%8.1f us idle at +%9.1f us   after %-70s before %s" % (g / 1e3, (sel[i][1] - t0) / 1e3, sel[i - 1][0][:70], sel[i][0][:70]))
