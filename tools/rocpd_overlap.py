#!/usr/bin/env python3
"""Kernels around (and overlapping, on any queue) the last three launches of a kernel in a rocprofv3 rocpd trace (developer tool):
    tools/rocpd_overlap.py <db> "<kernel name substring>" """
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = [c for c in cols if "queue" in c or "stream" in c]
rows = cur.execute("select %s, start, end, %s from kernels order by start" % (name_col, ",".join(qcol) if qcol else "0")).fetchall()
tgt = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
for i in tgt[-3:]:
    n, s, e = rows[i][:3]
    print("== %s  dur %.1f us  queue %s" % (n[:60], (e - s) / 1e3, rows[i][3:]))
    for j in range(max(0, i - 6), min(len(rows), i + 7)):
        m, s2, e2 = rows[j][:3]
        print("   %s%-70s start %+9.1f  end %+9.1f  dur %7.1f  q %s" % ("*" if j == i else " ", m[:70], (s2 - s) / 1e3, (e2 - s) / 1e3, (e2 - s2) / 1e3, rows[j][3:]))

    print("   kernels of ANY queue overlapping it:")
    for m, s2, e2, *q in rows:
        if s2 < e and e2 > s and (m, s2) != (n, s):
            print("      %-70s start %+9.1f end %+9.1f  q %s" % (m[:70], (s2 - s) / 1e3, (e2 - s) / 1e3, q))
