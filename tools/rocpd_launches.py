#!/usr/bin/env python3
"""List the launches of one kernel in a rocprofv3 rocpd trace, in order, split by position within a repeating group.

bench.py launches corr_fwd_glds<5,3> twice per step with the same grid (conv4: 1024 channels, then conv5: 2048), so
the --stats average of that kernel name mixes both; `rocpd_launches.py <db> "corr_fwd_glds<5" 2` prints the average
per position (position 1 = conv5, the launch bench.py's roofline object times)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]; period = int(sys.argv[3]) if len(sys.argv) > 3 else 1
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rows = [r for r in db.execute("select name, start, end from kernels order by start") if pat in r[0]]
rows = rows[skip * period:]
print("%d launches of %s (first %d groups skipped)" % (len(rows), rows[0][0][:80] if rows else pat, skip))
for ph in range(period):
    d = [(e - s) / 1e3 for _, s, e in rows[ph::period]]
    if d:
        print("position %d of %d: n=%d  avg %.2f us  min %.2f  max %.2f" % (ph, period, len(d), sum(d) / len(d), min(d), max(d)))
