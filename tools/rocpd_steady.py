#!/usr/bin/env python3
"""Per-kernel time over the LAST n steps of a bench.py rocprofv3 trace (steps delimited by a marker kernel
that runs `per_step` times per step), excluding warm-up / MIOpen find-mode launches."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
marker = sys.argv[3] if len(sys.argv) > 3 else "corr_fwd_mfma<3"
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[0]]
first = marks[-nsteps - 1] + 1  # just after the marker of step -(nsteps+1) .. approximates nsteps steps
last = marks[-1]
sel = rows[first:last + 1]
span = (sel[-1][2] - sel[0][1]) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in sel:
    agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print("steps=%d  wall span %.1f us/step  kernel-busy %.1f us/step (%d kernels/step)" % (nsteps, span / nsteps, tot / nsteps, len(sel) // nsteps))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    print("%-90s %6.1f calls/step %9.1f us/step %6.2f%%  avg %8.2f us" % (n[:90], c / nsteps, t / nsteps, 100 * t / tot, t / c))
