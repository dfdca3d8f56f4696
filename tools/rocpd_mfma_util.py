#!/usr/bin/env python3
"""MFMA utilisation per kernel from one rocprofv3 pass with --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES:
util = busy cycles (summed over the 1024 SIMDs) / (1024 x kernel duration x 2.4 GHz), over the last `nsteps` bench steps."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
marker = sys.argv[3] if len(sys.argv) > 3 else "corr_fwd_mfma<3"
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
have_time = "start" in cols and "end" in cols
rows = db.execute("select %s, counter_name, value, dispatch_id%s from counters_collection" % (kcol, ", start, end" if have_time else "")).fetchall()
per = collections.OrderedDict()
for r in rows:
    if r[1] != "SQ_VALU_MFMA_BUSY_CYCLES":
        continue
    d = per.setdefault(r[3], [r[0], 0.0, r[4] if have_time else None, r[5] if have_time else None])
    d[1] += r[2]
disp = sorted(per.items(), key=lambda kv: (kv[1][2] if have_time else kv[0]))
marks = [i for i, (_, d) in enumerate(disp) if marker in d[0]]
sel = disp[marks[-nsteps - 1] + 1: marks[-1] + 1] if len(marks) > nsteps else disp
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for _, (name, busy, s, e) in sel:
    a = agg[name]; a[0] += 1; a[1] += busy; a[2] += (e - s) if have_time else 0.0
print("time columns in counters_collection: %s; %d dispatches over %d steps" % (have_time, len(sel), nsteps))
print("%-100s %6s %12s %10s %8s" % ("kernel", "calls", "mfma_busy", "dur_us", "util"))
for name, (c, busy, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    util = busy / (1024 * dur * 1e-9 * 2.4e9) if dur > 0 else float("nan")
    print("%-100s %6.1f %12.0f %10.1f %8.3f" % (name[:100], c / nsteps, busy / nsteps, dur / nsteps / 1e3, util))
