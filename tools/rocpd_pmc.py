#!/usr/bin/env python3
"""Per-kernel average of the PMC counters in a rocprofv3 rocpd database (one --pmc pass)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("tables:", [t for t in tabs if not t[-36:-35] == "_"][:40]); sys.exit(0)
cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
rows = db.execute("select %s, counter_name, value, dispatch_id from %s" % (kcol, view)).fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for k, c, v, d in rows:
    agg[(k, c)][d] += v  # sum over instances (XCDs / SEs) of one dispatch
for (k, c), per in sorted(agg.items()):
    vals = sorted(per.values())
    print("%-70s %-12s dispatches %3d  avg %14.1f  min %14.1f  max %14.1f" % (k[:70], c, len(vals), sum(vals) / len(vals), vals[0], vals[-1]))
