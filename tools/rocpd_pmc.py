#!/usr/bin/env python3
"""Per-kernel average of the PMC counters in a rocprofv3 rocpd database (one --pmc pass).  Kernels are keyed by (name, grid) like
tools/rocpd_stats.py, so that two layers sharing one instantiation (conv5 and conv4 are both corr_wsplit_kernel<9>) get a row each;
`--by-name` restores one row per name.  The grid comes from the counter view's own columns when it has them, else from the kernel
trace of the same run (joined on dispatch_id: profile_round.sh always passes --kernel-trace beside --pmc)."""
import sqlite3, sys, collections
args = [a for a in sys.argv[1:] if not a.startswith("--")]
db = sqlite3.connect(args[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("tables:", [t for t in tabs if not t[-36:-35] == "_"][:40]); sys.exit(0)
cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
grid_of = {}
if "--by-name" not in sys.argv:
    gcol = [c for c in ("grid_size_x", "grid_x", "grid_size") if c in cols]
    try:
        if gcol:
            for d, g in db.execute("select dispatch_id, %s from %s" % (gcol[0], view)):
                grid_of[d] = g
        elif "kernels" in tabs:
            kc = [r[1] for r in db.execute("pragma table_info(kernels)")]
            if "dispatch_id" in kc and "grid_x" in kc:
                for d, g in db.execute("select dispatch_id, grid_x from kernels"):
                    grid_of[d] = g
    except sqlite3.Error:
        grid_of = {}
rows = db.execute("select %s, counter_name, value, dispatch_id from %s" % (kcol, view)).fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for k, c, v, d in rows:
    agg[(k, grid_of.get(d), c)][d] += v  # sum over instances (XCDs / SEs) of one dispatch
for (k, g, c), per in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][1] or 0, kv[0][2])):
    vals = sorted(per.values())
    name = k[:58] + (" [grid %d]" % g if g is not None else "")
    print("%-70s %-12s dispatches %3d  avg %14.1f  min %14.1f  max %14.1f" % (name[:70], c, len(vals), sum(vals) / len(vals), vals[0], vals[-1]))
