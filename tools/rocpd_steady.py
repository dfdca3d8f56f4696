#!/usr/bin/env python3
"""Per-kernel time over the LAST n steps of a bench.py rocprofv3 trace, excluding warm-up / find-mode launches.

    rocpd_steady.py <db> <nsteps> <marker> [top] [--per-step M] [--expect L]

A step ends with the marker kernel's launch (it runs M times per step, default 1; `psroi_pm_det_kernel` is the last pooling of
an inference step).  The window is REFUSED (exit 3, nothing summarised) unless every one of the n steps holds the same number of
launches -- and exactly L of them when --expect is given (the launch count of tools/rocpd_sequence.py's one step): a window that
straddles two kinds of step (round 4: inference steps + the training leg bench.py runs afterwards) is not a steady state.
Kernels are keyed by (name, grid, workgroup): two layers that share an instantiation (conv5 / conv4 correlations are both
corr_wsplit_kernel<9>) get a row each."""
import collections
import sqlite3
import sys


def load(db_path):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    grid = ", grid_x, grid_y, grid_z, workgroup_x" if "grid_x" in cols else ""
    rows = db.execute("select name, start, end%s from kernels order by start" % grid).fetchall()
    return [(r[0], r[1], r[2], ("%dx%dx%d/%d" % tuple(r[3:7])) if grid else "") for r in rows]


def step_windows(rows, marker, nsteps, per_step=1):
    """[(first, last)] row-index ranges of the last nsteps steps; a step ends at every per_step-th marker launch."""
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    ends = marks[::-1][::per_step][::-1]          # the LAST marker launch closes a step
    if len(ends) < nsteps + 1:
        return None
    ends = ends[-(nsteps + 1):]
    return [(ends[i] + 1, ends[i + 1]) for i in range(nsteps)]


def main():
    argv = [a for a in sys.argv[1:]]
    opts = {}
    for key in ("--per-step", "--expect"):
        if key in argv:
            i = argv.index(key)
            opts[key] = int(argv[i + 1])
            del argv[i:i + 2]
    rows = load(argv[0])
    nsteps = int(argv[1]) if len(argv) > 1 else 5
    marker = argv[2] if len(argv) > 2 else "psroi_pm_det_kernel"
    top = int(argv[3]) if len(argv) > 3 else 30
    wins = step_windows(rows, marker, nsteps, opts.get("--per-step", 1))
    if wins is None:
        print("REFUSED: fewer than %d + 1 steps delimited by '%s' in the trace" % (nsteps, marker))
        sys.exit(3)
    counts = [b - a + 1 for a, b in wins]
    expect = opts.get("--expect")
    if len(set(counts)) != 1 or (expect is not None and counts[0] != expect):
        print("REFUSED: launches per step %s%s -- the window is not %d identical steps" % (
            counts, "" if expect is None else " (expected %d)" % expect, nsteps))
        sys.exit(3)
    sel = rows[wins[0][0]:wins[-1][1] + 1]
    span = (sel[-1][2] - sel[0][1]) / 1e3
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for n, s, e, g in sel:
        a = agg[(n, g)]
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(v[1] for v in agg.values())
    print("steps=%d  wall span %.1f us/step  kernel-busy %.1f us/step (%d launches/step, every step)" % (
        nsteps, span / nsteps, tot / nsteps, counts[0]))
    for (n, g), (c, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-74s %-18s %5.1f calls/step %9.1f us/step %6.2f%%  avg %8.2f  min %8.2f  max %8.2f us" % (
            n[:74], g, c / nsteps, t / nsteps, 100 * t / tot, t / c, mn, mx))


if __name__ == "__main__":
    main()
