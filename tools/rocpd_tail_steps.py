#!/usr/bin/env python3
"""Per-step start offset / duration of the inference tail's kernels in a bench.py rocprofv3 trace (two streams: the proposal
layer runs beside the correlations and the head GEMMs) -- shows which launches overlapped and what that cost.

    rocpd_tail_steps.py <db> [nsteps]

A step's tail runs from its conv5 correlation launch (grid 196608 at the 600 px shape; the first corr_wsplit_kernel after a
pooling launch) to its last pooling launch.  REFUSED (exit 3) unless the printed steps all hold the same tag sequence: a window
that mixes inference steps with anything else (round 4: the training leg bench.py runs afterwards) is not printed."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = (("corr_wsplit_kernel", "corr"), ("proposal_select_sort", "sort"), ("proposal_sort_runs", "runs"), ("proposal_rank_scatter", "rank"),
         ("head_gemm_kernel<3", "rpn"), ("nms_mask", "mask"),
         ("nms_sweep", "sweep"), ("head_gemm_kernel<10", "head"), ("head_gemm_kernel<6", "trk"), ("psroi_pm_det_kernel", "psroi"), ("psroi_pm_kernel<32", "psroi"))
steps, cur = [], None
for n, s, e in rows:
    tag = next((t for k, t in short if k in n), None)
    if tag is None:
        continue
    if tag == "corr" and (cur is None or any(t == "psroi" for t, _, _ in cur)):
        cur = []; steps.append(cur)
    if cur is not None:
        cur.append((tag, s, e))
# what is launched behind a step's last pooling (the next step's trunk issues its class + box head GEMM ahead of the next conv5
# correlation) belongs to the next step, not to this tail
for st in steps:
    last = max((i for i, (t, _, _) in enumerate(st) if t == "psroi"), default=len(st) - 1)
    del st[last + 1:]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sel = steps[-n:]
seqs = {tuple(sorted(t for t, _, _ in st)) for st in sel}
if len(sel) < n or len(seqs) != 1:
    print("REFUSED: the last %d tails do not hold the same launches (%d distinct sets) -- not %d identical inference steps" % (n, len(seqs), n))
    import collections
    for q in seqs:
        print("   ", dict(collections.Counter(q)), " x %d steps" % sum(1 for st in sel if tuple(sorted(t for t, _, _ in st)) == q))
    sys.exit(3)
for st in sel:
    t0 = st[0][1]
    print("  ".join("%s@%d:%.0f" % (t, (s - t0) / 1e3, (e - s) / 1e3) for t, s, e in st), " | tail %.0f us" % ((max(e for _, _, e in st) - t0) / 1e3))
