#!/usr/bin/env python3
"""Per-step start offset / duration of the inference tail's kernels in a bench.py rocprofv3 trace (two streams: the proposal
layer runs beside the correlations and the head GEMMs) -- shows which launches overlapped and what that cost."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = (("corr_wsplit_kernel<9", "corr5"), ("corr_wsplit_kernel", "corr3/4"), ("proposal_select_sort", "sort"), ("proposal_sort_runs", "runs"), ("proposal_rank_scatter", "rank"),
         ("head_gemm_kernel<3", "rpn"), ("nms_mask", "mask"),
         ("nms_sweep", "sweep"), ("head_gemm_kernel<10", "head"), ("head_gemm_kernel<6", "trk/rpn"), ("psroi_pm_det_kernel", "psroi"), ("psroi_pm_kernel<32", "psroi"))
steps, cur = [], None
for n, s, e in rows:
    tag = next((t for k, t in short if k in n), None)
    if tag is None:
        continue
    if tag == "corr5" and (cur is None or any(t == "psroi" for t, _, _ in cur)):
        cur = []; steps.append(cur)
    if cur is not None:
        cur.append((tag, s, e))
for st in steps[-int(sys.argv[2]) if len(sys.argv) > 2 else -8:]:
    t0 = st[0][1]
    print("  ".join("%s@%d:%.0f" % (t, (s - t0) / 1e3, (e - s) / 1e3) for t, s, e in st), " | tail %.0f us" % ((max(e for _, _, e in st) - t0) / 1e3))
