#!/usr/bin/env python3
"""Randomised NMS soak against the oracle (developer tool, GPU box): sizes 1 .. 20000, cluster counts from "everything overlaps" to
"nothing does", thresholds, max_keep (two-phase / single-phase / keep-all), degenerate and duplicate boxes.  SEEDS cases, seed S0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np, torch
from dtt.ops import nms
from oracle import oracle_lib as O
dev = torch.device("cuda:0")
N, S0 = int(os.environ.get("SEEDS", 300)), int(os.environ.get("S0", 0))
bad = 0
t0 = time.time()
for s in range(S0, S0 + N):
    rng = np.random.RandomState(s)
    n = int(rng.choice([rng.randint(1, 200), rng.randint(200, 3000), rng.randint(3000, 13000), rng.randint(13000, 20001)]))
    clusters = max(1, int(n * rng.choice([0.002, 0.02, 0.1, 0.5, 2.0])))
    ctr = rng.uniform(0, rng.choice([300, 1000, 4000]), size=(clusters, 2))
    wh = rng.uniform(8, 250, size=(clusters, 2))
    which = rng.randint(0, clusters, size=n)
    boxes = np.concatenate([ctr[which] - wh[which] / 2, ctr[which] + wh[which] / 2], 1) + rng.normal(0, rng.choice([0.0, 0.5, 3.0]), size=(n, 4))
    if n > 10 and rng.rand() < 0.3:
        boxes[rng.randint(0, n, 5)] = boxes[0]                       # exact duplicates
        boxes[rng.randint(0, n), 2:] = boxes[rng.randint(0, n), :2] - 5  # a degenerate (negative-size) box
    dets = np.concatenate([boxes, np.sort(rng.uniform(0, 1, n))[::-1][:, None]], 1).astype(np.float32)
    thr = float(rng.choice([0.3, 0.5, 0.7, 0.9]))
    mk = int(rng.choice([0, 1, 100, 300, 2000, n]))
    ref = O.nms(dets, thr)
    got = nms(torch.from_numpy(dets).to(dev), thr, max_keep=mk).cpu().numpy().ravel()
    want = ref[:mk] if mk else ref
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH seed %d n %d clusters %d thr %.1f max_keep %d: got %d want %d first diff %s" % (
            s, n, clusters, thr, mk, len(got), len(want), np.nonzero(got[:min(len(got), len(want))] != want[:min(len(got), len(want))])[0][:3]), flush=True)
print("%d cases, %d mismatches, %.0f s" % (N, bad, time.time() - t0))
sys.exit(1 if bad else 0)
