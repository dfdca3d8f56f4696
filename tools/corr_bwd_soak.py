#!/usr/bin/env python3
"""Randomised soak of the streamed correlation gradients (developer tool, GPU box): random maps, window radius 1 .. 16 (radius > 8:
the quarter launches, PARTIAL instantiation), strides 1 / 2, pad != displacement, 1 .. 4 channel groups, planes and rows layouts,
both / single gradients -- against the oracle at 1e-4 and run-to-run bit identity under a concurrent stream.  SEEDS cases from S0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np, torch
from dtt.ops import correlation_backward_nhwc, correlation_output_shape
from oracle import oracle_lib as O
dev = torch.device("cuda:0")
N, S0 = int(os.environ.get("SEEDS", 120)), int(os.environ.get("S0", 0))
side = torch.cuda.Stream(); junk = torch.randn(32 << 20, device=dev)
bad = 0; t0 = time.time(); done = 0
for sd in range(S0, S0 + N):
    rng = np.random.RandomState(sd)
    s = int(rng.choice([1, 1, 2])); R = int(rng.randint(1, 17)); d = R * s
    pad = d if rng.rand() < 0.7 else max(0, d - s * int(rng.randint(1, 3)))
    B = int(rng.randint(1, 3)); C = 64 * int(rng.randint(1, 5)); H = int(rng.randint(5, 44)); W = int(rng.randint(5, 70))
    try:
        oc, oh, ow = correlation_output_shape(C, H, W, pad, 1, d, s, s)
    except Exception:
        continue
    if oh < 1 or ow < 1:
        continue
    x1 = rng.normal(size=(B, C, H, W)).astype(np.float32); x2 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    go = rng.normal(size=(B, oc, oh, ow)).astype(np.float32)
    cl = lambda a: torch.from_numpy(a).to(dev).contiguous(memory_format=torch.channels_last)
    t1, t2, gt = cl(x1), cl(x2), torch.from_numpy(go).to(dev)
    ga, gb = torch.full_like(t1, float("nan")), torch.full_like(t2, float("nan"))
    correlation_backward_nhwc(gt, t1, t2, ga, gb, pad, 1, d, s, s)
    g1, g2 = O.correlation_backward(go, x1, x2, pad, 1, d, s, s)
    e1, e2 = float(np.nanmax(np.abs(ga.cpu().numpy() - g1))), float(np.nanmax(np.abs(gb.cpu().numpy() - g2)))
    nan = int(torch.isnan(ga).sum() + torch.isnan(gb).sum())
    ld, col = oc + 24, 8
    rows = torch.zeros((B * oh * ow, ld), device=dev); rows[:, col:col + oc] = gt.permute(0, 2, 3, 1).reshape(-1, oc)
    ra, rb = torch.full_like(t1, float("nan")), torch.full_like(t2, float("nan"))
    with torch.cuda.stream(side):
        junk.mul_(1.0001)
    correlation_backward_nhwc(None, t1, t2, ra, None, pad, 1, d, s, s, rows=rows, col=col)
    correlation_backward_nhwc(None, t1, t2, None, rb, pad, 1, d, s, s, rows=rows, col=col)
    same = bool(torch.equal(ra, ga) and torch.equal(rb, gb))
    done += 1
    if e1 > 1e-4 or e2 > 1e-4 or nan or not same:
        bad += 1
        print("MISMATCH seed %d B %d C %d %dx%d pad %d d %d s %d: err %.2e %.2e nan %d rows/planes identical %s" % (sd, B, C, H, W, pad, d, s, e1, e2, nan, same), flush=True)
torch.cuda.synchronize()
print("%d cases, %d bad, %.0f s" % (done, bad, time.time() - t0))
sys.exit(1 if bad else 0)
