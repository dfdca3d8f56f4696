#!/usr/bin/env python3
"""Developer tool: race screen for the LDS-DMA correlation kernel -- the op is deterministic (fixed-order split-K
reduction), so any run-to-run difference is a synchronisation bug.  Runs the conv5 / conv4 / d=16 shapes many times
under memory pressure from a concurrent stream and compares every result bit for bit with the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
from dtt.ops import Correlation
dev = torch.device("cuda:0")
N = int(os.environ.get("N", 300))
side = torch.cuda.Stream()
junk = torch.randn(64 << 20, device=dev)
bad = 0
for (B, C, H, W, d) in [(2, 2048, 38, 67, 8), (4, 1024, 38, 67, 8), (1, 2048, 36, 63, 16), (2, 512, 19, 23, 4), (3, 64, 38, 67, 8)]:
    g = torch.Generator(device=dev).manual_seed(C + d)
    x1 = torch.relu(torch.randn(B, C, H, W, generator=g, device=dev)); x2 = torch.relu(torch.randn(B, C, H, W, generator=g, device=dev))
    corr = Correlation(d, 1, d, 1, 1)
    ref = corr(x1, x2).clone()
    # independent check of the first result: zero-displacement channel in float64
    R = d; D = 2 * R + 1
    c0 = (x1.double() * x2.double()).mean(1)
    assert float((ref[:, R * D + R].double() - c0).abs().max()) < 1e-5
    mism = 0
    for i in range(N):
        with torch.cuda.stream(side):
            junk.mul_(1.0001)          # HBM traffic + another stream's kernels in flight
        out = corr(x1, x2)
        if not torch.equal(out, ref):
            mism += 1
    torch.cuda.synchronize()
    print("B=%d C=%d %dx%d d=%d: %d / %d runs differ" % (B, C, H, W, d, mism, N), flush=True)
    bad += mism
sys.exit(1 if bad else 0)
