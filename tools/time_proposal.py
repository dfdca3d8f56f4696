#!/usr/bin/env python3
"""Proposal layer alone at the 600 x 1067 shapes (B images, 12 anchors, 38 x 67 map): wall time per call by HIP events, and -- run
under `rocprofv3 --kernel-trace --stats` -- the per-kernel split.  DTT_PROPOSAL_ONE_WG=1 selects the one-workgroup-per-image
selection for an A/B.  (developer tool, GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
from dtt.rpn import generate_anchors, proposal_forward

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 2))
H, W = 38, 67
g = torch.Generator(device="cpu").manual_seed(3)
base = torch.from_numpy(generate_anchors(scales=(4, 8, 16, 32))).float().to(dev)
A = base.size(0)
sharp = float(os.environ.get("SHARP", 2))
prob = torch.softmax(torch.randn(B, 2, A * H, W, generator=g) * sharp, 1).view(B, 2 * A, H, W).to(dev)
bbox = (torch.randn(B, 4 * A, H, W, generator=g) * 0.4).to(dev)
info = torch.tensor([[600.0, 1067.0, 0.8333]] * B).to(dev)


def timeit(name, fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    print("%-34s %9.1f us" % (name, s.elapsed_time(e) * 1000 / iters), flush=True)


cases = ((6000, 300), (12000, 2000)) if not os.environ.get("TEST_ONLY") else ((6000, 300),)
if os.environ.get("TRAIN_ONLY"):
    cases = ((12000, 2000),)
for pre, post in cases:
    fn = lambda: proposal_forward(prob, bbox, info, base, 16, pre, post, 0.7)
    timeit("proposal %d->%d (eager)" % (pre, post), fn)
    # the same call replayed from a HIP graph: what the layer costs inside the captured inference step
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            out = fn()
    torch.cuda.synchronize()
    timeit("proposal %d->%d (graph)" % (pre, post), gr.replay)
    print("kept", out[1].tolist())
