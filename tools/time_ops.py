#!/usr/bin/env python3
"""Quick per-op timing of the HIP hot path at the D&T 600 px shapes (developer tool, GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np
import torch
from dtt.ops import Correlation, _PSRoIPooling, nms, psroi_pool_vote, RoIAlignAvg
from dtt.rpn import generate_anchors, proposal_forward

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 2))
H, W = 38, 67


def timeit(name, fn, iters=50, warm=10, bytes_=None, flops=None):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / iters
    extra = ""
    if bytes_:
        extra += "  %.1f GB/s" % (bytes_ / us / 1e3)
    if flops:
        extra += "  %.2f TFLOP/s" % (flops / us / 1e6)
    print("%-34s %9.1f us%s" % (name, us, extra), flush=True)
    return us


g = torch.Generator(device="cpu").manual_seed(3)
f3 = torch.relu(torch.randn(B, 512, 75, 134, generator=g)).to(dev)
f4 = torch.relu(torch.randn(B, 1024, H, W, generator=g)).to(dev)
f5 = torch.relu(torch.randn(B, 2048, H, W, generator=g)).to(dev)
f3b, f4b, f5b = (torch.relu(x.roll((1, 2), (2, 3)) + 0.1 * torch.randn_like(x)) for x in (f3, f4, f5))
c3, c4, c5 = Correlation(8, 1, 8, 2, 2), Correlation(8, 1, 8, 1, 1), Correlation(8, 1, 8, 1, 1)
timeit("corr3 512ch 75x134 s2", lambda: c3(f3, f3b), bytes_=B * (41.16e6 + 0.82e6), flops=B * 0.211e9)
timeit("corr4 1024ch 38x67", lambda: c4(f4, f4b), bytes_=B * (20.86e6 + 2.94e6), flops=B * 1.507e9)
timeit("corr5 2048ch 38x67", lambda: c5(f5, f5b), bytes_=B * (41.71e6 + 2.94e6), flops=B * 3.014e9)

base = torch.from_numpy(generate_anchors(scales=(4, 8, 16, 32))).float()
A = base.size(0)
prob = torch.softmax(torch.randn(B, 2, A * H, W, generator=g) * 2, 1).view(B, 2 * A, H, W).to(dev)
bbox = (torch.randn(B, 4 * A, H, W, generator=g) * 0.4).to(dev)
info = torch.tensor([[600.0, 1067.0, 0.8333]] * B).to(dev)
timeit("proposal TEST 6000->300", lambda: proposal_forward(prob, bbox, info, base, 16, 6000, 300, 0.7))
timeit("proposal TRAIN 12000->2000", lambda: proposal_forward(prob, bbox, info, base, 16, 12000, 2000, 0.7))
rois, num = proposal_forward(prob, bbox, info, base, 16, 6000, 300, 0.7)
print("rois kept", num.tolist())
r = rois.view(-1, 5).contiguous()
cls = torch.randn(B, 31 * 49, H, W, generator=g).to(dev)
loc = torch.randn(B, 4 * 49, H, W, generator=g).to(dev)
pc, pl = _PSRoIPooling(7, 7, 1 / 16.0, 7, 31), _PSRoIPooling(7, 7, 1 / 16.0, 7, 4)
timeit("psroi cls 1519ch R=%d" % r.size(0), lambda: pc(cls, r), bytes_=B * (15.47e6 + 1.82e6))
timeit("psroi loc 196ch", lambda: pl(loc, r), bytes_=B * (2.0e6 + 0.235e6))
timeit("psroi cls + vote", lambda: psroi_pool_vote(cls, r, 7, 7, 1 / 16.0, 7, 31), bytes_=B * (15.47e6 + 1.82e6))
from dtt.ops import psroi_vote
timeit("psroi cls vote (channel-major)", lambda: psroi_vote(cls, r, 7, 7, 1 / 16.0, 7, 31), bytes_=B * (15.47e6 + 1.82e6))
clsg = cls.clone().requires_grad_(True)
out = pc(clsg, r)
go = torch.randn_like(out)
timeit("psroi cls backward", lambda: torch.autograd.grad(out, clsg, go, retain_graph=True))
top = torch.randn(B, 512, 36, 63, generator=g).to(dev)
ra = RoIAlignAvg(7, 7, 1 / 16.0)
with torch.no_grad():
    timeit("roialign avg 512ch", lambda: ra(top, r), bytes_=B * (4.64e6 + 30.1e6))
for n in (300, 6000, 12000):
    x = torch.rand(n, 2) * torch.tensor([1000.0, 550.0])
    wh = torch.rand(n, 2) * 150 + 16
    d = torch.cat([x, x + wh, torch.linspace(1, 0, n)[:, None]], 1).to(dev)
    timeit("nms N=%d (incl. host count read)" % n, lambda: nms(d, 0.7), iters=20)
f5g, f5bg = f5.clone().requires_grad_(True), f5b.clone().requires_grad_(True)
o = c5(f5g, f5bg)
go = torch.randn_like(o)
timeit("corr5 backward (simple)", lambda: torch.autograd.grad(o, (f5g, f5bg), go, retain_graph=True), iters=5, warm=1)
