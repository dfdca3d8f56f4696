#!/usr/bin/env python3
"""Timing of the hand-written R-FCN heads (csrc/heads.hip) at the 600 px D&T shapes, next to the library path they
replace (developer tool, GPU box).  B = images in the batch (4 = both legs of two frame pairs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np
import torch
import torch.nn.functional as F
from dtt.heads import PackedHeads, head_gemm, psroi_pm, pm_to_nchw
from dtt.ops import psroi_vote
from dtt.rpn import generate_anchors, proposal_forward

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 4))
H, W = int(os.environ.get("H", 38)), int(os.environ.get("W", 67))
ITERS = int(os.environ.get("ITERS", 50))


def timeit(name, fn, iters=ITERS, warm=10, bytes_=None, flops=None):
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
    print("%-44s %9.1f us%s" % (name, us, extra), flush=True)
    return us


g = torch.Generator().manual_seed(3)
cls = torch.nn.Conv2d(512, 31 * 49, 1).to(dev)
loc = torch.nn.Conv2d(512, 4 * 49, 1).to(dev)
torch.nn.init.normal_(cls.weight, 0, 0.01); torch.nn.init.normal_(loc.weight, 0, 0.01)
x = torch.relu(torch.randn(B, 512, H, W, generator=g)).to(dev)
x_cl = x.contiguous(memory_format=torch.channels_last)
rows = x.permute(0, 2, 3, 1).reshape(-1, 512).contiguous()
M = rows.shape[0]
both = PackedHeads([cls, loc])
only_cls = PackedHeads([cls])
only_loc = PackedHeads([loc])
fl_cls, fl_loc = 2.0 * M * 512 * 1519, 2.0 * M * 512 * 196
with torch.no_grad():
    timeit("library conv2d cls (NCHW)", lambda: cls(x), flops=fl_cls)
    timeit("library conv2d loc (NCHW)", lambda: loc(x), flops=fl_loc)
    timeit("library conv2d cls (channels-last)", lambda: cls(x_cl), flops=fl_cls)
    for p in (1, 2, 3, 4):
        timeit("head_gemm cls+loc passes=%d" % p, lambda: head_gemm(rows, both, passes=p), flops=fl_cls + fl_loc)
    out = torch.empty((M, only_cls.stride), device=dev)
    for p in (1, 2, 4):
        timeit("head_gemm cls passes=%d" % p, lambda: head_gemm(rows, only_cls, out=out, passes=p), flops=fl_cls)
    timeit("head_gemm loc (narrow config)", lambda: head_gemm(rows, only_loc), flops=fl_loc)

    # RoIs from the proposal layer on random RPN outputs (the shape the pipeline pools: 300 per image)
    rng = np.random.RandomState(0)
    base = torch.from_numpy(generate_anchors(scales=(4, 8, 16, 32))).float()
    A = base.shape[0]
    prob = torch.softmax(torch.randn(B, 2, A * H, W, generator=g) * 2, 1).view(B, 2 * A, H, W).to(dev)
    bbox = (torch.randn(B, 4 * A, H, W, generator=g) * 0.4).to(dev)
    info = torch.tensor([[H * 16.0, W * 16.0, 1.0]] * B, device=dev)
    rois, _ = proposal_forward(prob, bbox, info, base, 16, 6000, 300, 0.7)
    rois = rois.view(-1, 5).contiguous()
    R = rois.shape[0]
    pm = head_gemm(rows, both)
    nchw_cls = pm_to_nchw(pm, both.heads[0], B, H, W)
    nchw_loc = pm_to_nchw(pm, both.heads[1], B, H, W)
    by_cls = B * 1519 * H * W * 4 + R * 31 * 4
    by_loc = B * 196 * H * W * 4 + R * 4 * 4
    timeit("psroi_vote cls (plane kernel, NCHW)", lambda: psroi_vote(nchw_cls, rois, 7, 7, 1 / 16.0, 7, 31), bytes_=by_cls)
    timeit("psroi_pm  cls (position-major)", lambda: psroi_pm(pm, both.heads[0], B, H, W, rois, 1 / 16.0), bytes_=by_cls)
    timeit("psroi_vote loc (plane kernel, NCHW)", lambda: psroi_vote(nchw_loc, rois, 7, 7, 1 / 16.0, 7, 4), bytes_=by_loc)
    timeit("psroi_pm  loc (position-major)", lambda: psroi_pm(pm, both.heads[1], B, H, W, rois, 1 / 16.0), bytes_=by_loc)
    a = psroi_pm(pm, both.heads[0], B, H, W, rois, 1 / 16.0)
    b = psroi_vote(nchw_cls, rois, 7, 7, 1 / 16.0, 7, 31)
    print("cls vote identical to plane kernel:", bool(torch.equal(a, b)))
    ref = cls(x)
    print("cls head max |err| vs conv2d:", float((nchw_cls - ref).abs().max()))
