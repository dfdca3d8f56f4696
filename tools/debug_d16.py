#!/usr/bin/env python3
"""Isolate the configs[4] (d = 16) training-graph difference: correlation gradients rows / planes / oracle at full channel counts,
the tracking head's dX GEMM at K_in = 2880 (developer tool, GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np, torch
from dtt import _lib
from dtt._lib import check, ptr, stream_ptr
from dtt.ops import correlation_backward_nhwc, correlation_output_shape
from oracle import oracle_lib as O
dev = torch.device("cuda:0")
rng = np.random.RandomState(5)
cl = lambda a: torch.from_numpy(a).to(dev).contiguous(memory_format=torch.channels_last)
for (B, C, H, W, pad, d, s) in [(1, 1024, 36, 63, 16, 16, 1), (1, 2048, 36, 63, 16, 16, 1), (1, 512, 71, 125, 16, 16, 2), (2, 1024, 38, 67, 8, 8, 1)]:
    x1 = np.maximum(rng.normal(size=(B, C, H, W)), 0).astype(np.float32); x2 = np.maximum(rng.normal(size=(B, C, H, W)), 0).astype(np.float32)
    t1, t2 = cl(x1), cl(x2)
    oc, oh, ow = correlation_output_shape(C, H, W, pad, 1, d, s, s)
    gout = rng.normal(size=(B, oc, oh, ow)).astype(np.float32)
    gt = torch.from_numpy(gout).to(dev)
    ga, gb = torch.full_like(t1, float("nan")), torch.full_like(t2, float("nan"))
    correlation_backward_nhwc(gt, t1, t2, ga, gb, pad, 1, d, s, s)
    ld, col = oc + 40, 392 % 8 * 0 + 8
    rows = torch.zeros((B * oh * ow, ld), device=dev)
    rows[:, col:col + oc] = gt.permute(0, 2, 3, 1).reshape(-1, oc)
    ra, rb = torch.full_like(t1, float("nan")), torch.full_like(t2, float("nan"))
    correlation_backward_nhwc(None, t1, t2, ra, rb, pad, 1, d, s, s, rows=rows, col=col)
    g1, g2 = O.correlation_backward(gout, x1, x2, pad, 1, d, s, s)
    for name, a, r in (("planes g1", ga, g1), ("planes g2", gb, g2), ("rows   g1", ra, g1), ("rows   g2", rb, g2)):
        a = a.cpu().numpy()
        print("C=%d %dx%d d=%d s=%d  %s: max|diff| %.3e  rel %.3e  nan %d" % (C, H, W, d, s, name, np.nanmax(np.abs(a - r)),
              np.linalg.norm(np.nan_to_num(a - r)) / np.linalg.norm(r), int(np.isnan(a).sum())), flush=True)
# the tracking head's dX: gout (M, 224) @ wt.T with wt (2880, 224)
L = _lib.lib()
for (M, K, stride) in [(2268, 2880, 224), (5092, 1056, 224)]:
    gout = torch.randn(M, stride, device=dev); gout[:, 208:] = 0
    wt = torch.randn(K, stride, device=dev) * 0.01; wt[:, 196:] = 0
    gx = torch.full((M, K), float("nan"), device=dev)
    check(L.dtt_head_gemm(ptr(gout), stride, M, stride, ptr(wt), ptr(torch.zeros(K, device=dev)), K, ptr(gx), K, K, 0, stream_ptr(dev)), "dX")
    ref = (gout.double() @ wt.double().t())
    print("head dX M=%d K=%d: max|diff| %.3e rel %.3e nan %d" % (M, K, float((gx.double() - ref).abs().max()), float((gx.double() - ref).norm() / ref.norm()),
          int(torch.isnan(gx).sum())), flush=True)
