#!/usr/bin/env python3
"""head_gemm alone at the 600 px shape (developer tool; knobs: DTT_HEAD_NLOAD, DTT_HEAD_ABLATE, PASSES)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
from dtt.heads import PackedHeads, head_gemm
dev = torch.device("cuda:0")
B, H, W = int(os.environ.get("B", 4)), 38, 67
g = torch.Generator().manual_seed(3)
cls = torch.nn.Conv2d(512, 31 * 49, 1).to(dev); loc = torch.nn.Conv2d(512, 4 * 49, 1).to(dev)
rows = torch.relu(torch.randn(B * H * W, 512, generator=g)).to(dev)
both = PackedHeads([cls, loc])
out = torch.empty((rows.shape[0], both.stride), device=dev)
P = int(os.environ.get("PASSES", 0))
for _ in range(5):
    head_gemm(rows, both, out=out, passes=P)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = int(os.environ.get("ITERS", 30))
s.record()
for _ in range(n):
    head_gemm(rows, both, out=out, passes=P)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / n
fl = 2.0 * rows.shape[0] * 512 * (1519 + 196)
print("head_gemm cls+loc  NLOAD=%s ABLATE=%s PASSES=%d : %.1f us  %.1f TFLOP/s" % (os.environ.get("DTT_HEAD_NLOAD", "2"), os.environ.get("DTT_HEAD_ABLATE", "0"), P, us, fl / us / 1e6))
