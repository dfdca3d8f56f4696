#!/usr/bin/env python3
"""The RPN's two 1x1 heads + pairwise softmax as one launch (dtt_rpn_head_gemm) at the 600 px shape, both legs of two frame
pairs (developer tool, GPU box; run under rocprofv3 --kernel-trace for the kernel time; DTT_HEAD_KSPLIT=0 selects the
tile-per-wave form)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
import torch.nn.functional as F
from dtt.heads import PackedRPNHeads, rpn_head_gemm

dev = torch.device("cuda:0")
B, H, W, K, A = int(os.environ.get("B", 4)), 38, 67, 512, 12
ITERS = int(os.environ.get("ITERS", 50))
g = torch.Generator().manual_seed(5)
cls, box = torch.nn.Conv2d(K, 2 * A, 1).to(dev), torch.nn.Conv2d(K, 4 * A, 1).to(dev)
x = torch.relu(torch.randn(B, K, H, W, generator=g)).to(dev)
rows = x.permute(0, 2, 3, 1).reshape(B * H * W, K).contiguous()
packed = PackedRPNHeads(cls, box)
with torch.no_grad():
    for _ in range(10):
        prob, bbox = rpn_head_gemm(rows, packed, B, H, W)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS):
        prob, bbox = rpn_head_gemm(rows, packed, B, H, W)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / ITERS
    flops = 2.0 * B * H * W * K * 6 * A
    print("rpn_head_gemm  %d x %d x %d: %.1f us per call (back to back)  %.1f TFLOP/s" % (B * H * W, K, 6 * A, us, flops / us / 1e6))
    score = F.conv2d(x.double(), cls.weight.double(), cls.bias.double())
    want = F.softmax(score.view(B, 2, A * H, W), dim=1).view(B, 2 * A, H, W)
    print("max |prob - float64 conv2d + softmax| = %.2e" % float((prob.double() - want).abs().max()))
