#!/usr/bin/env python3
"""conv3 / conv4 / conv5 correlation at the 600 px shapes: NCHW op (banded product + reduce kernels) next to the
channels-last single-launch kernel (developer tool, GPU box; run under rocprofv3 --kernel-trace for kernel times)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
from dtt.ops import Correlation, correlation_forward_nhwc
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 2))
ITERS = int(os.environ.get("ITERS", 30))
D = int(os.environ.get("D", 8))


def timeit(name, fn, flops):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS):
        fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / ITERS
    print("%-46s %8.1f us  %6.1f TFLOP/s" % (name, us, flops / us / 1e6), flush=True)


g = torch.Generator().manual_seed(3)
for name, C, H, W, s in (("conv5", 2048, 38, 67, 1), ("conv4", 1024, 38, 67, 1), ("conv3", 512, 75, 134, 2)):
    f1 = torch.relu(torch.randn(B, C, H, W, generator=g)).to(dev)
    f2 = torch.relu(f1.roll((1, 2), (2, 3)) + 0.1 * torch.randn(B, C, H, W, device=dev))
    c1, c2 = f1.contiguous(memory_format=torch.channels_last), f2.contiguous(memory_format=torch.channels_last)
    R = D // s
    flops = 2.0 * C * (2 * R + 1) ** 2 * 38 * 67 * B
    layer = Correlation(D, 1, D, s, s)
    with torch.no_grad():
        timeit("%s NCHW op (2 kernels)" % name, lambda: layer(f1, f2), flops)
        timeit("%s channels-last (1 kernel)" % name, lambda: correlation_forward_nhwc(c1, c2, D, 1, D, s, s), flops)
        a, b = layer(f1, f2), correlation_forward_nhwc(c1, c2, D, 1, D, s, s)
        print("   max |diff| between the two: %.2e" % float((a - b).abs().max()))
