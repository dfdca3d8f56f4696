#!/usr/bin/env python3
"""Probe: MIOpen fused conv+bias+ReLU on channels-last tensors vs conv + separate epilogue (3x3 trunk shapes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch, torch.nn.functional as F
from dtt.fuse import bias_act_nhwc_, _rows
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (N, C, H, W, K, dil) in [(4, 64, 150, 267, 64, 1), (4, 128, 75, 134, 128, 1), (4, 256, 38, 67, 256, 1), (4, 512, 38, 67, 512, 2), (4, 2048, 38, 67, 512, 6)]:
    x = torch.relu(torch.randn(N, C, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, 3, 3, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.randn(K, device=dev)
    def sep():
        y = F.conv2d(x, w, None, 1, dil, dil)
        bias_act_nhwc_(_rows(y), b)
        return y
    t_sep = timeit(sep)
    t_conv = timeit(lambda: F.conv2d(x, w, None, 1, dil, dil))
    try:
        y2 = torch.miopen_convolution_relu(x, w, b, (1, 1), (dil, dil), (dil, dil), 1)
        t_f = timeit(lambda: torch.miopen_convolution_relu(x, w, b, (1, 1), (dil, dil), (dil, dil), 1))
        err = float((y2 - sep()).abs().max())
        cl = y2.is_contiguous(memory_format=torch.channels_last)
    except Exception as e:
        t_f, err, cl = float("nan"), str(e)[:60], None
    print("C=%4d K=%4d %3dx%3d dil %d: conv %7.1f  conv+ep %7.1f  miopen fused %7.1f us  (maxdiff %s, channels_last out %s)" % (C, K, H, W, dil, t_conv, t_sep, t_f, err, cl), flush=True)
