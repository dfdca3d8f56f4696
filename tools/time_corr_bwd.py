#!/usr/bin/env python3
"""conv3 / conv4 / conv5 correlation GRADIENTS at the 600 px shapes on channels-last maps: the band-stationary streamed kernels
(dtt_correlation_backward_nhwc_strided) next to round 1's (dtt_correlation_backward_nhwc), values compared, both timed with
events around the whole op (developer tool, GPU box; run under rocprofv3 --kernel-trace for per-kernel times).
DTT_CORR_BWD_ABLATE (1 no DMA, 2 no MFMA, 4 no stores, 8 no band loads) applies to the streamed kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
from dtt import _lib
from dtt._lib import check, ptr, stream_ptr
from dtt.ops import correlation_backward_nhwc, correlation_output_shape
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 2))
ITERS = int(os.environ.get("ITERS", 30))
D = int(os.environ.get("D", 8))
SHAPE = os.environ.get("SHAPE", "600")   # "600": 600x1067 frames (38x67 / 75x134 maps); "563": 563x1000 (36x63 / 71x125, BASELINE configs[4])
H5, W5, H3, W3 = (38, 67, 75, 134) if SHAPE == "600" else (36, 63, 71, 125)
OLD = not os.environ.get("NO_OLD")
L = _lib.lib()


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
    print("%-58s %8.1f us  %6.1f TFLOP/s  %.3f of the fp32 MFMA peak" % (name, us, flops / us / 1e6, flops / us / 1e6 / 157.3), flush=True)


g = torch.Generator().manual_seed(3)
ONLY = os.environ.get("ONLY", "")      # e.g. ONLY=conv5: one map (per-kernel counter averages of a --pmc pass then belong to it)
for name, C, H, W, s in (("conv5", 2048, H5, W5, 1), ("conv4", 1024, H5, W5, 1), ("conv3", 512, H3, W3, 2)):
    if ONLY and name != ONLY:
        continue
    f1 = torch.relu(torch.randn(B, C, H, W, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
    f2 = torch.relu(f1.roll((1, 2), (2, 3)) + 0.1 * torch.randn(B, C, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
    oc, oh, ow = correlation_output_shape(C, H, W, D, 1, D, s, s)
    gout = torch.randn(B, oc, oh, ow, generator=g).to(dev)
    flops = 2 * 2.0 * C * oc * oh * ow * B      # both gradients
    a1, a2 = torch.empty_like(f1), torch.empty_like(f2)
    b1, b2 = torch.empty_like(f1), torch.empty_like(f2)

    def old():
        check(L.dtt_correlation_backward_nhwc(ptr(gout), B, oc, oh, ow, ptr(f1), C, H, W, ptr(f2), ptr(b1), ptr(b2), D, 1, D, s, s,
                                              stream_ptr(dev)), "round-1 backward")
    timeit("%s gradients, streamed (band + 2 launches)" % name, lambda: correlation_backward_nhwc(gout, f1, f2, a1, a2, D, 1, D, s, s), flops)
    if not OLD:
        continue
    timeit("%s gradients, round 1 (2 launches)" % name, old, flops)
    if not os.environ.get("DTT_CORR_BWD_ABLATE"):
        print("   max |diff| between the two: %.2e / %.2e (max |g| %.2e)" % (float((a1 - b1).abs().max()), float((a2 - b2).abs().max()),
                                                                            float(b1.abs().max())))
