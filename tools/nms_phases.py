#!/usr/bin/env python3
"""Shader cycles of nms_sweep_kernel's phases (set-up / A staging / B serial walk / C kept rows over later columns) on the training
proposal layer (12000 -> 2000, B images), per image and per launch (first / second phase), on the -DDTT_NMS_TRACE build:
    DTT_HIP_LIBRARY=tools/_variants/nmstrace.so python tools/nms_phases.py        (BBOX_STD=0.05: heavily overlapping proposals)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
from dtt import _lib
from dtt.rpn import generate_anchors, proposal_forward
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 4)); H, W = 38, 67
g = torch.Generator(device="cpu").manual_seed(3)
base = torch.from_numpy(generate_anchors(scales=(4, 8, 16, 32))).float().to(dev)
A = base.size(0)
std = float(os.environ.get("BBOX_STD", 0.4))
prob = torch.softmax(torch.randn(B, 2, A * H, W, generator=g) * 2, 1).view(B, 2 * A, H, W).to(dev)
bbox = (torch.randn(B, 4 * A, H, W, generator=g) * std).to(dev)
info = torch.tensor([[600.0, 1067.0, 0.8333]] * B).to(dev)
L = _lib.lib()
fn = lambda: proposal_forward(prob, bbox, info, base, 16, 12000, 2000, 0.7)
for _ in range(3):
    out = fn()
buf = (ctypes.c_ulonglong * 512)()
L.dtt_nms_cycles_read(buf, 1)
N = 10
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(N):
    out = fn()
e.record(); torch.cuda.synchronize()
L.dtt_nms_cycles_read(buf, 0)
print("BBOX_STD=%g  kept %s  proposal layer %.1f us per call" % (std, out[1].tolist(), s.elapsed_time(e) * 1e3 / N))
for ph in (0, 1):
    for b in range(B):
        v = [buf[((32 if ph else 0) + b) * 8 + k] / N for k in range(4)]
        w = [buf[((32 if ph else 0) + b) * 8 + k] / N for k in range(4, 8)]
        print("phase %d image %d: set-up %7.0f  A %7.0f  B %7.0f  C %7.0f  total %7.0f cycles   inside B: settle %7.0f  row-OR %7.0f  fixpoint iterations %5.0f" % (
            ph + 1, b, v[0], v[1], v[2], v[3], sum(v), w[0], w[2], w[3]))
