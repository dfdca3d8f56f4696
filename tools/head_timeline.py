#!/usr/bin/env python3
"""Timeline of workgroup 0 of head_gemm from the shader-clock stamps of the DTT_HEAD_STAMP build
(tools/_variants/headstamp.so; developer tool):  DTT_HIP_LIBRARY=tools/_variants/headstamp.so python tools/head_timeline.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np
import torch
from dtt import _lib
from dtt.heads import PackedHeads, head_gemm
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
cls = torch.nn.Conv2d(512, 31 * 49, 1).to(dev); loc = torch.nn.Conv2d(512, 4 * 49, 1).to(dev)
rows = torch.relu(torch.randn(4 * 38 * 67, 512, generator=g)).to(dev)
both = PackedHeads([cls, loc])
out = torch.empty((rows.shape[0], both.stride), device=dev)
for _ in range(3):
    head_gemm(rows, both, out=out)
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * (8 * 512))()
L.dtt_head_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.dtt_head_stamps_read(buf, 8 * 512)
st = np.array(buf, dtype=np.uint64).reshape(8, 512).astype(np.int64)
t0 = st[0, 500]
print("compute waves: step  arrive(w0)  barrier-wait w0 w1 w2 w3 | step length (w0)")
prev = None
for s in range(40):
    a = st[:4, 2 * s]; b = st[:4, 2 * s + 1]
    if a[0] == 0:
        continue
    line = "step %2d  arrive %8d  wait %5d %5d %5d %5d" % (s, a[0] - t0, b[0] - a[0], b[1] - a[1], b[2] - a[2], b[3] - a[3])
    if prev is not None:
        line += "   | %6d" % (a[0] - prev)
    prev = a[0]
    print(line)
for p in range(3):
    if st[0, 400 + 4 * p]:
        print("pass %d: MFMAs issued at %d, stores issued at %d (+%d)" % (p, st[0, 400 + 4 * p] - t0, st[0, 401 + 4 * p] - t0, st[0, 401 + 4 * p] - st[0, 400 + 4 * p]))
print("loader waves (w4..): step  dma-issued  landed(+)  barrier-released(+)")
for s in range(34):
    row = []
    for w in range(4, 8):
        if st[w, 3 * s]:
            row.append("w%d %8d +%5d +%5d" % (w, st[w, 3 * s] - t0, st[w, 3 * s + 1] - st[w, 3 * s], st[w, 3 * s + 2] - st[w, 3 * s + 1]))
    if row:
        print("step %2d  " % s + "  ".join(row))
