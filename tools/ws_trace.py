#!/usr/bin/env python3
"""Phase timeline of the window-split correlation kernel (developer tool, GPU box): runs the conv5 shape with the
-DDTT_WS_TRACE build (tools/build_ws_trace.sh; DTT_HIP_LIBRARY=tools/_variants/wstrace.so) and prints, over all workgroups,
the shader cycles between the stamps of compute wave 0 (chunk phase 0), compute wave 4 (phase 1) and loader wave 8."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np
import torch
from dtt import _lib
from dtt.ops import correlation_forward_nhwc
dev = torch.device("cuda:0")
C = int(os.environ.get("C", 2048)); MW = int(os.environ.get("MW", 0))
g = torch.Generator().manual_seed(3)
f1 = torch.relu(torch.randn(2, C, 38, 67, generator=g)).to(dev)
f2 = torch.relu(f1.roll((1, 2), (2, 3)) + 0.1 * torch.randn(2, C, 38, 67, device=dev))
c1, c2 = f1.contiguous(memory_format=torch.channels_last), f2.contiguous(memory_format=torch.channels_last)
rows = torch.zeros(2 * 38 * 67, 296, device=dev)
for _ in range(5):
    correlation_forward_nhwc(c1, c2, 8, 1, 8, 1, 1, rows=rows, col=0, max_workgroups=MW)
torch.cuda.synchronize()
L = _lib.lib()
n = 512 * 3 * 16
buf = (ctypes.c_ulonglong * n)()
L.dtt_ws_trace_read.restype = ctypes.c_int
assert L.dtt_ws_trace_read(buf, n)
t = np.array(buf, dtype=np.uint64).reshape(512, 3, 16).astype(np.float64)
nwg = int((t[:, 0, 0] > 0).sum())
t = t[:nwg]
names = ["start", "setup done", "loop done", "E1 passed", "exchange done (E3)", "emit done", "E4 passed", "write-out done"]
wall0 = t[:, 0, 15].min()
print("%d workgroups; start spread %.2f us (100 MHz wall clock)" % (nwg, (t[:, 0, 15].max() - wall0) / 100.0))
for w, wn in enumerate(("compute wave 0 (phase 0)", "compute wave 4 (phase 1)", "loader wave 8")):
    print(wn)
    for i in range(1, 8):
        d = t[:, w, i] - t[:, w, i - 1]
        ok = (t[:, w, i] > 0) & (t[:, w, i - 1] > 0)
        if ok.any():
            print("   %-22s -> %-22s  cycles  min %8.0f  median %8.0f  max %8.0f" % (names[i - 1], names[i], d[ok].min(), np.median(d[ok]), d[ok].max()))
    tot = t[:, w, 7] - t[:, w, 0]
    print("   total cycles  min %.0f  median %.0f  max %.0f" % (tot.min(), np.median(tot), tot.max()))
