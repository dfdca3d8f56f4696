#!/usr/bin/env python3
"""Timeline of two workgroups of psroi_pm_bwd_rows_kernel from the shader-clock stamps of the DTT_PSROI_BWD_STAMP build (developer tool):
the whole library with psroi_bwd.hip compiled -DDTT_PSROI_BWD_STAMP into tools/_variants/pbstamp.so (the Makefile's flags, the other
objects from csrc/build/), then   DTT_HIP_LIBRARY=tools/_variants/pbstamp.so python tools/psroi_bwd_timeline.py
(tools/probes/psroi_bwd_instep_clock.py reads the same stamps behind a `bench.py --mode train` run: the launch inside the step.)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd"), os.path.join(ROOT, "tools")]
import numpy as np
import torch
from dtt import _lib
from dtt._lib import check, ptr, stream_ptr
from time_psroi_bwd import rois_like_training
dev = torch.device("cuda:0")
L = _lib.lib()
rng = np.random.RandomState(0)
B, H, W, stride, per = 4, 38, 67, 1792, 128
R = B * per
rois = torch.from_numpy(rois_like_training(rng, per, B, H, W)).to(dev)
g_cls = torch.from_numpy(rng.normal(size=(R, 31)).astype(np.float32)).to(dev)
g_loc = torch.from_numpy(rng.normal(size=(R, 4)).astype(np.float32)).to(dev)
add = torch.from_numpy(rng.normal(size=(B * H * W, 196)).astype(np.float32)).to(dev)
gm = torch.empty((B * H * W, stride), device=dev)
with torch.cuda.device(dev):
    for _ in range(int(os.environ.get("ITERS", "300"))):
        check(L.dtt_psroi_pm_backward_heads(ptr(g_cls), 31, 32, ptr(g_loc), 4, 4, ptr(rois), R, B, H, W, 7, 1 / 16.0, stride, stride,
                                            ptr(add), 1568, 196, ptr(gm), stream_ptr(dev)), "heads")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
L.dtt_psroi_bwd_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.dtt_psroi_bwd_stamps_read(buf, 256)
st = np.array(buf, dtype=np.uint64).reshape(4, 64).astype(np.int64)
names = {44: "run scan issued", 40: "  gradient rows landed (first batch)", 41: "  gradient rows in LDS", 42: "  RoI floats landed", 43: "  bin edges computed", 45: "barrier 2 passed", 0: "entry", 1: "init done (sync 1)", 2: "run scan + staging issued", 3: "sync 2", 4: "column edges in registers", 50: "end"}
for it in range(4):
    names[5 + 6 * it] = "px %d tests done" % it
    names[6 + 6 * it] = "px %d listed" % it
    names[7 + 6 * it] = "px %d walked" % it
    names[8 + 6 * it] = "px %d compact gradient added" % it
    names[9 + 6 * it] = "px %d row stored" % it
print("entries walked by wave 0 of workgroup 0 per pixel:", [int(x) for x in st[0, 56:60]])
for row in (0, 2):
    ticks, real = st[row, 50] - st[row, 0], st[row, 63] - st[row, 62]
    print("workgroup %s: %d shader-clock ticks in %d ticks of the 100 MHz clock = %.2f us -> %.0f MHz" % ("0" if row == 0 else "middle", ticks, real, real / 100.0, ticks / max(real, 1) * 100.0))
for row, label in enumerate(["workgroup 0 wave 0", "workgroup 0 last wave", "middle workgroup wave 0", "middle workgroup last wave"]):
    t0 = st[row, 0]
    print("== %s (shader-clock ticks since entry, step)" % label)
    prev = t0
    for i in sorted(names, key=lambda i: st[row, i]):
        if st[row, i]:
            print("  %-32s %8d  +%6d" % (names[i], st[row, i] - t0, st[row, i] - prev))
            prev = st[row, i]

wg = (ctypes.c_ulonglong * (1024 * 3))()
L.dtt_psroi_bwd_wg_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.dtt_psroi_bwd_wg_read(wg, 1024 * 3)
w = np.array(wg, dtype=np.uint64).reshape(1024, 3)
w = w[w[:, 0] > 0]
t0 = int(w[:, 0].min())
start, end = (w[:, 0].astype(np.int64) - t0) / 100.0, (w[:, 1].astype(np.int64) - t0) / 100.0
xcc = (w[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
print("== %d workgroups of the last launch (us since the first one entered): entry min %.2f median %.2f max %.2f   end min %.2f median %.2f max %.2f"
      % (len(w), start.min(), np.median(start), start.max(), end.min(), np.median(end), end.max()))
order = np.argsort(start)
print("entry times, sorted:", " ".join("%.1f" % x for x in start[order][::8]))
print("lifetime us: min %.2f median %.2f max %.2f" % ((end - start).min(), np.median(end - start), (end - start).max()))
for x in range(8):
    m = xcc == x
    if m.any():
        print("XCC %d: %d workgroups, entry %.2f .. %.2f, end %.2f .. %.2f" % (x, m.sum(), start[m].min(), start[m].max(), end[m].min(), end[m].max()))
