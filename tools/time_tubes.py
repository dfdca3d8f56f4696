#!/usr/bin/env python3
"""Test-side developer script (it uses the oracle, so it lives under tests/): time tube linking at a production-like size (30 classes x 300 frames, 300 detections and 300
tracklets per frame) on the GPU, and the numpy oracle on one class for scale."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from dtt.tubes import link_tubes
from oracle import tubes_oracle as to
from test_gpu_tubes import _random_video
dev = torch.device("cuda:0")
P, F, M = int(os.environ.get("P", 30)), int(os.environ.get("F", 300)), 300
rs = np.random.RandomState(0)
vids = [_random_video(rs, F, 150, 225, M, 3000, 0.0) for _ in range(P)]
nm = max(v[0].shape[1] for v in vids)
D = np.zeros((P, F, nm, 6), np.float32); N = np.zeros((P, F), np.int32); Tk = np.zeros((P, F, 2, M, 4), np.float32); Mc = np.full((P, F), -1, np.int32)
for p, (d, n, t, m) in enumerate(vids):
    D[p, :, :d.shape[1]] = d; N[p] = n; Tk[p] = t; Mc[p] = m
Dd, Td = torch.from_numpy(D).to(dev), torch.from_numpy(Tk).to(dev)
Nn, Mm = torch.from_numpy(N), torch.from_numpy(Mc)
for _ in range(2):
    out = link_tubes(Dd, Nn, Td, Mm)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    out = link_tubes(Dd, Nn, Td, Mm)
torch.cuda.synchronize()
gpu = (time.time() - t0) / 5
print("GPU: %d classes x %d frames, <=%d dets, %d tracklets: %.2f ms per video (paths per class: %s)" % (P, F, nm, M, gpu * 1e3, out[5][:6].tolist()))
t0 = time.time()
to.make_tubes(*vids[0])
cpu = time.time() - t0
print("oracle (numpy, 1 class): %.2f s -> %.1f s for %d classes; GPU/CPU = %.0fx" % (cpu, cpu * P, P, cpu * P / gpu))
