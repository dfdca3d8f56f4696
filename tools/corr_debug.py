#!/usr/bin/env python3
"""Test-side developer script (it uses the oracle, so it lives under tests/): locate forward-correlation mismatches against the oracle for one case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np, torch
from oracle import oracle_lib as O
from dtt.ops import Correlation
case = tuple(int(v) for v in sys.argv[1:10]) if len(sys.argv) >= 10 else (1, 16, 13, 19, 4, 1, 4, 1, 1)
B, C, H, W, pad, k, d, s1, s2 = case
rng = np.random.RandomState(sum(case))
x1 = np.maximum(rng.normal(size=(B, C, H, W)), 0).astype(np.float32)
x2 = np.maximum(np.roll(x1, (1, -2), (2, 3)) + 0.3 * rng.normal(size=x1.shape), 0).astype(np.float32)
ref = O.correlation_forward(x1, x2, pad, k, d, s1, s2)
dev = torch.device("cuda:0")
out = Correlation(pad, k, d, s1, s2)(torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev)).cpu().numpy()
bad = np.abs(out - ref) > 1e-4
print("case", case, "bad", bad.sum(), "of", bad.size)
D = int(round(ref.shape[1] ** 0.5))
b = bad.reshape(B, D, D, ref.shape[2], ref.shape[3])
print("bad by image", b.sum((1, 2, 3, 4)))
print("bad by dy", b.sum((0, 2, 3, 4)))
print("bad by dx", b.sum((0, 1, 3, 4)))
print("bad by y", b.sum((0, 1, 2, 4)))
print("bad by x", b.sum((0, 1, 2, 3)))
idx = np.argwhere(b)[:10]
for i in idx:
    print(tuple(i), "got", out.reshape(b.shape)[tuple(i)], "ref", ref.reshape(b.shape)[tuple(i)])
