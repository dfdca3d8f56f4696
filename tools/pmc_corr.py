#!/usr/bin/env python3
"""Launch only the correlation ops at the 600 px D&T shapes (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
from dtt.ops import Correlation, _PSRoIPooling
dev = torch.device("cuda:0"); B = 2
g = torch.Generator().manual_seed(3)
f5 = torch.relu(torch.randn(B, 2048, 38, 67, generator=g)).to(dev); f5b = torch.relu(torch.randn(B, 2048, 38, 67, generator=g)).to(dev)
f4 = torch.relu(torch.randn(B, 1024, 38, 67, generator=g)).to(dev); f4b = torch.relu(torch.randn(B, 1024, 38, 67, generator=g)).to(dev)
c = Correlation(8, 1, 8, 1, 1)
cls = torch.randn(B, 1519, 38, 67, generator=g).to(dev)
rois = torch.cat([torch.randint(0, B, (600, 1)).float(), torch.rand(600, 2) * 500, torch.rand(600, 2) * 400 + 550], 1).to(dev)
rois[:, 3] = rois[:, 1] + 200; rois[:, 4] = rois[:, 2] + 150
p = _PSRoIPooling(7, 7, 1 / 16.0, 7, 31)
for _ in range(5):
    c(f5, f5b); c(f4, f4b); p(cls, rois)
torch.cuda.synchronize()
