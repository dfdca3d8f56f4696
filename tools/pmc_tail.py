#!/usr/bin/env python3
"""Launch only the hand-written kernels of the inference tail at the 600 px D&T shapes (B = 2), for rocprofv3 --pmc passes:
channels-last correlation conv5 / conv4 / conv3, the class + box head GEMM, the tracking head GEMM, the three
position-major PSRoI poolings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
from dtt.config import cfg
from dtt.synth import build_model
from dtt.fuse import fuse_for_inference
from dtt.heads import head_gemm, psroi_pm, psroi_pm_det
from dtt.ops import correlation_forward_nhwc
dev = torch.device("cuda:0"); B = 2; H, W = 38, 67
g = torch.Generator().manual_seed(3)
cl = lambda c, h, w: torch.relu(torch.randn(B, c, h, w, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
maps = [(cl(2048, H, W), cl(2048, H, W), 1), (cl(1024, H, W), cl(1024, H, W), 1), (cl(512, 75, 134), cl(512, 75, 134), 2)]
model = build_model(101, cfg=cfg, seed=0).to(dev).eval()
fuse_for_inference(model, channels_last=True)
pm = model._pm_tail
top = torch.randn(2 * B * H * W, 512, generator=g).to(dev)
rows = pm.tracking_rows(B * H * W, dev)
rois = torch.cat([torch.randint(0, 2 * B, (1200, 1)).float(), torch.rand(1200, 2) * 500, torch.zeros(1200, 2)], 1).to(dev)
rois[:, 3] = rois[:, 1] + 200; rois[:, 4] = rois[:, 2] + 150
with torch.no_grad():
    for _ in range(5):
        col = 2 * pm.n_box
        for (a, b, s) in maps:      # conv5, conv4, conv3 (tools/pmc_conv5_json.py relies on this order)
            correlation_forward_nhwc(a, b, 8, 1, 8, s, s)
        det = head_gemm(top, pm.det)
        trk = head_gemm(rows, pm.trk)
        psroi_pm_det(det, pm.cls_head, pm.loc_head, 2 * B, H, W, rois, 1 / 16.0)      # class scores + box deltas + softmax: one launch
        psroi_pm(trk, pm.trk_head, B, H, W, rois[rois[:, 0] < B][:600].contiguous(), 1 / 16.0)
torch.cuda.synchronize()
