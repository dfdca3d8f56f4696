#!/usr/bin/env python3
"""Sizes of the RoIs the bench step pools (developer tool): the synthetic model's 300 proposals per image, in map pixels per bin."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
from dtt.config import apply_dataset_defaults, cfg
from dtt.fuse import fuse_for_inference
from dtt.synth import build_model, calibrate_batchnorm_, make_batch
apply_dataset_defaults("imagenet_vid")
dev = torch.device("cuda:0")
model = build_model(101, cfg=cfg).to(dev)
im, info, gt, nb = make_batch(2, 600, 1067, seed=3, device=dev)
calibrate_batchnorm_(model, im[:, 0]); model.eval(); fuse_for_inference(model, channels_last=True)
with torch.no_grad():
    out = model(im, info, gt, nb)
r = out[0].reshape(-1, 5).float()
w, h = (r[:, 3] - r[:, 1] + 1) / 16, (r[:, 4] - r[:, 2] + 1) / 16
area_bin = (torch.ceil(w / 7) + 1) * (torch.ceil(h / 7) + 1)
q = lambda t: [round(float(v), 1) for v in torch.quantile(t, torch.tensor([0.1, 0.5, 0.9, 0.99, 1.0], device=t.device))]
print("RoIs %d  width (map px) q10/50/90/99/max %s  height %s  pixels per bin (upper bound) %s" % (r.shape[0], q(w), q(h), q(area_bin)))
