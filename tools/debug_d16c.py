#!/usr/bin/env python3
"""Run-to-run and graph-to-graph differences of the trunk-map gradients at the configs[4] shape (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
if os.environ.get("DET"):
    torch.backends.cudnn.deterministic = True
if os.environ.get("BENCH"):
    torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
import test_gpu_train_fullsize as T
from dtt.fuse import fuse_for_training
from dtt.synth import build_model, calibrate_batchnorm_, make_batch
H, W, B, disp, roi = 563, 1000, 1, 16, "align"
c = T._cfg_for(disp, roi)
model = build_model(101, cfg=c).to(dev)
im, info, gt, nb = make_batch(B, H, W, seed=7, device=dev)
calibrate_batchnorm_(model, im[:, 0])
model.train(); fuse_for_training(model, channels_last=True)
model.RFCN_rpn.proposals = T._fixed_proposals(H, W, dev)
spy = T._MapGrads(model)
c.TRAIN.SAMPLER_RNG = "reference"
def run(pm, which=(7,)):
    model._train_pm = pm
    model.zero_grad(set_to_none=True)
    np.random.seed(99); torch.manual_seed(99)
    out = model(im, info, gt, nb)
    sum(out[i].mean() for i in which).backward()
    torch.cuda.synchronize()
    g = spy.grads()
    g["maps_fwd"] = [m.detach().clone() for m in spy.maps]
    return g
def cmp(tag, a, b):
    print(tag, "  ".join("%s %.2e" % (k, float((a[k].double() - b[k].double()).norm() / b[k].double().norm().clamp_min(1e-30))) for k in sorted(b) if k != "maps_fwd"),
          " fwd maps equal:", [bool(torch.equal(x, y)) for x, y in zip(a["maps_fwd"], b["maps_fwd"])], flush=True)
run(True)
p1, p2, n1, n2, p3 = run(True), run(True), run(False), run(False), run(True)
cmp("pm  vs pm  ", p1, p2); cmp("nchw vs nchw", n1, n2); cmp("pm  vs nchw", p1, n1); cmp("pm3 vs pm1 ", p3, p1); cmp("pm3 vs nchw2", p3, n2)
