#!/usr/bin/env python3
"""Which loss's gradient into the top / conv maps differs between the two training graphs at the configs[4] shape (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from dtt import _lib
from dtt._lib import check, ptr, stream_ptr
dev = torch.device("cuda:0")
L = _lib.lib()
for (M, K, N) in [(4536, 1792, 512), (4536, 96, 512), (4536, 512, 1776), (10184, 96, 512), (10184, 1792, 512), (2268, 224, 2880), (4536, 64, 512), (4536, 128, 512)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    out = torch.full((M, N), float("nan"), device=dev)
    check(L.dtt_head_gemm(ptr(x), K, M, K, ptr(w), ptr(b), N, ptr(out), N, N, 0, stream_ptr(dev)), "gemm")
    ref = x.double() @ w.double().t() + b.double()
    d = (out.double() - ref)
    bad = (d.abs() > 1e-3).nonzero()
    print("head gemm M=%d K=%d N=%d: max|diff| %.3e rel %.3e nan %d bad %d %s" % (M, K, N, float(d.abs().max()), float(d.norm() / ref.norm()), int(torch.isnan(out).sum()),
          bad.shape[0], bad[:4].tolist()), flush=True)
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
def run(pm, which):
    model._train_pm = pm
    model.zero_grad(set_to_none=True)
    np.random.seed(99); torch.manual_seed(99)
    out = model(im, info, gt, nb)
    sum(out[i].mean() for i in which).backward()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None and not n.startswith("RFCN_base.")}
    g.update(spy.grads())
    return g
run(True, (4, 5, 6, 7, 9))
for which, name in (((4,), "rpn cls"), ((5,), "rpn box"), ((6,), "rfcn cls"), ((7,), "rfcn box"), ((9,), "tracking")):
    a, b = run(True, which), run(False, which)
    print(name, "  ".join("%s %.2e" % (k.replace("RFCN_", "").replace("_map", ""), float((a[k].double() - b[k].double()).norm() / b[k].double().norm().clamp_min(1e-30)))
                          for k in sorted(b) if k in a and float(b[k].abs().max()) > 0), flush=True)
