#!/usr/bin/env python3
"""Per-call timing of the channels-last fused trunk in situ (kernels as chosen by MIOpen find mode in this process)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
from dtt import fuse
from dtt.config import cfg
from dtt.synth import build_model, calibrate_batchnorm_, make_batch
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
model = build_model(101, class_agnostic=True, cfg=cfg).to(dev).eval()
im, info, gt, nb = make_batch(2, 600, 1067, seed=1, device=dev)
calibrate_batchnorm_(model, im[:, 0])
x = torch.cat([im[:, 0], im[:, 1]], 0).contiguous()
for cl in (True, False):
    fuse.fuse_for_inference(model, channels_last=cl)
    trunk = model._fused_trunk
    for _ in range(3):
        trunk(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        trunk(x)
    b.record(); torch.cuda.synchronize()
    print("channels_last=%s trunk %.1f us" % (cl, a.elapsed_time(b) / 5 * 1e3))
    if not cl:
        break
    rec = collections.OrderedDict()
    def wrap(name, fn):
        def f(self, inp):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = fn(self, inp); e1.record(); torch.cuda.synchronize()
            key = (name, tuple(inp.shape), self.is_gemm, tuple(getattr(self, "w", getattr(self, "wt", None)).shape))
            rec.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
            return out
        return f
    oa, orw = fuse._NhwcConv.act, fuse._NhwcConv.raw
    fuse._NhwcConv.act, fuse._NhwcConv.raw = wrap("act", oa), wrap("raw", orw)
    for _ in range(3):
        trunk(x)
    fuse._NhwcConv.act, fuse._NhwcConv.raw = oa, orw
    tot = 0
    for k, v in rec.items():
        n = len(v) // 3
        avg = sum(v) / len(v)
        tot += avg * n
        print("%-4s in %-22s gemm=%d w %-20s n=%3d avg %8.1f us" % (k[0], k[1], k[2], k[3], n, avg))
    print("sum of conv calls %.1f us" % tot)
