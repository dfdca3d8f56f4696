#!/usr/bin/env python3
"""Per-layer timing of the fused inference trunk's convolutions under alternative formulations (NCHW MIOpen,
channels-last MIOpen, 1x1 as hipBLASLt GEMM with fused bias+ReLU epilogue, MIOpen fused conv+bias+relu)."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
import torch.nn.functional as F

from dtt import fuse
from dtt.config import cfg
from dtt.synth import build_model, calibrate_batchnorm_, make_batch


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    model = build_model(101, class_agnostic=True, cfg=cfg).to(dev).eval()
    im, info, gt, nb = make_batch(B, 600, 1067, seed=1, device=dev)
    calibrate_batchnorm_(model, im[:, 0])
    fuse.fuse_for_inference(model)
    x = torch.cat([im[:, 0], im[:, 1]], 0).contiguous()
    trunk = model._fused_trunk
    recs = []
    orig = fuse._FusedConv.conv

    def rec_conv(self, inp):
        recs.append((self, tuple(inp.shape)))
        return orig(self, inp)
    fuse._FusedConv.conv = rec_conv
    with torch.no_grad():
        trunk(x)
    fuse._FusedConv.conv = orig
    print("trunk NCHW fused: %.1f us" % timeit(lambda: trunk(x), 5, 2))
    groups = collections.OrderedDict()
    for fc, shp in recs:
        key = (shp, tuple(fc.w.shape), tuple(fc.kw["stride"]), tuple(fc.kw["padding"]), tuple(fc.kw["dilation"]))
        groups.setdefault(key, []).append(fc)
    tot = collections.Counter()
    print("%-22s %-20s %-8s %4s | %9s %9s %9s %9s %9s" % ("input", "weight", "s/p/d", "n", "nchw", "nchw+ep", "nhwc", "gemm+ep", "mio_fused"))
    for key, fcs in groups.items():
        shp, wshp, st, pd, dl = key
        fc = fcs[0]
        n = len(fcs)
        xin = torch.randn(shp, device=dev)
        xin_cl = xin.contiguous(memory_format=torch.channels_last)
        w_cl = fc.w.contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            t_nchw = timeit(lambda: F.conv2d(xin, fc.w, None, **fc.kw))
            t_ep = timeit(lambda: fuse.bias_act_(F.conv2d(xin, fc.w, None, **fc.kw), fc.b))
            t_nhwc = timeit(lambda: F.conv2d(xin_cl, w_cl, None, **fc.kw))
            t_gemm = float("nan")
            if wshp[2] == 1 and st == (1, 1):
                x2 = xin_cl.permute(0, 2, 3, 1).reshape(-1, shp[1])
                wt = fc.w.view(wshp[0], wshp[1]).t().contiguous()
                t_gemm = timeit(lambda: torch._addmm_activation(fc.b, x2, wt))
            try:
                t_mio = timeit(lambda: torch.miopen_convolution_relu(xin, fc.w, fc.b, fc.kw["stride"], fc.kw["padding"], fc.kw["dilation"], 1))
            except Exception as e:  # noqa: BLE001
                t_mio = float("nan")
        for k, v in (("nchw", t_nchw), ("nchw+ep", t_ep), ("nhwc", t_nhwc), ("gemm+ep", t_gemm if t_gemm == t_gemm else t_nhwc), ("mio", t_mio)):
            tot[k] += v * n
        print("%-22s %-20s %-8s %4d | %9.1f %9.1f %9.1f %9.1f %9.1f" % (shp, wshp, "%d/%d/%d" % (st[0], pd[0], dl[0]), n, t_nchw, t_ep, t_nhwc, t_gemm, t_mio))
    print("totals (us):", dict(tot))


if __name__ == "__main__":
    main()
