#!/usr/bin/env python3
"""Developer tool: time variant builds of the PSRoI forward (tools/_variants/*.so) at the 600 px D&T shape."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
dev = torch.device("cuda:0")
B, od, H, W, R = int(os.environ.get("B", 2)), int(os.environ.get("OD", 31)), 38, 67, 300
C = od * 49
g = torch.Generator().manual_seed(0)
feat = torch.randn(B, C, H, W, generator=g).to(dev)
x1 = torch.rand(B * R, generator=g) * 900; y1 = torch.rand(B * R, generator=g) * 450
rois = torch.stack([torch.arange(B).repeat_interleave(R).float(), x1, y1, x1 + 30 + torch.rand(B * R, generator=g) * 400,
                    y1 + 30 + torch.rand(B * R, generator=g) * 300], 1).to(dev).contiguous()
out = torch.empty(B * R, od, 7, 7, device=dev); mapc = torch.empty(B * R, od, 7, 7, dtype=torch.int32, device=dev)
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
ref = None
for so in sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "*.so"))):
    lib = ctypes.CDLL(so)
    lib.dtt_psroi_pool_forward.argtypes = [P, F, I, I, I, I, I, I, I, P, I, I, P, P, P]
    vote = torch.empty(B * R, od, device=dev); scratch = torch.empty(C * B * R, device=dev)
    if os.environ.get("TR"):
        lib.dtt_psroi_vote_forward.argtypes = [P, F, I, I, I, I, I, I, I, P, I, I, P, P, P]
    def run():
        if os.environ.get("TR"):
            assert lib.dtt_psroi_vote_forward(P(feat.data_ptr()), 1 / 16.0, B, B * R, H, W, C, 7, 7, P(rois.data_ptr()), 7, od,
                                              P(scratch.data_ptr()), P(vote.data_ptr()), P(torch.cuda.current_stream().cuda_stream)) == 1
            out.view(-1)[:vote.numel()].copy_(vote.view(-1)); return
        assert lib.dtt_psroi_pool_forward(P(feat.data_ptr()), 1 / 16.0, B, B * R, H, W, C, 7, 7, P(rois.data_ptr()), 7, od,
                                          P(out.data_ptr()), P(mapc.data_ptr()), P(torch.cuda.current_stream().cuda_stream)) == 1
    for _ in range(10): run()
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 200
    s0.record()
    for _ in range(N): run()
    s1.record(); torch.cuda.synchronize()
    us = s0.elapsed_time(s1) * 1e3 / N
    alg = B * (C * H * W * 4) + B * R * od * 49 * 4
    same = "" if ref is None else ("  identical=%s" % bool(torch.equal(out, ref)))
    ref = out.clone() if ref is None else ref
    print("%-20s %.1f us/launch  %.0f GB/s algorithmic%s" % (os.path.basename(so), us, alg / us / 1e3, same), flush=True)
