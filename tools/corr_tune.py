#!/usr/bin/env python3
"""Developer tool: time variant builds of the correlation kernel (tools/_variants/*.so) at the conv5 shape."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
dev = torch.device("cuda:0")
B, C, H, W = int(os.environ.get("B", 2)), int(os.environ.get("C", 2048)), 38, 67
x1 = torch.relu(torch.randn(B, C, H, W, device=dev)); x2 = torch.relu(torch.randn(B, C, H, W, device=dev))
out = torch.empty(B, 289, H, W, device=dev)
P, I, L, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t
for so in sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "*.so"))):
    lib = ctypes.CDLL(so)
    lib.dtt_correlation_forward_workspace_bytes.restype = Z
    lib.dtt_correlation_forward.argtypes = [P, I, I, I, I, L, P, I, I, I, P, P, Z, I, I, I, I, I, I, P]
    n = lib.dtt_correlation_forward_workspace_bytes(B, C, H, W, 8, 1, 8, 1, 1)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    lib.dtt_profile_attach.argtypes = [ctypes.c_char_p, P, P, I]
    N = 30
    evb = [torch.cuda.Event(enable_timing=True) for _ in range(N)]; eve = [torch.cuda.Event(enable_timing=True) for _ in range(N)]
    for e in evb + eve: e.record()
    torch.cuda.synchronize()
    ab = (P * N)(*[e.cuda_event for e in evb]); ae = (P * N)(*[e.cuda_event for e in eve])
    def run():
        ok = lib.dtt_correlation_forward(P(out.data_ptr()), B, 289, H, W, 289 * H * W, P(x1.data_ptr()), C, H, W, P(x2.data_ptr()),
                                         P(ws.data_ptr()), n, 8, 1, 8, 1, 1, 1, P(torch.cuda.current_stream().cuda_stream))
        assert ok == 1
    for _ in range(5): run()
    torch.cuda.synchronize()
    lib.dtt_profile_attach(b"corr_fwd_mfma", ab, ae, N)
    for _ in range(N): run()
    torch.cuda.synchronize()
    lib.dtt_profile_attach(None, None, None, 0)
    d = sorted(evb[i].elapsed_time(eve[i]) * 1e3 for i in range(N))
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(N): run()
    s1.record(); torch.cuda.synchronize()
    print("%-28s mfma kernel: median %.1f us  min %.1f us | whole op %.1f us (ws %.0f MB)" %
          (os.path.basename(so), d[N // 2], d[0], s0.elapsed_time(s1) * 1e3 / N, n / 1e6), flush=True)
