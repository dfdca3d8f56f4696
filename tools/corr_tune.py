#!/usr/bin/env python3
"""Developer tool: time variant builds of the correlation kernel (tools/_variants/*.so) at the conv5 shape."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
dev = torch.device("cuda:0")
B, C, H, W = int(os.environ.get("B", 2)), int(os.environ.get("C", 2048)), int(os.environ.get("H", 38)), int(os.environ.get("W", 67))
DISP = int(os.environ.get("D", 8)); OC = (2 * DISP + 1) ** 2
x1 = torch.relu(torch.randn(B, C, H, W, device=dev)); x2 = torch.relu(torch.randn(B, C, H, W, device=dev))
if os.environ.get("ZERO"): x1.zero_(); x2.zero_()
out = torch.empty(B, OC, H, W, device=dev)
P, I, L, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t
for so in sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "*.so"))):
    lib = ctypes.CDLL(so)
    lib.dtt_correlation_forward_workspace_bytes.restype = Z
    lib.dtt_correlation_forward.argtypes = [P, I, I, I, I, L, P, I, I, I, P, P, Z, I, I, I, I, I, I, P]
    n = lib.dtt_correlation_forward_workspace_bytes(B, C, H, W, DISP, 1, DISP, 1, 1)
    ws = torch.zeros(n + (1 << 20), dtype=torch.uint8, device=dev)
    lib.dtt_profile_attach.argtypes = [ctypes.c_char_p, P, P, I]
    N = 30
    evb = [torch.cuda.Event(enable_timing=True) for _ in range(N)]; eve = [torch.cuda.Event(enable_timing=True) for _ in range(N)]
    for e in evb + eve: e.record()
    torch.cuda.synchronize()
    ab = (P * N)(*[e.cuda_event for e in evb]); ae = (P * N)(*[e.cuda_event for e in eve])
    def run():
        ok = lib.dtt_correlation_forward(P(out.data_ptr()), B, OC, H, W, OC * H * W, P(x1.data_ptr()), C, H, W, P(x2.data_ptr()),
                                         P(ws.data_ptr()), n, DISP, 1, DISP, 1, 1, 1, P(torch.cuda.current_stream().cuda_stream))
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
    if "stamp" in os.path.basename(so):  # built with -DDTT_CORR_STAMP: per-workgroup timeline of the last launch
        import numpy as np
        st = ws[n:n + (1 << 20)].view(torch.int64).cpu().numpy().reshape(-1, 4)
        st = st[st[:, 0] != 0]
        t0 = st[:, 0].min()
        beg, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0  # us
        cu = ((st[:, 3] & 0xF) << 8) | (((st[:, 2] >> 13) & 7) << 5) | (((st[:, 2] >> 12) & 1) << 4) | ((st[:, 2] >> 8) & 0xF)
        print("  workgroups %d  span %.1f us  start: p50 %.1f p99 %.1f max %.1f | end: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | life mean %.1f" %
              (len(st), end.max(), np.median(beg), np.percentile(beg, 99), beg.max(), end.min(), np.percentile(end, 10), np.median(end),
               np.percentile(end, 90), end.max(), (end - beg).mean()))
        ids, cnt = np.unique(cu, return_counts=True)
        print("  distinct CUs %d; workgroups per CU histogram:" % len(ids), dict(zip(*np.unique(cnt, return_counts=True))))
        for k in sorted(set(cnt)):
            sel = np.isin(cu, ids[cnt == k])
            print("    CUs with %d WGs: end mean %.1f us, life mean %.1f us" % (k, end[sel].mean(), (end - beg)[sel].mean()))
        xcc = st[:, 3] & 0xF
        item = (st[:, 3] >> 8) & 0xFFFFFFFF; fixp = (st[:, 3] >> 40) & 1
        main = ((st[:, 3] >> 40) & 0xFFFFFF) / 100.0
        print("   time to end of channel loop (us): mean %.1f p10 %.1f p90 %.1f ; epilogue store (life - loop) mean %.1f" % (main.mean(), np.percentile(main, 10), np.percentile(main, 90), ((end - beg) - main).mean()))
        tile = item % 45; ks = (item // 45) % 8; nn = item // 360
        for name, key in (("xcc", xcc), ("tile_x", tile % 9), ("tile_y", tile // 9), ("ks", ks), ("n", nn), ("wg_fix", fixp)):
            print("   end by %-7s" % name, " ".join("%d:%.0f" % (k, end[key == k].mean()) for k in np.unique(key)))
        order = np.argsort(end)
        print("   slowest 12 (end us, xcc, cu, tile_x, tile_y, ks, n):", [(round(float(end[i]), 1), int(xcc[i]), int(cu[i]) & 0xFF, int(tile[i] % 9), int(tile[i] // 9), int(ks[i]), int(nn[i])) for i in order[-12:]])
        print("  per-XCC workgroups:", dict(zip(*np.unique(xcc, return_counts=True))))
