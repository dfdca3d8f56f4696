#!/usr/bin/env python3
"""Developer tool: randomized shape sweep of the hot-path ops against the oracle (GPU box).  Complements the fixed cases
of tests/test_gpu_ops.py: map sizes around tile / piece boundaries, channel counts around chunk boundaries, batch 1-3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np, torch
from oracle import oracle_lib as O
from dtt.ops import Correlation, _PSRoIPooling, nms


def run(N=150, seed=0):
    """Returns (correlation mismatches, psroi mismatches, nms mismatches) over N / N//3 / N//3 random cases."""
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(seed)
    bad = 0
    for it in range(N):
        B = rs.randint(1, 4); C = int(rs.choice([8, 16, 24, 40, 64, 12, 20])); H = rs.randint(2, 41); W = rs.randint(4, 70)
        d = int(rs.choice([4, 8, 8, 8, 3, 16])); s = int(rs.choice([1, 1, 1, 2])); pad = d if rs.rand() < 0.8 else d + 4 * rs.randint(0, 2)
        if (H + 2 * pad - 2 * d + s - 1) // s < 1 or (W + 2 * pad - 2 * d + s - 1) // s < 1:
            continue
        if it < int(os.environ.get("START", 0)) or it > int(os.environ.get("STOP", 10**9)):
            continue
        if os.environ.get("VERBOSE"):
            print("corr case", it, (B, C, H, W, pad, d, s), flush=True)
        x1 = np.maximum(rs.normal(size=(B, C, H, W)), 0).astype(np.float32)
        x2 = np.maximum(rs.normal(size=(B, C, H, W)), 0).astype(np.float32)
        ref = O.correlation_forward(x1, x2, pad, 1, d, s, s)
        t1 = torch.from_numpy(x1).to(dev).requires_grad_(True); t2 = torch.from_numpy(x2).to(dev).requires_grad_(True)
        out = Correlation(pad, 1, d, s, s)(t1, t2)
        torch.cuda.synchronize()
        err = float(np.abs(out.detach().cpu().numpy() - ref).max())
        go = rs.normal(size=ref.shape).astype(np.float32)
        e1 = e2 = 0.0
        if not os.environ.get("FWD_ONLY"):
            out.backward(torch.from_numpy(go).to(dev))
            torch.cuda.synchronize()
            g1, g2 = O.correlation_backward(go, x1, x2, pad, 1, d, s, s)
            e1 = float(np.abs(t1.grad.cpu().numpy() - g1).max()); e2 = float(np.abs(t2.grad.cpu().numpy() - g2).max())
        if err > 1e-5 or e1 > 1e-4 or e2 > 1e-4:
            bad += 1
            print("CORR MISMATCH", (B, C, H, W, pad, d, s), err, e1, e2, flush=True)
    print("correlation: %d cases, %d bad" % (N, bad), flush=True)
    badp = 0
    for it in range(N // 3):
        B = rs.randint(1, 4); od = rs.randint(1, 6); gs = int(rs.choice([3, 7])); H = rs.randint(3, 45); W = rs.randint(3, 70); R = rs.randint(1, 400)
        feat = rs.normal(size=(B, od * gs * gs, H, W)).astype(np.float32)
        x1 = rs.uniform(-30, W * 16, R); y1 = rs.uniform(-30, H * 16, R)
        rois = np.stack([rs.randint(0, B, R), x1, y1, x1 + rs.uniform(0, W * 12, R), y1 + rs.uniform(0, H * 12, R)], 1).astype(np.float32)
        ref, _ = O.psroi_pool_forward(feat, rois, gs, gs, 1 / 16.0, gs, od)
        out = _PSRoIPooling(gs, gs, 1 / 16.0, gs, od)(torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev))
        if not np.array_equal(out.cpu().numpy(), ref):
            badp += 1
            print("PSROI MISMATCH", (B, od, gs, H, W, R), flush=True)
    print("psroi: %d cases, %d bad" % (N // 3, badp), flush=True)
    badn = 0
    for it in range(N // 3):
        n = rs.randint(1, 3000); thr = float(rs.choice([0.3, 0.5, 0.7]))
        c = rs.uniform(0, 300, size=(n, 2)); wh = rs.uniform(5, 150, size=(n, 2))
        dets = np.concatenate([c - wh / 2, c + wh / 2, np.sort(rs.uniform(0, 1, n))[::-1][:, None]], 1).astype(np.float32)
        if rs.rand() < 0.3:
            dets[:, :4] = np.round(dets[:, :4])            # integer boxes: exact-threshold IoUs
        keep = nms(torch.from_numpy(dets).to(dev), thr).view(-1).cpu().numpy()
        if not np.array_equal(keep, O.nms(dets, thr).reshape(-1)):
            badn += 1
            print("NMS MISMATCH", n, thr, flush=True)
    print("nms: %d cases, %d bad" % (N // 3, badn), flush=True)
    return bad, badp, badn


if __name__ == "__main__":
    b = run(int(os.environ.get("N", 150)), int(os.environ.get("SEED", 0)))
    sys.exit(1 if any(b) else 0)
