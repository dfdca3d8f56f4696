"""Times the position-major PSRoI backward at the training step's shape (both legs of two frame pairs: 4 images x 38 x 67 pixels,
128 RoIs per image, class + box heads, the tracking branch's compact box-delta gradient added): the one-launch wave-per-pixel kernel
(csrc/psroi_bwd.hip) against what rounds 4 - 5 ran (one launch per head of the one-workgroup-per-pixel kernel + zero_ of the padding
columns + the add), and the tracking head's call.  Usage: python tools/time_psroi_bwd.py [--iters 200]
Environment: DTT_PSROI_BWD_WAVES=4|8|16, DTT_PSROI_BWD_PPW=n (developer sweeps)."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd"), os.path.join(ROOT, "tests")]

from dtt import _lib  # noqa: E402
from dtt._lib import check, ptr, stream_ptr  # noqa: E402


def rois_like_training(rng, per_image, batch, H, W):
    """sampled proposals: sides between 2 and 40 map pixels, sorted by image"""
    out = []
    for b in range(batch):
        w = rng.uniform(32, 640, size=per_image)
        h = rng.uniform(32, 500, size=per_image)
        x1 = rng.uniform(0, W * 16 - 32, size=per_image)
        y1 = rng.uniform(0, H * 16 - 32, size=per_image)
        out.append(np.stack([np.full(per_image, b), x1, y1, np.minimum(x1 + w, W * 16 - 1), np.minimum(y1 + h, H * 16 - 1)], 1))
    return np.concatenate(out).astype(np.float32)


def timed(fn, iters, dev):
    for _ in range(10):
        fn()
    torch.cuda.synchronize(dev)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize(dev)
    return s.elapsed_time(e) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--per-image", type=int, default=128)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    rng = np.random.RandomState(0)
    B, H, W, stride = args.batch, 38, 67, 1792
    R = B * args.per_image
    rois = torch.from_numpy(rois_like_training(rng, args.per_image, B, H, W)).to(dev)
    g_cls = torch.from_numpy(rng.normal(size=(R, 31)).astype(np.float32)).to(dev)
    g_loc = rng.normal(size=(R, 4)).astype(np.float32)
    g_loc[rng.rand(R) < 0.75] = 0
    g_loc = torch.from_numpy(g_loc).to(dev)
    add = torch.from_numpy(rng.normal(size=(B * H * W, 196)).astype(np.float32)).to(dev)
    edges = torch.empty((R * 29 + 2 * B,), dtype=torch.int32, device=dev)
    gm_new = torch.empty((B * H * W, stride), device=dev)
    gm_old = torch.empty((B * H * W, stride), device=dev)

    def new():
        check(L.dtt_psroi_pm_backward_heads(ptr(g_cls), 31, 32, ptr(g_loc), 4, 4, ptr(rois), R, B, H, W, 7, 1 / 16.0, stride, stride,
                                            ptr(add), 1568, 196, ptr(gm_new), stream_ptr(dev)), "heads")

    def old():
        os.environ["DTT_PSROI_BWD_OLD"] = "1"
        check(L.dtt_psroi_pm_backward(ptr(g_cls), ptr(rois), R, B, H, W, 7, 1 / 16.0, 31, 32, stride, ptr(gm_old), ptr(edges),
                                      stream_ptr(dev)), "cls")
        check(L.dtt_psroi_pm_backward(ptr(g_loc), ptr(rois), R, B, H, W, 7, 1 / 16.0, 4, 4, stride,
                                      ctypes.c_void_p(gm_old.data_ptr() + 4 * 1568), ptr(edges), stream_ptr(dev)), "loc")
        os.environ["DTT_PSROI_BWD_OLD"] = "0"
        gm_old[:, 1764:].zero_()
        gm_old[:, 1568:1764] += add

    with torch.cuda.device(dev):
        t_new, t_old = timed(new, args.iters, dev), timed(old, args.iters, dev)
        same = torch.equal(gm_new, gm_old)
        # tracking head: one pair's 2 x 38 x 67 pixels, a handful of ground-truth RoIs, 224-float rows
        Bt, Rt, st = B // 2 if B > 1 else 1, 8 * (B // 2 if B > 1 else 1), 224
        troi = torch.from_numpy(rois_like_training(rng, Rt // Bt, Bt, H, W)).to(dev)
        g_trk = torch.from_numpy(rng.normal(size=(Rt, 4)).astype(np.float32)).to(dev)
        gm_t = torch.empty((Bt * H * W, st), device=dev)
        gm_t2 = torch.empty((Bt * H * W, st), device=dev)

        def trk_new():
            check(L.dtt_psroi_pm_backward_heads(ptr(g_trk), 4, 4, None, 0, 0, ptr(troi), Rt, Bt, H, W, 7, 1 / 16.0, st, st, None, 0, 0,
                                                ptr(gm_t), stream_ptr(dev)), "trk")

        def trk_old():
            os.environ["DTT_PSROI_BWD_OLD"] = "1"
            check(L.dtt_psroi_pm_backward(ptr(g_trk), ptr(troi), Rt, Bt, H, W, 7, 1 / 16.0, 4, 4, st, ptr(gm_t2), ptr(edges), stream_ptr(dev)), "trk")
            os.environ["DTT_PSROI_BWD_OLD"] = "0"
            gm_t2[:, 196:].zero_()

        tt_new, tt_old = timed(trk_new, args.iters, dev), timed(trk_old, args.iters, dev)
        same_t = torch.equal(gm_t, gm_t2)
    mb = B * H * W * stride * 4 / 1e6
    print("detection heads, %d pixels x %d floats (%.1f MB written), %d RoIs: one launch %.1f us (%.2f TB/s, %.3f of 8 TB/s)   "
          "per-head launches + zero_ + add %.1f us (%.2f TB/s)   bit-identical: %s"
          % (B * H * W, stride, mb, R, t_new, mb / t_new, mb / t_new / 8, t_old, mb / t_old, same))
    print("tracking head, %d pixels x %d floats, %d RoIs: one launch %.1f us   old kernel + zero_ %.1f us   bit-identical: %s"
          % (Bt * H * W, st, Rt, tt_new, tt_old, same_t))
    print("env: DTT_PSROI_BWD_WAVES=%s DTT_PSROI_BWD_PPW=%s (wall time between stream events over back-to-back calls, launch gaps included)"
          % (os.environ.get("DTT_PSROI_BWD_WAVES", "-"), os.environ.get("DTT_PSROI_BWD_PPW", "-")))


if __name__ == "__main__":
    main()
