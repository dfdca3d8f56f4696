#!/usr/bin/env python3
"""What the PSRoI backward sees in the bench's training step (developer tool, GPU box): the RoIs handed to dtt.heads.PsroiPmFn in one
step of `bench.py --mode train`'s model -- sizes, degenerate boxes, and the (RoI, bin) pairs per map pixel (mean / max, where)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np, torch
from dtt.config import cfg, cfg_from_file, apply_dataset_defaults
from dtt.synth import build_model, calibrate_batchnorm_, make_batch
from dtt import heads
from oracle import oracle_lib as O  # noqa: F401  (bin edges through the oracle's arithmetic below)

apply_dataset_defaults("imagenet_vid")
cfg_from_file(os.path.join(ROOT, "cfgs", "res101.yml"))
dev = torch.device("cuda:0")
B, H, W = 2, 600, 1067
im, info, gt, nb = make_batch(B, H, W, seed=3)
model = build_model(101, cfg=cfg).to(dev)
calibrate_batchnorm_(model, im[:, 0].to(dev))
model.train()
from dtt.dist import prepare_replica
runner = prepare_replica(model, 1, channels_last=True, force_buckets=False)
seen = []
orig = heads.PsroiPmFn.forward


def spy(ctx, pm_map, rois, batch, height, width, spatial_scale, heads_, extract=None):
    seen.append((rois.detach().cpu().numpy().copy(), batch, height, width))
    return orig(ctx, pm_map, rois, batch, height, width, spatial_scale, heads_, extract)


heads.PsroiPmFn.forward = staticmethod(spy)
grads = []
orig_b = heads.PsroiPmFn.backward


def spy_b(ctx, *gvotes):
    grads.append((ctx.saved_tensors[0].clone(), [None if g is None else g.clone() for g in gvotes], ctx.geom, ctx.extract))
    return orig_b(ctx, *gvotes)


heads.PsroiPmFn.backward = staticmethod(spy_b)
from dtt.dist import make_optimizer
opt = make_optimizer(model, cfg, lr=1e-4)
STEPS = int(os.environ.get("STEPS", "1"))          # bench.py --mode train looks at steps 4 .. (3 warm-up + the timed ones)
for st_i in range(STEPS):
    seen.clear(); grads.clear()
    runner.zero_grad(set_to_none=True)
    out = runner(im.to(dev), info.to(dev), gt.to(dev), nb.to(dev))
    (out[4].mean() + out[5].mean() + out[6].mean() + out[7].mean() + out[9].mean()).backward()
    runner.finish_gradients()
    opt.step()
torch.cuda.synchronize()
print("after %d training steps:" % STEPS)
from dtt import _lib
from dtt._lib import check, ptr, stream_ptr
L = _lib.lib()
if hasattr(L, "dtt_psroi_bwd_wg_read"):      # the DTT_PSROI_BWD_STAMP build: the detection launch of the STEP (the last one so far)
    import ctypes
    buf = (ctypes.c_ulonglong * 256)(); wgb = (ctypes.c_ulonglong * (1024 * 3))()
    L.dtt_psroi_bwd_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.dtt_psroi_bwd_wg_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.dtt_psroi_bwd_stamps_read(buf, 256) and L.dtt_psroi_bwd_wg_read(wgb, 1024 * 3)
    st = np.array(buf, dtype=np.uint64).reshape(4, 64).astype(np.int64)
    for row in (0, 2):
        ticks, real = st[row, 50] - st[row, 0], st[row, 63] - st[row, 62]
        print("in the step: workgroup %s: %d shader-clock ticks in %.2f us -> %.0f MHz; prologue (to the first pixel) %d ticks" % (
            "0" if row == 0 else "middle", ticks, real / 100.0, ticks / max(real, 1) * 100.0, st[row, 4] - st[row, 0]))
    w_ = np.array(wgb, dtype=np.uint64).reshape(1024, 3); w_ = w_[w_[:, 0] > 0]
    t0 = int(w_[:, 0].min()); st_, en_ = (w_[:, 0].astype(np.int64) - t0) / 100.0, (w_[:, 1].astype(np.int64) - t0) / 100.0
    print("in the step: %d workgroups, entry %.2f .. %.2f us, end min %.2f median %.2f max %.2f us" % (len(w_), st_.min(), st_.max(), en_.min(), np.median(en_), en_.max()))
for rois_t, gv, geom, extract in grads:
    batch, height, width, scale, hd, M, stride = geom
    if len(hd) != 2:
        continue
    g0, g1 = gv[0].contiguous(), gv[1].contiguous()
    add = gv[2].contiguous() if extract is not None and gv[2] is not None else None
    gm = torch.empty((M, stride), device=dev)
    R = rois_t.shape[0]

    def call():
        check(L.dtt_psroi_pm_backward_heads(ptr(g0), hd[0]["od"], hd[0]["cp"], ptr(g1), hd[1]["od"], hd[1]["cp"], ptr(rois_t), R, batch, height, width, 7,
                                            scale, stride, stride, ptr(add) if add is not None else None, extract[0] if add is not None else 0,
                                            extract[1] if add is not None else 0, ptr(gm), stream_ptr(dev)), "heads")
    with torch.cuda.device(dev):
        for _ in range(10):
            call()
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(100):
            call()
        e_.record(); torch.cuda.synchronize()
    print("the step's own detection call (rois, vote gradients and compact gradient of the step), 100 back-to-back launches: %.1f us each; "
          "zero rows in the class gradient %d, in the box gradient %d; |g| max %.3e min nonzero %.3e"
          % (s_.elapsed_time(e_) * 10, int((g0.abs().sum(1) == 0).sum()), int((g1.abs().sum(1) == 0).sum()), float(g0.abs().max()),
             float(g0.abs()[g0 != 0].min())))
for rois, batch, h, w in seen:
    R = rois.shape[0]
    ww, hh = rois[:, 3] - rois[:, 1] + 1, rois[:, 4] - rois[:, 2] + 1
    print("PsroiPmFn call: %d RoIs over %d images of %d x %d pixels; sides px: width min %.0f median %.0f max %.0f, height min %.0f median %.0f max %.0f; "
          "%d boxes with both sides <= 16 px, %d all-zero rows; per image %s"
          % (R, batch, h, w, ww.min(), np.median(ww), ww.max(), hh.min(), np.median(hh), hh.max(), int(((ww <= 16) & (hh <= 16)).sum()),
             int((np.abs(rois[:, 1:]).sum(1) == 0).sum()), np.bincount(rois[:, 0].astype(int), minlength=batch).tolist()))
    hits = np.zeros((batch, h, w), np.int64)
    for r in rois:
        b = int(min(max(r[0], 0), batch - 1))
        x1, y1, x2, y2 = [np.float32(np.round(v)) for v in r[1:]]
        sw, sh = x1 * np.float32(1 / 16.), y1 * np.float32(1 / 16.)
        ew, eh = (x2 + 1) * np.float32(1 / 16.), (y2 + 1) * np.float32(1 / 16.)
        rw, rh = max(ew - sw, np.float32(0.1)), max(eh - sh, np.float32(0.1))
        bw, bh = rw / np.float32(7), rh / np.float32(7)
        for ph in range(7):
            hs, he = int(np.floor(ph * bh + sh)), int(np.ceil((ph + 1) * bh + sh))
            hs, he = min(max(hs, 0), h), min(max(he, 0), h)
            for pw in range(7):
                ws, we = int(np.floor(pw * bw + sw)), int(np.ceil((pw + 1) * bw + sw))
                ws, we = min(max(ws, 0), w), min(max(we, 0), w)
                if he > hs and we > ws:
                    hits[b, hs:he, ws:we] += 1
    am = np.unravel_index(hits.argmax(), hits.shape)
    print("   (RoI, bin) pairs per pixel: mean %.1f, median %.0f, 99th percentile %.0f, max %d at image %d pixel (%d, %d); pixels above 128: %d"
          % (hits.mean(), np.median(hits), np.percentile(hits, 99), hits.max(), am[0], am[1], am[2], int((hits > 128).sum())))
