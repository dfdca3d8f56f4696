"""Golden vectors from the REFERENCE's own operator kernels (oracle/_ref: built by oracle/build_ref.sh from the sources
under /root/reference, see its header), run on an MI355X:

    gpurun -- 'python tests/golden/make_golden_ref_kernels.py gpurun_out/ref_kernels.npz'
    cp gpurun_out/ref_kernels.npz tests/golden/ref_kernels.npz

Seeded inputs + the outputs the reference kernels produced (forward outputs, index maps, keep lists, gradients).
tests/test_oracle_ref_golden.py checks the CPU oracle against them without a GPU: forwards / index outputs bit for bit,
gradients (float atomics in the reference: summation order is free) to 1e-5.  Data only -- no reference source."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import ref_kernels as RK  # noqa: E402


def rois_for(rng, n, batch, im_h, im_w):
    x1 = rng.uniform(-20, im_w - 10, size=n); y1 = rng.uniform(-20, im_h - 10, size=n)
    w = rng.uniform(1, im_w * 0.7, size=n); h = rng.uniform(1, im_h * 0.7, size=n)
    r = np.stack([rng.randint(0, batch, size=n), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    r[0, 1:] = [0, 0, im_w - 1, im_h - 1]; r[1, 1:] = [5, 5, 5, 5]; r[2, 1:] = [im_w + 50, im_h + 50, im_w + 90, im_h + 90]
    r[3, 1:] = [16, 32, 16 + 7 * 16 - 1, 32 + 7 * 16 - 1]; r[4, 1:] = [100.5, 50.5, 30.5, 20.5]; r[5, 1:] = [-40, -40, 30, 30]
    r[6, 1:] = np.round(r[6, 1:])
    return r


def main(path):
    rng = np.random.RandomState(2024)
    out = {}
    # correlation: (name, B, C, H, W, pad, k, d, s1, s2)
    for name, B, C, H, W, pad, k, d, s1, s2 in [("a", 2, 37, 9, 11, 3, 1, 3, 1, 1), ("b", 1, 64, 12, 14, 4, 1, 4, 1, 2),
                                                 ("c", 1, 33, 13, 10, 2, 1, 2, 2, 2), ("d", 1, 8, 10, 10, 3, 3, 2, 1, 1),
                                                 ("e", 1, 40, 7, 15, 0, 1, 2, 1, 1)]:
        x1 = rng.normal(size=(B, C, H, W)).astype(np.float32); x2 = rng.normal(size=(B, C, H, W)).astype(np.float32)
        y = RK.correlation_forward(x1, x2, pad, k, d, s1, s2)
        out.update({"corr_%s_cfg" % name: np.array([pad, k, d, s1, s2]), "corr_%s_x1" % name: x1, "corr_%s_x2" % name: x2,
                    "corr_%s_out" % name: y})
        if s1 == 1 and k == 1:
            g = rng.normal(size=y.shape).astype(np.float32)
            g1, g2 = RK.correlation_backward(g, x1, x2, pad, k, d, s1, s2)
            out.update({"corr_%s_gout" % name: g, "corr_%s_g1" % name: g1, "corr_%s_g2" % name: g2})
    # PSRoI pooling
    for name, B, od, g, H, W, n in [("a", 1, 2, 7, 10, 12, 20), ("b", 2, 4, 3, 9, 11, 17)]:
        feat = rng.normal(size=(B, od * g * g, H, W)).astype(np.float32)
        rois = rois_for(rng, n, B, H * 16, W * 16)
        y, m = RK.psroi_pool_forward(feat, rois, g, g, 1 / 16.0, g, od)
        top = rng.normal(size=y.shape).astype(np.float32)
        out.update({"psroi_%s_cfg" % name: np.array([g, od]), "psroi_%s_feat" % name: feat, "psroi_%s_rois" % name: rois,
                    "psroi_%s_out" % name: y, "psroi_%s_map" % name: m, "psroi_%s_top" % name: top,
                    "psroi_%s_grad" % name: RK.psroi_pool_backward(top, rois, feat.shape, g, g, 1 / 16.0, g, od, m)})
    # RoI align / max pool / crop on one map
    B, C, H, W, n = 2, 6, 14, 19, 28
    feat = rng.normal(size=(B, C, H, W)).astype(np.float32)
    rois = rois_for(rng, n, B, H * 16, W * 16)
    ya = RK.roi_align_forward(feat, rois, 7, 7, 1 / 16.0)
    yp, arg = RK.roi_pool_forward(feat, rois, 7, 7, 1 / 16.0)
    top = rng.normal(size=ya.shape).astype(np.float32)
    grid = rng.uniform(-1.3, 1.3, size=(B * 3, 5, 5, 2)).astype(np.float32)
    yc = RK.roi_crop_forward(feat, grid)
    gc = rng.normal(size=yc.shape).astype(np.float32)
    out.update({"roi_feat": feat, "roi_rois": rois, "roi_top": top, "align_out": ya,
                "align_grad": RK.roi_align_backward(top, rois, feat.shape, 7, 7, 1 / 16.0),
                "pool_out": yp, "pool_argmax": arg, "pool_grad": RK.roi_pool_backward(top, rois, arg, feat.shape, 7, 7, 1 / 16.0),
                "crop_grid": grid, "crop_out": yc, "crop_gout": gc, "crop_grad": RK.roi_crop_backward(feat, grid, gc)})
    # NMS: sorted boxes with ties and a duplicate
    for name, nb, thr in [("a", 700, 0.7), ("b", 129, 0.3)]:
        x1 = rng.uniform(0, 600, nb); y1 = rng.uniform(0, 360, nb)
        d = np.stack([x1, y1, x1 + rng.uniform(4, 240, nb), y1 + rng.uniform(4, 180, nb),
                      np.round(np.sort(rng.uniform(0, 1, nb))[::-1], 2)], 1).astype(np.float32)
        d[nb // 2, :4] = d[nb // 3, :4]
        out.update({"nms_%s_dets" % name: d, "nms_%s_thresh" % name: np.float32(thr), "nms_%s_keep" % name: RK.nms(d, thr)})
    np.savez_compressed(path, **out)
    print("wrote %s: %d arrays, %d bytes" % (path, len(out), os.path.getsize(path)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kernels.npz"))
