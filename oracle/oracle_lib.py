"""ctypes binding of oracle/libdtt_oracle.so (numpy in, numpy out).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdtt_oracle.so")
_lib = None

c_f = ctypes.POINTER(ctypes.c_float)
c_i = ctypes.POINTER(ctypes.c_int)
c_u64 = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "dtt_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdtt_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_f)


def _ip(a):
    return a.ctypes.data_as(c_i) if a is not None else None


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(ctypes.c_int(int(n)))


def correlation_output_shape(C, H, W, pad, k, d, s1, s2):
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    ok = lib().oracle_correlation_output_shape(C, H, W, pad, k, d, s1, s2, ctypes.byref(oc),
                                               ctypes.byref(oh), ctypes.byref(ow))
    if not ok:
        raise ValueError("invalid correlation geometry")
    return oc.value, oh.value, ow.value


def correlation_forward(x1, x2, pad, k, d, s1, s2):
    x1, p1 = _f(x1)
    x2, p2 = _f(x2)
    B, C, H, W = x1.shape
    oc, oh, ow = correlation_output_shape(C, H, W, pad, k, d, s1, s2)
    out = np.zeros((B, oc, oh, ow), dtype=np.float32)
    ok = lib().oracle_correlation_forward(out.ctypes.data_as(c_f), p1, p2, B, C, H, W, pad, k, d, s1, s2)
    assert ok == 1
    return out


def correlation_backward(gout, x1, x2, pad, k, d, s1, s2):
    x1, p1 = _f(x1)
    x2, p2 = _f(x2)
    gout, pg = _f(gout)
    B, C, H, W = x1.shape
    g1 = np.zeros_like(x1)
    g2 = np.zeros_like(x2)
    ok = lib().oracle_correlation_backward(g1.ctypes.data_as(c_f), g2.ctypes.data_as(c_f), pg, p1, p2,
                                           B, C, H, W, pad, k, d, s1, s2)
    assert ok == 1
    return g1, g2


def psroi_pool_forward(feat, rois, ph, pw, scale, group, od):
    feat, pf = _f(feat)
    rois, pr = _f(rois)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, od, ph, pw), dtype=np.float32)
    mc = np.zeros((R, od, ph, pw), dtype=np.int32)
    lib().oracle_psroi_pool_forward(pf, ctypes.c_float(scale), R, H, W, C, ph, pw, pr, group, od,
                                    out.ctypes.data_as(c_f), _ip(mc))
    return out, mc


def psroi_pool_backward(top_diff, rois, feat_shape, ph, pw, scale, group, od, mapping=None):
    top_diff, pt = _f(top_diff)
    rois, pr = _f(rois)
    B, C, H, W = feat_shape
    R = rois.shape[0]
    g = np.zeros((B, C, H, W), dtype=np.float32)
    if mapping is not None:
        mapping = np.ascontiguousarray(mapping, dtype=np.int32)
    lib().oracle_psroi_pool_backward(pt, _ip(mapping), B, R, ctypes.c_float(scale), C, H, W, pw, ph, od,
                                     group, g.ctypes.data_as(c_f), pr)
    return g


def nms(dets, thresh, return_mask=False):
    dets, pd = _f(dets)
    n, dim = (dets.shape[0], dets.shape[1]) if dets.ndim == 2 and dets.shape[0] else (0, 5)
    keep = np.zeros((max(n, 1),), dtype=np.int32)
    num = ctypes.c_int(0)
    cb = (n + 63) // 64
    mask = np.zeros((max(n, 1), max(cb, 1)), dtype=np.uint64) if return_mask else None
    if n > 0:
        lib().oracle_nms(_ip(keep), ctypes.byref(num), pd, n, dim, ctypes.c_float(thresh),
                         mask.ctypes.data_as(c_u64) if return_mask else None)
    keep = keep[: num.value].copy()
    return (keep, mask[:n, :cb]) if return_mask else keep


def roi_align_forward(feat, rois, ah, aw, scale):
    feat, pf = _f(feat)
    rois, pr = _f(rois)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, ah, aw), dtype=np.float32)
    lib().oracle_roi_align_forward(pf, ctypes.c_float(scale), R, H, W, C, ah, aw, pr, out.ctypes.data_as(c_f))
    return out


def roi_align_backward(top_diff, rois, feat_shape, ah, aw, scale):
    top_diff, pt = _f(top_diff)
    rois, pr = _f(rois)
    B, C, H, W = feat_shape
    g = np.zeros((B, C, H, W), dtype=np.float32)
    lib().oracle_roi_align_backward(pt, ctypes.c_float(scale), B, rois.shape[0], H, W, C, ah, aw, pr,
                                    g.ctypes.data_as(c_f))
    return g


def roi_pool_forward(feat, rois, ph, pw, scale):
    feat, pf = _f(feat)
    rois, pr = _f(rois)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, ph, pw), dtype=np.float32)
    am = np.zeros((R, C, ph, pw), dtype=np.int32)
    lib().oracle_roi_pool_forward(pf, ctypes.c_float(scale), R, H, W, C, ph, pw, pr,
                                  out.ctypes.data_as(c_f), _ip(am))
    return out, am


def roi_pool_backward(top_diff, rois, argmax, feat_shape, ph, pw, scale):
    top_diff, pt = _f(top_diff)
    rois, pr = _f(rois)
    argmax = np.ascontiguousarray(argmax, dtype=np.int32)
    B, C, H, W = feat_shape
    g = np.zeros((B, C, H, W), dtype=np.float32)
    lib().oracle_roi_pool_backward(pt, ctypes.c_float(scale), B, rois.shape[0], H, W, C, ph, pw, pr,
                                   g.ctypes.data_as(c_f), _ip(argmax))
    return g


def roi_crop_forward(images, grids):
    images, pi = _f(images)
    grids, pg = _f(grids)
    ib, ic, ih, iw = images.shape
    ob, oh, ow, _ = grids.shape
    out = np.zeros((ob, ic, oh, ow), dtype=np.float32)
    lib().oracle_roi_crop_forward(ic, ow, oh, ob, ic, ih, iw, ib, pi, pg, out.ctypes.data_as(c_f))
    return out


def roi_crop_backward(images, grids, gout):
    images, pi = _f(images)
    grids, pg = _f(grids)
    gout, po = _f(gout)
    ib, ic, ih, iw = images.shape
    ob, oh, ow, _ = grids.shape
    g = np.zeros_like(images)
    lib().oracle_roi_crop_backward(ic, ow, oh, ob, ic, ih, iw, ib, pi, pg, g.ctypes.data_as(c_f), po)
    return g
