"""ctypes binding of oracle/_ref/libdtt_ref_kernels.so: the REFERENCE's own operator kernels, translated by ROCm's
hipify-perl at build time from the sources under /root/reference (oracle/build_ref.sh) and run on the GPU.
TEST INFRASTRUCTURE ONLY (same rule as oracle_lib.py): a second checker beside the CPU oracle, never the product.

Only the kernels + their extern "C" launchers are the reference's; what the reference's TH/THC cffi shims do around
them (allocate and zero the outputs / scratch tensors, hand over sizes and strides) is restated here with torch
tensors, each function citing the shim it follows (paths relative to the reference's lib/model/).
numpy in, numpy out, same signatures as oracle_lib.py.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


def so_path(fma=False):
    return os.path.join(_HERE, "_ref", "libdtt_ref_kernels_fma.so" if fma else "libdtt_ref_kernels.so")


def available(fma=False):
    return os.path.exists(so_path(fma))


def lib(fma=False):
    """fma=False: built -ffp-contract=off (the declared semantics the oracle restates); fma=True: hipcc's default
    contraction, the analogue of nvcc's -fmad=true."""
    if fma not in _LIBS:
        L = ctypes.CDLL(so_path(fma))
        L.Correlation_forward_cuda_kernel.argtypes = [_P] + [_I] * 8 + [_P] + [_I] * 7 + [_P] + [_I] * 5 + [_P, _P] + \
            [_I] * 6 + [_P]
        L.Correlation_backward_cuda_kernel.argtypes = [_P] + [_I] * 8 + [_P] + [_I] * 7 + [_P] + [_I] * 4 + \
            [_P] + [_I] * 4 + [_P] + [_I] * 5 + [_P, _P] + [_I] * 6 + [_P]
        L.PSROIPoolForwardLauncher.argtypes = [_P, _F, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P]
        L.PSROIPoolBackwardLauncher.argtypes = [_P, _P, _I, _I, _F, _I, _I, _I, _I, _I, _I, _P, _P, _P]
        L.ROIAlignForwardLaucher.argtypes = [_P, _F, _I, _I, _I, _I, _I, _I, _P, _P, _P]
        L.ROIAlignBackwardLaucher.argtypes = [_P, _F, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]
        L.ROIPoolForwardLaucher.argtypes = [_P, _F, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]
        L.ROIPoolBackwardLaucher.argtypes = [_P, _F, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]
        L.BilinearSamplerBHWD_updateOutput_cuda_kernel.argtypes = [_I] * 8 + [_P] + [_I] * 4 + [_P] + [_I] * 4 + \
            [_P] + [_I] * 4 + [_P]
        L.BilinearSamplerBHWD_updateGradInput_cuda_kernel.argtypes = [_I] * 8 + ([_P] + [_I] * 4) * 5 + [_P]
        L.nms_cuda_compute.argtypes = [_P, _P, _P, _I, _I, _F]
        L.nms_cuda_compute.restype = None
        _LIBS[fma] = L
    return _LIBS[fma]


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _cu(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=_dev(), dtype=dtype).contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _done(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


# ------------------------------------------------------------------------------------------------ correlation
def correlation_forward(x1, x2, pad, k, d, s1, s2, mult=1, fma=False):
    """correlation/src/correlation_cuda.c:11-95 (Correlation_forward_cuda) around correlation_cuda_kernel.cu."""
    a, b = _cu(x1), _cu(x2)
    B, C, H, W = a.shape
    kr = (k - 1) // 2
    br = kr + d
    pH, pW = H + 2 * pad, W + 2 * pad
    oc = ((d // s2) * 2 + 1) ** 2
    oh = int(np.ceil(np.float32(pH - 2 * br) / np.float32(s1)))
    ow = int(np.ceil(np.float32(pW - 2 * br) / np.float32(s1)))
    r1 = torch.zeros(B, pH, pW, C, device=a.device)
    r2 = torch.zeros(B, pH, pW, C, device=a.device)
    out = torch.zeros(B, oc, oh, ow, device=a.device)
    ok = lib(fma).Correlation_forward_cuda_kernel(_ptr(out), *out.shape, *out.stride(), _ptr(a), C, H, W, *a.stride(),
                                                   _ptr(b), b.shape[1], *b.stride(), _ptr(r1), _ptr(r2), pad, k, d, s1,
                                                   s2, mult, None)
    assert ok, "reference correlation forward reported failure"
    return _done(out)


def correlation_backward(gout, x1, x2, pad, k, d, s1, s2, mult=1, fma=False):
    """correlation/src/correlation_cuda.c:97-183 (Correlation_backward_cuda)."""
    a, b, g = _cu(x1), _cu(x2), _cu(gout)
    B, C, H, W = a.shape
    pH, pW = H + 2 * pad, W + 2 * pad
    r1 = torch.zeros(B, pH, pW, C, device=a.device)
    r2 = torch.zeros(B, pH, pW, C, device=a.device)
    g1, g2 = torch.zeros_like(a), torch.zeros_like(b)
    ok = lib(fma).Correlation_backward_cuda_kernel(_ptr(g), *g.shape, *g.stride(), _ptr(a), C, H, W, *a.stride(),
                                                    _ptr(b), *b.stride(), _ptr(g1), *g1.stride(), _ptr(g2), C,
                                                    *g2.stride(), _ptr(r1), _ptr(r2), pad, k, d, s1, s2, mult, None)
    assert ok, "reference correlation backward reported failure"
    torch.cuda.synchronize()
    return g1.cpu().numpy(), g2.cpu().numpy()


# ------------------------------------------------------------------------------------------------ PSRoI pooling
def psroi_pool_forward(feat, rois, ph, pw, scale, group, od, fma=False):
    """psroi_pooling/src/psroi_pooling_cuda.c:7-37 + functions/psroi_pool.py:18-33 (zeroed output / mapping)."""
    f, r = _cu(feat), _cu(rois)
    n = r.shape[0]
    out = torch.zeros(n, od, ph, pw, device=f.device)
    mapping = torch.zeros(n, od, ph, pw, dtype=torch.int32, device=f.device)
    lib(fma).PSROIPoolForwardLauncher(_ptr(f), scale, n, f.shape[2], f.shape[3], f.shape[1], ph, pw, _ptr(r), group, od,
                                      _ptr(out), _ptr(mapping), None)
    torch.cuda.synchronize()
    return out.cpu().numpy(), mapping.cpu().numpy()


def psroi_pool_backward(top_diff, rois, feat_shape, ph, pw, scale, group, od, mapping, fma=False):
    """psroi_pooling/src/psroi_pooling_cuda.c:39-77 + functions/psroi_pool.py:35-45."""
    g, r, m = _cu(top_diff), _cu(rois), _cu(mapping, torch.int32)
    B, C, H, W = feat_shape
    bottom = torch.zeros(B, C, H, W, device=g.device)
    lib(fma).PSROIPoolBackwardLauncher(_ptr(g), _ptr(m), B, r.shape[0], scale, C, H, W, pw, ph, od, _ptr(bottom), _ptr(r),
                                       None)
    return _done(bottom)


# ------------------------------------------------------------------------------------------------ RoI align
def roi_align_forward(feat, rois, ah, aw, scale, fma=False):
    """roi_align/src/roi_align_cuda.c:8-36 + functions/roi_align.py:16-30."""
    f, r = _cu(feat), _cu(rois)
    out = torch.zeros(r.shape[0], f.shape[1], ah, aw, device=f.device)
    lib(fma).ROIAlignForwardLaucher(_ptr(f), scale, r.shape[0], f.shape[2], f.shape[3], f.shape[1], ah, aw, _ptr(r),
                                    _ptr(out), None)
    return _done(out)


def roi_align_backward(top_diff, rois, feat_shape, ah, aw, scale, fma=False):
    """roi_align/src/roi_align_cuda.c:38-67 + functions/roi_align.py:32-47."""
    g, r = _cu(top_diff), _cu(rois)
    B, C, H, W = feat_shape
    bottom = torch.zeros(B, C, H, W, device=g.device)
    lib(fma).ROIAlignBackwardLaucher(_ptr(g), scale, B, r.shape[0], H, W, C, ah, aw, _ptr(r), _ptr(bottom), None)
    return _done(bottom)


# ------------------------------------------------------------------------------------------------ RoI max pooling
def roi_pool_forward(feat, rois, ph, pw, scale, fma=False):
    """roi_pooling/src/roi_pooling_cuda.c:7-42 + functions/roi_pool.py:15-28."""
    f, r = _cu(feat), _cu(rois)
    n = r.shape[0]
    out = torch.zeros(n, f.shape[1], ph, pw, device=f.device)
    argmax = torch.zeros(n, f.shape[1], ph, pw, dtype=torch.int32, device=f.device)
    lib(fma).ROIPoolForwardLaucher(_ptr(f), scale, n, f.shape[2], f.shape[3], f.shape[1], ph, pw, _ptr(r), _ptr(out),
                                   _ptr(argmax), None)
    torch.cuda.synchronize()
    return out.cpu().numpy(), argmax.cpu().numpy()


def roi_pool_backward(top_diff, rois, argmax, feat_shape, ph, pw, scale, fma=False):
    """roi_pooling/src/roi_pooling_cuda.c:44-79 + functions/roi_pool.py:30-38."""
    g, r, am = _cu(top_diff), _cu(rois), _cu(argmax, torch.int32)
    B, C, H, W = feat_shape
    bottom = torch.zeros(B, C, H, W, device=g.device)
    lib(fma).ROIPoolBackwardLaucher(_ptr(g), scale, B, r.shape[0], H, W, C, ph, pw, _ptr(r), _ptr(bottom), _ptr(am), None)
    return _done(bottom)


# ------------------------------------------------------------------------------------------------ RoI crop
def roi_crop_forward(images, grids, fma=False):
    """roi_crop/src/roi_crop_cuda.c:14-52 + functions/roi_crop.py:7-13: images (B,C,H,W), grids (R,Ho,Wo,2) = (y, x)."""
    im, gr = _cu(images), _cu(grids)
    ob, oh, ow, _ = gr.shape
    out = torch.zeros(ob, im.shape[1], oh, ow, device=im.device)
    gs, os_ = gr.stride(), out.stride()
    ok = lib(fma).BilinearSamplerBHWD_updateOutput_cuda_kernel(
        out.shape[1], out.shape[3], out.shape[2], out.shape[0], im.shape[1], im.shape[2], im.shape[3], im.shape[0],
        _ptr(im), *im.stride(), _ptr(gr), gs[0], gs[3], gs[1], gs[2], _ptr(out), *os_, None)
    assert ok, "reference roi_crop forward reported failure"
    return _done(out)


def roi_crop_backward(images, grids, gout, fma=False):
    """roi_crop/src/roi_crop_cuda.c:54-107 + functions/roi_crop.py:15-21; returns the image gradient (the grid
    gradient buffer is handed over zeroed, as the reference does)."""
    im, gr, go = _cu(images), _cu(grids), _cu(gout)
    gim, ggr = torch.zeros_like(im), torch.zeros_like(gr)
    gs, ggs = gr.stride(), ggr.stride()
    ok = lib(fma).BilinearSamplerBHWD_updateGradInput_cuda_kernel(
        go.shape[1], go.shape[3], go.shape[2], go.shape[0], im.shape[1], im.shape[2], im.shape[3], im.shape[0],
        _ptr(im), *im.stride(), _ptr(gr), gs[0], gs[3], gs[1], gs[2], _ptr(gim), *gim.stride(),
        _ptr(ggr), ggs[0], ggs[3], ggs[1], ggs[2], _ptr(go), *go.stride(), None)
    assert ok, "reference roi_crop backward reported failure"
    return _done(gim)


# ------------------------------------------------------------------------------------------------ NMS
def nms(dets, thresh, fma=False):
    """nms/src/nms_cuda.c:8-18 + nms/nms_gpu.py:6-11: dets (N,5) sorted by descending score; returns kept indices.
    `boxes_host` is given as host memory (the kernel file copies it to the device itself, nms_cuda_kernel.cu:93-100)."""
    d = np.ascontiguousarray(dets, dtype=np.float32)
    n = d.shape[0]
    if n == 0:
        return np.zeros((0,), np.int32)
    keep = torch.zeros(n, dtype=torch.int32, device=_dev())
    num = torch.zeros(1, dtype=torch.int32, device=_dev())
    lib(fma).nms_cuda_compute(_ptr(keep), _ptr(num), d.ctypes.data_as(ctypes.c_void_p), n, d.shape[1], float(thresh))
    torch.cuda.synchronize()
    return keep[:int(num.item())].cpu().numpy()
