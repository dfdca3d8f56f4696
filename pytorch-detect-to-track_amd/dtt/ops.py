"""Operator layer of the D&T hot path: same class names, constructor arguments and forward signatures
as the reference's op packages, backed by libdtt_hip.so through its C ABI.

Reference surface mirrored here (paths relative to the reference's lib/model/):
  Correlation          correlation/modules/correlation.py:5-19   (+ functions/correlation.py:18-50)
  _PSRoIPooling        psroi_pooling/modules/psroi_pool.py:7-18   (+ functions/psroi_pool.py:18-45)
  RoIAlign/Avg/Max     roi_align/modules/roi_align.py:6-42        (+ functions/roi_align.py:7-47)
  _RoIPooling          roi_pooling/modules/roi_pool.py:5-14       (+ functions/roi_pool.py:6-38)
  _RoICrop             roi_crop/modules/roi_crop.py:4-8           (+ functions/roi_crop.py:7-21)
  nms                  nms/nms_wrapper.py:11-18                   (+ nms_gpu.py:6-11)

The reference builds an old-style stateful autograd Function per call; these are static
torch.autograd.Function subclasses.  All ops are GPU-only and raise if the library is missing.
"""
import ctypes
import os

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import check, ptr, require_f32_contig, require_gpu, stream_ptr


def _workspace(nbytes, device):
    return torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------ correlation
def correlation_output_shape(channels, height, width, pad_size, kernel_size, max_displacement, stride1, stride2):
    """Output (channels, height, width) per correlation_cuda.c:25-34."""
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    ok = _lib.lib().dtt_correlation_output_shape(channels, height, width, pad_size, kernel_size, max_displacement,
                                                 stride1, stride2, ctypes.byref(oc), ctypes.byref(oh),
                                                 ctypes.byref(ow))
    check(ok, "correlation_output_shape")
    return oc.value, oh.value, ow.value


def correlation_forward_into(out, input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2,
                             corr_multiply=1):
    """Write the correlation of two (B,C,H,W) maps into `out`, which may be a channel slice
    (B, oc, oh, ow) of a larger contiguous tensor (used to fill the tracking concat buffer in place)."""
    require_gpu(out, input1, input2)
    require_f32_contig("input1", input1)
    require_f32_contig("input2", input2)
    if input1.shape != input2.shape:
        raise ValueError("correlation: input shapes differ: %s vs %s" % (tuple(input1.shape), tuple(input2.shape)))
    B, C, H, W = input1.shape
    oc, oh, ow = correlation_output_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    if tuple(out.shape) != (B, oc, oh, ow) or out.dtype != torch.float32:
        raise ValueError("correlation: out must be float32 %s" % ((B, oc, oh, ow),))
    if out.stride()[1:] != (oh * ow, ow, 1):
        raise ValueError("correlation: out must be a dense channel slice")
    L = _lib.lib()
    nbytes = L.dtt_correlation_forward_workspace_bytes(B, C, H, W, pad_size, kernel_size, max_displacement, stride1,
                                                       stride2)
    ws = _workspace(nbytes, input1.device)
    with torch.cuda.device(input1.device):
        check(L.dtt_correlation_forward(ptr(out), B, oc, oh, ow, out.stride(0), ptr(input1), C, H, W, ptr(input2),
                                        ptr(ws), nbytes, pad_size, kernel_size, max_displacement, stride1, stride2,
                                        corr_multiply, stream_ptr(input1.device)), "correlation forward")
    return out


def correlation_forward_rows(rows, col, input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2,
                             corr_multiply=1):
    """Write the correlation of two (B,C,H,W) maps into columns [col, col + oc) of `rows`, a position-major
    (B*oh*ow, ld) fp32 matrix (row = output pixel, batch-major): the tracking head's GEMM input (dtt.heads)."""
    require_gpu(rows, input1, input2)
    require_f32_contig("input1", input1)
    require_f32_contig("input2", input2)
    if input1.shape != input2.shape:
        raise ValueError("correlation: input shapes differ: %s vs %s" % (tuple(input1.shape), tuple(input2.shape)))
    B, C, H, W = input1.shape
    oc, oh, ow = correlation_output_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    if rows.dim() != 2 or rows.dtype != torch.float32 or rows.stride(1) != 1 or rows.shape[0] != B * oh * ow \
            or col < 0 or col + oc > rows.shape[1]:
        raise ValueError("correlation: rows must be float32 (%d, >= %d) with unit column stride" % (B * oh * ow, col + oc))
    L = _lib.lib()
    nbytes = L.dtt_correlation_forward_workspace_bytes(B, C, H, W, pad_size, kernel_size, max_displacement, stride1,
                                                       stride2)
    ws = _workspace(nbytes, input1.device)
    ld = rows.stride(0)
    out = ctypes.c_void_p(rows.data_ptr() + 4 * col)
    with torch.cuda.device(input1.device):
        check(L.dtt_correlation_forward_strided(out, B, oc, oh, ow, oh * ow * ld, 1, ld, ptr(input1), C, H, W, ptr(input2),
                                                ptr(ws), nbytes, pad_size, kernel_size, max_displacement, stride1,
                                                stride2, corr_multiply, stream_ptr(input1.device)),
              "correlation forward (position-major)")
    return rows


def correlation_forward_nhwc(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2, rows=None, col=0,
                             max_workgroups=0):
    """Correlation of two channels-last (B, C, H, W)-shaped maps (memory order B, H, W, C -- the channels-last trunk's
    own layout) without any layout change in front of the op (`dtt_correlation_forward_nhwc`: the window-split kernel,
    one launch, no workspace).
    rows is None: returns the reference's (B, D*D, oh, ow) NCHW tensor.  Otherwise writes columns [col, col + D*D) of
    `rows`, a position-major (B*oh*ow, ld) matrix (the tracking head's GEMM input, dtt.heads).
    max_workgroups: 0 = plan one round over every CU; n = plan for n CUs (other kernels run beside this one)."""
    require_gpu(input1, input2)
    if input1.shape != input2.shape or input1.dim() != 4:
        raise ValueError("correlation: input shapes differ: %s vs %s" % (tuple(input1.shape), tuple(input2.shape)))
    for name, t in (("input1", input1), ("input2", input2)):
        if t.dtype != torch.float32 or not t.is_contiguous(memory_format=torch.channels_last):
            raise ValueError("%s must be float32 in channels-last memory" % name)
    B, C, H, W = input1.shape
    oc, oh, ow = correlation_output_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    L = _lib.lib()
    if rows is None:
        out = torch.empty((B, oc, oh, ow), dtype=torch.float32, device=input1.device)
        optr, sb, sc, sp, ret = ptr(out), oc * oh * ow, oh * ow, 1, out
    else:
        if rows.dim() != 2 or rows.dtype != torch.float32 or rows.stride(1) != 1 or rows.shape[0] != B * oh * ow \
                or col < 0 or col + oc > rows.shape[1]:
            raise ValueError("correlation: rows must be float32 (%d, >= %d) with unit column stride" % (B * oh * ow, col + oc))
        ld = rows.stride(0)
        optr, sb, sc, sp, ret = ctypes.c_void_p(rows.data_ptr() + 4 * col), oh * ow * ld, 1, ld, rows
    with torch.cuda.device(input1.device):
        check(L.dtt_correlation_forward_nhwc(optr, B, oc, oh, ow, sb, sc, sp, ptr(input1), C, H, W, ptr(input2),
                                             pad_size, kernel_size, max_displacement, stride1, stride2,
                                             int(max_workgroups), stream_ptr(input1.device)),
              "correlation forward (channels-last)")
    return ret


class CorrelationFunction(Function):
    @staticmethod
    def forward(ctx, input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply):
        require_gpu(input1, input2)
        ctx.save_for_backward(input1, input2)
        ctx.params = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
        B, C, H, W = input1.shape
        oc, oh, ow = correlation_output_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
        out = torch.empty((B, oc, oh, ow), dtype=torch.float32, device=input1.device)
        return correlation_forward_into(out, input1, input2, pad_size, kernel_size, max_displacement, stride1,
                                        stride2, corr_multiply)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply = ctx.params
        grad_output = grad_output.contiguous()
        B, C, H, W = input1.shape
        g1 = torch.empty_like(input1)
        g2 = torch.empty_like(input2)
        with torch.cuda.device(input1.device):
            check(_lib.lib().dtt_correlation_backward(ptr(grad_output), grad_output.shape[0], grad_output.shape[1],
                                                      grad_output.shape[2], grad_output.shape[3], ptr(input1), C, H,
                                                      W, ptr(input2), ptr(g1), ptr(g2), pad_size, kernel_size,
                                                      max_displacement, stride1, stride2, corr_multiply,
                                                      stream_ptr(input1.device)), "correlation backward")
        return g1, g2, None, None, None, None, None, None


def _is_channels_last(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


def corr_bwd_stream_enabled():
    """DTT_CORR_BWD_STREAM=0 (developer A/B switch): the band-stationary streamed gradient kernels off, round 1's kernels instead."""
    return os.environ.get("DTT_CORR_BWD_STREAM", "1") != "0"


def correlation_backward_nhwc(grad_output, input1, input2, g1, g2, pad_size, kernel_size, max_displacement, stride1, stride2,
                              rows=None, col=0, phase=3, workspace=None):
    """Both correlation gradients on channels-last maps, written into the channels-last tensors g1 / g2 (either may be None).
    grad_output: the reference's (B, D*D, oh, ow) tensor -- or, with `rows`, columns [col, col + D*D) of a position-major
    (B*oh*ow, ld) matrix (the gradient of the tracking head's input rows: read where it lies).  Channels % 64 == 0: the
    band-stationary streamed kernels (`dtt_correlation_backward_nhwc_phase`, workspace from the caching allocator); other
    channel counts: round 1's kernels (`dtt_correlation_backward_nhwc`; contiguous grad_output, both gradients).
    phase (streamed kernels only): 1 = lay out the band words and RETURN the workspace tensor (issued on the current stream: the
    caller may run it on a second stream), 2 = the gradients from `workspace`, 3 = both."""
    L = _lib.lib()
    B, C, H, W = input1.shape
    oc, oh, ow = correlation_output_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    dev = input1.device
    streamed = bool(L.dtt_correlation_backward_stream_supported(C, kernel_size, max_displacement, stride1, stride2)) and corr_bwd_stream_enabled()
    if phase != 3 and not streamed:
        raise ValueError("correlation backward: phases are a feature of the streamed kernels (channels % 64 == 0)")
    with torch.cuda.device(dev):
        if not streamed:
            # round 1's kernels: a contiguous (B, D*D, oh, ow) gradOutput and both gradient outputs, dense channels-last.  A rows-form
            # gradient is gathered into that tensor first, a gradient the caller does not want goes to a scratch map (this path is the
            # developer switch's and the fallback of channel counts that are not a multiple of 64: correctness first)
            if max_displacement // max(stride2, 1) > 8:
                raise ValueError("correlation backward (channels-last): window radius %d needs the streamed kernels (channels %% 64 == 0, "
                                 "DTT_CORR_BWD_STREAM not 0)" % (max_displacement // max(stride2, 1)))
            if rows is not None:
                if rows.dim() != 2 or rows.dtype != torch.float32 or rows.shape[0] != B * oh * ow or col < 0 or col + oc > rows.shape[1]:
                    raise ValueError("correlation backward: rows must be float32 (%d, >= %d)" % (B * oh * ow, col + oc))
                grad_output = rows[:, col:col + oc].reshape(B, oh, ow, oc).permute(0, 3, 1, 2)
            grad_output = grad_output.contiguous()
            t1 = g1 if g1 is not None else torch.empty_like(input1)
            t2 = g2 if g2 is not None else torch.empty_like(input2)
            for t in (t1, t2):   # the kernel writes dense NHWC: a channel- or spatially-sliced destination would be overrun
                if t.shape != input1.shape or t.dtype != torch.float32 or not t.is_contiguous(memory_format=torch.channels_last):
                    raise ValueError("correlation backward (channels-last): the gradient maps must be dense channels-last float32 %s "
                                     "(batch slices are; channel or spatial slices are not)" % (tuple(input1.shape),))
            check(L.dtt_correlation_backward_nhwc(ptr(grad_output), B, oc, oh, ow, ptr(input1), C, H, W, ptr(input2), ptr(t1), ptr(t2),
                                                  pad_size, kernel_size, max_displacement, stride1, stride2, stream_ptr(dev)),
                  "correlation backward (channels-last)")
            return
        if rows is None:
            grad_output = grad_output.contiguous()
            gptr, sb, sc, sp = ptr(grad_output), oc * oh * ow, oh * ow, 1
        else:
            if rows.dim() != 2 or rows.dtype != torch.float32 or rows.stride(1) != 1 or rows.shape[0] != B * oh * ow \
                    or col < 0 or col + oc > rows.shape[1]:
                raise ValueError("correlation backward: rows must be float32 (%d, >= %d) with unit column stride" % (B * oh * ow, col + oc))
            ld = rows.stride(0)
            gptr, sb, sc, sp = ctypes.c_void_p(rows.data_ptr() + 4 * col), oh * ow * ld, 1, ld
        nbytes = int(L.dtt_correlation_backward_workspace_bytes(B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2))
        ws = workspace if phase == 2 else torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=dev)
        if ws is None or ws.numel() * 4 < nbytes:
            raise ValueError("correlation backward: phase 2 needs the workspace phase 1 returned")
        which = 3 if phase == 1 else (1 if g1 is not None else 0) | (2 if g2 is not None else 0)
        check(L.dtt_correlation_backward_nhwc_phase(gptr, sb, sc, sp, B, oc, oh, ow, ptr(input1), C, H, W, ptr(input2),
                                                    ptr(g1) if g1 is not None else None, ptr(g2) if g2 is not None else None,
                                                    pad_size, kernel_size, max_displacement, stride1, stride2, which, phase,
                                                    ptr(ws), ws.numel() * 4, stream_ptr(dev)),
              "correlation backward (channels-last, streamed)")
        return ws if phase == 1 else None


class CorrelationNHWCFunction(Function):
    """The same op on channels-last maps, for a channels-last training trunk: forward = the window-split kernel
    (`dtt_correlation_forward_nhwc`, NCHW output as the reference's), backward = the matrix-core gradients reading and
    writing channels-last (`dtt_correlation_backward_nhwc`) -- no layout conversion of the feature maps or of their
    gradients in either direction.  Geometry: kernel_size 1, stride1 == stride2, max_displacement / stride <= 8,
    channels % 16 == 0 (`Correlation.forward` routes everything else through the NCHW functions)."""

    MAX_RADIUS = 16  # window radius max_displacement / stride the channels-last kernels take (gradients: channels % 64 == 0; round 1's
                     # gradient kernels, which take the other channel counts, stop at 8)

    @staticmethod
    def supports(input1, input2, kernel_size, max_displacement, stride1, stride2, pad_size=None):
        """The geometries both channels-last kernels take (dtt_correlation_forward_nhwc / _backward_nhwc*): everything else goes
        through `.contiguous()` + CorrelationFunction.  pad_size None: the caller's padding is not checked (legacy callers)."""
        if not (input1.is_cuda and input1.dtype == torch.float32 and input1.shape == input2.shape and
                _is_channels_last(input1) and _is_channels_last(input2) and kernel_size == 1 and stride1 == stride2 and
                stride2 > 0 and input1.size(1) % 16 == 0):
            return False
        radius = max_displacement // stride2
        wide = input1.size(1) % 64 == 0 and corr_bwd_stream_enabled()    # radius 9 .. 16: only the streamed gradient kernels take it
        if not 1 <= radius <= (CorrelationNHWCFunction.MAX_RADIUS if wide else 8):
            return False
        # the kernels address the stride lattice: displacement and (displacement - padding) must be multiples of the stride
        if max_displacement % stride2 != 0 or (pad_size is not None and (max_displacement - pad_size) % stride2 != 0):
            return False
        return True

    @staticmethod
    def forward(ctx, input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2):
        ctx.save_for_backward(input1, input2)
        ctx.params = (pad_size, kernel_size, max_displacement, stride1, stride2)
        return correlation_forward_nhwc(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        pad_size, kernel_size, max_displacement, stride1, stride2 = ctx.params
        grad_output = grad_output.contiguous()
        B, C, H, W = input1.shape
        g1 = torch.empty_like(input1)   # (channels-last, as the inputs)
        g2 = torch.empty_like(input2)
        assert _is_channels_last(g1) and _is_channels_last(g2)
        correlation_backward_nhwc(grad_output, input1, input2, g1, g2, pad_size, kernel_size, max_displacement, stride1, stride2)
        return g1, g2, None, None, None, None, None


class CorrelationPairNHWCFunction(Function):
    """CorrelationNHWCFunction for two legs of ONE channels-last batch tensor `maps` (n_legs * B, C, H, W): images
    [i*B, (i+1)*B) against images [j*B, (j+1)*B).  Slicing the batch under autograd would hand the trunk an NCHW zeros
    tensor with the channels-last gradient copied in (SliceBackward does not keep the memory format: a strided copy of the
    whole map per leg and NCHW gradients into a channels-last trunk); here the gradient of `maps` is allocated channels-last
    once and the two gradient kernels write their legs of it in place."""

    @staticmethod
    def forward(ctx, maps, B, i, j, pad_size, kernel_size, max_displacement, stride1, stride2):
        ctx.save_for_backward(maps)
        ctx.params = (B, i, j, pad_size, kernel_size, max_displacement, stride1, stride2)
        return correlation_forward_nhwc(maps[i * B:(i + 1) * B], maps[j * B:(j + 1) * B], pad_size, kernel_size,
                                        max_displacement, stride1, stride2)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (maps,) = ctx.saved_tensors
        B, i, j, pad_size, kernel_size, max_displacement, stride1, stride2 = ctx.params
        grad_output = grad_output.contiguous()
        n, C, H, W = maps.shape
        # (two legs: every image's gradient is written by one of the two kernels, which zero-fill their outputs themselves)
        g = torch.empty_like(maps) if n == 2 * B else torch.zeros_like(maps)
        assert _is_channels_last(g) or n == 1
        g1, g2 = g[i * B:(i + 1) * B], g[j * B:(j + 1) * B]
        in1, in2 = maps[i * B:(i + 1) * B], maps[j * B:(j + 1) * B]
        correlation_backward_nhwc(grad_output, in1, in2, g1, g2, pad_size, kernel_size, max_displacement, stride1, stride2)
        return g, None, None, None, None, None, None, None, None


class Correlation(nn.Module):
    """correlation/modules/correlation.py:5-13 (same argument order and defaults).  Channels-last inputs (a channels-last
    trunk) take the channels-last kernels when the geometry allows; the result is the same NCHW tensor either way."""

    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        self.pad_size = pad_size
        self.kernel_size = kernel_size
        self.max_displacement = max_displacement
        self.stride1 = stride1
        self.stride2 = stride2
        self.corr_multiply = corr_multiply

    def forward(self, input1, input2):
        if CorrelationNHWCFunction.supports(input1, input2, self.kernel_size, self.max_displacement, self.stride1, self.stride2,
                                            self.pad_size):
            return CorrelationNHWCFunction.apply(input1, input2, self.pad_size, self.kernel_size, self.max_displacement,
                                                 self.stride1, self.stride2)
        if input1.is_cuda and not input1.is_contiguous():
            input1 = input1.contiguous()
        if input2.is_cuda and not input2.is_contiguous():
            input2 = input2.contiguous()
        return CorrelationFunction.apply(input1, input2, self.pad_size, self.kernel_size, self.max_displacement,
                                         self.stride1, self.stride2, self.corr_multiply)

    def pair(self, maps, B, i, j):
        """Legs i and j of a (n_legs * B, C, H, W) batch tensor; channels-last maps stay whole under autograd
        (CorrelationPairNHWCFunction), anything else is sliced and goes through forward()."""
        a, b = maps[i * B:(i + 1) * B], maps[j * B:(j + 1) * B]
        if i != j and CorrelationNHWCFunction.supports(a, b, self.kernel_size, self.max_displacement, self.stride1, self.stride2,
                                                       self.pad_size):
            return CorrelationPairNHWCFunction.apply(maps, B, i, j, self.pad_size, self.kernel_size, self.max_displacement,
                                                     self.stride1, self.stride2)
        return self.forward(a, b)

    def extra_repr(self):
        return "pad_size=%d, kernel_size=%d, max_displacement=%d, stride1=%d, stride2=%d" % (
            self.pad_size, self.kernel_size, self.max_displacement, self.stride1, self.stride2)


# ---------------------------------------------------------------------------------- PSRoI pooling
def _check_rois(rois):
    if rois.dim() != 2 or rois.size(1) != 5:
        # psroi_pooling_cuda.c:17-21 returns 0 when rois.size(1) != 5
        raise ValueError("rois must have shape (R, 5) [batch_idx, x1, y1, x2, y2], got %s" % (tuple(rois.shape),))
    require_f32_contig("rois", rois)


class PSRoIPoolFunction(Function):
    @staticmethod
    def forward(ctx, features, rois, pooled_height, pooled_width, spatial_scale, group_size, output_dim):
        require_gpu(features, rois)
        require_f32_contig("features", features)
        _check_rois(rois)
        B, C, H, W = features.shape
        R = rois.size(0)
        out = torch.empty((R, output_dim, pooled_height, pooled_width), dtype=torch.float32, device=features.device)
        with torch.cuda.device(features.device):
            check(_lib.lib().dtt_psroi_pool_forward(ptr(features), spatial_scale, B, R, H, W, C, pooled_height,
                                                    pooled_width, ptr(rois), group_size, output_dim, ptr(out), None,
                                                    stream_ptr(features.device)), "psroi_pool forward")
        ctx.save_for_backward(rois)
        ctx.cfg = (pooled_height, pooled_width, spatial_scale, group_size, output_dim, tuple(features.shape))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        pooled_height, pooled_width, spatial_scale, group_size, output_dim, fshape = ctx.cfg
        B, C, H, W = fshape
        grad_output = grad_output.contiguous()
        grad_input = torch.empty(fshape, dtype=torch.float32, device=grad_output.device)
        with torch.cuda.device(grad_output.device):
            check(_lib.lib().dtt_psroi_pool_backward(ptr(grad_output), None, B, rois.size(0), spatial_scale, C, H, W,
                                                     pooled_width, pooled_height, output_dim, group_size,
                                                     ptr(grad_input), ptr(rois), stream_ptr(grad_output.device)),
                  "psroi_pool backward")
        return grad_input, None, None, None, None, None, None


class _PSRoIPooling(nn.Module):
    """psroi_pooling/modules/psroi_pool.py:7-18."""

    def __init__(self, pooled_height, pooled_width, spatial_scale, group_size, output_dim):
        super().__init__()
        self.pooled_width = int(pooled_width)
        self.pooled_height = int(pooled_height)
        self.spatial_scale = float(spatial_scale)
        self.group_size = int(group_size)
        self.output_dim = int(output_dim)

    def forward(self, features, rois):
        return PSRoIPoolFunction.apply(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale,
                                       self.group_size, self.output_dim)


def psroi_pool_vote(features, rois, pooled_height, pooled_width, spatial_scale, group_size, output_dim):
    """Inference-only fused PSRoI pool + AvgPool2d((P,P)) vote (rfcn.py:62-64, 136-140): returns
    (pooled (R, od, P, P), vote (R, od))."""
    require_gpu(features, rois)
    require_f32_contig("features", features)
    _check_rois(rois)
    B, C, H, W = features.shape
    R = rois.size(0)
    pooled = torch.empty((R, output_dim, pooled_height, pooled_width), dtype=torch.float32, device=features.device)
    vote = torch.empty((R, output_dim), dtype=torch.float32, device=features.device)
    with torch.cuda.device(features.device):
        check(_lib.lib().dtt_psroi_pool_vote_forward(ptr(features), spatial_scale, B, R, H, W, C, pooled_height,
                                                     pooled_width, ptr(rois), group_size, output_dim, ptr(pooled),
                                                     ptr(vote), stream_ptr(features.device)), "psroi_pool vote")
    return pooled, vote


def psroi_vote(features, rois, pooled_height, pooled_width, spatial_scale, group_size, output_dim):
    """Inference-only: the (R, od) vote of `psroi_pool_vote` without materialising the pooled tensor in the reference
    layout (the bins live in a channel-major scratch buffer; bit-identical vote)."""
    require_gpu(features, rois)
    require_f32_contig("features", features)
    _check_rois(rois)
    B, C, H, W = features.shape
    R = rois.size(0)
    scratch = torch.empty(max(C * R, 1), dtype=torch.float32, device=features.device)
    vote = torch.empty((R, output_dim), dtype=torch.float32, device=features.device)
    with torch.cuda.device(features.device):
        check(_lib.lib().dtt_psroi_vote_forward(ptr(features), spatial_scale, B, R, H, W, C, pooled_height, pooled_width,
                                                ptr(rois), group_size, output_dim, ptr(scratch), ptr(vote),
                                                stream_ptr(features.device)), "psroi vote")
    return vote


# -------------------------------------------------------------------------------------------- NMS
def nms(dets, thresh, force_cpu=False, max_keep=0):
    """nms/nms_wrapper.py:11-18: dets (N, 5) [x1,y1,x2,y2,score] sorted by descending score ->
    int32 (n_keep, 1) indices, or [] for empty input.  The greedy sweep runs on the device; the only
    host synchronisation is reading n_keep to size the result (the reference does the same,
    nms_gpu.py:10 `keep[:num_out[0]]`)."""
    if dets.shape[0] == 0:
        return []
    if force_cpu:
        raise RuntimeError("force_cpu=True: there is no CPU NMS in this package (GPU-only hot path)")
    require_gpu(dets)
    require_f32_contig("dets", dets)
    n, dim = dets.shape
    L = _lib.lib()
    keep = torch.empty((n,), dtype=torch.int32, device=dets.device)
    num = torch.empty((1,), dtype=torch.int32, device=dets.device)
    nbytes = L.dtt_nms_workspace_bytes(n)
    ws = _workspace(nbytes, dets.device)
    with torch.cuda.device(dets.device):
        check(L.dtt_nms(ptr(keep), ptr(num), ptr(dets), n, dim, float(thresh), int(max_keep), ptr(ws), nbytes,
                        stream_ptr(dets.device)), "nms")
    return keep[: int(num.item())].view(-1, 1)


# -------------------------------------------------------------------------------------- RoI Align
class RoIAlignFunction(Function):
    @staticmethod
    def forward(ctx, features, rois, aligned_height, aligned_width, spatial_scale):
        require_gpu(features, rois)
        require_f32_contig("features", features)
        _check_rois(rois)
        B, C, H, W = features.shape
        R = rois.size(0)
        out = torch.empty((R, C, aligned_height, aligned_width), dtype=torch.float32, device=features.device)
        with torch.cuda.device(features.device):
            check(_lib.lib().dtt_roi_align_forward_planes(ptr(features), spatial_scale, B, R, H, W, C, aligned_height,
                                                   aligned_width, ptr(rois), ptr(out), 0,
                                                   stream_ptr(features.device)), "roi_align forward")
        ctx.save_for_backward(rois)
        ctx.cfg = (aligned_height, aligned_width, spatial_scale, tuple(features.shape))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        ah, aw, scale, fshape = ctx.cfg
        B, C, H, W = fshape
        grad_output = grad_output.contiguous()
        grad_input = torch.zeros(fshape, dtype=torch.float32, device=grad_output.device)
        with torch.cuda.device(grad_output.device):
            check(_lib.lib().dtt_roi_align_backward(ptr(grad_output), scale, B, rois.size(0), H, W, C, ah, aw,
                                                    ptr(rois), ptr(grad_input), stream_ptr(grad_output.device)),
                  "roi_align backward")
        return grad_input, None, None, None, None


def _roi_align_pooled(features, rois, h, w, scale, mode):
    require_gpu(features, rois)
    require_f32_contig("features", features)
    _check_rois(rois)
    B, C, H, W = features.shape
    out = torch.empty((rois.size(0), C, h, w), dtype=torch.float32, device=features.device)
    with torch.cuda.device(features.device):
        check(_lib.lib().dtt_roi_align_forward_planes(ptr(features), scale, B, rois.size(0), H, W, C, h, w, ptr(rois),
                                                      ptr(out), mode, stream_ptr(features.device)), "roi_align forward")
    return out


class RoIAlign(nn.Module):
    """roi_align/modules/roi_align.py:6-16."""

    def __init__(self, aligned_height, aligned_width, spatial_scale):
        super().__init__()
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        return RoIAlignFunction.apply(features, rois, self.aligned_height, self.aligned_width, self.spatial_scale)


class RoIAlignAvg(RoIAlign):
    """roi_align/modules/roi_align.py:18-29: samples on (h+1)x(w+1), then avg_pool2d(2, stride 1).
    Without autograd the pooling is fused into the sampling kernel."""

    def forward(self, features, rois):
        if torch.is_grad_enabled() and features.requires_grad:
            x = RoIAlignFunction.apply(features, rois, self.aligned_height + 1, self.aligned_width + 1,
                                       self.spatial_scale)
            return torch.nn.functional.avg_pool2d(x, kernel_size=2, stride=1)
        return _roi_align_pooled(features, rois, self.aligned_height, self.aligned_width, self.spatial_scale, 1)


class RoIAlignMax(RoIAlign):
    """roi_align/modules/roi_align.py:31-42."""

    def forward(self, features, rois):
        if torch.is_grad_enabled() and features.requires_grad:
            x = RoIAlignFunction.apply(features, rois, self.aligned_height + 1, self.aligned_width + 1,
                                       self.spatial_scale)
            return torch.nn.functional.max_pool2d(x, kernel_size=2, stride=1)
        return _roi_align_pooled(features, rois, self.aligned_height, self.aligned_width, self.spatial_scale, 2)


# ---------------------------------------------------------------------------------- RoI max pool
class RoIPoolFunction(Function):
    @staticmethod
    def forward(ctx, features, rois, pooled_height, pooled_width, spatial_scale):
        require_gpu(features, rois)
        require_f32_contig("features", features)
        _check_rois(rois)
        B, C, H, W = features.shape
        R = rois.size(0)
        out = torch.empty((R, C, pooled_height, pooled_width), dtype=torch.float32, device=features.device)
        argmax = torch.empty((R, C, pooled_height, pooled_width), dtype=torch.int32, device=features.device)
        with torch.cuda.device(features.device):
            check(_lib.lib().dtt_roi_pool_forward(ptr(features), spatial_scale, R, H, W, C, pooled_height,
                                                  pooled_width, ptr(rois), ptr(out), ptr(argmax),
                                                  stream_ptr(features.device)), "roi_pool forward")
        ctx.save_for_backward(rois, argmax)
        ctx.cfg = (pooled_height, pooled_width, spatial_scale, tuple(features.shape))
        ctx.mark_non_differentiable(argmax)
        return out, argmax

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output, _grad_argmax):
        rois, argmax = ctx.saved_tensors
        ph, pw, scale, fshape = ctx.cfg
        B, C, H, W = fshape
        grad_output = grad_output.contiguous()
        grad_input = torch.zeros(fshape, dtype=torch.float32, device=grad_output.device)
        with torch.cuda.device(grad_output.device):
            check(_lib.lib().dtt_roi_pool_backward(ptr(grad_output), scale, B, rois.size(0), H, W, C, ph, pw,
                                                   ptr(rois), ptr(grad_input), ptr(argmax),
                                                   stream_ptr(grad_output.device)), "roi_pool backward")
        return grad_input, None, None, None, None


class _RoIPooling(nn.Module):
    """roi_pooling/modules/roi_pool.py:5-14."""

    def __init__(self, pooled_height, pooled_width, spatial_scale):
        super().__init__()
        self.pooled_width = int(pooled_width)
        self.pooled_height = int(pooled_height)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        return RoIPoolFunction.apply(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale)[0]


# --------------------------------------------------------------------------------------- RoI crop
class RoICropFunction(Function):
    @staticmethod
    def forward(ctx, input1, input2):
        """input1: images (B,C,H,W); input2: grids (R,Ho,Wo,2) holding (y, x) in [-1, 1]."""
        require_gpu(input1, input2)
        require_f32_contig("input1", input1)
        require_f32_contig("input2", input2)
        ib, ic, ih, iw = input1.shape
        ob, oh, ow, two = input2.shape
        if two != 2:
            raise ValueError("grid must have shape (R, Ho, Wo, 2)")
        out = torch.empty((ob, ic, oh, ow), dtype=torch.float32, device=input1.device)
        with torch.cuda.device(input1.device):
            check(_lib.lib().dtt_roi_crop_forward(ic, ow, oh, ob, ic, ih, iw, ib, ptr(input1), ptr(input2), ptr(out),
                                                  stream_ptr(input1.device)), "roi_crop forward")
        ctx.save_for_backward(input1, input2)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        ib, ic, ih, iw = input1.shape
        ob, oh, ow, _ = input2.shape
        grad_output = grad_output.contiguous()
        grad_input1 = torch.zeros_like(input1)
        with torch.cuda.device(input1.device):
            check(_lib.lib().dtt_roi_crop_backward(ic, ow, oh, ob, ic, ih, iw, ib, ptr(input1), ptr(input2),
                                                   ptr(grad_input1), ptr(grad_output), stream_ptr(input1.device)),
                  "roi_crop backward")
        # the reference never writes the grid gradient (roi_crop_cuda_kernel.cu:154-192): zeros
        return grad_input1, torch.zeros_like(input2)


class _RoICrop(nn.Module):
    """roi_crop/modules/roi_crop.py:4-8."""

    def __init__(self, layout="BHWD"):
        super().__init__()

    def forward(self, input1, input2):
        return RoICropFunction.apply(input1, input2)


def affine_grid_gen(rois, input_size, grid_size):
    """Sampling grids for the 'crop' pooling mode (net_utils.py:143-165, `_affine_grid_gen`).

    rois (R,5) rows [batch, x1, y1, x2, y2] in image pixels; input_size = (H, W) of the stride-16 feature map.
    Returns (R, grid_size, grid_size, 2) holding (x, y) in the [-1, 1] corner-aligned convention of the torch 0.3
    `F.affine_grid` the reference calls: x = theta00 * u + theta02, y = theta11 * v + theta12 with u, v =
    linspace(-1, 1, grid_size).  Written out directly so the result does not depend on the align_corners default.
    """
    rois = rois.detach()
    height, width = int(input_size[0]), int(input_size[1])
    x1, y1, x2, y2 = (rois[:, k] / 16.0 for k in (1, 2, 3, 4))
    sx, tx = (x2 - x1) / (width - 1), (x1 + x2 - width + 1) / (width - 1)
    sy, ty = (y2 - y1) / (height - 1), (y1 + y2 - height + 1) / (height - 1)
    lin = torch.linspace(-1.0, 1.0, grid_size, device=rois.device, dtype=rois.dtype) if grid_size > 1 else \
        torch.full((1,), -1.0, device=rois.device, dtype=rois.dtype)
    gx = sx[:, None, None] * lin[None, None, :] + tx[:, None, None]
    gy = sy[:, None, None] * lin[None, :, None] + ty[:, None, None]
    return torch.stack([gx.expand(-1, grid_size, grid_size), gy.expand(-1, grid_size, grid_size)], 3).contiguous()


def roi_crop_pool(base_feat, rois, pooling_size, max_pool=True):
    """The 'crop' branch of the detector's RoI pooling (faster_rcnn.py:73-80): affine grids from the RoIs, (x, y)
    swapped to the (y, x) order `_RoICrop` samples with, bilinear crop on the HIP kernel, optional 2x2 max pool
    (cfg.CROP_RESIZE_WITH_MAX_POOL, grid_size = 2 * POOLING_SIZE)."""
    grid_size = pooling_size * 2 if max_pool else pooling_size
    grid_xy = affine_grid_gen(rois.view(-1, 5), base_feat.shape[2:], grid_size)
    grid_yx = torch.stack([grid_xy[..., 1], grid_xy[..., 0]], 3).contiguous()
    pooled = RoICropFunction.apply(base_feat, grid_yx.detach())
    if max_pool:
        pooled = torch.nn.functional.max_pool2d(pooled, 2, 2)
    return pooled
