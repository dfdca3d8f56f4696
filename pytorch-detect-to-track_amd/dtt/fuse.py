"""Inference-time fusion of the frozen-BatchNorm ResNet trunk.

The reference keeps BatchNorm frozen everywhere (eval mode + requires_grad False, faster_rcnn/resnet.py:290-295,
325-330), so at inference every `conv -> BN -> ReLU` is `relu(conv(x; w * s) + t)` with s = gamma / sqrt(var + eps),
t = beta - mean * s folded once, and the end of a bottleneck is `relu(conv3'(.) + t3 + shortcut)`.  The convolutions
still run on MIOpen / rocBLAS; the per-channel bias, the residual add and the ReLU become ONE in-place HIP pass
(`dtt_bias_act_inplace`) instead of the 2-3 elementwise launches (7 full-tensor passes per bottleneck) of the
unfused graph.  Weights are not modified: the folded copies live next to the original modules, the state_dict /
checkpoint layout is untouched, and `unfuse()` (or `.train()`) goes back to the reference graph.
"""
import os

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import check, ptr, stream_ptr


def bias_act_(x, bias, residual=None, relu=True):
    """In place: x = act(x + bias[None, :, None, None] (+ residual)).  x (N,C,H,W) contiguous fp32 on the GPU."""
    assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
    N, C = x.shape[0], x.shape[1]
    hw = x.numel() // (N * C)
    if residual is not None:
        assert residual.shape == x.shape and residual.is_contiguous()
    L = _lib.lib()
    per = max(1, 65535 // C)  # images per launch (grid.y limit)
    with torch.cuda.device(x.device):
        for n0 in range(0, N, per):
            n1 = min(N, n0 + per)
            check(L.dtt_bias_act_inplace(ptr(x[n0:n1]), ptr(bias), ptr(residual[n0:n1]) if residual is not None else None,
                                         n1 - n0, C, hw, int(relu), stream_ptr(x.device)), "bias_act")
    return x


def _fold(conv, bn):
    """(w', b') with conv'(x) = bn(conv(x)) for a frozen BatchNorm."""
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    w = conv.weight * s.view(-1, 1, 1, 1)
    b = bn.bias - bn.running_mean * s
    if conv.bias is not None:
        b = b + conv.bias * s
    return w.detach().contiguous(), b.detach().contiguous()


class _FusedConv:
    def __init__(self, conv, bn=None):
        if bn is not None:
            self.w, self.b = _fold(conv, bn)
        else:
            self.w = conv.weight.detach()
            self.b = conv.bias.detach().contiguous() if conv.bias is not None else torch.zeros(
                conv.out_channels, device=conv.weight.device)
        self.kw = dict(stride=conv.stride, padding=conv.padding, dilation=conv.dilation)

    def conv(self, x):
        return F.conv2d(x, self.w, None, **self.kw)


class _FusedBottleneck:
    def __init__(self, blk):
        self.c1, self.c2, self.c3 = _FusedConv(blk.conv1, blk.bn1), _FusedConv(blk.conv2, blk.bn2), _FusedConv(blk.conv3, blk.bn3)
        self.down = None
        if blk.downsample is not None:
            self.down = _FusedConv(blk.downsample[0], blk.downsample[1])
            self.b3 = (self.c3.b + self.down.b).contiguous()  # both biases land in the single final pass
        else:
            self.b3 = self.c3.b

    def __call__(self, x):
        out = bias_act_(self.c1.conv(x), self.c1.b)
        out = bias_act_(self.c2.conv(out), self.c2.b)
        out = self.c3.conv(out)
        res = x if self.down is None else self.down.conv(x)
        return bias_act_(out, self.b3, res.contiguous())


class FusedTrunk:
    """Callable replacement for resnet._im_to_head at inference."""

    def __init__(self, model):
        b = model.RFCN_base
        self.stem = _FusedConv(b[0], b[1])
        self.pool = b[3]
        self.stages = [[_FusedBottleneck(blk) for blk in b[i]] for i in (4, 5, 6, 7)]
        self.top = _FusedConv(b[8])  # RFCN_net (3x3, dilation 6, has its own bias) + the ReLU after it

    @torch.no_grad()
    def __call__(self, x):
        x = self.pool(bias_act_(self.stem.conv(x), self.stem.b))
        feats = []
        for stage in self.stages:
            for blk in stage:
                x = blk(x)
            feats.append(x)
        top = bias_act_(self.top.conv(feats[3]), self.top.b)
        return feats[1], feats[2], feats[3], top


# ------------------------------------------------------------------------------------------------ training
class _BiasActFn(torch.autograd.Function):
    """y = relu(x + bias[c] (+ residual)) written over x (a fresh convolution output); bias is a frozen-BatchNorm
    constant.  Backward is one threshold pass; the saved tensor is the output the next convolution keeps anyway."""

    @staticmethod
    def forward(ctx, x, bias, residual):
        if x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last):
            # channels-last trunk: same pass over the (pixels, C) rows
            res = None
            if residual is not None:
                res = _rows(residual if residual.is_contiguous(memory_format=torch.channels_last)
                            else residual.contiguous(memory_format=torch.channels_last))
            bias_act_nhwc_(_rows(x), bias, relu=True, residual2d=res)
        else:
            bias_act_(x, bias, residual, relu=True)
        ctx.mark_dirty(x)
        ctx.save_for_backward(x)
        ctx.has_res = residual is not None
        return x

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        if not g.is_contiguous() and not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous()
        gi = torch.ops.aten.threshold_backward(g, y, 0)
        return gi, None, (gi if ctx.has_res else None)


def _dense_rows(t):
    """(rows, cols) of a tensor whose MEMORY is `size(0)` dense rows (contiguous or channels-last), else None."""
    if t.dim() >= 1 and (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))):
        return t.size(0), t.numel() // max(t.size(0), 1)
    return None


def scale_rows_batch(tensors, scales):
    """[t_i * s_i] for per-row scales s_i (size(0) entries, any shape) in ONE launch per 48 tensors (dtt_scale_rows_batch);
    results keep each tensor's memory format.  Falls back to a PyTorch multiply per tensor off the GPU / for layouts that
    are not dense rows."""
    import ctypes
    if not tensors:
        return []
    ok = all(t.is_cuda and t.dtype == torch.float32 and s.is_cuda and s.dtype == torch.float32 and s.is_contiguous() and
             s.numel() == t.size(0) and _dense_rows(t) is not None and t.numel() > 0 for t, s in zip(tensors, scales))
    if not ok or os.environ.get("DTT_FOLD_BATCH", "1") == "0":   # (env: developer A/B switch)
        return [t * s for t, s in zip(tensors, scales)]
    n = len(tensors)
    outs = [torch.empty_like(t) for t in tensors]   # (preserves contiguous / channels-last strides)
    if any(o.stride() != t.stride() for o, t in zip(outs, tensors)):
        return [t * s for t, s in zip(tensors, scales)]
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    shapes = [_dense_rows(t) for t in tensors]
    dev = tensors[0].device
    with torch.cuda.device(dev):
        check(_lib.lib().dtt_scale_rows_batch(n, P(*[t.data_ptr() for t in tensors]), P(*[s.data_ptr() for s in scales]),
                                              P(*[o.data_ptr() for o in outs]), I(*[r for r, _ in shapes]),
                                              I(*[c for _, c in shapes]), stream_ptr(dev)), "scale_rows_batch")
    return outs


class _FoldScalesFn(torch.autograd.Function):
    """(w_1 .. w_n, s_1 .. s_n) -> (w_1 * s_1, .. , w_n * s_n) with one launch per 48 filters in both directions
    (dtt_scale_rows_batch; torch._foreach_mul takes its per-tensor path for broadcast operands: ~100 launch-bound multiplies
    per direction, 1.6 ms of a 44 ms step); the scales are frozen-BatchNorm constants (no gradient)."""

    @staticmethod
    def forward(ctx, *tensors):
        n = len(tensors) // 2
        ctx.scales = tensors[n:]
        return tuple(scale_rows_batch([t.detach() for t in tensors[:n]], list(tensors[n:])))

    @staticmethod
    def backward(ctx, *grads):
        n = len(grads)
        idx = [i for i, g in enumerate(grads) if g is not None]
        out = [None] * n
        for i, o in zip(idx, scale_rows_batch([grads[i] for i in idx], [ctx.scales[i] for i in idx])):
            out[i] = o
        return tuple(out) + (None,) * n


class FusedTrainTrunk:
    """Training-time trunk with the frozen BatchNorm folded out of the activation path.

    The reference trains with every BatchNorm frozen (eval mode, no gradients: resnet.py:290-295, 325-343), so
    `bn(conv(x; w))` is `conv(x; w * s) + t` with constants s, t.  The trainable weights stay the parameters w
    (autograd sees the product w * s, checkpoints are unchanged); what disappears are the BatchNorm forward pass,
    its backward scale pass, the separate ReLU and the residual add over every activation tensor: one in-place HIP
    pass forward, one threshold pass backward.  Blocks without trainable parameters (conv1 + the FIXED_BLOCKS
    stages) run under no_grad on cached folded weights.
    """

    def __init__(self, model, channels_last=False):
        b = model.RFCN_base
        self.model = model
        self.channels_last = channels_last
        self.pool = b[3]
        self.stem = _FusedConv(b[0], b[1])
        self.frozen, self.live = [], []
        for idx in (4, 5, 6, 7):
            blocks = list(b[idx])
            if any(p.requires_grad for p in b[idx].parameters()):
                self.live.append(blocks)
            else:
                assert not self.live, "a frozen stage after a trainable one is not supported"
                self.frozen.append([_FusedBottleneck(blk) for blk in blocks])
        self.n_frozen = len(self.frozen)
        # constants of the trainable blocks: per conv the scale s (C,1,1,1) and shift t
        self.convs, self.scales, self.shifts = [], [], []
        for blocks in self.live:
            for blk in blocks:
                pairs = [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)]
                if blk.downsample is not None:
                    pairs.append((blk.downsample[0], blk.downsample[1]))
                for conv, bn in pairs:
                    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
                    self.convs.append(conv)
                    self.scales.append(sc.view(-1, 1, 1, 1).contiguous())
                    self.shifts.append((bn.bias - bn.running_mean * sc).detach().contiguous())
        if channels_last:
            # keep the trainable filters themselves in channels-last memory (values, names and checkpoints are
            # unchanged): no per-step layout copy of every weight and of every weight gradient
            for conv in self.convs:
                w = conv.weight
                if not w.is_contiguous(memory_format=torch.channels_last):
                    w.data = w.data.contiguous(memory_format=torch.channels_last)
                    if w.grad is not None:
                        w.grad = None

    @staticmethod
    def run_block(blk, x, w, t):
        """One bottleneck on folded weights w = [w1*s1, w2*s2, w3*s3 (, wd*sd)] and shifts t."""
        kw = lambda c: dict(stride=c.stride, padding=c.padding, dilation=c.dilation)
        out = _BiasActFn.apply(F.conv2d(x, w[0], None, **kw(blk.conv1)), t[0], None)
        out = _BiasActFn.apply(F.conv2d(out, w[1], None, **kw(blk.conv2)), t[1], None)
        out = F.conv2d(out, w[2], None, **kw(blk.conv3))
        if blk.downsample is not None:
            res = F.conv2d(x, w[3], None, **kw(blk.downsample[0]))
            return _BiasActFn.apply(out, t[2] + t[3], res)
        return _BiasActFn.apply(out, t[2], x)

    def __call__(self, x):
        with torch.no_grad():
            x = self.pool(bias_act_(self.stem.conv(x), self.stem.b))
            feats = []
            for stage in self.frozen:
                for blk in stage:
                    x = blk(x)
                feats.append(x)
        # all folded weights in a few launches, forward and backward (the parameters already have the trunk's layout)
        ws = _FoldScalesFn.apply(*[c.weight for c in self.convs], *self.scales)
        if self.channels_last:
            # MIOpen's fp32 backward kernels are NHWC implicit GEMMs; feeding them NCHW costs a transpose on each side
            # (the frozen stages' output crosses over once, through the tiled transpose: the strided copy behind
            #  .contiguous(memory_format=...) moves the 164 MB of a 600 px batch at 1.4 TB/s)
            x = nchw_to_nhwc(x) if x.is_cuda and x.is_contiguous() else x.contiguous(memory_format=torch.channels_last)
        k = 0
        for blocks in self.live:
            for blk in blocks:
                n = 4 if blk.downsample is not None else 3
                x = self.run_block(blk, x, ws[k:k + n], self.shifts[k:k + n])
                k += n
            feats.append(x)
        top = F.relu(self.model.RFCN_net(feats[3]), inplace=True)
        if self.channels_last:
            # (the hand-written heads of the training graph read the channels-last `top` rows directly: dtt/model.py, `_train_pm`)
            keep_top = getattr(self.model, "_train_pm", False)
            if os.environ.get("DTT_TRAIN_CORR_CL", "1") != "0":   # (env: developer A/B switch)
                # the correlations read channels-last maps and hand channels-last gradients back (dtt.ops.Correlation ->
                # CorrelationNHWCFunction): no NHWC -> NCHW copies of conv3 / conv4 / conv5, none of their gradients
                return feats[1], feats[2], feats[3], (top if keep_top else _to_nchw(top))
            return _to_nchw(feats[1]), _to_nchw(feats[2]), _to_nchw(feats[3]), (top if keep_top else _to_nchw(top))
        return feats[1], feats[2], feats[3], top


def fuse_for_training(model, channels_last=False):
    """Build the training-time fused trunk from the model's current (frozen) BatchNorm statistics; call again after
    loading a checkpoint."""
    model._fused_train_trunk = FusedTrainTrunk(model, channels_last=channels_last)
    # RFCN_cls_net + RFCN_bbox_net as ONE hand-written MFMA GEMM with a hand-written backward, position-major PSRoI pooling
    # forward / backward (dtt.heads.HeadGemmFn / PsroiPmFn): needs the channels-last `top`, class-agnostic boxes and at most
    # 32 classes (the pooling kernel's lane layout).  DTT_TRAIN_PM=0: the library convolutions + NCHW pooling kernels.
    # An odd anchor count (the 9-anchor default of the non-imagenet datasets) keeps the library graph: the packed RPN heads pair anchors.
    model._train_pm = bool(channels_last and os.environ.get("DTT_TRAIN_PM", "1") != "0" and getattr(model, "class_agnostic", False)
                           and getattr(model, "n_classes", 99) <= 32 and model.RFCN_rpn.RPN_cls_score.weight.shape[0] % 4 == 0)
    return model


# ------------------------------------------------------------------------------------------------ channels-last inference
def maxpool3s2_bias_relu_nhwc(x, bias):
    """relu(MaxPool2d(3, 2, padding 0, ceil_mode=True)(x) + bias) for a channels-last x, one kernel."""
    assert x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    n, c, h, w = x.shape
    oh, ow = (h - 3 + 1) // 2 + 1, (w - 3 + 1) // 2 + 1
    y = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        check(_lib.lib().dtt_maxpool3s2_bias_relu_nhwc(ptr(x), ptr(bias), ptr(y), n, h, w, c, stream_ptr(x.device)),
              "maxpool3s2_bias_relu")
    return y


def bias_act_nhwc_(y2d, bias, relu=True, residual2d=None):
    """In place on a (rows, C) row-major view: y = act(y + bias[None, :] (+ residual))."""
    L = _lib.lib()
    with torch.cuda.device(y2d.device):
        check(L.dtt_bias_act_nhwc_inplace(ptr(y2d), ptr(bias), ptr(residual2d) if residual2d is not None else None,
                                          y2d.shape[0], y2d.shape[1], int(relu), stream_ptr(y2d.device)), "bias_act_nhwc")
    return y2d


_GEMM_WS = {}
_GEMM_TUNED = set()


def gemm_bias_act_(out2d, a2d, wt, bias, residual2d=None, relu=True):
    """out2d (rows, n) = act(a2d (rows, k) @ wt (k, n) + bias (+ residual2d)); residual2d may be out2d itself.
    One hipBLASLt launch (dtt_gemm_bias_act): bias, residual add and ReLU live in the GEMM epilogue."""
    assert out2d.is_contiguous() and a2d.is_contiguous() and wt.is_contiguous() and out2d.dtype == torch.float32
    dev = out2d.device
    ws = _GEMM_WS.get(dev)
    if ws is None:
        ws = _GEMM_WS[dev] = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
    L = _lib.lib()
    key = (dev, a2d.shape[0], a2d.shape[1], wt.shape[1], bool(relu), residual2d is not None)
    if key not in _GEMM_TUNED:
        # first product of this shape: let the library time its candidates on these operands (explicit, synchronising
        # entry point; the hot one below never allocates or waits)
        _GEMM_TUNED.add(key)
        scratch = torch.empty((a2d.shape[0], wt.shape[1]), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.dtt_gemm_tune(ptr(a2d), ptr(wt), ptr(bias), ptr(residual2d) if residual2d is not None else None,
                                  a2d.shape[0], a2d.shape[1], wt.shape[1], int(relu), ptr(scratch), ptr(ws), ws.numel(),
                                  stream_ptr(dev)), "gemm_tune")
    with torch.cuda.device(dev):
        check(L.dtt_gemm_bias_act(ptr(out2d), ptr(a2d), ptr(wt), ptr(bias), ptr(residual2d) if residual2d is not None else None,
                                  a2d.shape[0], a2d.shape[1], wt.shape[1], int(relu), ptr(ws), ws.numel(),
                                  stream_ptr(dev)), "gemm_bias_act")
    return out2d


def _transpose_batched(src, dst, batch, rows, cols):
    with torch.cuda.device(src.device):
        check(_lib.lib().dtt_transpose_batched(ptr(src), ptr(dst), batch, rows, cols, stream_ptr(src.device)),
              "transpose_batched")
    return dst


def nhwc_to_nchw(x):
    """Channels-last (N,C,H,W)-shaped tensor -> NCHW-contiguous copy, by the tiled HIP transpose (torch's strided
    copy kernel moves these maps at 1.4 TB/s)."""
    n, c, h, w = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    return _transpose_batched(x, torch.empty((n, c, h, w), dtype=x.dtype, device=x.device), n, h * w, c)


def nchw_to_nhwc(x):
    """NCHW-contiguous tensor -> channels-last tensor of the same logical shape."""
    n, c, h, w = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty((n, c, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    return _transpose_batched(x, out, n, c, h * w)


class _ToNCHWFn(torch.autograd.Function):
    """nhwc_to_nchw with the reverse transpose as its backward (training trunk -> D&T operators)."""

    @staticmethod
    def forward(ctx, x):
        return nhwc_to_nchw(x)

    @staticmethod
    def backward(ctx, g):
        return nchw_to_nhwc(g.contiguous())


def _to_nchw(x):
    if x.shape[1] == 1 or x.shape[2] * x.shape[3] == 1 or not x.is_contiguous(memory_format=torch.channels_last) \
            or x.is_contiguous():
        return x.contiguous()
    return _ToNCHWFn.apply(x) if x.requires_grad else nhwc_to_nchw(x)


def _rows(x):
    """(N,C,H,W) channels-last tensor -> its (N*H*W, C) row-major view."""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])


def _from_rows(y2d, n, h, w):
    return y2d.view(n, h, w, y2d.shape[1]).permute(0, 3, 1, 2)


_WINO_G = {2: torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64),
           4: torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                            [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)}
_WINO_BUF = {}


def winograd_weights(w, m=2):
    """(K, C, 3, 3) filters -> U ((m+2)^2, C, K) = G g G^T per (k, c), laid out for the row-major batched GEMMs
    (computed in float64, stored fp32).  m = 2: F(2x2, 3x3); m = 4: F(4x4, 3x3), points 0, +-1, +-2, infinity."""
    G = _WINO_G[m].to(device=w.device)
    u = torch.einsum("ia,kcab,jb->ijck", G, w.double(), G)
    return u.reshape((m + 2) ** 2, w.shape[1], w.shape[0]).float().contiguous()


def winograd_conv3x3_nhwc(x, u, bias, dilation=1, relu=True, m=2):
    """3x3, stride 1, padding == dilation convolution of a channels-last map in Winograd form: input transform (HIP) ->
    (m+2)^2 batched GEMMs (hipBLASLt, dtt_gemm_batched) -> output transform + bias (+ ReLU) (HIP)."""
    assert x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    n, c, h, w = x.shape
    k = u.shape[2]
    t2 = (m + 2) ** 2
    assert u.shape[0] == t2 and u.shape[1] == c
    L = _lib.lib()
    dev = x.device
    tiles = L.dtt_winograd_tiles(n, h, w, dilation, m)
    key = (dev, t2 * tiles, c, k)
    bufs = _WINO_BUF.get(key)
    if bufs is None:
        bufs = _WINO_BUF[key] = (torch.empty(t2 * tiles * c, device=dev), torch.empty(t2 * tiles * k, device=dev))
    v, mm = bufs
    ws = _GEMM_WS.get(dev)
    if ws is None:
        ws = _GEMM_WS[dev] = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
    y = torch.empty((n, k, h, w), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
    st = stream_ptr(dev)
    with torch.cuda.device(dev):
        check(L.dtt_winograd_input_transform(ptr(x), ptr(v), n, h, w, c, dilation, m, st), "winograd input transform")
        bkey = (dev, "batched", t2, tiles, c, k)
        if bkey not in _GEMM_TUNED:
            _GEMM_TUNED.add(bkey)
            check(L.dtt_gemm_batched_tune(ptr(mm), ptr(v), ptr(u), t2, tiles, c, k, ptr(ws), ws.numel(), st), "gemm_batched_tune")
        check(L.dtt_gemm_batched(ptr(mm), ptr(v), ptr(u), t2, tiles, c, k, ptr(ws), ws.numel(), st), "gemm_batched")
        check(L.dtt_winograd_output_transform(ptr(mm), ptr(bias), ptr(y), n, h, w, k, dilation, m, int(relu), st),
              "winograd output transform")
    return y


def _time_us(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


class _NhwcConv:
    """One folded convolution for the channels-last trunk: 1x1 stride-1 layers are plain GEMMs over the (pixels, C)
    view (hipBLASLt, bias + ReLU in the GEMM epilogue); 3x3 stride-1 layers run in Winograd form (HIP transforms around
    batched GEMMs) when that beats MIOpen on the layer's own input (timed once per map size); everything else is
    MIOpen's NHWC kernel + the fused bias / ReLU pass."""

    def __init__(self, conv, bn=None, extra_bias=None):
        base = _FusedConv(conv, bn)
        self.b = base.b if extra_bias is None else (base.b + extra_bias).contiguous()
        self.kw = base.kw
        self.is_gemm = conv.kernel_size == (1, 1) and conv.stride == (1, 1)
        self.u = None          # Winograd-domain filters {m: U}, for 3x3 / stride 1 / padding == dilation layers
        self.pick = {}         # per map size (C, H, W): 0 = MIOpen, 2 / 4 = the Winograd variant that won the timing
        if self.is_gemm:
            self.wt = base.w.view(base.w.shape[0], base.w.shape[1]).t().contiguous()   # (Cin, Cout)
            self.zero_b = torch.zeros_like(self.b)
        else:
            self.w = base.w.contiguous(memory_format=torch.channels_last)
            if (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.groups == 1 and
                    conv.padding == conv.dilation and conv.dilation[0] == conv.dilation[1] and
                    base.w.shape[0] % 4 == 0 and base.w.shape[1] % 4 == 0 and os.environ.get("DTT_WINOGRAD", "1") != "0"):
                variants = (2, 4) if os.environ.get("DTT_WINOGRAD_F4", "1") != "0" else (2,)
                self.u = {m: winograd_weights(base.w, m) for m in variants}
                self.dil = conv.dilation[0]

    def raw(self, x):
        """Convolution without bias; x and the result are channels-last."""
        if self.is_gemm:
            n, _, h, w = x.shape
            a = _rows(x)
            out = torch.empty((a.shape[0], self.wt.shape[1]), dtype=a.dtype, device=a.device)
            return _from_rows(gemm_bias_act_(out, a, self.wt, self.zero_b, relu=False), n, h, w)
        return F.conv2d(x, self.w, None, **self.kw)

    def act(self, x):
        """relu(conv(x) + b)."""
        if self.is_gemm:
            n, _, h, w = x.shape
            a = _rows(x)
            out = torch.empty((a.shape[0], self.wt.shape[1]), dtype=a.dtype, device=a.device)
            return _from_rows(gemm_bias_act_(out, a, self.wt, self.b), n, h, w)
        if self.u is not None:
            key = tuple(x.shape[1:])   # per map size, whatever the batch: the same images give the same arithmetic
            pick = self.pick.get(key)
            if pick is None:   # first time at this map size: the fastest path wins (all timed on this input)
                times = {0: _time_us(lambda: bias_act_nhwc_(_rows(F.conv2d(x, self.w, None, **self.kw)), self.b))}
                for m, u in self.u.items():
                    t2 = (m + 2) ** 2
                    tiles = _lib.lib().dtt_winograd_tiles(x.shape[0], x.shape[2], x.shape[3], self.dil, m)
                    if 4 * t2 * tiles * max(x.shape[1], u.shape[2]) > (1 << 30):
                        continue   # transform buffers out of proportion
                    times[m] = _time_us(lambda: winograd_conv3x3_nhwc(x, u, self.b, self.dil, True, m))
                pick = self.pick[key] = min(times, key=times.get)
                if os.environ.get("DTT_WINOGRAD_VERBOSE"):
                    print("[dtt] conv3x3 %s dil %d: %s -> %s" % (key, self.dil, ", ".join(
                        "%s %.1f us" % ("direct" if m == 0 else "F(%d,3)" % m, t) for m, t in sorted(times.items())),
                        "direct" if pick == 0 else "F(%d,3)" % pick))
            if pick:
                return winograd_conv3x3_nhwc(x, self.u[pick], self.b, self.dil, True, pick)
        y = F.conv2d(x, self.w, None, **self.kw)
        bias_act_nhwc_(_rows(y), self.b)
        return y


class _NhwcBottleneck:
    def __init__(self, blk):
        self.c1, self.c2 = _NhwcConv(blk.conv1, blk.bn1), _NhwcConv(blk.conv2, blk.bn2)
        self.down = _NhwcConv(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
        # both shifts of the block's last step land in the single final pass
        self.c3 = _NhwcConv(blk.conv3, blk.bn3, extra_bias=self.down.b if self.down is not None else None)

    def __call__(self, x):
        h = self.c2.act(self.c1.act(x))
        n, _, hh, ww = h.shape
        # relu(conv3(h) + shift + shortcut) in ONE GEMM: the shortcut rides in as the C operand and is overwritten in
        # place -- for identity shortcuts that is the block input, which nothing reads afterwards (stage outputs only
        # ever feed downsample blocks, which write into their own projection buffer)
        res = _rows(x) if self.down is None else _rows(self.down.raw(x))
        y = gemm_bias_act_(res, _rows(h), self.c3.wt, self.c3.b, residual2d=res, relu=True)
        return _from_rows(y, n, hh, ww)


class TrunkExtras:
    """What the channels-last inference trunk computes on the way for the position-major tail, handed to `_RFCN.forward` as a
    return value (not as attributes that outlive the call): `top_rows` / `top_hw` the channels-last rows of `top` and its map size,
    `det_rows` the class + box head GEMM's output when it was issued inside the trunk, `rpn_rows` the channels-last rows of
    relu(RPN_Conv(top)) for the one-launch RPN heads, or `rpn_conv1` the same map in NCHW for the library heads."""
    __slots__ = ("top_rows", "top_hw", "det_rows", "rpn_rows", "rpn_conv1")

    def __init__(self):
        self.top_rows = self.top_hw = self.det_rows = self.rpn_rows = self.rpn_conv1 = None


class FusedTrunkNHWC:
    """Inference trunk in channels-last layout.  Same folded weights as FusedTrunk; the 1x1 convolutions (two thirds
    of the layers) become GEMMs whose epilogue already applies bias + ReLU, the residual add is the GEMM's beta * C
    term, and MIOpen's NHWC 3x3 kernels need no layout transposes around them.  The four feature maps the D&T ops
    consume are returned NCHW-contiguous (one transpose each)."""

    def __init__(self, model):
        b = model.RFCN_base
        self.stem = _NhwcConv(b[0], b[1])
        self.pool = b[3]
        k = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        self.pool_is_3s2 = (isinstance(self.pool, torch.nn.MaxPool2d) and k(self.pool.kernel_size) == (3, 3) and
                            k(self.pool.stride) == (2, 2) and k(self.pool.padding) == (0, 0) and k(self.pool.dilation) == (1, 1)
                            and self.pool.ceil_mode)
        self.stages = [[_NhwcBottleneck(blk) for blk in b[i]] for i in (4, 5, 6, 7)]
        self.top = _NhwcConv(b[8])
        # the RPN's 3x3 convolution + ReLU (rpn.py:60-61) reads the same channels-last map: computed here, where it can
        # take the Winograd path, and handed to _RPN.head() through `rpn_conv1`
        self.rpn_conv = _NhwcConv(model.RFCN_rpn.RPN_Conv)
        self.pm_heads = False     # set by fuse_for_inference when the position-major tail is active
        self.pm_tail = None       # dtt.heads.PositionMajorTail: when set, the class + box head GEMM is issued in here

    @torch.no_grad()
    def __call__(self, x):
        x = x.contiguous(memory_format=torch.channels_last)
        # stem: relu(maxpool(conv(x)) + b) instead of maxpool(relu(conv(x) + b)) -- the same values bit for bit (adding one
        # bias per channel and clamping at zero are monotonic, so they commute with the window maximum; the pool pads with
        # -inf), but the bias / ReLU pass runs on the pooled map: 41 MB instead of 164 MB read and written (46 -> 12 us) --
        # and, for the reference's MaxPool2d(3, 2, 0, ceil_mode=True), inside the pooling kernel itself
        x = self.stem.raw(x)
        if self.pool_is_3s2:
            x = maxpool3s2_bias_relu_nhwc(x, self.stem.b)    # dtt_maxpool3s2_bias_relu_nhwc: pool + shift + ReLU in one pass
        else:
            x = self.pool(x)
            bias_act_nhwc_(_rows(x), self.stem.b)
        feats = []
        for stage in self.stages:
            for blk in stage:
                x = blk(x)
            feats.append(x)
        top = self.top.act(feats[3])
        ex = TrunkExtras()
        if self.pm_heads:
            # the position-major tail reads channels-last memory directly (head GEMM over the `top` rows, channels-last
            # correlation kernel over conv3 / conv4 / conv5): no layout hand-over at all
            ex.top_rows, ex.top_hw = _rows(top), (top.shape[2], top.shape[3])
            if self.pm_tail is not None and os.environ.get("DTT_DET_EARLY", "1") != "0":   # (developer A/B switch)
                # The class + box head GEMM runs HERE, ahead of the RPN: its grid is one workgroup on every CU, so it
                # cannot share the chip with the proposal layer's kernels (256 + 4 workgroups: the last 4 wait for, or
                # squeeze in beside, the others -- 150 -> 200 us when it ran after the correlations, under the NMS sweep).
                from .heads import head_gemm
                ex.det_rows = head_gemm(ex.top_rows, self.pm_tail.det)
        if self.pm_heads and self.pm_tail is not None and self.pm_tail.rpn is not None and os.environ.get("DTT_RPN_FUSED", "1") != "0":
            # channels-last rows of relu(RPN_Conv(top)) for the one-launch RPN heads (dtt.heads.rpn_head_gemm): no
            # NHWC -> NCHW hand-over at all
            ex.rpn_rows = _rows(self.rpn_conv.act(top))
        else:
            ex.rpn_conv1 = _to_nchw(self.rpn_conv.act(top))
        if self.pm_heads:
            return feats[1], feats[2], feats[3], top, ex
        return _to_nchw(feats[1]), _to_nchw(feats[2]), _to_nchw(feats[3]), _to_nchw(top), ex


def fuse_for_inference(model, channels_last=True):
    """Build the fused trunk from the model's current weights (call again after loading a checkpoint)."""
    model._fused_trunk = FusedTrunkNHWC(model) if channels_last else FusedTrunk(model)
    model._pm_tail = None
    if channels_last and os.environ.get("DTT_PM_HEADS", "1") != "0":
        # hand-written heads + position-major PSRoI pooling (dtt.heads); needs the class-agnostic box head of D&T
        # (rfcn.py:52-53, resnet.py:311: 4 * 49 box channels) and at most 32 classes
        if model.n_reg_classes == 1 and model.n_classes <= 32 and model.RFCN_cls_net.weight.shape[1] % 32 == 0:
            from .heads import PositionMajorTail
            model._pm_tail = PositionMajorTail(model)
            model._fused_trunk.pm_heads = True
            model._fused_trunk.pm_tail = model._pm_tail
    return model


def unfuse(model):
    model._fused_trunk = None
    model._fused_train_trunk = None
    model._pm_tail = None
    return model
