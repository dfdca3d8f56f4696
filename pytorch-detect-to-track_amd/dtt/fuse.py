"""Inference-time fusion of the frozen-BatchNorm ResNet trunk.

The reference keeps BatchNorm frozen everywhere (eval mode + requires_grad False, faster_rcnn/resnet.py:290-295,
325-330), so at inference every `conv -> BN -> ReLU` is `relu(conv(x; w * s) + t)` with s = gamma / sqrt(var + eps),
t = beta - mean * s folded once, and the end of a bottleneck is `relu(conv3'(.) + t3 + shortcut)`.  The convolutions
still run on MIOpen / rocBLAS; the per-channel bias, the residual add and the ReLU become ONE in-place HIP pass
(`dtt_bias_act_inplace`) instead of the 2-3 elementwise launches (7 full-tensor passes per bottleneck) of the
unfused graph.  Weights are not modified: the folded copies live next to the original modules, the state_dict /
checkpoint layout is untouched, and `unfuse()` (or `.train()`) goes back to the reference graph.
"""
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


def fuse_for_inference(model):
    """Build the fused trunk from the model's current weights (call again after loading a checkpoint)."""
    model._fused_trunk = FusedTrunk(model)
    return model


def unfuse(model):
    model._fused_trunk = None
    return model
