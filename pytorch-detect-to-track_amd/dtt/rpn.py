"""RPN-side layers of the D&T hot path on the device.

Mirrors (paths relative to the reference's lib/model/rpn/):
  generate_anchors        generate_anchors.py:45-56
  _ProposalLayer          proposal_layer.py:29-161     -> one dtt_proposal_forward call for the batch
  _AnchorTargetLayer      anchor_target_layer.py:30-191 -> assign / (host numpy RNG) / disable / finish
  bbox_transform_inv, clip_boxes, bbox_transform_batch, bbox_overlaps_batch (bbox_transform.py) as plain
  tensor code for the callers around the path (test-time decoding, target layers).
Constructor arguments, forward inputs (tuples) and output layouts are the reference's.
"""
import math

import numpy as np
import torch
from torch import nn

from . import _lib
from ._lib import check, ptr, require_gpu, stream_ptr


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """Anchor windows around the 0-based reference box (0, 0, base-1, base-1): for every ratio (outer)
    and scale (inner), same centre, width = round(sqrt(area/ratio)) * scale, height =
    round(width0 * ratio) * scale.  Rounding is half-to-even as numpy's (generate_anchors.py:83-94)."""
    ratios = np.asarray(ratios, dtype=np.float64).reshape(-1)
    scales = np.asarray(scales, dtype=np.float64).reshape(-1)
    ctr = (base_size - 1) / 2.0
    area = float(base_size * base_size)
    w0 = np.round(np.sqrt(area / ratios))
    h0 = np.round(w0 * ratios)
    ws = (w0[:, None] * scales[None, :]).reshape(-1)
    hs = (h0[:, None] * scales[None, :]).reshape(-1)
    half_w, half_h = 0.5 * (ws - 1), 0.5 * (hs - 1)
    return np.stack([ctr - half_w, ctr - half_h, ctr + half_w, ctr + half_h], axis=1)


class _ProposalLayer(nn.Module):
    """proposal_layer.py:29-161.  forward(input) with input = (rpn_cls_prob (B,2A,H,W), rpn_bbox_pred
    (B,4A,H,W), im_info (B,3), cfg_key) -> rois (B, post_nms_topN, 5)."""

    def __init__(self, feat_stride, scales, ratios, cfg=None):
        super().__init__()
        if cfg is None:
            from .config import cfg as _cfg
            cfg = _cfg
        self._cfg = cfg
        self._feat_stride = int(feat_stride)
        anchors = torch.from_numpy(generate_anchors(scales=np.array(scales), ratios=np.array(ratios))).float()
        self.register_buffer("_anchors", anchors, persistent=False)
        self._num_anchors = anchors.size(0)

    def forward(self, input):
        scores, bbox_deltas, im_info, cfg_key = input
        c = self._cfg[cfg_key]
        return proposal_forward(scores, bbox_deltas, im_info, self._anchors, self._feat_stride,
                                c.RPN_PRE_NMS_TOP_N, c.RPN_POST_NMS_TOP_N, c.RPN_NMS_THRESH)[0]

    # The same layer in two phases (dtt_proposal_select_sort / dtt_proposal_decode_nms): phase 1 needs the scores only, so a
    # caller that overlaps the proposal layer with other work starts it before the box deltas exist (dtt/model.py).
    def select(self, scores, cfg_key):
        """Phase 1 on the current stream; returns the handle `finish` takes."""
        require_gpu(scores)
        scores = scores.detach().float().contiguous()
        B, twoA, H, W = scores.shape
        A = self._num_anchors
        if twoA != 2 * A:
            raise ValueError("proposal: cls_prob %s does not match %d anchors" % (tuple(scores.shape), A))
        c = self._cfg[cfg_key]
        L = _lib.lib()
        nbytes = L.dtt_proposal_workspace_bytes(B, A, H, W, int(c.RPN_PRE_NMS_TOP_N))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=scores.device)
        with torch.cuda.device(scores.device):
            check(L.dtt_proposal_select_sort(ptr(scores), B, A, H, W, int(c.RPN_PRE_NMS_TOP_N), ptr(ws), nbytes,
                                             stream_ptr(scores.device)), "proposal select / sort")
        return ws, scores, (B, A, H, W)

    def finish(self, handle, bbox_deltas, im_info, cfg_key):
        """Phase 2 on the current stream (which must be ordered after phase 1) -> rois (B, post_nms_topN, 5)."""
        ws, scores, (B, A, H, W) = handle
        require_gpu(bbox_deltas)
        bbox_deltas = bbox_deltas.detach().float().contiguous()
        if tuple(bbox_deltas.shape) != (B, 4 * A, H, W):
            raise ValueError("proposal: bbox_pred %s does not match cls_prob %s" % (tuple(bbox_deltas.shape), tuple(scores.shape)))
        dev = bbox_deltas.device
        c = self._cfg[cfg_key]
        im_info = im_info.detach().to(dev, torch.float32).contiguous()
        anchors = self._anchors.to(dev, torch.float32).contiguous()
        rois = torch.empty((B, int(c.RPN_POST_NMS_TOP_N), 5), dtype=torch.float32, device=dev)
        num = torch.empty((B,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            check(_lib.lib().dtt_proposal_decode_nms(ptr(bbox_deltas), ptr(im_info), ptr(anchors), B, A, H, W, self._feat_stride,
                                                     int(c.RPN_PRE_NMS_TOP_N), int(c.RPN_POST_NMS_TOP_N), float(c.RPN_NMS_THRESH),
                                                     ptr(rois), ptr(num), ptr(ws), ws.numel(), stream_ptr(dev)),
                  "proposal decode / nms")
        return rois


def proposal_forward(cls_prob, bbox_pred, im_info, anchors, feat_stride, pre_nms_topN, post_nms_topN, nms_thresh):
    """Functional form: returns (rois (B, post, 5), num_valid int32 (B,))."""
    require_gpu(cls_prob, bbox_pred)
    cls_prob = cls_prob.detach().float().contiguous()
    bbox_pred = bbox_pred.detach().float().contiguous()
    dev = cls_prob.device
    im_info = im_info.detach().to(dev, torch.float32).contiguous()
    anchors = anchors.to(dev, torch.float32).contiguous()
    B, twoA, H, W = cls_prob.shape
    A = anchors.size(0)
    if twoA != 2 * A or tuple(bbox_pred.shape) != (B, 4 * A, H, W):
        raise ValueError("proposal: cls_prob %s / bbox_pred %s do not match %d anchors" %
                         (tuple(cls_prob.shape), tuple(bbox_pred.shape), A))
    L = _lib.lib()
    rois = torch.empty((B, int(post_nms_topN), 5), dtype=torch.float32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    nbytes = L.dtt_proposal_workspace_bytes(B, A, H, W, int(pre_nms_topN))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(L.dtt_proposal_forward(ptr(cls_prob), ptr(bbox_pred), ptr(im_info), ptr(anchors), B, A, H, W,
                                     int(feat_stride), int(pre_nms_topN), int(post_nms_topN), float(nms_thresh),
                                     ptr(rois), ptr(num), ptr(ws), nbytes, stream_ptr(dev)), "proposal forward")
    return rois, num


def subsample_disable_lists(labels_np, counts, rpn_batchsize, fg_fraction, rng=np.random):
    """Host half of the anchor-target layer (anchor_target_layer.py:118-141): given the pre-subsampling
    labels and per-image [fg, bg] counts, draw numpy permutations in the reference's order and return
    (disable indices per image, fg/bg counts after subsampling)."""
    num_fg = int(fg_fraction * rpn_batchsize)
    disable, after = [], []
    for i in range(labels_np.shape[0]):
        sum_fg, sum_bg = int(counts[i][0]), int(counts[i][1])
        dis = []
        fg_left = sum_fg
        if sum_fg > num_fg:
            fg_inds = np.nonzero(labels_np[i] == 1)[0]
            perm = rng.permutation(fg_inds.size)
            dis.append(fg_inds[perm[: fg_inds.size - num_fg]])
            fg_left = num_fg
        num_bg = rpn_batchsize - sum_fg  # pre-subsampling fg count, as in the reference
        bg_left = sum_bg
        if sum_bg > num_bg:
            bg_inds = np.nonzero(labels_np[i] == 0)[0]
            perm = rng.permutation(bg_inds.size)
            dis.append(bg_inds[perm[: bg_inds.size - num_bg]])
            bg_left = max(num_bg, 0)  # a negative quota (fg alone overflows the batch) disables every bg
        disable.append(np.concatenate(dis).astype(np.int32) if dis else np.zeros((0,), np.int32))
        after.append((fg_left, bg_left))
    return disable, after


class _AnchorTargetLayer(nn.Module):
    """anchor_target_layer.py:30-191.  forward(input) with input = (rpn_cls_score (B,2A,H,W) [shape only],
    gt_boxes (B,G,5), im_info (B,3), num_boxes) -> [labels (B,1,A*H,W), bbox_targets, bbox_inside_weights,
    bbox_outside_weights (B,4A,H,W)].

    Random subsampling consumes numpy's global RNG exactly as the reference does, which costs one small
    device->host copy (labels + 2B counts) per call; everything else stays on the device."""

    def __init__(self, feat_stride, scales, ratios, cfg=None):
        super().__init__()
        if cfg is None:
            from .config import cfg as _cfg
            cfg = _cfg
        self._cfg = cfg
        self._feat_stride = int(feat_stride)
        anchors = torch.from_numpy(generate_anchors(scales=np.array(scales), ratios=np.array(ratios))).float()
        self.register_buffer("_anchors", anchors, persistent=False)
        self._num_anchors = anchors.size(0)
        self._allowed_border = 0

    def forward(self, input, out=None):
        rpn_cls_score, gt_boxes, im_info, _num_boxes = input
        T = self._cfg.TRAIN
        if T.RPN_POSITIVE_WEIGHT >= 0 and not (0 < T.RPN_POSITIVE_WEIGHT < 1):
            raise AssertionError("RPN_POSITIVE_WEIGHT must be negative (uniform weighting) or in (0, 1)")   # :149-150
        return anchor_target_forward(gt_boxes, im_info, self._anchors, rpn_cls_score.size(2), rpn_cls_score.size(3),
                                     self._feat_stride, T.RPN_BATCHSIZE, T.RPN_FG_FRACTION, T.RPN_NEGATIVE_OVERLAP,
                                     T.RPN_POSITIVE_OVERLAP, T.RPN_CLOBBER_POSITIVES,
                                     T.RPN_BBOX_INSIDE_WEIGHTS[0], positive_weight=T.RPN_POSITIVE_WEIGHT,
                                     mode=getattr(T, "SAMPLER_RNG", "device"), out=out)


def anchor_target_forward(gt_boxes, im_info, anchors, height, width, feat_stride, rpn_batchsize=256,
                          fg_fraction=0.5, negative_overlap=0.3, positive_overlap=0.7, clobber_positives=False,
                          inside_weight=1.0, rng=np.random, positive_weight=-1.0, mode="reference", keys=None, out=None):
    """positive_weight < 0: every sampled anchor weighs 1 / num_examples (anchor_target_layer.py:143-147).  In (0, 1): the
    branch the reference asserts on but never finishes (:148-150 leave positive_weights / negative_weights undefined, a
    NameError at :152) is completed the way the py-faster-rcnn layer it was ported from defines it: positives share
    `positive_weight`, negatives 1 - positive_weight, i.e. p / num_positives and (1 - p) / num_negatives -- counted, like
    num_examples, on the LAST image of the batch (:144 `labels[i]`).
    out: the four result tensors to write into (contiguous float32, e.g. one leg's slices of the buffers `RpnLossFn` reads)."""
    require_gpu(gt_boxes)
    dev = gt_boxes.device
    gt = gt_boxes.detach()[:, :, :5].float().contiguous()
    anchors = anchors.to(dev, torch.float32).contiguous()
    B, G, _ = gt.shape
    A = anchors.size(0)
    n = A * height * width
    if mode == "device":
        return _anchor_target_device(gt, im_info, anchors, B, G, A, height, width, feat_stride, rpn_batchsize, fg_fraction,
                                     negative_overlap, positive_overlap, clobber_positives, inside_weight, positive_weight, rng, keys, out)
    info0 = im_info[0].detach().cpu()
    im_h0, im_w0 = int(info0[0]), int(info0[1])  # long(im_info[0][0]) (anchor_target_layer.py:85-86)
    L = _lib.lib()
    labels = torch.empty((B, n), dtype=torch.int32, device=dev)
    argmax = torch.empty((B, n), dtype=torch.int32, device=dev)
    counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
    scratch = torch.empty((B * G,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(L.dtt_anchor_target_assign(ptr(gt), im_h0, im_w0, ptr(anchors), B, G, A, height, width,
                                         int(feat_stride), float(negative_overlap), float(positive_overlap),
                                         int(bool(clobber_positives)), ptr(labels), ptr(argmax), ptr(counts),
                                         ptr(scratch), stream_ptr(dev)), "anchor_target assign")
        counts_h = counts.cpu().numpy()
        num_fg = int(fg_fraction * rpn_batchsize)
        need_labels = any(c[0] > num_fg or c[1] > rpn_batchsize - c[0] for c in counts_h)
        if need_labels:
            disable, after = subsample_disable_lists(labels.cpu().numpy(), counts_h, rpn_batchsize, fg_fraction, rng)
            offs = np.zeros((B + 1,), dtype=np.int32)
            offs[1:] = np.cumsum([d.size for d in disable])
            if offs[-1] > 0:
                dis_t = torch.from_numpy(np.concatenate(disable)).to(dev)
                off_t = torch.from_numpy(offs).to(dev)
                check(L.dtt_anchor_target_disable(ptr(labels), ptr(dis_t), ptr(off_t), B, n, stream_ptr(dev)),
                      "anchor_target disable")
        else:
            after = [(int(c[0]), int(c[1])) for c in counts_h]
        num_examples = after[B - 1][0] + after[B - 1][1]  # LAST image only (anchor_target_layer.py:154)
        w = float(np.float32(1.0) / np.float32(num_examples)) if num_examples > 0 else math.inf
        w_pos = w_neg = w
        if positive_weight >= 0:
            n_pos, n_neg = after[B - 1]
            w_pos = float(np.float32(positive_weight) / np.float32(n_pos)) if n_pos > 0 else math.inf
            w_neg = float(np.float32(1.0 - positive_weight) / np.float32(n_neg)) if n_neg > 0 else math.inf
        labels_out, targets, inside, outside = _anchor_target_outputs(out, B, A, height, width, dev)
        check(L.dtt_anchor_target_finish(ptr(gt), im_h0, im_w0, ptr(anchors), ptr(labels), ptr(argmax), B, G, A,
                                         height, width, int(feat_stride), float(inside_weight), w_pos, w_neg,
                                         ptr(labels_out), ptr(targets), ptr(inside), ptr(outside), stream_ptr(dev)),
              "anchor_target finish")
    return [labels_out, targets, inside, outside]


def _anchor_target_outputs(out, B, A, height, width, dev):
    """The layer's four outputs: fresh tensors, or the caller's (checked)."""
    shapes = [(B, 1, A * height, width)] + [(B, 4 * A, height, width)] * 3
    if out is None:
        return [torch.empty(s, dtype=torch.float32, device=dev) for s in shapes]
    out = list(out)
    for t, s in zip(out, shapes):
        if tuple(t.shape) != s or t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
            raise ValueError("anchor_target: out tensors must be contiguous float32 %s on %s" % (shapes, dev))
    return out


def _anchor_target_device(gt, im_info, anchors, B, G, A, height, width, feat_stride, rpn_batchsize, fg_fraction, negative_overlap,
                          positive_overlap, clobber_positives, inside_weight, positive_weight, rng, keys, out=None):
    """mode == "device" (cfg.TRAIN.SAMPLER_RNG, the RoI sampler's counterpart): `dtt_anchor_target_device` -- no host read, no
    upload that waits for the stream.  The reference's numpy permutations (anchor_target_layer.py:124-141) become one 32-bit
    random key per anchor, drawn without looking at the labels: of a class over its quota the candidates with the smallest
    (key, anchor index) stay.  The keys come from the DEVICE generator, seeded per call by one integer drawn from `rng` (numpy's
    global generator by default): a seeded run is reproducible as before, but it is not the reference's sample."""
    dev = gt.device
    n = A * height * width
    if keys is None:
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(rng.randint(0, 2 ** 31 - 1)))
        keys = torch.randint(0, 2 ** 31 - 1, (B, n), dtype=torch.int32, device=dev, generator=gen)
    if keys.dtype != torch.int32 or tuple(keys.shape) != (B, n) or keys.device != dev or not keys.is_contiguous():
        raise ValueError("anchor_target: keys must be a contiguous int32 tensor of shape (%d, %d) on %s" % (B, n, dev))
    im_info = im_info.detach().to(dev, torch.float32).contiguous()
    i32 = dict(dtype=torch.int32, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    labels, argmax = torch.empty((B, n), **i32), torch.empty((B, n), **i32)
    counts, scratch, weights = torch.empty((B, 4), **i32), torch.empty((B * G,), **i32), torch.empty((2,), **f32)
    labels_out, targets, inside, outside = _anchor_target_outputs(out, B, A, height, width, dev)
    with torch.cuda.device(dev):
        check(_lib.lib().dtt_anchor_target_device(ptr(gt), ptr(im_info), ptr(anchors), ptr(keys), B, G, A, height, width,
                                                  int(feat_stride), int(rpn_batchsize), int(fg_fraction * rpn_batchsize),
                                                  float(negative_overlap), float(positive_overlap), int(bool(clobber_positives)),
                                                  float(inside_weight), float(positive_weight), ptr(labels), ptr(argmax), ptr(counts),
                                                  ptr(scratch), ptr(weights), ptr(labels_out), ptr(targets), ptr(inside), ptr(outside),
                                                  stream_ptr(dev)), "anchor_target (device)")
    return [labels_out, targets, inside, outside]


# ------------------------------------------------------------------- box algebra for the callers
def bbox_transform_inv(boxes, deltas, batch_size=None):
    """bbox_transform.py:108-134 on (B, N, 4) boxes and (B, N, 4k) deltas."""
    widths = boxes[:, :, 2] - boxes[:, :, 0] + 1.0
    heights = boxes[:, :, 3] - boxes[:, :, 1] + 1.0
    ctr_x = boxes[:, :, 0] + 0.5 * widths
    ctr_y = boxes[:, :, 1] + 0.5 * heights
    dx, dy, dw, dh = deltas[:, :, 0::4], deltas[:, :, 1::4], deltas[:, :, 2::4], deltas[:, :, 3::4]
    pcx = dx * widths.unsqueeze(2) + ctr_x.unsqueeze(2)
    pcy = dy * heights.unsqueeze(2) + ctr_y.unsqueeze(2)
    pw = torch.exp(dw) * widths.unsqueeze(2)
    ph = torch.exp(dh) * heights.unsqueeze(2)
    out = torch.empty_like(deltas)
    out[:, :, 0::4] = pcx - 0.5 * pw
    out[:, :, 1::4] = pcy - 0.5 * ph
    out[:, :, 2::4] = pcx + 0.5 * pw
    out[:, :, 3::4] = pcy + 0.5 * ph
    return out


def clip_boxes(boxes, im_shape, batch_size=None):
    """bbox_transform.py:156-173 (3-D branch), vectorised over the batch."""
    wmax = (im_shape[:, 1] - 1).view(-1, 1, 1)
    hmax = (im_shape[:, 0] - 1).view(-1, 1, 1)
    zero = torch.zeros_like(wmax)
    boxes[:, :, 0::4] = torch.min(torch.max(boxes[:, :, 0::4], zero), wmax)
    boxes[:, :, 1::4] = torch.min(torch.max(boxes[:, :, 1::4], zero), hmax)
    boxes[:, :, 2::4] = torch.min(torch.max(boxes[:, :, 2::4], zero), wmax)
    boxes[:, :, 3::4] = torch.min(torch.max(boxes[:, :, 3::4], zero), hmax)
    return boxes


def bbox_overlaps_batch(anchors, gt_boxes):
    """bbox_transform.py:256-296 (3-D anchors branch): anchors (B,N,4|5), gt (B,K,>=4) -> (B,N,K)."""
    if anchors.size(2) == 5:
        anchors = anchors[:, :, 1:5]
    gt = gt_boxes[:, :, :4]
    gx = gt[:, :, 2] - gt[:, :, 0] + 1
    gy = gt[:, :, 3] - gt[:, :, 1] + 1
    ax = anchors[:, :, 2] - anchors[:, :, 0] + 1
    ay = anchors[:, :, 3] - anchors[:, :, 1] + 1
    g_area = (gx * gy).unsqueeze(1)
    a_area = (ax * ay).unsqueeze(2)
    b = anchors.unsqueeze(2)
    q = gt.unsqueeze(1)
    iw = (torch.min(b[..., 2], q[..., 2]) - torch.max(b[..., 0], q[..., 0]) + 1).clamp(min=0)
    ih = (torch.min(b[..., 3], q[..., 3]) - torch.max(b[..., 1], q[..., 1]) + 1).clamp(min=0)
    ov = iw * ih / (a_area + g_area - iw * ih)
    ov = ov.masked_fill(((gx == 1) & (gy == 1)).unsqueeze(1), 0)
    ov = ov.masked_fill(((ax == 1) & (ay == 1)).unsqueeze(2), -1)
    return ov


def bbox_transform_batch(ex_rois, gt_rois):
    """bbox_transform.py:54-70 (3-D branch): (B,N,4), (B,N,4) -> (B,N,4)."""
    ew = ex_rois[:, :, 2] - ex_rois[:, :, 0] + 1.0
    eh = ex_rois[:, :, 3] - ex_rois[:, :, 1] + 1.0
    ecx = ex_rois[:, :, 0] + 0.5 * ew
    ecy = ex_rois[:, :, 1] + 0.5 * eh
    gw = gt_rois[:, :, 2] - gt_rois[:, :, 0] + 1.0
    gh = gt_rois[:, :, 3] - gt_rois[:, :, 1] + 1.0
    gcx = gt_rois[:, :, 0] + 0.5 * gw
    gcy = gt_rois[:, :, 1] + 0.5 * gh
    return torch.stack(((gcx - ecx) / ew, (gcy - ecy) / eh, torch.log(gw / ew), torch.log(gh / eh)), 2)
