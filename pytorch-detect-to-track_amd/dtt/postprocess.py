"""Test-time decoding and per-class NMS of the D&T outputs (reference test_net.py:239-301) on the device.

`decode_detections` un-normalises the box deltas, applies them to the RoIs and clips (test_net.py:243-266);
`class_nms` runs the 30 per-class threshold / sort / NMS passes and the max_per_image cut as ONE launch per batch
of images (dtt_class_nms) instead of 30 NMS round trips per frame pair; `to_all_boxes` converts the dense result
into the `all_boxes[j][i]` numpy lists the reference's evaluators consume.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, require_f32_contig, require_gpu, stream_ptr
from .rpn import bbox_transform_inv, clip_boxes


def decode_detections(rois, bbox_pred, im_info, cfg, class_agnostic=True):
    """rois (B,R,5), bbox_pred (B,R,4 or 4*ncls), im_info (B,3) -> boxes (B,R,4k) in original-image pixels."""
    boxes = rois[:, :, 1:5]
    deltas = bbox_pred
    if cfg.TEST.BBOX_REG and cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED:
        stds = torch.tensor(cfg.TRAIN.BBOX_NORMALIZE_STDS, device=deltas.device, dtype=deltas.dtype)
        means = torch.tensor(cfg.TRAIN.BBOX_NORMALIZE_MEANS, device=deltas.device, dtype=deltas.dtype)
        shp = deltas.shape
        deltas = (deltas.reshape(-1, 4) * stds + means).view(shp)
    pred = clip_boxes(bbox_transform_inv(boxes, deltas), im_info)
    return pred / im_info[:, 2].view(-1, 1, 1)


def class_nms(scores, boxes, score_thresh=0.05, nms_thresh=0.3, max_per_image=100, class_agnostic=True):
    """scores (I,R,ncls), boxes (I,R,4|4*ncls) -> dets (I,ncls,R,5) [x1,y1,x2,y2,score] in kept order, counts (I,ncls)."""
    require_gpu(scores, boxes)
    scores = scores.detach().float().contiguous()
    boxes = boxes.detach().float().contiguous()
    require_f32_contig("scores", scores)
    I, R, C = scores.shape
    if tuple(boxes.shape) != (I, R, 4 if class_agnostic else 4 * C):
        raise ValueError("class_nms: boxes %s do not match scores %s" % (tuple(boxes.shape), tuple(scores.shape)))
    dets = torch.zeros((I, C, R, 5), dtype=torch.float32, device=scores.device)
    counts = torch.empty((I, C), dtype=torch.int32, device=scores.device)
    with torch.cuda.device(scores.device):
        check(_lib.lib().dtt_class_nms(ptr(scores), ptr(boxes), I, R, C, int(bool(class_agnostic)), float(score_thresh),
                                       float(nms_thresh), int(max_per_image), ptr(dets), ptr(counts),
                                       stream_ptr(scores.device)), "class_nms")
    return dets, counts


def to_all_boxes(dets, counts):
    """-> list over images of list over classes of (n, 5) numpy arrays (class 0 empty), as test_net.py builds."""
    d, c = dets.cpu().numpy(), counts.cpu().numpy()
    return [[d[i, j, : c[i, j]].copy() if j > 0 else np.zeros((0, 5), np.float32) for j in range(d.shape[1])]
            for i in range(d.shape[0])]
