"""One-image minibatch blobs (reference: lib/roi_data_layer/minibatch.py:20-88)."""
import numpy as np
import numpy.random as npr

from ..config import cfg
from .blob import im_list_to_blob, imread_bgr, prep_im_for_blob


def get_minibatch(roidb, num_classes):
    """roidb: a one-entry list.  Returns {'data' (1,H,W,3) BGR mean-subtracted, 'gt_boxes' (G,6) = [x1,y1,x2,y2,cls,
    track_id] in scaled pixels, 'im_info' (1,3) = [H, W, scale], 'img_id'}.  Draws one value from numpy's global RNG
    for the scale index, like the reference (minibatch.py:24-25)."""
    assert len(roidb) == 1, "Single batch only"
    scale_inds = npr.randint(0, high=len(cfg.TRAIN.SCALES), size=1)
    assert cfg.TRAIN.BATCH_SIZE % 1 == 0
    entry = roidb[0]
    im = imread_bgr(entry["image"])
    if entry["flipped"]:
        im = im[:, ::-1, :]
    im, scale = prep_im_for_blob(im, cfg.PIXEL_MEANS, cfg.TRAIN.SCALES[scale_inds[0]], cfg.TRAIN.MAX_SIZE)
    blob = im_list_to_blob([im])
    if cfg.TRAIN.USE_ALL_GT:
        gt_inds = np.where(entry["gt_classes"] != 0)[0]
    else:  # minibatch.py:43-44 (operator precedence as written there)
        gt_inds = np.where(entry["gt_classes"] != 0 & np.all(entry["gt_overlaps"].toarray() > -1.0, axis=1))[0]
    gt_boxes = np.empty((len(gt_inds), 6), dtype=np.float32)
    gt_boxes[:, 0:4] = entry["boxes"][gt_inds, :] * scale
    gt_boxes[:, 4] = entry["gt_classes"][gt_inds]
    gt_boxes[:, 5] = entry["track_id"][gt_inds]
    return {"data": blob, "gt_boxes": gt_boxes,
            "im_info": np.array([[blob.shape[1], blob.shape[2], scale]], dtype=np.float32), "img_id": entry["img_id"]}
