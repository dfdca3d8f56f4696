"""Image database base class (reference: lib/datasets/imdb.py:21-131; the proposal-recall evaluation, which needs the
compiled cython_bbox helper, is not part of the D&T path and is left out)."""
import os

from PIL import Image

from ..config import cfg


class imdb(object):
    def __init__(self, name, classes=None):
        self._name = name
        self._classes = classes if classes else []
        self._image_index = []
        self._obj_proposer = "gt"
        self._roidb = None
        self._roidb_handler = self.default_roidb
        self.config = {}

    name = property(lambda self: self._name)
    classes = property(lambda self: self._classes)
    num_classes = property(lambda self: len(self._classes))
    image_index = property(lambda self: self._image_index)
    num_images = property(lambda self: len(self._image_index))

    @property
    def roidb_handler(self):
        return self._roidb_handler

    @roidb_handler.setter
    def roidb_handler(self, val):
        self._roidb_handler = val

    def set_proposal_method(self, method):
        """imdb.py:62-64: 'gt' -> self.gt_roidb, 'rpn' -> self.rpn_roidb, ..."""
        self.roidb_handler = getattr(self, method + "_roidb")

    @property
    def roidb(self):
        """List of dicts (boxes, gt_overlaps, gt_classes, flipped, ...), built once by the handler."""
        if self._roidb is None:
            self._roidb = self.roidb_handler()
        return self._roidb

    @property
    def cache_path(self):
        path = os.path.abspath(os.path.join(cfg.DATA_DIR, "cache"))
        os.makedirs(path, exist_ok=True)
        return path

    def image_path_at(self, i):
        raise NotImplementedError

    def image_id_at(self, i):
        raise NotImplementedError

    def default_roidb(self):
        raise NotImplementedError

    def evaluate_detections(self, all_boxes, output_dir=None):
        raise NotImplementedError

    def _get_widths(self):
        widths = []
        for i in range(self.num_images):
            with Image.open(self.image_path_at(i)) as im:
                widths.append(im.size[0])
        return widths

    def append_flipped_images(self):
        """imdb.py:113-131: mirrored copies of every entry (boxes reflected; only boxes / gt_overlaps / gt_classes /
        flipped are carried over, as in the reference -- the D&T drivers switch flipping off, trainval_net.py:191)."""
        widths = self._get_widths()
        for i in range(self.num_images):
            boxes = self.roidb[i]["boxes"].copy()
            x1, x2 = boxes[:, 0].copy(), boxes[:, 2].copy()
            boxes[:, 0] = widths[i] - x2 - 1
            boxes[:, 2] = widths[i] - x1 - 1
            assert (boxes[:, 2] >= boxes[:, 0]).all()
            self.roidb.append({"boxes": boxes, "gt_overlaps": self.roidb[i]["gt_overlaps"],
                               "gt_classes": self.roidb[i]["gt_classes"], "flipped": True})
        self._image_index = self._image_index * 2
