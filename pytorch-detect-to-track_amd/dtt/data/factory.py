"""Dataset registry (reference: lib/datasets/factory.py:56-77; only the ImageNet VID / DET databases of the D&T path --
the VOC / COCO / Visual Genome readers of the faster-rcnn code base the reference grew from are not part of it).
The devkit lives under cfg.DATA_DIR/ILSVRC (the reference resolves 'data/ILSVRC' against the working directory)."""
import os

from ..config import cfg

_SPLITS = ("train", "val", "test")


def list_imdbs():
    return ["imagenet_%s_%s" % (k, s) for k in ("vid", "det") for s in _SPLITS]


def get_imdb(name):
    """'imagenet_vid_{train,val,test}' / 'imagenet_det_{train,val,test}' -> imagenet_detect instance."""
    from .imagenet_detect import imagenet_detect
    parts = name.split("_")
    if len(parts) != 3 or parts[0] != "imagenet" or parts[1] not in ("vid", "det") or parts[2] not in _SPLITS:
        raise KeyError("Unknown dataset: {}".format(name))
    return imagenet_detect(parts[2], os.path.join(cfg.DATA_DIR, "ILSVRC"), parts[1].upper())
