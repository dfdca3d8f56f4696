"""Frame-pair roidb construction (reference: lib/roi_data_layer/roidb.py:13-173)."""
import numpy as np
from PIL import Image

from ..config import cfg
from .factory import get_imdb
from .imdb import imdb as _imdb_base


def create_roi_pairs(roidb, training, duplicate_frames=False):
    """roidb.py:13-52.  Consecutive entries form a pair when they come from the same video snippet; for training they
    must also share at least one track id and agree on `flipped`.  duplicate_frames (DET still images): every entry is
    paired with itself (training only -- at test time the reference reads snippet names it never set; same here)."""
    pairs = []
    if duplicate_frames:
        print("Duplicating frames for each roidb entry.")
        if not training:
            raise NameError("video_snippet1")  # the reference fails the same way (roidb.py:47, names unbound)
        pairs = [(e, e) for e in roidb]
    else:
        for a, b in zip(roidb[:-1], roidb[1:]):
            if a["video_snippet"] != b["video_snippet"]:
                continue
            if training and not (set(a["track_id"]) & set(b["track_id"]) and a["flipped"] == b["flipped"]):
                continue
            pairs.append((a, b))
    print("Pairs in roidb: {}".format(len(pairs)))
    assert len(pairs) <= len(roidb), "Something is wrong. Too many frame pairs."
    return pairs


def prepare_roidb(imdb):
    """roidb.py:54-88: image id / path / size and the per-box max overlap + class."""
    roidb = imdb.roidb
    for i in range(len(imdb.image_index)):
        e = roidb[i]
        e["img_id"] = imdb.image_id_at(i)
        e["image"] = imdb.image_path_at(i)
        if not imdb.name.startswith("coco"):
            with Image.open(e["image"]) as im:
                e["width"], e["height"] = im.size
        ov = e["gt_overlaps"].toarray()
        e["max_overlaps"] = ov.max(axis=1)
        e["max_classes"] = ov.argmax(axis=1)
        assert all(e["max_classes"][e["max_overlaps"] == 0] == 0)
        assert all(e["max_classes"][e["max_overlaps"] > 0] != 0)


def rank_roidb_ratio(roidb_pairs):
    """roidb.py:91-115: width / height of the first frame, clamped to [0.5, 2] (clamped entries get need_crop = 1);
    returns (sorted ratios, argsort)."""
    ratios = np.empty(len(roidb_pairs))
    for i, pair in enumerate(roidb_pairs):
        first = pair[0]
        r = first["width"] / float(first["height"])
        first["need_crop"] = int(r > 2 or r < 0.5)
        ratios[i] = min(max(r, 0.5), 2.0)
    order = np.argsort(ratios)
    return ratios[order], order


def filter_roidb(roidb):
    """roidb.py:117-129: drop entries without boxes (in place)."""
    print("before filtering, there are %d images..." % len(roidb))
    roidb[:] = [e for e in roidb if len(e["boxes"]) != 0]
    print("after filtering, there are %d images..." % len(roidb))
    return roidb


def combined_roidb(imdb_names, training=True, duplicate_frames=False):
    """roidb.py:131-173: '+'-joined dataset names -> (imdb, frame pairs, sorted ratios, ratio order).  Training pairs are
    shuffled with numpy seed 123."""
    names = imdb_names.split("+")

    def one(name):
        db = get_imdb(name)
        print("Loaded dataset `{:s}` for training".format(db.name))
        db.set_proposal_method(cfg.TRAIN.PROPOSAL_METHOD)
        print("Set proposal method: {:s}".format(cfg.TRAIN.PROPOSAL_METHOD))
        if cfg.TRAIN.USE_FLIPPED:
            print("Appending horizontally-flipped training examples...")
            db.append_flipped_images()
        prepare_roidb(db)
        return db.roidb

    roidbs = [one(n) for n in names]
    roidb = roidbs[0]
    for extra in roidbs[1:]:
        roidb.extend(extra)
    imdb = _imdb_base(imdb_names, get_imdb(names[1]).classes) if len(names) > 1 else get_imdb(imdb_names)
    if training:
        roidb = filter_roidb(roidb)
    pairs = create_roi_pairs(roidb, training, duplicate_frames)
    if training:
        np.random.seed(123)
        np.random.shuffle(pairs)
    ratio_list, ratio_index = rank_roidb_ratio(pairs)
    return imdb, pairs, ratio_list, ratio_index
