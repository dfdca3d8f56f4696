"""VOC-style average precision on ImageNet VID annotations (reference: lib/datasets/vid_eval.py:16-238)."""
import os
import pickle
import xml.etree.ElementTree as ET

import numpy as np

WNIDS = ("__background__",
         "n02691156", "n02419796", "n02131653", "n02834778", "n01503061", "n02924116", "n02958343", "n02402425",
         "n02084071", "n02121808", "n02503517", "n02118333", "n02510455", "n02342885", "n02374451", "n02129165",
         "n01674464", "n02484322", "n03790512", "n02324045", "n02509815", "n02411705", "n01726692", "n02355227",
         "n02129604", "n04468005", "n01662784", "n04530566", "n02062744", "n02391049")
CLASSES = ("__background__",
           "airplane", "antelope", "bear", "bicycle", "bird", "bus", "car", "cattle", "dog", "domestic_cat", "elephant",
           "fox", "giant_panda", "hamster", "horse", "lion", "lizard", "monkey", "motorcycle", "rabbit", "red_panda",
           "sheep", "snake", "squirrel", "tiger", "train", "turtle", "watercraft", "whale", "zebra")
_WNID_TO_CLASS = dict(zip(WNIDS, CLASSES))


def parse_vid_rec(filename):
    """vid_eval.py:39-57: the objects of one annotation file as dicts (name, difficult = 0, bbox [xmin,ymin,xmax,ymax])."""
    objects = []
    for obj in ET.parse(filename).findall("object"):
        box = obj.find("bndbox")
        objects.append({"name": _WNID_TO_CLASS[obj.find("name").text], "difficult": 0,
                        "bbox": [int(box.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")]})
    return objects


def vid_ap(rec, prec, use_07_metric=False):
    """vid_eval.py:60-90: 11-point (VOC07) or exact area under the monotone precision envelope."""
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            ap += (np.max(prec[rec >= t]) if np.sum(rec >= t) else 0) / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def vid_eval(detpath, annopath, imagesetfile, classname, cachedir, ovthresh=0.5, use_07_metric=False):
    """vid_eval.py:93-238.  detpath.format(classname): detections `image_id score x1 y1 x2 y2`; annopath.format(image):
    XML annotation; imagesetfile: one image name (first token) per line.  Returns (rec, prec, ap); an empty detection
    file gives (1e-4, 1e-4, 1e-4) like the reference.  Annotations are cached in cachedir/annots.pkl."""
    os.makedirs(cachedir, exist_ok=True)
    cachefile = os.path.join(cachedir, "annots.pkl")
    print(cachefile)
    with open(imagesetfile) as f:
        imagenames = [x.strip().split(" ")[0] for x in f.readlines()]
    if not os.path.isfile(cachefile):
        recs = {}
        for i, name in enumerate(imagenames):
            recs[name] = parse_vid_rec(annopath.format(name))
            if i % 100 == 0:
                print("Reading annotation for {:d}/{:d}".format(i + 1, len(imagenames)))
        print("Saving cached annotations to {:s}".format(cachefile))
        with open(cachefile, "wb") as f:
            pickle.dump(recs, f)
    else:
        with open(cachefile, "rb") as f:
            recs = pickle.load(f)
    # ground truth of this class per image
    gt_boxes, npos = {}, 0
    for name in imagenames:
        b = np.array([o["bbox"] for o in recs[name] if o["name"] == classname], dtype=float).reshape(-1, 4)
        gt_boxes[name] = b
        npos += len(b)  # every object counts: the parser marks nothing difficult (vid_eval.py:50)
    with open(detpath.format(classname)) as f:
        rows = [x.strip().split(" ") for x in f.readlines()]
    if not rows:
        return 0.0001, 0.0001, 0.0001
    image_ids = [r[0].replace("val/", "") for r in rows]
    confidence = np.array([float(r[1]) for r in rows])
    boxes = np.array([[float(z) for z in r[2:]] for r in rows])
    order = np.argsort(-confidence)  # the reference's order, ties included (vid_eval.py:186)
    rank_of = np.empty(len(order), dtype=np.int64)
    rank_of[order] = np.arange(len(order))
    # Matching is independent per image (a detection can only claim ground truth of its own image), so walk each
    # image's detections in global rank order and write the verdicts back at their ranks.
    by_image = {}
    for k in order:
        by_image.setdefault(image_ids[k], []).append(k)
    tp = np.zeros(len(order)); fp = np.zeros(len(order))
    for name, dets in by_image.items():
        g = gt_boxes[name]
        claimed = np.zeros(len(g), dtype=bool)
        g_area = (g[:, 2] - g[:, 0] + 1.0) * (g[:, 3] - g[:, 1] + 1.0)
        for k in dets:
            bb = boxes[k]
            hit = -1
            if len(g):
                iw = np.maximum(np.minimum(g[:, 2], bb[2]) - np.maximum(g[:, 0], bb[0]) + 1.0, 0.0)
                ih = np.maximum(np.minimum(g[:, 3], bb[3]) - np.maximum(g[:, 1], bb[1]) + 1.0, 0.0)
                inter = iw * ih
                ov = inter / ((bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0) + g_area - inter)
                j = int(np.argmax(ov))
                if ov[j] > ovthresh:
                    hit = j
            if hit >= 0 and not claimed[hit]:
                claimed[hit] = True
                tp[rank_of[k]] = 1.0
            else:
                fp[rank_of[k]] = 1.0  # below the threshold, or a second detection of an already claimed object
    tp_cum, fp_cum = np.cumsum(tp), np.cumsum(fp)
    with np.errstate(divide="ignore", invalid="ignore"):  # a class without ground truth: nan recall, as in the reference
        rec = tp_cum / float(npos)
    prec = tp_cum / np.maximum(tp_cum + fp_cum, np.finfo(np.float64).eps)
    return rec, prec, vid_ap(rec, prec, use_07_metric)
