"""ImageNet VID / DET image database (reference: lib/datasets/imagenet_detect.py:12-323, Python-2 source restated).

Directory layout under the devkit root (the reference's, unchanged):
    ImageSets/{VID,DET}/<split>.txt      lines `index frame_id` or `index start_frame frame_id snippet_len`
    Annotations/{VID,DET}/<split>/<index>.xml
    Data/{VID,DET}/<split>/<index>.JPEG
"""
import os
import pickle
import uuid
import xml.etree.ElementTree as ET

import numpy as np
import scipy.sparse

from .imdb import imdb
from .vid_eval import CLASSES, WNIDS, vid_eval


class imagenet_detect(imdb):
    def __init__(self, image_set, devkit_path, det_or_vid):
        super().__init__("imagenet_" + det_or_vid.lower() + image_set)  # (sic: no separator, imagenet_detect.py:21)
        self._det_vid = det_or_vid
        self._image_set = image_set
        self._root_path = self._devkit_path = self._data_path = devkit_path
        self._classes = CLASSES
        self._classes_map = WNIDS
        print("Number of classes: {}".format(self.num_classes))
        self._class_to_ind = dict(zip(self.classes, range(self.num_classes)))
        self._image_ext = ".JPEG"
        assert os.path.exists(devkit_path), "imagenet devkit path does not exist: {}".format(devkit_path)
        self._load_image_set_index()
        self._roidb_handler = self.gt_roidb
        self._salt = str(uuid.uuid4())
        self.config = {"cleanup": True, "use_salt": True, "top_k": 2000, "use_diff": False, "rpn_file": None}

    # ------------------------------------------------------------------------------------------------ paths / index
    def image_path_at(self, i):
        return self.image_path_from_index(self._image_index[i])

    def image_id_at(self, i):
        return i

    def image_path_from_index(self, index):
        path = os.path.join(self._data_path, "Data", self._det_vid, self._image_set, index + self._image_ext)
        assert os.path.exists(path), "Path does not exist: {}".format(path)
        return path

    def _load_image_set_index(self):
        """imagenet_detect.py:94-115: two-column lists carry (index, frame id); four-column lists (index, first frame of
        the snippet, frame id, snippet length)."""
        path = os.path.join(self._data_path, "ImageSets", self._det_vid, self._image_set + ".txt")
        assert os.path.exists(path), "Path does not exist: {}".format(path)
        with open(path) as f:
            rows = [line.strip().split(" ") for line in f.readlines()]
        self._image_index = [r[0] for r in rows]
        if len(rows[0]) == 2:
            self._frame_id = [int(r[1]) for r in rows]
            self._frame_len = [-1] * len(rows)
        else:
            self._start_frame_id = [int(r[1]) for r in rows]
            self._frame_id = [int(r[2]) for r in rows]
            self._frame_len = [int(r[3]) for r in rows]

    # ------------------------------------------------------------------------------------------------ ground truth
    def gt_roidb(self):
        """Ground-truth roidb, cached in cfg.DATA_DIR/cache/<name>_gt_roidb.pkl (imagenet_detect.py:117-134)."""
        cache_file = os.path.join(self.cache_path, self.name + "_gt_roidb.pkl")
        if os.path.exists(cache_file):
            with open(cache_file, "rb") as fid:
                roidb = pickle.load(fid)
            print("{} gt roidb loaded from {}".format(self.name, cache_file))
            return roidb
        roidb = [self._load_vid_annotation(i, index) for i, index in enumerate(self.image_index)]
        with open(cache_file, "wb") as fid:
            pickle.dump(roidb, fid, pickle.HIGHEST_PROTOCOL)
        print("wrote gt roidb to {}".format(cache_file))
        return roidb

    def _load_vid_annotation(self, idx, index):
        """One annotation file -> roidb entry (imagenet_detect.py:154-230): boxes uint16 clipped to the image, classes by
        WordNet id (objects of other classes are dropped), one-hot gt_overlaps (sparse), track ids (object order for DET)."""
        rec = {"image": self.image_path_from_index(index), "frame_id": self._frame_id[idx]}
        parts = index.split("/")
        rec["video_snippet"] = parts[0] if len(parts) < 3 else parts[1]
        rec["frame_snippet_len"] = self._frame_len[idx]
        tree = ET.parse(os.path.join(self._data_path, "Annotations", self._det_vid, self._image_set, index + ".xml"))
        size = tree.find("size")
        rec["height"] = float(size.find("height").text)
        rec["width"] = float(size.find("width").text)
        wnid_to_ind = dict(zip(self._classes_map, range(self.num_classes)))
        boxes, classes, tracks = [], [], []
        for ix, obj in enumerate(tree.findall("object")):
            wnid = obj.find("name").text
            if wnid not in wnid_to_ind:
                continue
            bb = obj.find("bndbox")
            boxes.append([max(float(bb.find("xmin").text), 0), max(float(bb.find("ymin").text), 0),
                          min(float(bb.find("xmax").text), rec["width"] - 1),
                          min(float(bb.find("ymax").text), rec["height"] - 1)])
            classes.append(wnid_to_ind[wnid.lower().strip()])
            tracks.append(ix if self._det_vid == "DET" else int(obj.find("trackid").text))
        boxes = np.array(boxes, dtype=np.float64).reshape(-1, 4).astype(np.uint16)
        gt_classes = np.array(classes, dtype=np.int32)
        overlaps = np.zeros((len(classes), self.num_classes), dtype=np.float32)
        overlaps[np.arange(len(classes)), gt_classes] = 1.0
        overlaps = scipy.sparse.csr_matrix(overlaps)
        assert (boxes[:, 2] >= boxes[:, 0]).all()
        rec.update({"boxes": boxes, "gt_classes": gt_classes, "gt_overlaps": overlaps,
                    "max_classes": overlaps.argmax(axis=1), "max_overlaps": overlaps.max(axis=1), "flipped": False,
                    "track_id": np.array(tracks, dtype=np.uint16)})
        return rec

    # ------------------------------------------------------------------------------------------------ evaluation
    def _results_template(self):
        base = os.path.join(self._devkit_path, "results")
        os.makedirs(base, exist_ok=True)
        return os.path.join(base, "det_" + self._image_set + "_{:s}.txt")

    def _write_results(self, all_boxes):
        """imagenet_detect.py:244-261: one file per class, `index score x1 y1 x2 y2` with 1-based pixel coordinates."""
        for cls_ind, cls in enumerate(self.classes):
            if cls == "__background__":
                continue
            print("Writing {} Imagenet vid results file".format(cls))
            with open(self._results_template().format(cls), "wt") as f:
                for im_ind, index in enumerate(self._image_index):
                    dets = all_boxes[cls_ind][im_ind]
                    if len(dets) == 0:
                        continue
                    for k in range(dets.shape[0]):
                        f.write("{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n".format(
                            index, dets[k, -1], dets[k, 0] + 1, dets[k, 1] + 1, dets[k, 2] + 1, dets[k, 3] + 1))

    def _do_python_eval(self, output_dir="output"):
        """imagenet_detect.py:263-303: per-class AP at IoU 0.5 against the VID annotations, mean AP printed and returned."""
        annopath = os.path.join(self._devkit_path, "Annotations", "VID", self._image_set, "{:s}.xml")
        imagesetfile = os.path.join(self._devkit_path, "ImageSets", "VID", self._image_set + ".txt")
        cachedir = os.path.join(self._devkit_path, "annotations_cache")
        os.makedirs(output_dir, exist_ok=True)
        aps = []
        for cls in self._classes[1:]:
            rec, prec, ap = vid_eval(self._results_template().format(cls), annopath, imagesetfile, cls, cachedir,
                                     ovthresh=0.5)
            aps.append(ap)
            print("AP for {} = {:.4f}".format(cls, ap))
            with open(os.path.join(output_dir, cls + "_pr.pkl"), "wb") as f:
                pickle.dump({"rec": rec, "prec": prec, "ap": ap}, f)
        print("Mean AP = {:.4f}".format(np.mean(aps)))
        print("~~~~~~~~\nResults:")
        for ap in aps:
            print("{:.3f}".format(ap))
        print("{:.3f}\n~~~~~~~~".format(np.mean(aps)))
        return aps

    def evaluate_detections(self, all_boxes, roidb, output_dir):
        """imagenet_detect.py:305-318: all_boxes[class][pair] = (n, 5) detections of the pair's first frame; image names
        are rebuilt from the pairs' paths (last three components, extension stripped)."""
        self._roidb = roidb
        self._image_index = [os.path.splitext("/".join(pair[0]["image"].split("/")[-3:]))[0] for pair in roidb]
        self._write_results(all_boxes)
        aps = self._do_python_eval(output_dir)
        if self.config["cleanup"]:
            for cls in self._classes[1:]:
                os.remove(self._results_template().format(cls))
        return aps

    def competition_mode(self, on):
        self.config["use_salt"] = self.config["cleanup"] = not on
