"""A tiny synthetic ILSVRC-layout devkit (ImageSets / Annotations / Data for VID train, VID test and DET train) for the
data-layer tests.  Deterministic: image content and boxes come from a seeded generator, images are stored losslessly
(PNG bytes under the reference's .JPEG names; PIL reads by content).  Used by tests/golden/make_golden_data.py (which
feeds it to the REFERENCE's data layer) and by tests/test_data_layer_cpu.py (which feeds it to dtt.data)."""
import os

import numpy as np
from PIL import Image

WNIDS = ["n02691156", "n02419796", "n02131653", "n02834778", "n01503061", "n02924116", "n02958343"]
UNKNOWN = "n00000000"  # not one of the 30 VID classes: dropped by the readers

# (video, snippet, width, height, frames); each frame = list of (wnid index or -1 for UNKNOWN, track id, box fractions)
VIDEOS = {
    "train": [
        ("vidA", "snip0", 80, 60, [[(0, 0, (.1, .2, .5, .7)), (1, 1, (.5, .1, .9, .6))],
                                   [(0, 0, (.15, .2, .55, .7)), (1, 1, (.5, .15, .9, .65))],
                                   [(0, 0, (.2, .25, .6, .75))],
                                   [(2, 2, (.3, .3, .7, .9))],              # shares no track with the frame before
                                   [(2, 2, (.35, .3, .75, .9)), (-1, 3, (.0, .0, .3, .3))]]),
        ("vidB", "snip1", 150, 50, [[(3, 0, (.3, .2, .5, .8))], [(3, 0, (.32, .2, .52, .8)), (4, 1, (.7, .1, .95, .9))],
                                    [(3, 0, (.34, .2, .54, .8)), (4, 1, (.6, .1, .85, .9))],
                                    [(3, 0, (.0, .2, .2, .8)), (4, 1, (.5, .1, .75, .9))]]),     # a box touching x = 0
        ("vidC", "snip2", 40, 100, [[(5, 0, (.2, .4, .8, .6))], [(5, 0, (.2, .42, .8, .62))],
                                    [(5, 0, (.2, .0, .8, .2))],                                  # touches y = 0
                                    [(5, 0, (.2, .05, .8, .95))]]),                              # taller than the crop
        ("vidD", "snip3", 60, 60, [[(6, 0, (.2, .2, .6, .6))], [], [(6, 0, (.25, .2, .65, .6))],
                                   [(6, 0, (.3, .2, .7, .6)), (-1, 1, (.5, .5, .9, .9))]]),
        ("vidE", "snip4", 50, 70, [[(1, 4, (.1, .1, .5, .5))], [(1, 4, (.12, .1, .52, .5))], [(1, 4, (.14, .1, .54, .5))]]),
    ],
    "test": [
        ("vidT", "snipT", 72, 54, [[(0, 0, (.1, .2, .5, .7))], [], [(0, 0, (.2, .2, .6, .7)), (3, 1, (.5, .5, .9, .9))]]),
        ("vidU", "snipU", 54, 72, [[(2, 0, (.3, .3, .8, .8))], [(2, 0, (.3, .32, .8, .82))]]),
    ],
}
DET_TRAIN = [("det_0001", 64, 48, [(0, (.1, .1, .6, .7)), (6, (.5, .3, .9, .9))]), ("det_0002", 48, 64, [(4, (.2, .2, .7, .8))]),
             ("det_0003", 70, 50, [])]


def _write_image(path, w, h, rng):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)).save(path, format="PNG")


def _write_xml(path, w, h, objs, with_track):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    parts = ["<annotation><size><width>%d</width><height>%d</height></size>" % (w, h)]
    for wn, track, (fx1, fy1, fx2, fy2) in objs:
        name = UNKNOWN if wn < 0 else WNIDS[wn]
        box = (int(round(fx1 * (w - 1))), int(round(fy1 * (h - 1))), int(round(fx2 * (w - 1))), int(round(fy2 * (h - 1))))
        trk = "<trackid>%d</trackid>" % track if with_track else ""
        parts.append("<object>%s<name>%s</name><bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax>"
                     "</bndbox></object>" % ((trk, name) + box))
    parts.append("</annotation>")
    with open(path, "w") as f:
        f.write("".join(parts))


def build_devkit(root):
    """Creates <root>/ILSVRC/... and returns its path."""
    rng = np.random.RandomState(77)
    dev = os.path.join(root, "ILSVRC")
    for split, videos in VIDEOS.items():
        lines = []
        for video, snippet, w, h, frames in videos:
            for k, objs in enumerate(frames):
                index = "%s/%s/%06d" % (video, snippet, k)
                _write_image(os.path.join(dev, "Data", "VID", split, index + ".JPEG"), w, h, rng)
                _write_xml(os.path.join(dev, "Annotations", "VID", split, index + ".xml"), w, h, objs, True)
                lines.append("%s %d %d %d" % (index, 1, k, len(frames)) if split == "train" else "%s %d" % (index, k))
        os.makedirs(os.path.join(dev, "ImageSets", "VID"), exist_ok=True)
        with open(os.path.join(dev, "ImageSets", "VID", split + ".txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    lines = []
    for k, (index, w, h, objs) in enumerate(DET_TRAIN):
        _write_image(os.path.join(dev, "Data", "DET", "train", index + ".JPEG"), w, h, rng)
        _write_xml(os.path.join(dev, "Annotations", "DET", "train", index + ".xml"), w, h, [(c, 0, b) for c, b in objs], False)
        lines.append("%s %d" % (index, k))
    os.makedirs(os.path.join(dev, "ImageSets", "DET"), exist_ok=True)
    with open(os.path.join(dev, "ImageSets", "DET", "train.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return dev


def synthetic_detections(roidb_pairs, num_classes, seed=5):
    """all_boxes[class][pair] for the evaluation: jittered ground truth of the pair's first frame + random boxes."""
    rng = np.random.RandomState(seed)
    all_boxes = [[np.zeros((0, 5), np.float32) for _ in roidb_pairs] for _ in range(num_classes)]
    for i, pair in enumerate(roidb_pairs):
        e = pair[0]
        for j in range(1, num_classes):
            rows = []
            for b, c in zip(e["boxes"], e["gt_classes"]):
                if c == j and rng.rand() < 0.8:
                    rows.append(list(b.astype(np.float64) + rng.randint(-3, 4, 4)) + [rng.uniform(0.3, 1.0)])
                    if rng.rand() < 0.3:  # a duplicate detection of the same object
                        rows.append(list(b.astype(np.float64) + rng.randint(-2, 3, 4)) + [rng.uniform(0.2, 0.9)])
            if rng.rand() < 0.15:
                x, y = rng.uniform(0, 30, 2)
                rows.append([x, y, x + rng.uniform(5, 30), y + rng.uniform(5, 30), rng.uniform(0.05, 0.6)])
            if rows:
                all_boxes[j][i] = np.array(rows, dtype=np.float32)
    return all_boxes


def entry_summary(e, dev):
    """Comparable view of one roidb entry (paths relative to the devkit)."""
    return {"image": os.path.relpath(e["image"], dev), "frame_id": int(e["frame_id"]), "video_snippet": e["video_snippet"],
            "frame_snippet_len": int(e["frame_snippet_len"]), "width": int(e["width"]), "height": int(e["height"]),
            "boxes": np.asarray(e["boxes"]).astype(np.int64), "gt_classes": np.asarray(e["gt_classes"]).astype(np.int64),
            "track_id": np.asarray(e["track_id"]).astype(np.int64), "flipped": bool(e["flipped"]),
            "gt_overlaps": np.asarray(e["gt_overlaps"].toarray(), dtype=np.float32),
            "max_classes": np.asarray(e["max_classes"]).reshape(-1).astype(np.int64),
            "max_overlaps": np.asarray(e["max_overlaps"], dtype=np.float32).reshape(-1), "img_id": int(e["img_id"]),
            "need_crop": int(e.get("need_crop", -1))}


def run_data_layer(api, data_dir, out_dir):
    """Drives one data-layer implementation (api: namespace with combined_roidb, roibatchLoader, vid_eval, parse_vid_rec,
    write_results) over the synthetic devkit with fixed seeds; returns a flat dict of comparable numpy arrays / strings.
    The same function runs the reference (tests/golden/make_golden_data.py) and dtt.data (tests/test_data_layer_cpu.py)."""
    import pickle
    dev = os.path.join(data_dir, "ILSVRC")
    out = {}

    def put_pairs(tag, pairs):
        out[tag + "_n"] = np.array(len(pairs))
        for i, pair in enumerate(pairs):
            for k, e in enumerate(pair):
                for key, val in entry_summary(e, dev).items():
                    out["%s_%d_%d_%s" % (tag, i, k, key)] = np.array(val)

    def put_items(tag, ds, n):
        for i in range(n):
            item = ds[i]
            for name, t in zip(("data", "im_info", "gt", "num"), item):
                out["%s_%d_%s" % (tag, i, name)] = t.cpu().numpy()

    np.random.seed(3)
    imdb, pairs, ratio_list, ratio_index = api.combined_roidb("imagenet_vid_train")
    out["train_name"] = np.array(imdb.name); out["train_num_classes"] = np.array(imdb.num_classes)
    put_pairs("train", pairs)
    out["train_ratio_list"] = np.asarray(ratio_list, dtype=np.float64); out["train_ratio_index"] = np.asarray(ratio_index)
    ds = api.roibatchLoader(pairs, ratio_list, ratio_index, 2, imdb.num_classes, training=True)
    out["train_ratio_list_batch"] = ds.ratio_list_batch.numpy()
    np.random.seed(11)
    put_items("train_item", ds, len(ds))
    out["train_rng_after"] = np.array(np.random.randint(0, 1 << 30))  # same RNG consumption

    np.random.seed(4)
    imdb_d, pairs_d, rl_d, ri_d = api.combined_roidb("imagenet_det_train", duplicate_frames=True)
    put_pairs("det", pairs_d)
    out["det_ratio_list"] = np.asarray(rl_d, dtype=np.float64); out["det_ratio_index"] = np.asarray(ri_d)
    ds_d = api.roibatchLoader(pairs_d, rl_d, ri_d, 1, imdb_d.num_classes, training=True)
    np.random.seed(12)
    put_items("det_item", ds_d, len(ds_d))

    imdb_t, pairs_t, rl_t, ri_t = api.combined_roidb("imagenet_vid_test", False)
    put_pairs("test", pairs_t)
    ds_t = api.roibatchLoader(pairs_t, rl_t, ri_t, 1, imdb_t.num_classes, training=False)
    np.random.seed(13)
    put_items("test_item", ds_t, len(ds_t))

    # evaluation: detections -> results files -> per-class precision / recall / AP
    all_boxes = synthetic_detections(pairs_t, imdb_t.num_classes)
    text = api.write_results(imdb_t, all_boxes, pairs_t)
    if text is not None:
        out["results_text"] = np.array(text)
    annopath = os.path.join(dev, "Annotations", "VID", "test", "{:s}.xml")
    imageset = os.path.join(dev, "ImageSets", "VID", "test.txt")
    cachedir = os.path.join(out_dir, "annotations_cache")
    os.makedirs(cachedir, exist_ok=True)
    with open(imageset) as f:
        names = [x.strip().split(" ")[0] for x in f.readlines()]
    recs = {n: api.parse_vid_rec(annopath.format(n)) for n in names}
    out["parsed_recs"] = np.array(repr(sorted((n, [(o["name"], o["difficult"], list(o["bbox"])) for o in r])
                                              for n, r in recs.items())))
    with open(os.path.join(cachedir, "annots.pkl"), "wb") as f:  # the reference writes this cache in text mode (py2)
        pickle.dump(recs, f)
    index = [os.path.splitext("/".join(p[0]["image"].split("/")[-3:]))[0] for p in pairs_t]
    det_tmpl = os.path.join(out_dir, "det_test_{:s}.txt")
    for j, cls in enumerate(imdb_t.classes):
        if j == 0:
            continue
        with open(det_tmpl.format(cls), "wt") as f:
            for i, name in enumerate(index):
                for row in all_boxes[j][i]:
                    f.write("{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n".format(name, row[-1], row[0] + 1, row[1] + 1,
                                                                                  row[2] + 1, row[3] + 1))
    aps = []
    for j, cls in enumerate(imdb_t.classes):
        if j == 0:
            continue
        for metric07 in (False, True):
            rec, prec, ap = api.vid_eval(det_tmpl, annopath, imageset, cls, cachedir, ovthresh=0.5, use_07_metric=metric07)
            tag = "eval_%s_%d" % (cls, int(metric07))
            out[tag + "_rec"] = np.asarray(rec, dtype=np.float64); out[tag + "_prec"] = np.asarray(prec, dtype=np.float64)
            out[tag + "_ap"] = np.asarray(ap, dtype=np.float64)
            if not metric07:
                aps.append(float(ap))
    out["eval_map"] = np.array(np.mean(aps))
    return out
