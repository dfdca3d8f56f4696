#!/usr/bin/env python3
"""Evaluation driver with the reference's command line (test_net.py:38-86): loads
`{load_dir}/{net}/{dataset}/rfcn_detect_track_{checksession}_{checkepoch}_{checkpoint}.pth` (or runs random
weights when the file does not exist and `--dataset synthetic`), runs the D&T forward on frame pairs, decodes the
boxes (test_net.py:239-266) and applies the per-class NMS + top-100 cut (test_net.py:274-301) -- the latter as
ONE device launch per pair instead of 30 NMS round trips.  Writes `detections.pkl` with the reference's
`all_boxes[class][pair]` layout.  `--dataset imagenet_vid` reads the VID test split of an ILSVRC devkit under
cfg.DATA_DIR/ILSVRC (dtt/data) and finishes with the VOC-style mAP (imagenet_detect.py:263-318); `--dataset synthetic`
(the default: no dataset ships with this repo) runs seeded synthetic pairs.
"""
import argparse
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Test a Detect-to-Track R-FCN network")
    p.add_argument("--dataset", dest="dataset", default="synthetic", type=str)
    p.add_argument("--cfg", dest="cfg_file", default="cfgs/res101.yml", type=str)
    p.add_argument("--net", dest="net", default="res101", type=str)
    p.add_argument("--set", dest="set_cfgs", default=None, nargs=argparse.REMAINDER)
    p.add_argument("--load_dir", dest="load_dir", default="output/models", type=str)
    p.add_argument("--cuda", dest="cuda", action="store_true")
    p.add_argument("--ls", dest="large_scale", action="store_true")
    p.add_argument("--mGPUs", dest="mGPUs", action="store_true")
    p.add_argument("--cag", dest="class_agnostic", action="store_true")
    p.add_argument("--checksession", dest="checksession", default=1, type=int)
    p.add_argument("--checkepoch", dest="checkepoch", default=1, type=int)
    p.add_argument("--checkpoint", dest="checkpoint", default=16470, type=int)
    p.add_argument("--bs", dest="batch_size", default=1, type=int)
    p.add_argument("--vis", dest="vis", action="store_true")
    p.add_argument("--num_pairs", default=10, type=int)
    p.add_argument("--height", default=600, type=int)
    p.add_argument("--width", default=1067, type=int)
    p.add_argument("--out_dir", default="output/detections", type=str)
    p.add_argument("--link_tubes", action="store_true",
                   help="treat the pairs as consecutive frame pairs of one video and link class tubes (demo.py:432-489)")
    p.add_argument("--online_tubes", action="store_true",
                   help="also run the demo's incremental tube linker + temporal labelling (online_tubes.py) over the pairs")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    from dtt.config import apply_dataset_defaults, cfg, cfg_from_file, cfg_from_list
    from dtt.fuse import fuse_for_inference
    from dtt.postprocess import class_nms, decode_detections, to_all_boxes
    from dtt.synth import build_model, calibrate_batchnorm_, make_batch
    print("Called with args:")
    print(args)
    apply_dataset_defaults("imagenet_vid" if args.dataset == "synthetic" else args.dataset)
    cfg_file = os.path.join(ROOT, "cfgs", "{}_ls.yml".format(args.net)) if args.large_scale else os.path.join(ROOT, args.cfg_file)
    cfg_from_file(cfg_file)
    if args.set_cfgs:
        cfg_from_list(args.set_cfgs)
    np.random.seed(cfg.RNG_SEED)
    imdb = pairs = data_iter = None
    if args.dataset != "synthetic":
        # test_net.py:131-144: imagenet_vid_test pairs through the test-mode loader, one pair per step
        from dtt.data import combined_roidb, roibatchLoader
        if args.dataset not in ("imagenet_vid", "imagenet_vid+imagenet_det"):
            raise KeyError("Unknown dataset: {}".format(args.dataset))
        cfg.TRAIN.USE_FLIPPED = False
        imdb, pairs, ratio_list, ratio_index = combined_roidb("imagenet_vid_test", False)
        imdb.competition_mode(on=True)
        print("{:d} roidb frame pairs".format(len(pairs)))
        loader = torch.utils.data.DataLoader(roibatchLoader(pairs, ratio_list, ratio_index, 1, imdb.num_classes,
                                                            training=False), batch_size=1, shuffle=False, num_workers=0)
        data_iter = iter(loader)
        args.num_pairs, args.batch_size = len(pairs), 1
    dev = torch.device("cuda:0")
    layers = {"res50": 50, "res101": 101, "res152": 152}[args.net]
    model = build_model(layers, class_agnostic=args.class_agnostic, cfg=cfg).to(dev)
    load_name = os.path.join(args.load_dir, args.net, args.dataset,
                             "rfcn_detect_track_{}_{}_{}.pth".format(args.checksession, args.checkepoch, args.checkpoint))
    if os.path.exists(load_name):
        ck = torch.load(load_name, map_location=dev)
        model.load_state_dict(ck["model"])
        if "pooling_mode" in ck:
            cfg.POOLING_MODE = ck["pooling_mode"]
        print("load model successfully! (%s)" % load_name)
    else:
        print("no checkpoint at %s: running random-init weights" % load_name)
        im, _, _, _ = make_batch(1, args.height, args.width, seed=1, device=dev)
        calibrate_batchnorm_(model, im[:, 0])
    model.eval()
    fuse_for_inference(model)
    max_per_image, thresh = 100, 0.05  # test_net.py:194-199 (vis off)
    n_classes = model.n_classes
    all_boxes = [[[] for _ in range(args.num_pairs)] for _ in range(n_classes)]
    det_time = nms_time = 0.0
    vid_boxes, vid_scores, vid_trk = [], [], []
    for i in range(args.num_pairs):
        if data_iter is not None:
            im, info, gt, nb = (t.to(dev) for t in next(data_iter))
        else:
            im, info, gt, nb = make_batch(args.batch_size, args.height, args.width, seed=10 + i, device=dev)
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.no_grad():
            rois, cls_prob, bbox_pred, tracking_pred = model(im, info, gt, nb)[:4]
            boxes = decode_detections(rois[0], bbox_pred[0], info[:, 0], cfg, args.class_agnostic)   # leg 0 (:277)
            if args.link_tubes or args.online_tubes:   # demo.py:432-473: both legs' boxes / scores and the tracked boxes of snippet 0
                legs = [decode_detections(rois[l], bbox_pred[l], info[:, l], cfg, args.class_agnostic)[0] for l in range(2)]
                vid_boxes.append(torch.stack(legs, 0))
                vid_scores.append(torch.stack([cls_prob[0][0], cls_prob[1][0]], 0))
                R = rois.size(2)
                vid_trk.append(decode_detections(rois[0][:1], tracking_pred.view(-1, R, 4)[:1], info[:, 0][:1], cfg, True)[0])
        torch.cuda.synchronize()
        t1 = time.time()
        dets, counts = class_nms(cls_prob[0], boxes, thresh, cfg.TEST.NMS, max_per_image, args.class_agnostic)
        per_image = to_all_boxes(dets, counts)
        t2 = time.time()
        for j in range(1, n_classes):
            all_boxes[j][i] = per_image[0][j]
        det_time += t1 - t0
        nms_time += t2 - t1
        sys.stdout.write("im_detect: {:d}/{:d} {:.3f}s {:.3f}s   \r".format(i + 1, args.num_pairs, t1 - t0, t2 - t1))
        sys.stdout.flush()
    if args.link_tubes:
        from dtt.tubes import VideoPostProcessor
        torch.cuda.synchronize()
        t0 = time.time()
        vp = VideoPostProcessor(torch.stack(vid_boxes, 0), torch.stack(vid_scores, 0), torch.stack(vid_trk, 0),
                                ["__background__"] + ["class_%d" % j for j in range(1, n_classes)])
        paths = vp.build_class_paths()
        torch.cuda.synchronize()
        n_tubes = [0 if p is None else int(p["idx"].shape[0]) for p in paths]
        print("\ntube linking: %d classes, %d frames, %d tubes in %.1f ms" %
              (n_classes - 1, vp.num_frames, sum(n_tubes), (time.time() - t0) * 1e3))
        os.makedirs(args.out_dir, exist_ok=True)
        with open(os.path.join(args.out_dir, "tubes.pkl"), "wb") as f:
            pickle.dump([None if p is None else {k: v.cpu().numpy() for k, v in p.items()} for p in paths], f,
                        pickle.HIGHEST_PROTOCOL)
    if args.online_tubes:   # demo.py:487-489: incremental linking + temporal labelling of the video demo
        from dtt.online_tubes import VideoPostProcessor as OnlineVideoPostProcessor
        t0 = time.time()
        ovp = OnlineVideoPostProcessor(torch.stack(vid_boxes, 0), torch.stack(vid_scores, 0), torch.stack(vid_trk, 0),
                                       ["__background__"] + ["class_%d" % j for j in range(1, n_classes)], "video")
        ovp.class_paths(path_score_thresh=0.5)
        print("online tube linking: %d tubes above 0.5 in %.1f ms" % (len(ovp.path_boxes), (time.time() - t0) * 1e3))
        os.makedirs(args.out_dir, exist_ok=True)
        with open(os.path.join(args.out_dir, "online_tubes.pkl"), "wb") as f:
            pickle.dump({"labels": ovp.path_labels.numpy(), "starts": ovp.path_starts.numpy(), "ends": ovp.path_ends.numpy(),
                         "boxes": [b.numpy() for b in ovp.path_boxes], "scores": [x.numpy() for x in ovp.path_scores]}, f,
                        pickle.HIGHEST_PROTOCOL)
    os.makedirs(args.out_dir, exist_ok=True)
    with open(os.path.join(args.out_dir, "detections.pkl"), "wb") as f:
        pickle.dump(all_boxes, f, pickle.HIGHEST_PROTOCOL)
    print("\nmean detect time %.4fs, mean per-class NMS time %.4fs over %d pairs" %
          (det_time / args.num_pairs, nms_time / args.num_pairs, args.num_pairs))
    if imdb is not None:
        print("Evaluating detections")  # test_net.py:309-310
        empty = np.zeros((0, 5), dtype=np.float32)
        boxes = [[(b if len(b) else empty) for b in per_class] for per_class in all_boxes]
        aps = imdb.evaluate_detections(boxes, pairs, args.out_dir)
        return float(np.mean(aps))


if __name__ == "__main__":
    main()
