#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference's own Python (build container only).

The reference lives at /root/reference (read-only, absent on the GPU box); its RPN-side modules
import under py3.10 / torch 2.10 once two shims are in place (a tiny ``easydict`` stand-in on
sys.path, and ``yaml.load`` defaulting to SafeLoader -- config.py:374 calls it without a Loader).
``model.nms.nms_wrapper`` is pre-seeded with the CPU oracle NMS because the reference's own import
chain ends in a CUDA cffi extension.  Nothing from the reference is copied: only inputs and the
outputs it computes are stored.

    python tests/golden/make_golden.py          # rewrites the fixtures in place
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("DTT_REFERENCE", "/root/reference")

EASYDICT_SHIM = '''
class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}); d.update(kw)
        for k, v in d.items():
            setattr(self, k, v)
    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)
    __setitem__ = __setattr__
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)
'''


def import_reference():
    shim = tempfile.mkdtemp(prefix="dtt_shim_")
    os.makedirs(os.path.join(shim, "easydict"))
    with open(os.path.join(shim, "easydict", "__init__.py"), "w") as f:
        f.write(EASYDICT_SHIM)
    sys.path.insert(0, shim)
    sys.path.insert(0, os.path.join(REF, "lib"))
    sys.path.insert(0, ROOT)
    import yaml

    _load = yaml.load
    yaml.load = lambda f, Loader=yaml.SafeLoader: _load(f, Loader=Loader)
    import torch
    from oracle import oracle_lib

    def nms_stub(dets, thresh, force_cpu=False):
        if dets.shape[0] == 0:
            return []
        keep = oracle_lib.nms(dets.detach().cpu().numpy(), float(thresh))
        return torch.from_numpy(keep.astype(np.int32)).view(-1, 1)

    m = types.ModuleType("model.nms.nms_wrapper")
    m.nms = nms_stub
    import model.nms  # noqa: F401  (package itself imports fine)

    sys.modules["model.nms.nms_wrapper"] = m
    from model.utils.config import cfg, cfg_from_file

    cfg_from_file(os.path.join(REF, "cfgs", "res101.yml"))
    return cfg


def t2n(t):
    return t.detach().cpu().numpy()


def main():
    cfg = import_reference()
    import torch
    from model.rpn import bbox_transform as bt
    from model.rpn.anchor_target_layer import _AnchorTargetLayer
    from model.rpn.generate_anchors import generate_anchors
    from model.rpn.proposal_layer import _ProposalLayer

    out = {}
    # ---------------------------------------------------------------- 1. anchors
    out["anchors_s8_16_32"] = generate_anchors(scales=np.array([8, 16, 32]), ratios=np.array([0.5, 1, 2]))
    out["anchors_s4_8_16_32"] = generate_anchors(scales=np.array([4, 8, 16, 32]), ratios=np.array([0.5, 1, 2]))
    np.savez_compressed(os.path.join(HERE, "anchors.npz"), **out)

    # ---------------------------------------------------------------- 2. box algebra
    rng = np.random.RandomState(3)
    B, N, K = 2, 40, 6
    xy = rng.uniform(0, 500, size=(B, N, 2)).astype(np.float32)
    wh = rng.uniform(1, 300, size=(B, N, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 2)
    deltas = rng.normal(0, 0.5, size=(B, N, 4)).astype(np.float32)
    deltas[0, 0] = 0  # identity
    deltas[0, 1] = [3.0, -3.0, 4.0, -4.0]  # large
    im_info = np.array([[600, 1067, 0.8333], [480, 640, 1.0]], dtype=np.float32)
    inv = bt.bbox_transform_inv(torch.from_numpy(boxes), torch.from_numpy(deltas), B)
    clipped = bt.clip_boxes(inv.clone(), torch.from_numpy(im_info), B)
    anchors2d = boxes[0]
    gt = np.zeros((B, K, 5), dtype=np.float32)
    gxy = rng.uniform(0, 400, size=(B, K - 2, 2))
    gwh = rng.uniform(20, 300, size=(B, K - 2, 2))
    gt[:, : K - 2, :2] = gxy
    gt[:, : K - 2, 2:4] = gxy + gwh
    gt[:, : K - 2, 4] = rng.randint(1, 31, size=(B, K - 2))
    gt[0, 0, :4] = anchors2d[5]  # exact match -> IoU 1
    anchors2d_e = anchors2d.copy()
    anchors2d_e[7] = [10, 10, 10, 10]  # degenerate 1x1 anchor -> overlap -1
    ov = bt.bbox_overlaps_batch(torch.from_numpy(anchors2d_e), torch.from_numpy(gt))
    ov2 = bt.bbox_overlaps(torch.from_numpy(anchors2d_e), torch.from_numpy(gt[0, :, :4]))
    gt_sel = gt[:, rng.randint(0, K - 2, size=N), :4]
    enc = bt.bbox_transform_batch(torch.from_numpy(anchors2d), torch.from_numpy(gt_sel))
    np.savez_compressed(os.path.join(HERE, "bbox.npz"), boxes=boxes, deltas=deltas, im_info=im_info,
                        inv=t2n(inv), clipped=t2n(clipped), anchors2d=anchors2d_e, gt=gt,
                        overlaps_batch=t2n(ov), overlaps=t2n(ov2), enc_anchors=anchors2d,
                        enc_gt=gt_sel, enc=t2n(enc))

    # ---------------------------------------------------------------- 3. proposal layer end to end
    scales, ratios = [4, 8, 16, 32], [0.5, 1, 2]
    A = len(scales) * len(ratios)
    layer = _ProposalLayer(16, scales, ratios)
    cases = {}
    for name, (Bp, H, W, key, pre, post, seed) in {
        "test_19x32": (2, 19, 32, "TEST", 6000, 300, 11),
        "train_19x32": (2, 19, 32, "TRAIN", 12000, 2000, 12),
        "test_6x8": (2, 6, 8, "TEST", 6000, 300, 13),      # K*A < pre_nms_topN, guard quirk
        "test_small_pre": (3, 10, 12, "TEST", 500, 50, 14),  # pre < K*A, post < survivors
    }.items():
        r = np.random.RandomState(seed)
        logits = r.normal(0, 2, size=(Bp, 2, A * H, W)).astype(np.float32)
        prob = torch.softmax(torch.from_numpy(logits), 1).view(Bp, 2 * A, H, W)
        bbox = r.normal(0, 0.4, size=(Bp, 4 * A, H, W)).astype(np.float32)
        info = np.tile(np.array([[H * 16.0, W * 16.0, 1.0]], dtype=np.float32), (Bp, 1))
        info[-1, :2] -= 7  # last image slightly smaller -> per-image clipping
        cfg[key].RPN_PRE_NMS_TOP_N = pre
        cfg[key].RPN_POST_NMS_TOP_N = post
        rois = layer((prob, torch.from_numpy(bbox), torch.from_numpy(info), key))
        sflat = t2n(prob[:, A:].permute(0, 2, 3, 1).contiguous().view(Bp, -1))
        nties = int(sum(len(s) - len(np.unique(s)) for s in sflat))
        cases[name + "/cls_prob"] = t2n(prob)
        cases[name + "/bbox_pred"] = bbox
        cases[name + "/im_info"] = info
        cases[name + "/params"] = np.array([16, pre, post], dtype=np.int32)
        cases[name + "/nms_thresh"] = np.array([cfg[key].RPN_NMS_THRESH], dtype=np.float32)
        cases[name + "/rois"] = t2n(rois)
        cases[name + "/score_ties"] = np.array([nties], dtype=np.int32)
    cases["scales"] = np.array(scales)
    cases["ratios"] = np.array(ratios)
    np.savez_compressed(os.path.join(HERE, "proposal.npz"), **cases)

    # ---------------------------------------------------------------- 4. anchor target layer
    atl = _AnchorTargetLayer(16, scales, ratios)
    cases = {}
    for name, (Bp, H, W, seed) in {"b2_19x32": (2, 19, 32, 21), "b3_12x20": (3, 12, 20, 22),
                                   "b2_38x67": (2, 38, 67, 23)}.items():
        r = np.random.RandomState(seed)
        G = 30
        gtb = np.zeros((Bp, G, 5), dtype=np.float32)
        nb = r.randint(1, 6, size=Bp)
        for b in range(Bp):
            x1 = r.uniform(0, W * 16 * 0.6, size=nb[b])
            y1 = r.uniform(0, H * 16 * 0.6, size=nb[b])
            w = r.uniform(32, W * 16 * 0.4, size=nb[b])
            h = r.uniform(32, H * 16 * 0.4, size=nb[b])
            gtb[b, : nb[b], 0] = np.floor(x1)
            gtb[b, : nb[b], 1] = np.floor(y1)
            gtb[b, : nb[b], 2] = np.minimum(np.floor(x1 + w), W * 16 - 1)
            gtb[b, : nb[b], 3] = np.minimum(np.floor(y1 + h), H * 16 - 1)
            gtb[b, : nb[b], 4] = r.randint(1, 31, size=nb[b])
        info = np.tile(np.array([[H * 16.0, W * 16.0, 1.0]], dtype=np.float32), (Bp, 1))
        score = torch.zeros(Bp, 2 * A, H, W)
        np.random.seed(cfg.RNG_SEED)  # trainval_net.py:183
        lab, tgt, inw, outw = atl((score, torch.from_numpy(gtb), torch.from_numpy(info),
                                   torch.from_numpy(nb.astype(np.int64))))
        cases[name + "/gt_boxes"] = gtb
        cases[name + "/im_info"] = info
        cases[name + "/hw"] = np.array([H, W], dtype=np.int32)
        cases[name + "/labels"] = t2n(lab)
        cases[name + "/bbox_targets"] = t2n(tgt)
        cases[name + "/inside"] = t2n(inw)
        cases[name + "/outside"] = t2n(outw)
    cases["scales"] = np.array(scales)
    cases["ratios"] = np.array(ratios)
    cases["rng_seed"] = np.array([cfg.RNG_SEED])
    np.savez_compressed(os.path.join(HERE, "anchor_target.npz"), **cases)
    # ---------------------------------------------------------------- 5. RoI / tracking target samplers
    # proposal_target_layer_cascade.py:130 uses the torch-0.3 `Tensor.index(idx)`; give the harness's torch that
    # method (== advanced indexing) so the reference file runs unmodified.
    torch.Tensor.index = lambda self, idx: self[idx]  # (current torch has an unrelated Tensor.index; harness only)
    from model.rpn.proposal_target_layer_cascade import _ProposalTargetLayer
    from model.rpn.tracking_proposal_target_layer import _TrackingProposalTargetLayer
    cases = {}
    r = np.random.RandomState(31)
    Bp, G, R = 2, 30, 300
    gtb = np.zeros((2, Bp, G, 6), dtype=np.float32)  # (legs, B, G, 6)
    nbx = np.zeros((2, Bp, 1), dtype=np.int64)
    for b in range(Bp):
        n = r.randint(2, 6)
        ids = r.permutation(10)[:n] + 1
        for leg in range(2):
            order = r.permutation(n) if leg == 1 else np.arange(n)
            keep = n if not (leg == 1 and b == 1) else n - 1   # one track disappears in frame t+tau of image 1
            for j, o in enumerate(order[:keep]):
                x1, y1 = r.uniform(0, 600), r.uniform(0, 300)
                w, h = r.uniform(60, 300), r.uniform(60, 250)
                gtb[leg, b, j] = [x1, y1, x1 + w, y1 + h, r.randint(1, 31), ids[o]]
            nbx[leg, b, 0] = keep
    rois = np.zeros((Bp, R, 5), dtype=np.float32)
    xy = r.uniform(0, 700, size=(Bp, R, 2))
    wh = r.uniform(30, 300, size=(Bp, R, 2))
    rois[:, :, 1:3] = xy
    rois[:, :, 3:5] = xy + wh
    for b in range(Bp):  # some proposals close to gt boxes -> foreground
        for j in range(int(nbx[0, b, 0])):
            for k in range(8):
                rois[b, 20 * j + k, 1:5] = gtb[0, b, j, :4] + r.normal(0, 6, size=4)
        rois[b, :, 0] = b
    np.random.seed(cfg.RNG_SEED)
    ptl = _ProposalTargetLayer(31)
    o = ptl(torch.from_numpy(rois), torch.from_numpy(gtb[0][:, :, :5].copy()), torch.from_numpy(nbx[0]))
    for name, t in zip(("rois", "labels", "targets", "inside", "outside"), o):
        cases["pt/" + name] = t2n(t)
    cases["pt/in_rois"] = rois
    cases["gt_boxes"] = gtb
    cases["num_boxes"] = nbx
    ttl = _TrackingProposalTargetLayer(31)
    o = ttl(torch.from_numpy(gtb), torch.from_numpy(nbx))
    for name, t in zip(("rois", "labels", "targets", "inside", "outside"), o):
        cases["tt/" + name] = t2n(t)
    cases["rng_seed"] = np.array([cfg.RNG_SEED])
    np.savez_compressed(os.path.join(HERE, "targets.npz"), **cases)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
