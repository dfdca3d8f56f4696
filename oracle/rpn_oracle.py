"""numpy restatement of the reference's RPN-side Python (lib/model/rpn/*).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package never does.

Pinned by executing the reference itself (the CUDA ops of oracle/dtt_oracle.c are pinned against the
reference's own kernels, oracle/build_ref.sh): the reference modules restated here can be imported in the
build container, and tests/golden/make_golden.py runs them there to produce the fixtures in tests/golden/*.npz that tests/test_oracle_rpn.py checks this
file against.

Declared semantics where the reference leans on unspecified library behaviour:
  * sort order (proposal_layer.py:125 ``torch.sort(scores, 1, True)``): descending score, ties broken
    by the LOWER flattened anchor index first (stable).
  * exp / log (bbox_transform.py:121-122, 28-29): correctly rounded binary32, computed as
    float32(f(float64(x))).  torch's own float32 exp/log may differ from this by 1 ulp.
All other arithmetic is IEEE binary32 with one rounding per operation, in the reference's operation
order (numpy float32 arrays give exactly that).
"""
import numpy as np

f32 = np.float32


# ------------------------------------------------------------------------------------------ anchors
def _whctrs(box):
    # generate_anchors.py:58-67
    w = box[2] - box[0] + 1
    h = box[3] - box[1] + 1
    return w, h, box[0] + 0.5 * (w - 1), box[1] + 0.5 * (h - 1)


def _mkanchors(ws, hs, xc, yc):
    # generate_anchors.py:69-81
    ws = np.asarray(ws, dtype=np.float64)[:, None]
    hs = np.asarray(hs, dtype=np.float64)[:, None]
    return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """generate_anchors.py:45-56 -- ratio-major, then scale; float64, 0-based window (0,0,15,15)."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    base = np.array([1, 1, base_size, base_size], dtype=np.float64) - 1
    w, h, xc, yc = _whctrs(base)
    size_ratios = (w * h) / ratios  # generate_anchors.py:88-92 (np.round = half to even)
    ws = np.round(np.sqrt(size_ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = _mkanchors(ws, hs, xc, yc)
    out = []
    for r in ratio_anchors:  # generate_anchors.py:96-105
        w, h, xc, yc = _whctrs(r)
        out.append(_mkanchors(w * scales, h * scales, xc, yc))
    return np.vstack(out)


def shifted_anchors(base_anchors, height, width, feat_stride):
    """proposal_layer.py:80-93 / anchor_target_layer.py:67-79: all[k*A + a] = base[a] + shift[k],
    k = h*W + w, shift = (16w, 16h, 16w, 16h)."""
    base = np.asarray(base_anchors, dtype=f32)
    sx, sy = np.meshgrid(np.arange(width) * feat_stride, np.arange(height) * feat_stride)
    shifts = np.stack([sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()], 1).astype(f32)
    return (base[None, :, :] + shifts[:, None, :]).reshape(-1, 4)


# ------------------------------------------------------------------------------------- box algebra
def _exp32(x):
    return np.exp(x.astype(np.float64)).astype(f32)


def _log32(x):
    return np.log(x.astype(np.float64)).astype(f32)


def bbox_transform_inv(boxes, deltas):
    """bbox_transform.py:108-134.  boxes (..., 4), deltas (..., 4) float32 -> (..., 4)."""
    boxes = boxes.astype(f32)
    deltas = deltas.astype(f32)
    widths = boxes[..., 2] - boxes[..., 0] + f32(1.0)
    heights = boxes[..., 3] - boxes[..., 1] + f32(1.0)
    ctr_x = boxes[..., 0] + f32(0.5) * widths
    ctr_y = boxes[..., 1] + f32(0.5) * heights
    dx, dy, dw, dh = (deltas[..., i] for i in range(4))
    pcx = dx * widths + ctr_x
    pcy = dy * heights + ctr_y
    pw = _exp32(dw) * widths
    ph = _exp32(dh) * heights
    out = np.empty_like(deltas)
    out[..., 0] = pcx - f32(0.5) * pw
    out[..., 1] = pcy - f32(0.5) * ph
    out[..., 2] = pcx + f32(0.5) * pw
    out[..., 3] = pcy + f32(0.5) * ph
    return out


def clip_boxes(boxes, im_info):
    """bbox_transform.py:156-173 (3-D branch): clamp to [0, w-1] x [0, h-1], im_info[i]=[h,w,scale]."""
    boxes = boxes.copy()
    for i in range(boxes.shape[0]):
        wmax = f32(im_info[i, 1]) - f32(1)
        hmax = f32(im_info[i, 0]) - f32(1)
        boxes[i, :, 0] = np.clip(boxes[i, :, 0], f32(0), wmax)
        boxes[i, :, 1] = np.clip(boxes[i, :, 1], f32(0), hmax)
        boxes[i, :, 2] = np.clip(boxes[i, :, 2], f32(0), wmax)
        boxes[i, :, 3] = np.clip(boxes[i, :, 3], f32(0), hmax)
    return boxes


def bbox_overlaps_batch(anchors, gt_boxes):
    """bbox_transform.py:208-254 (2-D anchors branch): (N,4) x (B,K,>=4) -> (B,N,K) float32."""
    anchors = anchors.astype(f32)
    gt = gt_boxes[:, :, :4].astype(f32)
    gx = gt[:, :, 2] - gt[:, :, 0] + f32(1)
    gy = gt[:, :, 3] - gt[:, :, 1] + f32(1)
    g_area = (gx * gy)[:, None, :]
    ax = anchors[:, 2] - anchors[:, 0] + f32(1)
    ay = anchors[:, 3] - anchors[:, 1] + f32(1)
    a_area = (ax * ay)[None, :, None]
    g_zero = (gx == 1) & (gy == 1)
    a_zero = (ax == 1) & (ay == 1)
    b = anchors[None, :, None, :]
    q = gt[:, None, :, :]
    iw = np.minimum(b[..., 2], q[..., 2]) - np.maximum(b[..., 0], q[..., 0]) + f32(1)
    iw[iw < 0] = 0
    ih = np.minimum(b[..., 3], q[..., 3]) - np.maximum(b[..., 1], q[..., 1]) + f32(1)
    ih[ih < 0] = 0
    ua = a_area + g_area - (iw * ih)
    ov = (iw * ih / ua).astype(f32)
    ov[np.broadcast_to(g_zero[:, None, :], ov.shape)] = 0
    ov[np.broadcast_to(a_zero[None, :, None], ov.shape)] = -1
    return ov


def bbox_transform_batch(ex_rois, gt_rois):
    """bbox_transform.py:36-75 (2-D ex_rois branch): (N,4), (B,N,4) -> (B,N,4)."""
    ex = ex_rois.astype(f32)
    gt = gt_rois.astype(f32)
    ew = ex[:, 2] - ex[:, 0] + f32(1.0)
    eh = ex[:, 3] - ex[:, 1] + f32(1.0)
    ecx = ex[:, 0] + f32(0.5) * ew
    ecy = ex[:, 1] + f32(0.5) * eh
    gw = gt[:, :, 2] - gt[:, :, 0] + f32(1.0)
    gh = gt[:, :, 3] - gt[:, :, 1] + f32(1.0)
    gcx = gt[:, :, 0] + f32(0.5) * gw
    gcy = gt[:, :, 1] + f32(0.5) * gh
    dx = (gcx - ecx[None, :]) / ew
    dy = (gcy - ecy[None, :]) / eh
    dw = _log32(gw / ew[None, :])
    dh = _log32(gh / eh[None, :])
    return np.stack([dx, dy, dw, dh], 2).astype(f32)


# ---------------------------------------------------------------------------------- proposal layer
def sort_desc_stable(scores):
    """Declared order for proposal_layer.py:125: descending, lower index first on ties."""
    return np.argsort(-scores.astype(f32), axis=1, kind="stable")


def proposal_layer(cls_prob, bbox_pred, im_info, base_anchors, feat_stride, pre_nms_topN,
                   post_nms_topN, nms_thresh, nms_fn):
    """proposal_layer.py:49-161.  nms_fn(dets (N,5) float32, thresh) -> int keep indices (oracle NMS).
    Returns (rois (B, post_nms_topN, 5), num_valid (B,))."""
    cls_prob = np.asarray(cls_prob, dtype=f32)
    bbox_pred = np.asarray(bbox_pred, dtype=f32)
    B = bbox_pred.shape[0]
    A = np.asarray(base_anchors).shape[0]
    H, W = cls_prob.shape[2], cls_prob.shape[3]
    scores = cls_prob[:, A:, :, :]  # fg probabilities, proposal_layer.py:67
    anchors = shifted_anchors(base_anchors, H, W, feat_stride)  # (K*A, 4)
    deltas = bbox_pred.transpose(0, 2, 3, 1).reshape(B, -1, 4)  # proposal_layer.py:98-99
    scores = scores.transpose(0, 2, 3, 1).reshape(B, -1)  # proposal_layer.py:102-103
    proposals = bbox_transform_inv(np.broadcast_to(anchors[None], deltas.shape), deltas)
    proposals = clip_boxes(proposals, np.asarray(im_info, dtype=f32))
    order = sort_desc_stable(scores)
    out = np.zeros((B, post_nms_topN, 5), dtype=f32)
    nvalid = np.zeros((B,), dtype=np.int32)
    for i in range(B):
        o = order[i]
        if 0 < pre_nms_topN < scores.size:  # proposal_layer.py:138-139 (numel of the WHOLE batch)
            o = o[:pre_nms_topN]
        p = proposals[i][o]
        s = scores[i][o].reshape(-1, 1)
        keep = np.asarray(nms_fn(np.hstack([p, s]).astype(f32), nms_thresh), dtype=np.int64).reshape(-1)
        if post_nms_topN > 0:
            keep = keep[:post_nms_topN]
        p = p[keep]
        out[i, :, 0] = i
        out[i, : p.shape[0], 1:] = p
        nvalid[i] = p.shape[0]
    return out, nvalid


# ----------------------------------------------------------------------------- anchor target layer
def anchor_target_assign(gt_boxes, im_info, base_anchors, height, width, feat_stride,
                         negative_overlap=0.3, positive_overlap=0.7, clobber_positives=False):
    """anchor_target_layer.py:58-116, up to (not including) the random subsampling.
    Returns dict(inds_inside, anchors, labels (B, N_in) in {1,0,-1}, argmax (B, N_in))."""
    gt_boxes = np.asarray(gt_boxes, dtype=f32)
    B = gt_boxes.shape[0]
    all_anchors = shifted_anchors(base_anchors, height, width, feat_stride)
    im_h = int(im_info[0][0])  # long(im_info[0][0]) -- image 0 only, anchor_target_layer.py:85-86
    im_w = int(im_info[0][1])
    keep = ((all_anchors[:, 0] >= 0) & (all_anchors[:, 1] >= 0)
            & (all_anchors[:, 2] < im_w) & (all_anchors[:, 3] < im_h))
    inds_inside = np.nonzero(keep)[0]
    anchors = all_anchors[inds_inside]
    labels = np.full((B, inds_inside.size), -1, dtype=f32)
    overlaps = bbox_overlaps_batch(anchors, gt_boxes[:, :, :5])
    max_overlaps = overlaps.max(2)
    argmax_overlaps = overlaps.argmax(2)  # first maximal index, as torch.max on CPU
    gt_max = overlaps.max(1)
    if not clobber_positives:
        labels[max_overlaps < f32(negative_overlap)] = 0
    gt_max = gt_max.copy()
    gt_max[gt_max == 0] = f32(1e-5)
    keep_n = (overlaps == gt_max[:, None, :]).sum(2)
    if keep_n.sum() > 0:
        labels[keep_n > 0] = 1
    labels[max_overlaps >= f32(positive_overlap)] = 1
    if clobber_positives:
        labels[max_overlaps < f32(negative_overlap)] = 0
    return dict(inds_inside=inds_inside, anchors=anchors, labels=labels, argmax=argmax_overlaps,
                total=all_anchors.shape[0])


def anchor_target_subsample(labels, rpn_batchsize=256, fg_fraction=0.5, rng=np.random):
    """anchor_target_layer.py:118-141: consumes np.random.permutation exactly as the reference does
    (fg then bg, image by image, only when over quota)."""
    labels = labels.copy()
    num_fg = int(fg_fraction * rpn_batchsize)
    sum_fg = (labels == 1).sum(1)
    sum_bg = (labels == 0).sum(1)
    for i in range(labels.shape[0]):
        if sum_fg[i] > num_fg:
            fg_inds = np.nonzero(labels[i] == 1)[0]
            perm = rng.permutation(fg_inds.size)
            labels[i, fg_inds[perm[: fg_inds.size - num_fg]]] = -1
        num_bg = rpn_batchsize - sum_fg[i]  # uses the PRE-subsampling fg count (reference quirk)
        if sum_bg[i] > num_bg:
            bg_inds = np.nonzero(labels[i] == 0)[0]
            perm = rng.permutation(bg_inds.size)
            labels[i, bg_inds[perm[: bg_inds.size - num_bg]]] = -1
    return labels


def anchor_target_finish(assign, labels, gt_boxes, num_anchors, height, width, inside_weight=1.0):
    """anchor_target_layer.py:142-189: targets, weights, _unmap and the four output layouts."""
    gt_boxes = np.asarray(gt_boxes, dtype=f32)
    B = gt_boxes.shape[0]
    A = num_anchors
    inds = assign["inds_inside"]
    total = assign["total"]
    argmax = assign["argmax"]
    gt_sel = np.stack([gt_boxes[b, argmax[b], :4] for b in range(B)], 0)
    targets = bbox_transform_batch(assign["anchors"], gt_sel)
    inside = np.zeros(labels.shape, dtype=f32)
    inside[labels == 1] = f32(inside_weight)
    num_examples = int((labels[B - 1] >= 0).sum())  # LAST image only (reference quirk, :154)
    w = f32(1.0) / f32(num_examples) if num_examples > 0 else f32(np.inf)
    outside = np.zeros(labels.shape, dtype=f32)
    outside[labels == 1] = w
    outside[labels == 0] = w

    def unmap(data, fill):
        if data.ndim == 2:
            ret = np.full((B, total), fill, dtype=f32)
            ret[:, inds] = data
        else:
            ret = np.full((B, total, data.shape[2]), fill, dtype=f32)
            ret[:, inds, :] = data
        return ret

    labels_u = unmap(labels, -1)
    targets_u = unmap(targets, 0)
    inside_u = unmap(inside, 0)
    outside_u = unmap(outside, 0)
    labels_o = labels_u.reshape(B, height, width, A).transpose(0, 3, 1, 2).reshape(B, 1, A * height, width)
    targets_o = targets_u.reshape(B, height, width, A * 4).transpose(0, 3, 1, 2)
    inside_o = np.repeat(inside_u[:, :, None], 4, 2).reshape(B, height, width, 4 * A).transpose(0, 3, 1, 2)
    outside_o = np.repeat(outside_u[:, :, None], 4, 2).reshape(B, height, width, 4 * A).transpose(0, 3, 1, 2)
    return (np.ascontiguousarray(labels_o), np.ascontiguousarray(targets_o),
            np.ascontiguousarray(inside_o), np.ascontiguousarray(outside_o))


def anchor_target_layer(gt_boxes, im_info, base_anchors, height, width, feat_stride,
                        rpn_batchsize=256, fg_fraction=0.5, negative_overlap=0.3,
                        positive_overlap=0.7, clobber_positives=False, inside_weight=1.0,
                        rng=np.random):
    """anchor_target_layer.py:48-191 end to end."""
    a = anchor_target_assign(gt_boxes, im_info, base_anchors, height, width, feat_stride,
                             negative_overlap, positive_overlap, clobber_positives)
    labels = anchor_target_subsample(a["labels"], rpn_batchsize, fg_fraction, rng)
    return anchor_target_finish(a, labels, gt_boxes, np.asarray(base_anchors).shape[0], height, width,
                                inside_weight)


# ------------------------------------------------------------------------ test-time per-class NMS
def class_nms(scores, boxes, nms_fn, score_thresh=0.05, nms_thresh=0.3, max_per_image=100, class_agnostic=True):
    """test_net.py:274-301 for ONE image: scores (R, ncls), boxes (R, 4 | 4*ncls) -> list over classes of
    (n, 5) arrays.  Sort order: descending score, ties by lower RoI index (declared; torch.sort is unspecified)."""
    scores = np.asarray(scores, dtype=f32)
    boxes = np.asarray(boxes, dtype=f32)
    ncls = scores.shape[1]
    out = [np.zeros((0, 5), f32)]
    for j in range(1, ncls):
        inds = np.nonzero(scores[:, j] > f32(score_thresh))[0]
        if inds.size == 0:
            out.append(np.zeros((0, 5), f32))
            continue
        s = scores[inds, j]
        order = np.argsort(-s, kind="stable")
        b = boxes[inds] if class_agnostic else boxes[inds][:, 4 * j:4 * j + 4]
        dets = np.hstack([b, s[:, None]]).astype(f32)[order]
        keep = np.asarray(nms_fn(dets, nms_thresh), dtype=np.int64).reshape(-1)
        out.append(dets[keep])
    if max_per_image > 0:
        allscores = np.hstack([o[:, -1] for o in out[1:]])
        if allscores.size > max_per_image:
            th = np.sort(allscores)[-max_per_image]
            out = [o[o[:, -1] >= th] if k > 0 else o for k, o in enumerate(out)]
    return out
