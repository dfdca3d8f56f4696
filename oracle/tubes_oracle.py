"""CPU restatement of the reference's zero-jump Viterbi tube linker -- TEST INFRASTRUCTURE ONLY.

Follows lib/model/utils/tracking_utils.py: `_make_tubes` (:86-124: per-frame NMS 0.3, top max_per_image),
`_zero_jump_link` (:127-264: K-path Viterbi with box removal), `_score_of_edge` (:268-290: pairwise score sum + 1.0
when a tracklet links the two boxes) and `bbox_overlaps` (model/rpn/bbox_transform.py:175-206).  Pinned against the
reference itself: tests/golden/make_golden_tubes.py runs the reference module on CPU tensors and stores its outputs in
tests/golden/tubes.npz (tests/test_oracle_tubes.py).  `smooth_scores` goes through cv2.filter2D in the reference;
cv2 is not in this image, so that single step is pinned only to this file's restatement of OpenCV's default
BORDER_REFLECT_101 correlation.

Declared semantics where the reference leans on unspecified library behaviour: `torch.max(dim)` and the descending
`torch.sort` pick the LOWEST index among equal values (what PyTorch's CPU kernels do, and what the fixtures pin).
"""
import numpy as np

from . import oracle_lib

f32 = np.float32
GAUSS5 = np.array([1.0, 4.0, 6.0, 4.0, 1.0]) / 16.0   # tracking_utils.py:239


def bbox_overlaps(anchors, gt):
    """bbox_transform.py:175-206: (N,4) x (K,4) -> (N,K) float32, +1 widths, no zero-area special cases."""
    a, g = anchors.astype(f32), gt.astype(f32)
    g_area = ((g[:, 2] - g[:, 0] + f32(1)) * (g[:, 3] - g[:, 1] + f32(1)))[None, :]
    a_area = ((a[:, 2] - a[:, 0] + f32(1)) * (a[:, 3] - a[:, 1] + f32(1)))[:, None]
    iw = np.minimum(a[:, None, 2], g[None, :, 2]) - np.maximum(a[:, None, 0], g[None, :, 0]) + f32(1)
    iw[iw < 0] = 0
    ih = np.minimum(a[:, None, 3], g[None, :, 3]) - np.maximum(a[:, None, 1], g[None, :, 1]) + f32(1)
    ih[ih < 0] = 0
    ua = a_area + g_area - iw * ih
    with np.errstate(divide="ignore", invalid="ignore"):
        return (iw * ih / ua).astype(f32)


def score_of_edge(b1, s1, trk1, b2, s2, trk2):
    """tracking_utils.py:268-290.  trk = (boxes in frame t, predicted boxes in frame t+1) of frame t's tracklets."""
    score = (s1.astype(f32)[:, None] + s2.astype(f32)[None, :]).astype(f32)
    if trk1 is not None and trk2 is not None:
        o1 = bbox_overlaps(b1, trk1[0])
        o2 = bbox_overlaps(b2, trk1[1])
        track = (np.rint(o1).astype(f32) @ np.rint(o2).astype(f32).T).astype(f32)   # torch.round: half to even
        score[track > 0] += f32(1.0)
    return score


def filter2d_reflect101(v, kernel=GAUSS5):
    """cv2.filter2D(v, -1, kernel) for a 1-D float32 signal: correlation, anchor at the centre, BORDER_REFLECT_101."""
    v64 = np.asarray(v, dtype=np.float64).reshape(-1)
    n, r = len(v64), len(kernel) // 2
    idx = np.arange(-r, n + r)
    if n == 1:
        idx = np.zeros_like(idx)
    else:
        period = 2 * (n - 1)
        idx = np.mod(idx, period)
        idx = np.where(idx >= n, period - idx, idx)
    p = v64[idx]
    return np.array([np.dot(p[i:i + len(kernel)], kernel) for i in range(n)]).astype(f32)


def zero_jump_link(boxes, scores, tracked):
    """tracking_utils.py:127-264.  boxes / scores / tracked: lists over the F frames; the last frame is unused (None).
    Returns the reference's dict of stacked arrays."""
    F = len(boxes)
    T = F - 1
    boxes = [None if b is None else b.astype(f32).copy() for b in boxes]
    scores = [None if s is None else s.astype(f32).copy() for s in scores]
    idxs = [None if b is None else np.arange(len(b)) for b in boxes]
    for i in range(T):
        if boxes[i] is None or len(boxes[i]) == 0:
            raise RuntimeError("empty frame %d" % i)                    # :166-169
    out = {"total_score": [], "boxes": [], "idx": [], "smooth_scores": [], "scores": []}
    empty = False
    while not empty:
        data_scores = [np.zeros(len(boxes[i]), f32) for i in range(T)]
        data_index = [None] * T
        for i in range(T - 2, -1, -1):                                  # :207-218
            edge = score_of_edge(boxes[i], scores[i], tracked[i], boxes[i + 1], scores[i + 1], tracked[i + 1])
            edge = (edge + data_scores[i + 1][None, :]).astype(f32)
            data_index[i] = np.argmax(edge, axis=1)                     # first maximum
            data_scores[i] = edge[np.arange(edge.shape[0]), data_index[i]]
        cur = int(np.argmax(data_scores[0]))                            # sort descending, take the first (:222-223)
        score = data_scores[0][cur]
        p_idx, p_box, p_sc = [idxs[0][cur]], [boxes[0][cur]], [scores[0][cur]]
        for j in range(T - 1):                                          # :228-232
            cur = int(data_index[j][cur])
            p_idx.append(idxs[j + 1][cur]); p_box.append(boxes[j + 1][cur]); p_sc.append(scores[j + 1][cur])
        p_sc = np.array(p_sc, f32)
        out["total_score"].append(np.array([[f32(score) / f32(F)]], f32))   # score / num_frames (:233)
        out["idx"].append(np.array(p_idx, np.int64))
        out["boxes"].append(np.concatenate([np.array(p_box, f32), p_sc[:, None]], 1))
        top = np.sort(p_sc)[::-1]
        mean_top = f32(np.mean(top[:int(np.ceil(0.5 * len(top)))], dtype=f32))   # :237
        out["smooth_scores"].append((filter2d_reflect101(p_sc) + mean_top).astype(f32))
        out["scores"].append((p_sc + mean_top).astype(f32))
        for j in range(T):                                              # :248-258 remove the covered boxes
            keep = np.nonzero(idxs[j] != p_idx[j])[0]
            if len(keep) == 0:
                empty = True
                continue
            boxes[j], scores[j], idxs[j] = boxes[j][keep], scores[j][keep], idxs[j][keep]
    return {k: np.stack(v) for k, v in out.items()}


def make_tubes(dets, n, trk=None, m=None, max_per_image=25, nms_thresh=0.3):
    """tracking_utils.py:86-124.  dets (F, Nmax, >=5) rows [x1,y1,x2,y2,score,...] in priority order, n (F,) row counts;
    trk (F, 2, Mmax, 4) tracklet boxes with m (F,) counts (-1: the frame has none); trk=None: no tracking term."""
    F = len(n)
    boxes, scores, tracked = [None] * F, [None] * F, [None] * F
    for f in range(F - 1):
        d = np.asarray(dets[f][:n[f]], f32)
        keep = oracle_lib.nms(np.ascontiguousarray(d[:, :5]), float(nms_thresh)).astype(np.int64).reshape(-1)
        keep = keep[:max_per_image]
        boxes[f], scores[f] = d[keep, :4], d[keep, 4]
        if trk is not None and m[f] >= 0:
            tracked[f] = (np.asarray(trk[f, 0, :m[f]], f32), np.asarray(trk[f, 1, :m[f]], f32))
    return zero_jump_link(boxes, scores, tracked)


# ---------------------------------------------------------------------------------------------------------------------
# Front end of VideoPostProcessor: tracking_utils.py:19-52 (__init__), 320-386 (_process_frame_pairs), 294-318
# (_keep_top_k) and 54-84 (build_class_paths).  Pinned by the "video/*" entries of tests/golden/tubes.npz.
def process_frame_pairs(pred_boxes, scores, pred_trk_boxes, max_per_image=400):
    """tracking_utils.py:320-386.  pred_boxes (P,2,R,>=4), scores (P,2,R,C), pred_trk_boxes (P,R,4).
    Returns aboxes[c][f] (list of (n,6) arrays or None), tracks[f] = (boxes_t, boxes_t+1) or None."""
    P, _, R, C = scores.shape
    F = P + 1
    aboxes = [[None] * F for _ in range(C)]
    tracks = [None] * F
    for i_pair in range(P - 1):                                        # sic: the last pair is never read (:322)
        s0 = scores[i_pair, 0]
        tracklets = np.nonzero(s0[:, 1:].max(axis=1) > f32(0.01))[0]   # :325-326
        if tracklets.size:
            tracks[i_pair] = (pred_boxes[i_pair, 0][tracklets][:, :4].astype(f32), pred_trk_boxes[i_pair][tracklets].astype(f32))
        for i_leg in range(2):
            f = i_pair + i_leg
            boxes, sc = pred_boxes[i_pair, i_leg], scores[i_pair, i_leg]
            for c in range(1, C):
                order = np.argsort(-sc[:, c], kind="stable")[:max_per_image]          # CONF_THRESH starts at -inf: all rows
                entry = np.concatenate([boxes[order][:, :4], sc[order, c][:, None], sc[order, 0][:, None]], 1).astype(f32)
                aboxes[c][f] = entry if aboxes[c][f] is None else np.concatenate([aboxes[c][f], entry], 0)
    return aboxes, tracks


def keep_top_k(frames, top_k):
    """tracking_utils.py:294-318 for one class: threshold = the (top_k + 1)-th best score of the whole video; frames
    keep their rows >= threshold (a frame with none keeps everything).  The reference indexes scores[min(n, top_k)],
    which is out of range unless the class has MORE than top_k detections -- reproduced as IndexError."""
    lst = [b for b in frames if b is not None]
    if not lst:
        return frames, None
    allsc = np.sort(np.concatenate(lst, 0)[:, 4])[::-1]
    if min(allsc.size, top_k) >= allsc.size:
        raise IndexError("index %d is out of bounds for dimension 0 with size %d" % (min(allsc.size, top_k), allsc.size))
    thresh = allsc[min(allsc.size, top_k)]
    out = []
    for b in frames:
        if b is not None and len(b):
            keep = np.nonzero(b[:, 4] >= thresh)[0]
            out.append(b[keep] if keep.size else b)
        else:
            out.append(b)
    return out, thresh


def build_class_paths(pred_boxes, scores, pred_trk_boxes, max_per_image=25):
    """VideoPostProcessor(pred_boxes, scores, pred_trk_boxes, classes).build_class_paths(): dict class -> paths dict."""
    P, _, R, C = scores.shape
    F = P + 1
    aboxes, tracks = process_frame_pairs(pred_boxes, scores, pred_trk_boxes)
    paths, thresh = {}, np.full(C, -np.inf, f32)
    for c in range(1, C):
        aboxes[c], t = keep_top_k(aboxes[c], 160 * F)
        if t is not None:
            thresh[c] = t
    for c in range(1, C):
        frames = [f for f in range(F) if aboxes[c][f] is not None and len(aboxes[c][f])]
        if not frames:
            continue
        fb = [aboxes[c][f] for f in frames]
        nmax = max(len(b) for b in fb)
        dets = np.zeros((len(fb), nmax, 6), f32)
        n = np.zeros(len(fb), np.int32)
        for i, b in enumerate(fb):
            dets[i, :len(b)], n[i] = b, len(b)
        mmax = max([len(tracks[f][0]) for f in frames if tracks[f] is not None] + [1])
        trk = np.zeros((len(fb), 2, mmax, 4), f32)
        m = np.full(len(fb), -1, np.int32)
        for i, f in enumerate(frames):
            if tracks[f] is not None:
                m[i] = len(tracks[f][0])
                trk[i, 0, :m[i]], trk[i, 1, :m[i]] = tracks[f]
        paths[c] = make_tubes(dets, n, trk, m, max_per_image=max_per_image)
    return paths, aboxes, thresh
