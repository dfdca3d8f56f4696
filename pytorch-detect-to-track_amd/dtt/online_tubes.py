"""Online (incremental) tube linking + temporal class labelling of the video demo
(reference: lib/model/utils/online_tubes.py:20-550, `VideoPostProcessor`; used by demo.py:487-489).

Same constructor, `class_paths()` result and attributes as the reference class.  What is organised differently:

  * the per-frame, per-class candidate selection (score > 0, 50 best, NMS at 0.3, 10 best: online_tubes.py:182-223) runs
    for ALL frame pairs and classes at once as batched tensor ops on the device the detections live on (one stable sort,
    one batched IoU, a 50-step vectorised greedy sweep) instead of P x C python iterations with an NMS round trip each;
    the IoU is the NMS kernel's arithmetic, one rounding per operation, so the kept sets are those of `nms`;
  * the linker itself is sequential in time and works on <= 10 boxes per frame: it runs on the host, on CPU tensors.

The reference's behaviour is kept where it is visible in the output, including its quirks: a path is kept or retired
according to the staleness counter at its UNSORTED position (online_tubes.py:427-430); live paths are ranked by the mean of
their five most recent scores (:396-398 sorts along a size-1 dimension); paths retired before the last
frame pair are dropped (the dead list is rebuilt every frame, :414-420); a gap of d frames is filled with d copies of the
box after the gap (:311-317); `pred_trk_boxes` is stored and never read (:26).
"""
from collections import deque

import torch

JUMPGAP = 5       # frames a path may go unmatched before it is retired (online_tubes.py:32)
ALPHA_L = 3.0     # label-switch penalty of the temporal labelling (online_tubes.py:33)
PRE_NMS_TOP, POST_NMS_TOP, NMS_THRESH, LINK_IOU = 50, 10, 0.3, 0.1


def _pairwise_iou_nms(b):
    """(..., K, 4) -> (..., K, K) IoU with the NMS kernel's arithmetic (nms_cuda_kernel.cu:31-39)."""
    x1, y1, x2, y2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    w = (torch.minimum(x2[..., :, None], x2[..., None, :]) - torch.maximum(x1[..., :, None], x1[..., None, :]) + 1).clamp_(min=0)
    h = (torch.minimum(y2[..., :, None], y2[..., None, :]) - torch.maximum(y1[..., :, None], y1[..., None, :]) + 1).clamp_(min=0)
    inter = w * h
    return inter / (area[..., :, None] + area[..., None, :] - inter)


def select_candidates(boxes, scores):
    """boxes (P, R, 4), scores (P, R, C) of the first frame of every pair -> per (pair, class >= 1) the RoI indices of
    the <= 10 boxes the reference's per-frame selection keeps, in its order, and their number.
    Returns idx (P, C, 10) int64 (padding -1) and count (P, C)."""
    P, R, C = scores.shape
    K = min(PRE_NMS_TOP, R)
    s = scores.permute(0, 2, 1)                                           # (P, C, R)
    order = torch.sort(s, dim=2, descending=True, stable=True)[1][..., :K]  # ties: lower RoI index first
    top = torch.gather(s, 2, order)
    valid = top > 0.0
    b = boxes[:, None, :, :].expand(P, C, R, 4).gather(2, order[..., None].expand(P, C, K, 4))
    over = _pairwise_iou_nms(b) > NMS_THRESH
    keep = torch.zeros_like(valid)
    dead = ~valid
    for i in range(K):  # greedy sweep, all (pair, class) problems in lockstep
        k_i = ~dead[..., i]
        keep[..., i] = k_i
        dead = dead | (over[..., i, :] & k_i[..., None])
    rank = torch.cumsum(keep.long(), dim=2) - 1
    take = keep & (rank < POST_NMS_TOP)
    idx = torch.full((P, C, POST_NMS_TOP), -1, dtype=torch.long, device=scores.device)
    pc = torch.nonzero(take, as_tuple=True)
    idx[pc[0], pc[1], rank[pc]] = order[pc]
    return idx, take.sum(dim=2)


def _overlaps(a, b):
    """online_tubes.py:228-258 (the class's own IoU: +1 widths, clamp at 0), a (N,4) x b (K,4) -> (N,K)."""
    area_b = ((b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)).view(1, -1)
    area_a = ((a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)).view(-1, 1)
    iw = (torch.min(a[:, None, 2], b[None, :, 2]) - torch.max(a[:, None, 0], b[None, :, 0]) + 1).clamp(min=0)
    ih = (torch.min(a[:, None, 3], b[None, :, 3]) - torch.max(a[:, None, 1], b[None, :, 1]) + 1).clamp(min=0)
    return iw * ih / (area_a + area_b - iw * ih)


class _Path(object):
    __slots__ = ("boxes", "scores", "all_scores", "path_score", "found_at", "count", "last_found")

    def __init__(self, box, score, all_score, t):
        self.boxes, self.scores, self.all_scores = [box], [score], [all_score]
        self.path_score = score.clone().view(1, 1)
        self.found_at, self.count, self.last_found = [t], 1, 0


def _fill_gaps(paths):
    """online_tubes.py:261-323: keep paths seen in more than JUMPGAP frames; repeat the box after a gap of d frames d
    times.  -> list of dicts with stacked tensors."""
    out = []
    for p in paths:
        if len(p.found_at) <= JUMPGAP:
            continue
        boxes, scores, alls = [], [], []
        for i in range(len(p.scores)):
            d = p.found_at[i] - p.found_at[max(0, i - 1)]
            reps = 1 if (i == 0 or d == 1) else d
            boxes += [p.boxes[i]] * reps; scores += [p.scores[i]] * reps; alls += [p.all_scores[i]] * reps
        out.append({"start": torch.tensor([p.found_at[0]]), "end": torch.tensor([p.found_at[-1]]),
                    "boxes": torch.stack(boxes, 0), "scores": torch.stack(scores, 0).view(-1, 1),
                    "all_scores": torch.stack(alls, 0), "path_score": p.path_score.clone(),
                    "found_at": torch.tensor(p.found_at).view(-1, 1), "count": [p.count], "last_found": [p.last_found]})
    return out


def incremental_linking(frames_boxes, frames_scores, frames_all_scores):
    """online_tubes.py:326-550 for one class.  frames_*[t]: (n_t, 4), (n_t,), (n_t, C) CPU tensors of frame pair t
    (n_t >= 1).  Returns the reference's dict of lists (start, end, boxes, scores, all_scores, path_score, found_at,
    count, last_found), best path first."""
    live, dead = [], []
    for t, (fb, fs, fa) in enumerate(zip(frames_boxes, frames_scores, frames_all_scores)):
        n = fb.size(0)
        assert n > 0, "Must have boxes for class to build tubes. Check your filter threshold."
        if t == 0:
            live = [_Path(fb[b], fs[b], fa[b], 0) for b in range(n)]
            continue
        last = torch.stack([p.boxes[-1] for p in live], 0)
        edge = fs.view(1, n).expand(len(live), n) * (_overlaps(last, fb) > LINK_IOU).float()
        edge = edge.clone()
        covered = torch.zeros(n, dtype=torch.bool)
        order_score = torch.zeros(1, len(live))
        for lp, p in enumerate(live):
            if p.last_found >= JUMPGAP:
                continue
            row = edge[lp]
            if row.sum() > 0:
                m_score, j = row.max(0)
                j = int(j)
                p.count += 1
                p.boxes.append(fb[j]); p.scores.append(fs[j]); p.all_scores.append(fa[j])
                p.path_score = p.path_score + m_score
                p.found_at.append(t)
                p.last_found = 0
                edge[:, j] = 0.0
                covered[j] = True
            else:
                p.last_found += 1
            # mean of the JUMPGAP most recent scores: the reference sorts an (n, 1) tensor along its last dimension
            # (a no-op, online_tubes.py:396-398), so "the best five" is really "the last five"
            order_score[0, lp] = torch.stack(p.scores[-JUMPGAP:]).mean()
        inds = torch.sort(order_score, descending=True)[1].view(-1).tolist()
        stale = [p.last_found >= JUMPGAP for p in live]      # read at the UNSORTED position, as the reference does
        new_live, dead = [], []
        for pos, olp in enumerate(inds):
            (dead if stale[pos] else new_live).append(live[olp])
        new_live += [_Path(fb[b], fs[b], fa[b], t) for b in range(n) if not covered[b]]
        live = new_live
    paths = _fill_gaps(live) + _fill_gaps(dead)
    score = torch.zeros(len(paths))
    for i, p in enumerate(paths):
        s = torch.sort(p["scores"].view(-1), descending=True)[0]
        score[i] = s[:min(20, s.numel())].mean()
    order = torch.sort(score, descending=True)[1].tolist()
    keys = ("start", "end", "boxes", "scores", "all_scores", "path_score", "found_at", "count", "last_found")
    return {k: [paths[i][k] for i in order] for k in keys}


def dpEM_max(M, alpha_l=ALPHA_L):
    """online_tubes.py:53-92: Viterbi labelling of one path.  M (frames, classes-1) class scores along the path; a label
    switch between consecutive frames costs alpha_l.  Returns (labels + 1 per frame, frame indices, cumulative scores D)."""
    M = M.t()
    r, c = M.shape
    D = torch.zeros(r, c + 1)
    D[:, 1:] = M
    phi = torch.zeros(r, c, dtype=torch.long)
    switch = alpha_l * (1.0 - torch.eye(r))              # switch[i, k] = alpha_l * (k != i)
    for j in range(1, c + 1):
        best, arg = torch.max(D[:, j - 1].view(1, r) - switch, dim=1)
        D[:, j] = D[:, j] + best
        phi[:, j - 1] = arg
    D = D[:, 1:]
    i = int(torch.max(D[:, -1], dim=0)[1])
    labels, frames = deque([i + 1]), deque([c - 1])
    for j in range(c - 1, 0, -1):
        i = int(phi[i, j])
        labels.appendleft(i + 1)
        frames.appendleft(j - 1)
    return torch.tensor(list(labels), dtype=torch.float32), torch.tensor(list(frames), dtype=torch.float32), D


def extract_action(p, q, D, action):
    """online_tubes.py:96-131: the runs of frames labelled `action` along one path -> (starts, ends, mean score gain per
    run (n,1), label (n,1), path score (n,1)); five empty tensors when the label never occurs."""
    inds = torch.nonzero(p == action)
    if inds.numel() == 0:
        e = torch.zeros(0)
        return e, e, e, e, e
    diff = (torch.cat([inds, (inds[-1] + 1).view(-1, 1)], 0) - torch.cat([(inds[0] - 2).view(-1, 1), inds], 0)).view(-1)
    ts = torch.nonzero(diff > 1).view(-1)
    inds = inds.view(-1)
    te = torch.cat([ts[1:] - 1, torch.tensor([inds.size(0) - 1])]) if ts.numel() > 1 else torch.tensor([inds.size(0) - 1])
    ts, te = inds[ts], inds[te]
    q_s, q_e = q[ts].long(), q[te].long()
    # The reference indexes the (classes - 1)-row table with 1-based class ids (online_tubes.py:121-127): rows are off
    # by one, and the LAST class raises IndexError there.  The indexing is kept (same numbers wherever the reference
    # produces any); only the out-of-range row is clamped so that a video containing the last class does not abort.
    top = D.size(0) - 1
    row = D[min(action, top)]
    scores = ((row[q_e] - row[q_s]) / ((te - ts).float() + 1e-6)).view(-1, 1)
    label = torch.full((ts.size(0), 1), float(action))
    total = torch.ones(ts.size(0), 1) * D[min(int(p[-1]), top), int(q[-1])] / p.size(0)
    return ts, te, scores, label, total


class VideoPostProcessor(object):
    """online_tubes.py:20-51.  pred_boxes (P, 2, R, 4) class-agnostic boxes of both frames of every pair, scores
    (P, 2, R, C), pred_trk_boxes (stored, unused), classes (C names, background first)."""

    def __init__(self, pred_boxes, scores, pred_trk_boxes, classes, video_id=""):
        print("Starting post-processing on video id {}".format(video_id))
        self.video_id = video_id
        self.pred_boxes, self.scores, self.pred_trk_boxes = pred_boxes, scores, pred_trk_boxes
        self.num_frame_pairs = pred_boxes.size(0)
        self.num_frames = self.num_frame_pairs + 1
        self.classes, self.num_classes = classes, len(classes)
        self.class_agnostic = True
        self.jumpgap, self.alpha_l = JUMPGAP, ALPHA_L
        self.all_paths = [None] * self.num_classes

    def generate_paths(self):
        """online_tubes.py:182-223: candidates of every (pair, class) in one batched pass, then one linker run per class."""
        boxes0, scores0 = self.pred_boxes[:, 0], self.scores[:, 0]
        idx, count = select_candidates(boxes0, scores0)
        assert int(count[:, 1:].min()) > 0, "No detections found for this class."
        idx, count = idx.cpu(), count.cpu()
        boxes0, scores0 = boxes0.float().cpu(), scores0.float().cpu()
        for c in range(1, self.num_classes):
            fb, fs, fa = [], [], []
            for t in range(self.num_frame_pairs):
                sel = idx[t, c, :int(count[t, c])]
                fb.append(boxes0[t][sel]); fs.append(scores0[t][sel, c]); fa.append(scores0[t][sel])
            self.all_paths[c] = incremental_linking(fb, fs, fa)

    def get_tubes(self):
        """online_tubes.py:134-180: label every path over time and cut it into single-class tubes."""
        keys = ("starts", "ends", "ts", "te", "dpActionScore", "label", "dpPathScore", "path_total_score", "path_boxes",
                "path_scores", "video_id")
        acc = {k: [] for k in keys}
        for c in range(1, self.num_classes):
            paths = self.all_paths[c]
            if paths is None:
                continue
            for i in range(len(paths["count"])):
                p, q, D = dpEM_max(paths["all_scores"][i][:, 1:], self.alpha_l)
                ts, te, sc, lab, tot = extract_action(p, q, D, c)
                for k in range(ts.numel()):
                    acc["starts"].append(paths["start"][i]); acc["ends"].append(paths["end"][i])
                    acc["ts"].append(int(ts[k])); acc["te"].append(int(te[k]))
                    acc["dpActionScore"].append(sc[k]); acc["label"].append(lab[k]); acc["dpPathScore"].append(tot[k])
                    acc["path_total_score"].append(float(paths["scores"][i].mean()))
                    acc["path_boxes"].append(paths["boxes"][i]); acc["path_scores"].append(paths["scores"][i])
                    acc["video_id"].append(self.video_id)
        cat = lambda xs: torch.cat(xs, 0) if xs else torch.zeros(0)
        return {"starts": cat(acc["starts"]), "ends": cat(acc["ends"]), "ts": torch.tensor(acc["ts"], dtype=torch.long),
                "te": torch.tensor(acc["te"], dtype=torch.long), "dpActionScore": cat(acc["dpActionScore"]),
                "label": cat(acc["label"]), "dpPathScore": cat(acc["dpPathScore"]),
                "path_total_score": torch.tensor(acc["path_total_score"]), "path_boxes": acc["path_boxes"],
                "path_scores": acc["path_scores"], "video_id": acc["video_id"]}

    def class_paths(self, path_score_thresh=0.0):
        """online_tubes.py:36-51: build the paths, label them, keep the tubes whose path score exceeds the threshold
        (attributes path_total_score / path_scores / path_boxes / path_starts / path_ends / path_labels)."""
        self.generate_paths()
        tubes = self.get_tubes()
        keep = torch.nonzero(tubes["dpPathScore"] > path_score_thresh).view(-1)
        self.path_total_score = tubes["path_total_score"][keep]
        self.path_scores = [tubes["path_scores"][i] for i in keep.tolist()]
        self.path_boxes = [tubes["path_boxes"][i] for i in keep.tolist()]
        self.path_starts, self.path_ends = tubes["starts"][keep], tubes["ends"][keep]
        self.path_labels = tubes["label"][keep]
        return tubes
