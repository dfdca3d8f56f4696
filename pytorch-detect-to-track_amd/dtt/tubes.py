"""Zero-jump Viterbi tube linking on the GPU -- the device counterpart of the reference's
`VideoPostProcessor._make_tubes / _zero_jump_link / _score_of_edge` (lib/model/utils/tracking_utils.py:86-290).

`make_tubes(frame_dets, tracks)` keeps the reference method's contract (same arguments' meaning, same result dict:
'total_score' (K,1,1), 'boxes' (K,T,5), 'idx' (K,T), 'smooth_scores' (K,T), 'scores' (K,T)); `link_tubes` is the
batched form the kernels are built for: all classes of a video in three launches (`dtt_tube_link`), nothing returns to
the host in between.  There is no CPU fallback.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

KMAX = 32  # boxes per frame the kernels keep (one 32-bit link mask per box); the reference uses 25
_GAUSS5 = (1.0 / 16.0, 4.0 / 16.0, 6.0 / 16.0, 4.0 / 16.0, 1.0 / 16.0)   # tracking_utils.py:239


def link_tubes(dets, n, trk=None, m=None, max_per_image=25, nms_thresh=0.3):
    """dets (P, F, Nmax, S>=5) float32 rows [x1,y1,x2,y2,score,...] in NMS priority order, n (P, F) int32 row counts;
    trk (P, F, 2, Mmax, 4) tracklet boxes (in frame t / predicted in frame t+1), m (P, F) int32 counts (-1: none).
    Returns kept_boxes (P,T,32,4), kept_scores (P,T,32), kept_n (P,T), path_idx (P,32,T) int32, path_total (P,32),
    n_paths (P,) -- all on the device; T = F - 1 (the last frame only closes the last pair, as in the reference)."""
    _lib.require_gpu(dets)
    assert dets.dim() == 4 and dets.dtype == torch.float32 and dets.is_contiguous()
    P, F, Nmax, S = dets.shape
    T = F - 1
    dev = dets.device
    n = n.to(device=dev, dtype=torch.int32).contiguous()
    if trk is not None:
        assert trk.shape[:3] == (P, F, 2) and trk.shape[4] == 4 and trk.is_contiguous() and trk.dtype == torch.float32
        m = m.to(device=dev, dtype=torch.int32).contiguous()
    kb = torch.empty(P, T, KMAX, 4, device=dev)
    ks = torch.empty(P, T, KMAX, device=dev)
    kn = torch.empty(P, T, dtype=torch.int32, device=dev)
    pidx = torch.zeros(P, KMAX, max(T, 1), dtype=torch.int32, device=dev)
    ptot = torch.zeros(P, KMAX, device=dev)
    npaths = torch.zeros(P, dtype=torch.int32, device=dev)
    L = _lib.lib()
    nbytes = L.dtt_tube_link_workspace_bytes(P, F)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(L.dtt_tube_link(ptr(dets), S, ptr(n), ptr(trk) if trk is not None else None,
                              ptr(m) if trk is not None else None, P, F, Nmax, trk.shape[3] if trk is not None else 0,
                              max_per_image, float(nms_thresh), ptr(kb), ptr(ks), ptr(kn), ptr(pidx), ptr(ptot),
                              ptr(npaths), ptr(ws), nbytes, stream_ptr(dev)), "tube_link")
    return kb, ks, kn, pidx, ptot, npaths


def _reflect101_index(n, r):
    idx = np.arange(-r, n + r)
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    idx = np.mod(idx, period)
    return np.where(idx >= n, period - idx, idx)


def paths_from_link(kb, ks, pidx, ptot, k_paths):
    """Assemble the reference's result dict for ONE problem from the device outputs (tracking_utils.py:224-264)."""
    T = kb.shape[0]
    dev = kb.device
    idx = pidx[:k_paths, :T].long()                                    # (K, T)
    tt = torch.arange(T, device=dev)[None, :].expand_as(idx)
    boxes = kb[tt, idx]                                                # (K, T, 4)
    raw = ks[tt, idx]                                                  # (K, T)
    top, _ = torch.sort(raw, dim=1, descending=True)
    half = int(np.ceil(0.5 * T))
    mean_top = top[:, :half].mean(dim=1, keepdim=True)                 # :237
    pad = torch.from_numpy(_reflect101_index(T, 2)).to(dev)
    padded = raw.double()[:, pad]                                      # cv2.filter2D default border
    smooth = sum(_GAUSS5[i] * padded[:, i:i + T] for i in range(5)).float() + mean_top
    return {"total_score": ptot[:k_paths].view(-1, 1, 1), "boxes": torch.cat([boxes, raw[..., None]], dim=2), "idx": idx,
            "smooth_scores": smooth, "scores": raw + mean_top}


def make_tubes(frame_dets, tracks=None, max_per_image=25, nms_thresh=0.3):
    """One class of one video.  frame_dets: sequence of F tensors (n_f, >=5), rows in NMS priority order (the last frame
    is not used, as in the reference); tracks: optional sequence of F entries, each None or a pair (boxes in frame t,
    predicted boxes in frame t+1) of (M_f, 4) tensors."""
    F = len(frame_dets)
    dev = frame_dets[0].device
    S = min(int(d.shape[1]) for d in frame_dets[:F - 1])
    nmax = max(int(d.shape[0]) for d in frame_dets[:F - 1])
    dets = torch.zeros(1, F, max(nmax, 1), S, device=dev)
    n = torch.zeros(1, F, dtype=torch.int32)
    for f in range(F - 1):
        k = int(frame_dets[f].shape[0])
        n[0, f] = k
        if k:
            dets[0, f, :k] = frame_dets[f][:, :S]
    trk = m = None
    if tracks is not None:
        mmax = max([int(t[0].shape[0]) for t in tracks if t is not None] + [1])
        trk = torch.zeros(1, F, 2, mmax, 4, device=dev)
        m = torch.full((1, F), -1, dtype=torch.int32)
        for f, t in enumerate(tracks):
            if t is not None:
                m[0, f] = int(t[0].shape[0])
                trk[0, f, 0, :m[0, f]], trk[0, f, 1, :m[0, f]] = t[0], t[1]
    kb, ks, kn, pidx, ptot, npaths = link_tubes(dets, n, trk, m, max_per_image, nms_thresh)
    k_paths = int(npaths[0])
    if k_paths == 0:
        empty = (kn[0] == 0).nonzero().view(-1)
        raise RuntimeError("ERROR: Found empty box at %d" % (int(empty[0]) if empty.numel() else -1))   # :166-169
    return paths_from_link(kb[0], ks[0], pidx[0], ptot[0], k_paths)


class VideoPostProcessor:
    """Device counterpart of the reference's `VideoPostProcessor` (tracking_utils.py:18-386): same constructor
    arguments, `build_class_paths()` returns the per-class path dicts (index 0 = background = None).

    Everything stays on the GPU as padded tensors: the per-pair / per-leg / per-class Python loops of
    `_process_frame_pairs` become one batched sort + gather, `_keep_top_k` one sort per class, and all classes are
    linked by a single `link_tubes` call.  One host read (the per-frame detection counts) decides which frames take
    part, as the reference's `nonempty_frames` does."""

    def __init__(self, pred_boxes, scores, pred_trk_boxes, classes, max_per_image=400):
        _lib.require_gpu(pred_boxes, scores, pred_trk_boxes)
        self.pred_boxes, self.scores, self.pred_trk_boxes = pred_boxes, scores, pred_trk_boxes
        self.classes = classes
        self.num_classes = len(classes)
        self.num_frames = pred_boxes.size(0) + 1
        self.max_per_image = max_per_image
        self.max_per_set = 160 * self.num_frames
        self._paths = np.ndarray((self.num_classes,), dtype=object)
        self._process_frame_pairs()
        self._keep_top_k()

    # tracking_utils.py:320-386, all pairs / legs / classes at once
    def _process_frame_pairs(self):
        P, _, R, C = self.scores.shape
        F, Pm, dev = self.num_frames, self.scores.size(0) - 1, self.scores.device
        k = min(R, self.max_per_image)
        S = self.scores[:Pm]                                                    # (Pm, 2, R, C)   (the last pair is never read)
        cls_scores = S[..., 1:].permute(0, 1, 3, 2)                             # (Pm, 2, C-1, R)
        vals, order = torch.sort(cls_scores, dim=-1, descending=True, stable=True)
        vals, order = vals[..., :k], order[..., :k]
        boxes = self.pred_boxes[:Pm, :, :, :4]                                  # class agnostic (:356-357)
        gidx = order.unsqueeze(-1).expand(-1, -1, -1, -1, 4)
        b = torch.gather(boxes.unsqueeze(2).expand(-1, -1, C - 1, -1, -1), 3, gidx)
        bg = torch.gather(S[..., 0].unsqueeze(2).expand(-1, -1, C - 1, -1), 3, order)
        entry = torch.cat([b, vals.unsqueeze(-1), bg.unsqueeze(-1)], dim=-1)    # (Pm, 2, C-1, k, 6)
        # frame f = [leg 1 of pair f-1, leg 0 of pair f] in that order (pairs are visited in order, leg 0 first)
        dets = torch.zeros(C - 1, F, 2 * k, 6, device=dev)
        n = torch.zeros(C - 1, F, dtype=torch.int32, device=dev)
        if Pm > 0:
            leg0 = entry[:, 0].permute(1, 0, 2, 3)                              # (C-1, Pm, k, 6) -> frames 0..Pm-1
            leg1 = entry[:, 1].permute(1, 0, 2, 3)                              #                 -> frames 1..Pm
            dets[:, 1:Pm + 1, :k] = leg1
            dets[:, 0, :k] = leg0[:, 0]
            dets[:, 1:Pm, k:2 * k] = leg0[:, 1:]
            n[:, 0] = k
            n[:, 1:Pm] = 2 * k
            n[:, Pm] = k
        self._dets, self._n = dets, n
        # tracklets: RoIs of leg 0 whose best foreground score exceeds 0.01 (:324-345), in RoI order
        tmask = S[:, 0, :, 1:].max(dim=2).values > 0.01 if Pm > 0 else torch.zeros(0, R, dtype=torch.bool, device=dev)
        _, torder = torch.sort(tmask.to(torch.int8), dim=1, descending=True, stable=True)
        trk = torch.zeros(F, 2, R, 4, device=dev)
        m = torch.full((F,), -1, dtype=torch.int32, device=dev)
        if Pm > 0:
            ti = torder.unsqueeze(-1).expand(-1, -1, 4)
            trk[:Pm, 0] = torch.gather(self.pred_boxes[:Pm, 0, :, :4], 1, ti)
            trk[:Pm, 1] = torch.gather(self.pred_trk_boxes[:Pm], 1, ti)
            cnt = tmask.sum(dim=1).to(torch.int32)
            m[:Pm] = torch.where(cnt > 0, cnt, torch.full_like(cnt, -1))
        self._trk, self._m = trk, m

    # tracking_utils.py:294-318 for every class
    def _keep_top_k(self):
        C1, F, N2, _ = self._dets.shape
        dev = self._dets.device
        valid = torch.arange(N2, device=dev)[None, None, :] < self._n[:, :, None]
        sc = torch.where(valid, self._dets[..., 4], torch.full_like(self._dets[..., 4], float("-inf")))
        total = int(self._n[0].sum()) if C1 else 0                              # same count for every class
        top_k = self.max_per_set
        if C1 and min(total, top_k) >= total:
            raise IndexError("index %d is out of bounds for dimension 0 with size %d" % (min(total, top_k), total))  # :312
        flat, _ = torch.sort(sc.reshape(C1, -1), dim=1, descending=True)
        thresh = flat[:, min(total, top_k)] if C1 else flat.new_zeros(0)
        self.CONF_THRESH = torch.cat([thresh.new_full((1,), float("-inf")), thresh])
        keep = valid & (sc >= thresh[:, None, None])
        none_kept = keep.sum(dim=2, keepdim=True) == 0
        keep = torch.where(none_kept, valid, keep)                              # `if keep.numel()==0: continue` (:318)
        _, order = torch.sort(keep.to(torch.int8), dim=2, descending=True, stable=True)
        self._dets = torch.gather(self._dets, 2, order.unsqueeze(-1).expand(-1, -1, -1, 6)).contiguous()
        self._n = keep.sum(dim=2).to(torch.int32)

    # tracking_utils.py:54-84
    def build_class_paths(self, max_per_image=25, nms_thresh=0.3):
        C1, F = self._n.shape
        n_host = self._n.cpu().numpy()
        groups = {}
        for c in range(C1):                                                     # classes sharing the same non-empty frames
            frames = tuple(int(f) for f in np.nonzero(n_host[c] > 0)[0])
            if len(frames) >= 2:
                groups.setdefault(frames, []).append(c)
            elif len(frames) == 1:
                raise RuntimeError("class %d has detections in a single frame" % (c + 1))
        for frames, cls in groups.items():
            fi = torch.tensor(frames, device=self._dets.device)
            ci = torch.tensor(cls, device=self._dets.device)
            dets = self._dets[ci][:, fi].contiguous()
            n = self._n[ci][:, fi].contiguous()
            trk = self._trk[fi].unsqueeze(0).expand(len(cls), -1, -1, -1, -1).contiguous()
            m = self._m[fi].unsqueeze(0).expand(len(cls), -1).contiguous()
            kb, ks, kn, pidx, ptot, npaths = link_tubes(dets, n, trk, m, max_per_image, nms_thresh)
            npaths = npaths.cpu().tolist()
            for j, c in enumerate(cls):
                if npaths[j] == 0:
                    raise RuntimeError("ERROR: Found empty box")                 # :166-169
                self._paths[c + 1] = paths_from_link(kb[j], ks[j], pidx[j], ptot[j], npaths[j])
        return self._paths
