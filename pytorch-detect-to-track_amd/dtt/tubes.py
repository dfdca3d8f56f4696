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
