"""Training-time target samplers that sit between the proposal layer and PSRoI pooling
(SURVEY.md section 8f rank 1), on the device (csrc/targets.hip through the C ABI; no CPU path).

  _ProposalTargetLayer          rpn/proposal_target_layer_cascade.py:20-208
  _TrackingProposalTargetLayer  rpn/tracking_proposal_target_layer.py:20-196

RoI sampling draws from numpy's global generator, as the reference does.  cfg.TRAIN.SAMPLER_RNG selects how:
  "device"     (default) the host draws, per image, one uniform per candidate + one per output slot WITHOUT looking at
               the device; the kernel turns them into the reference's selection (a uniform subset of the foreground
               candidates without replacement, background slots with replacement) -- nothing is read back inside the
               training step;
  "reference"  the host reads the two candidate counts of each image (8 bytes) and consumes the generator exactly as
               the reference (`np.random.permutation(fg)`, `np.random.rand(bg_n)`): for a given seed the sampled RoIs
               are the reference's, bit for bit (pinned by tests/golden/targets.npz).
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import _lib
from ._lib import check, ptr, require_gpu, stream_ptr


def _f4(v):
    return (ctypes.c_float * 4)(*[float(x) for x in v])


class _ProposalTargetLayer(nn.Module):
    """forward(all_rois (B,R,5), gt_boxes (B,G,5), num_boxes) ->
    rois (B,N,5), labels (B,N), bbox_targets (B,N,4), inside weights, outside weights; N = TRAIN.BATCH_SIZE."""

    def __init__(self, nclasses, cfg=None):
        super().__init__()
        if cfg is None:
            from .config import cfg as _cfg
            cfg = _cfg
        self._cfg = cfg
        self._num_classes = nclasses
        self.last_status = None   # int32 (B,) on the device: 1 = an image had neither fg nor bg candidates (the reference raises)
        self._pending_status = []  # the status words of every call since the last status_flag() / check_status(): one per leg

    def forward(self, all_rois, gt_boxes, num_boxes):
        T = self._cfg.TRAIN
        require_gpu(all_rois, gt_boxes)
        dev = gt_boxes.device
        all_rois = all_rois.detach().float().contiguous()
        gt = gt_boxes.detach().float().contiguous()
        B, R, _ = all_rois.shape
        G, gs = gt.shape[1], gt.shape[2]
        N = R + G
        n_out = int(T.BATCH_SIZE / 1)
        fg_per_image = int(np.round(T.FG_FRACTION * n_out)) or 1
        L = _lib.lib()
        i32 = dict(dtype=torch.int32, device=dev)
        assign, fg_list, bg_list = (torch.empty((B, N), **i32) for _ in range(3))
        counts = torch.empty((B, 2), **i32)
        st = stream_ptr(dev)
        with torch.cuda.device(dev):
            check(L.dtt_proposal_target_assign(ptr(all_rois), ptr(gt), B, R, G, gs, float(T.FG_THRESH), float(T.BG_THRESH_HI),
                                               float(T.BG_THRESH_LO), ptr(assign), ptr(fg_list), ptr(bg_list), ptr(counts), st),
                  "proposal_target assign")
            pos = fgn = ufg = ubg = None
            if getattr(T, "SAMPLER_RNG", "device") == "reference":
                ch = counts.cpu().numpy()                       # the one host read of this mode
                pos_h = np.zeros((B, n_out), dtype=np.int32)
                fgn_h = np.zeros((B,), dtype=np.int32)
                for i in range(B):                                # proposal_target_layer_cascade.py:137-186
                    nf, nbg = int(ch[i, 0]), int(ch[i, 1])
                    if nf > 0 and nbg > 0:
                        k = min(fg_per_image, nf)
                        p_fg = np.random.permutation(nf)[:k]
                        p_bg = np.floor(np.random.rand(n_out - k) * nbg).astype(np.int64)
                    elif nf > 0:
                        k, p_fg, p_bg = n_out, np.floor(np.random.rand(n_out) * nf).astype(np.int64), np.zeros((0,), np.int64)
                    elif nbg > 0:
                        k, p_fg, p_bg = 0, np.zeros((0,), np.int64), np.floor(np.random.rand(n_out) * nbg).astype(np.int64)
                    else:
                        raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")
                    pos_h[i] = np.concatenate([p_fg, p_bg])
                    fgn_h[i] = k
                pos = torch.from_numpy(pos_h).to(dev)
                fgn = torch.from_numpy(fgn_h).to(dev)
            else:
                # (through pinned memory, non-blocking: a plain .to(dev) from pageable memory makes the host wait for the stream)
                ufg = torch.from_numpy(np.random.rand(B, N)).pin_memory().to(dev, non_blocking=True)
                ubg = torch.from_numpy(np.random.rand(B, n_out)).pin_memory().to(dev, non_blocking=True)
            f32 = dict(dtype=torch.float32, device=dev)
            rois = torch.empty((B, n_out, 5), **f32)
            labels = torch.empty((B, n_out), **f32)
            targets, inside, outside = (torch.empty((B, n_out, 4), **f32) for _ in range(3))
            status = torch.empty((B,), **i32)
            check(L.dtt_proposal_target_sample(ptr(all_rois), ptr(gt), B, R, G, gs, ptr(assign), ptr(fg_list), ptr(bg_list),
                                               ptr(counts), ptr(pos), ptr(fgn), ptr(ufg), ptr(ubg), n_out, fg_per_image,
                                               _f4(T.BBOX_NORMALIZE_MEANS), _f4(T.BBOX_NORMALIZE_STDS), _f4(T.BBOX_INSIDE_WEIGHTS),
                                               int(bool(T.BBOX_NORMALIZE_TARGETS_PRECOMPUTED)), ptr(rois), ptr(labels),
                                               ptr(targets), ptr(inside), ptr(outside), ptr(status), st),
                  "proposal_target sample")
        self.last_status = status
        self._pending_status = self._pending_status[-7:] + [status]
        return rois, labels, targets, inside, outside

    def status_flag(self):
        """A (1,) float tensor on the device: non-zero iff a call since the last status_flag() / check_status() (every leg of the
        step) met an image with neither foreground nor background candidates.  No host read: the caller folds it into a
        synchronisation it makes anyway (and, on several ranks, all-reduces it so that every rank aborts together)."""
        sts, self._pending_status = self._pending_status, []
        if not sts:
            return None
        return torch.stack([(st != 0).any() for st in sts]).any().float().view(1)

    def check_status(self):
        """proposal_target_layer_cascade.py:186 raises when an image has neither foreground nor background candidates.  The
        "device" sampler cannot raise inside the step without reading the device (the kernel fills candidate 0 and flags the
        image instead): call this -- or read status_flag() -- right after the forward, BEFORE the loss is used (the reference
        raises before backward: a flagged step must not update the weights)."""
        flag = self.status_flag()
        if flag is not None and bool(flag.item()):
            raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")


class _TrackingProposalTargetLayer(nn.Module):
    """forward(gt_boxes (2,B,G,6) [x1,y1,x2,y2,cls,track_id], num_boxes (2,B,1)) ->
    tracking rois (B,G,5) = frame-t GT boxes, labels (B,G), targets (B,G,4), inside, outside weights.

    Reference behaviour kept as is: targets / labels are listed for the matched tracks sorted by track id and
    packed to the front, while the RoIs stay in the original GT order (tracking_proposal_target_layer.py:171-185)."""

    def __init__(self, nclasses, cfg=None):
        super().__init__()
        if cfg is None:
            from .config import cfg as _cfg
            cfg = _cfg
        self._cfg = cfg

    def forward(self, gt_boxes, num_boxes):
        T = self._cfg.TRAIN
        require_gpu(gt_boxes)
        dev = gt_boxes.device
        gt = gt_boxes.detach().float().contiguous()
        _, B, G, six = gt.shape
        if six != 6:
            raise ValueError("tracking targets need (2, B, G, 6) boxes [x1,y1,x2,y2,cls,track_id]")
        nb = num_boxes.reshape(2, B).to(device=dev, dtype=torch.int64).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        rois = torch.empty((B, G, 5), **f32)
        labels = torch.empty((B, G), **f32)
        targets, inside, outside = (torch.empty((B, G, 4), **f32) for _ in range(3))
        with torch.cuda.device(dev):
            check(_lib.lib().dtt_tracking_target(ptr(gt), ptr(nb), B, G, _f4(T.BBOX_NORMALIZE_MEANS), _f4(T.BBOX_NORMALIZE_STDS),
                                                 _f4(T.BBOX_INSIDE_WEIGHTS), int(bool(T.BBOX_NORMALIZE_TARGETS_PRECOMPUTED)),
                                                 ptr(rois), ptr(labels), ptr(targets), ptr(inside), ptr(outside), stream_ptr(dev)),
                  "tracking_target")
        return rois, labels, targets, inside, outside
