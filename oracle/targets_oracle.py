"""TEST INFRASTRUCTURE -- not part of the product path (only tests/ import this).

Torch-tensor restatement of the two training-time target samplers, written against
  _ProposalTargetLayer          rpn/proposal_target_layer_cascade.py:20-208
  _TrackingProposalTargetLayer  rpn/tracking_proposal_target_layer.py:20-196
and pinned by tests/test_host_logic_cpu.py against fixtures produced by RUNNING the reference's own layers
(tests/golden/make_golden.py -> tests/golden/targets.npz; same numpy seed -> same sampled RoIs).  The product
(dtt/targets.py) runs HIP kernels (csrc/targets.hip); tests/test_gpu_targets.py compares the two.
`device_rule_sample` restates the device-side selection rule of dtt_proposal_target_sample (no reference counterpart:
the reference draws after reading the candidate counts back to the host).
"""
import numpy as np
import torch
from torch import nn

from dtt.rpn import bbox_overlaps_batch, bbox_transform_batch


class _ProposalTargetLayer(nn.Module):
    """forward(all_rois (B,R,5), gt_boxes (B,G,5), num_boxes) ->
    rois (B,N,5), labels (B,N), bbox_targets (B,N,4), inside weights, outside weights; N = TRAIN.BATCH_SIZE."""

    def __init__(self, nclasses, cfg=None):
        super().__init__()
        if cfg is None:
            from dtt.config import cfg as _cfg
            cfg = _cfg
        self._cfg = cfg
        self._num_classes = nclasses
        T = cfg.TRAIN
        self.register_buffer("means", torch.tensor(T.BBOX_NORMALIZE_MEANS, dtype=torch.float32), persistent=False)
        self.register_buffer("stds", torch.tensor(T.BBOX_NORMALIZE_STDS, dtype=torch.float32), persistent=False)
        self.register_buffer("inside_w", torch.tensor(T.BBOX_INSIDE_WEIGHTS, dtype=torch.float32), persistent=False)

    def forward(self, all_rois, gt_boxes, num_boxes):
        T = self._cfg.TRAIN
        dev = gt_boxes.device
        gt_append = torch.zeros_like(gt_boxes)
        gt_append[:, :, 1:5] = gt_boxes[:, :, :4]
        all_rois = torch.cat([all_rois, gt_append], 1)  # gt boxes join the candidates (:42-46)
        rois_per_image = int(T.BATCH_SIZE / 1)
        fg_per_image = int(np.round(T.FG_FRACTION * rois_per_image)) or 1

        overlaps = bbox_overlaps_batch(all_rois, gt_boxes[:, :, :5])
        max_ov, assign = overlaps.max(2)
        B = overlaps.size(0)
        labels_all = torch.gather(gt_boxes[:, :, 4], 1, assign)
        # host side: which candidates to keep (numpy RNG, :137-186)
        mo = max_ov.detach().cpu().numpy()
        keep = np.zeros((B, rois_per_image), dtype=np.int64)
        n_fg = np.zeros((B,), dtype=np.int64)
        for i in range(B):
            fg = np.nonzero(mo[i] >= T.FG_THRESH)[0]
            bg = np.nonzero((mo[i] < T.BG_THRESH_HI) & (mo[i] >= T.BG_THRESH_LO))[0]
            if fg.size > 0 and bg.size > 0:
                fg_n = min(fg_per_image, fg.size)
                fg = fg[np.random.permutation(fg.size)[:fg_n]]
                bg_n = rois_per_image - fg_n
                bg = bg[np.floor(np.random.rand(bg_n) * bg.size).astype(np.int64)]
            elif fg.size > 0:
                fg = fg[np.floor(np.random.rand(rois_per_image) * fg.size).astype(np.int64)]
                fg_n, bg = rois_per_image, bg[:0]
            elif bg.size > 0:
                bg = bg[np.floor(np.random.rand(rois_per_image) * bg.size).astype(np.int64)]
                fg_n, fg = 0, fg[:0]
            else:
                raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")
            keep[i] = np.concatenate([fg, bg])
            n_fg[i] = fg_n
        keep_t = torch.from_numpy(keep).to(dev)
        n_fg_t = torch.from_numpy(n_fg).to(dev)
        labels = torch.gather(labels_all, 1, keep_t)
        pos = torch.arange(rois_per_image, device=dev).unsqueeze(0)
        labels = torch.where(pos < n_fg_t.unsqueeze(1), labels, torch.zeros_like(labels))  # bg labels -> 0 (:193-194)
        rois = torch.gather(all_rois, 1, keep_t.unsqueeze(2).expand(-1, -1, 5)).clone()
        rois[:, :, 0] = torch.arange(B, device=dev, dtype=rois.dtype).unsqueeze(1)
        gt_sel = torch.gather(gt_boxes, 1, torch.gather(assign, 1, keep_t).unsqueeze(2).expand(-1, -1, gt_boxes.size(2)))
        targets = bbox_transform_batch(rois[:, :, 1:5], gt_sel[:, :, :4])
        if T.BBOX_NORMALIZE_TARGETS_PRECOMPUTED:
            targets = (targets - self.means.to(dev)) / self.stds.to(dev)
        fgmask = (labels > 0).unsqueeze(2).to(targets.dtype)
        bbox_targets = targets * fgmask
        inside = self.inside_w.to(dev).view(1, 1, 4) * fgmask
        outside = (inside > 0).float()
        return rois, labels, bbox_targets, inside, outside


class _TrackingProposalTargetLayer(nn.Module):
    """forward(gt_boxes (2,B,G,6) [x1,y1,x2,y2,cls,track_id], num_boxes (2,B,1)) ->
    tracking rois (B,G,5) = frame-t GT boxes, labels (B,G), targets (B,G,4), inside, outside weights.

    Reference behaviour kept as is: targets / labels are listed for the matched tracks sorted by track id and
    packed to the front, while the RoIs stay in the original GT order (tracking_proposal_target_layer.py:171-185)."""

    def __init__(self, nclasses, cfg=None):
        super().__init__()
        if cfg is None:
            from dtt.config import cfg as _cfg
            cfg = _cfg
        self._cfg = cfg
        T = cfg.TRAIN
        self.register_buffer("means", torch.tensor(T.BBOX_NORMALIZE_MEANS, dtype=torch.float32), persistent=False)
        self.register_buffer("stds", torch.tensor(T.BBOX_NORMALIZE_STDS, dtype=torch.float32), persistent=False)
        self.register_buffer("inside_w", torch.tensor(T.BBOX_INSIDE_WEIGHTS, dtype=torch.float32), persistent=False)

    def forward(self, gt_boxes, num_boxes):
        dev = gt_boxes.device
        _, B, G, _ = gt_boxes.shape
        nb = num_boxes.reshape(2, B).to(dev)
        idx = torch.arange(G, device=dev).view(1, G)
        v0 = idx < nb[0].view(B, 1)
        v1 = idx < nb[1].view(B, 1)
        id0, id1 = gt_boxes[0, :, :, 5], gt_boxes[1, :, :, 5]
        corr = (id0.unsqueeze(2) == id1.unsqueeze(1)) & v0.unsqueeze(2) & v1.unsqueeze(1)  # (B, G_t, G_t+tau)
        has0, has1 = corr.any(2), corr.any(1)
        ok = has0.any(1) & has1.any(1)
        big = torch.finfo(gt_boxes.dtype).max

        def packed(frame, has, ids):
            key = torch.where(has, ids, torch.full_like(ids, big))
            order = torch.sort(key, dim=1, descending=False, stable=True)[1]
            g = torch.gather(frame, 1, order.unsqueeze(2).expand(-1, -1, frame.size(2)))
            keepn = has.sum(1, keepdim=True)
            return g * (idx < keepn).unsqueeze(2).to(g.dtype)

        r0 = packed(gt_boxes[0], has0, id0) * ok.view(B, 1, 1).to(gt_boxes.dtype)
        r1 = packed(gt_boxes[1], has1, id1) * ok.view(B, 1, 1).to(gt_boxes.dtype)
        labels = r0[:, :, 4]
        rois = torch.zeros((B, G, 5), dtype=gt_boxes.dtype, device=dev)
        rois[:, :, 0] = torch.arange(B, device=dev, dtype=gt_boxes.dtype).unsqueeze(1)
        rois[:, :, 1:] = gt_boxes[0, :, :, :4]
        rois = rois * ok.view(B, 1, 1).to(rois.dtype)
        targets = bbox_transform_batch(r0[:, :, :4], r1[:, :, :4])
        if self._cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED:
            targets = (targets - self.means.to(dev)) / self.stds.to(dev)
        fgmask = (labels > 0).unsqueeze(2).to(targets.dtype)
        targets = targets * fgmask
        inside = self.inside_w.to(dev).view(1, 1, 4) * fgmask
        return rois, labels, targets, inside, (inside > 0).float()


def device_rule_sample(max_ov, fg_thresh, bg_hi, bg_lo, u_fg, u_bg, n_out, fg_per_image):
    """Candidate indices the device picks from the uniforms (csrc/targets.hip `pt_sample`, pos == NULL) for ONE image:
    max_ov (N,) numpy, u_fg (N,), u_bg (n_out,) float64.  Returns (indices (n_out,), fg_n)."""
    fg = np.nonzero(max_ov >= fg_thresh)[0]
    bg = np.nonzero((max_ov < bg_hi) & (max_ov >= bg_lo))[0]
    if fg.size and bg.size:
        fg_n = min(fg_per_image, fg.size)
        order = np.lexsort((np.arange(fg.size), u_fg[:fg.size]))     # ascending key, ties by position
        sel_fg = fg[order[:fg_n]]
        sel_bg = bg[np.minimum(np.floor(u_bg[:n_out - fg_n] * bg.size).astype(np.int64), bg.size - 1)]
        return np.concatenate([sel_fg, sel_bg]), fg_n
    if fg.size:
        return fg[np.minimum(np.floor(u_bg * fg.size).astype(np.int64), fg.size - 1)], n_out
    if bg.size:
        return bg[np.minimum(np.floor(u_bg * bg.size).astype(np.int64), bg.size - 1)], 0
    raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")
