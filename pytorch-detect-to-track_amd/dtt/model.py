"""Detect-to-Track R-FCN graph that hosts the hot-path ops (PyTorch-ROCm is plumbing here: the trunk and
the dense convolutions run on MIOpen / rocBLAS; everything the reference implemented as custom CUDA or
per-image Python loops goes through libdtt_hip.so).

Drop-in contract (paths relative to the reference's lib/model/):
  * `_RPN(din).forward(base_feat, im_info, gt_boxes, num_boxes)` -> (rois, loss_cls, loss_box)   rpn/rpn.py:58-107
  * `_RFCN.forward(im_data (B,2,3,H,W), im_info (B,2,3), gt_boxes (B,2,G,6), num_boxes (B,2,1))` -> the 10-tuple
    of faster_rcnn/rfcn.py:249-250
  * `resnet(classes, num_layers, pretrained, pretrained_rfcn, class_agnostic)` + `.create_architecture()`
    faster_rcnn/resnet.py:247-312, with the same state_dict keys (RFCN_base.{0,1,4,5,6,7}.*,
    RFCN_base.RFCN_net.* and its alias RFCN_net.*, RFCN_rpn.*, RFCN_cls_net.*, RFCN_bbox_net.*, corr_bbox_net.*)
    so `rfcn_detect_track_*.pth` checkpoints load and save unchanged.
Differences that are deliberate: `num_layers` is honoured (the reference always builds ResNet-101,
resnet.py:259); `rois_label` is reshaped to (n_legs, B, -1) instead of the hard-wired view(2, 2, -1)
(rfcn.py:220); the three correlations write straight into the tracking concat buffer.
"""
import math

import os

import torch
import torch.nn.functional as F
from torch import nn

from .config import cfg as _global_cfg
from .ops import (Correlation, RoIAlignAvg, _PSRoIPooling, _RoIPooling, correlation_forward_into, correlation_output_shape,
                  psroi_pool_vote, psroi_vote, roi_crop_pool)
from .rpn import _AnchorTargetLayer, _ProposalLayer
from .targets import _ProposalTargetLayer, _TrackingProposalTargetLayer


def _smooth_l1_loss(bbox_pred, bbox_targets, bbox_inside_weights, bbox_outside_weights, sigma=1.0, dim=(1,)):
    """utils/net_utils.py:73-87."""
    s2 = sigma ** 2
    d = bbox_inside_weights * (bbox_pred - bbox_targets)
    ad = d.abs()
    quad = (ad < 1.0 / s2).detach().float()
    loss = bbox_outside_weights * (d.pow(2) * (s2 / 2.0) * quad + (ad - 0.5 / s2) * (1.0 - quad))
    for i in sorted(dim, reverse=True):
        loss = loss.sum(i)
    return loss.mean()


# ----------------------------------------------------------------------------------------- trunk
class Bottleneck(nn.Module):
    """Caffe-style bottleneck: the stride sits on the first 1x1 conv (resnet.py:66-107)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1, dilate_first_conv=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=dilation if dilation > 1 else 1,
                               bias=False, dilation=dilation if dilation > 1 else 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(out + res)


_DEPTHS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def _make_stage(inplanes, planes, blocks, stride=1, dilation=1):
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                             nn.BatchNorm2d(planes * 4))
    layers = [Bottleneck(inplanes, planes, stride, down, dilation=dilation)]
    layers += [Bottleneck(planes * 4, planes, dilation=dilation) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


def _trunk(num_layers):
    """conv1, bn1, relu, maxpool(ceil), layer1..3, dilated layer4 (stride 16 overall, resnet.py:110-125)."""
    d = _DEPTHS[num_layers]
    mods = [nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2, padding=0, ceil_mode=True),
            _make_stage(64, 64, d[0]), _make_stage(256, 128, d[1], stride=2), _make_stage(512, 256, d[2], stride=2),
            _make_stage(1024, 512, d[3], stride=1, dilation=2)]
    seq = nn.Sequential(*mods)
    for m in seq.modules():
        if isinstance(m, nn.Conv2d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2.0 / n))
    return seq


# ------------------------------------------------------------------------------------------- RPN
class _RPN(nn.Module):
    """rpn/rpn.py:16-107."""

    def __init__(self, din, cfg=None):
        super().__init__()
        self._cfg = cfg or _global_cfg
        c = self._cfg
        self.din = din
        self.anchor_scales = c.ANCHOR_SCALES
        self.anchor_ratios = c.ANCHOR_RATIOS
        self.feat_stride = c.FEAT_STRIDE[0]
        self.RPN_Conv = nn.Conv2d(self.din, 512, 3, 1, 1, bias=True)
        self.nc_score_out = len(self.anchor_scales) * len(self.anchor_ratios) * 2
        self.RPN_cls_score = nn.Conv2d(512, self.nc_score_out, 1, 1, 0)
        self.nc_bbox_out = len(self.anchor_scales) * len(self.anchor_ratios) * 4
        self.RPN_bbox_pred = nn.Conv2d(512, self.nc_bbox_out, 1, 1, 0)
        self.RPN_proposal = _ProposalLayer(self.feat_stride, self.anchor_scales, self.anchor_ratios, cfg=c)
        self.RPN_anchor_target = _AnchorTargetLayer(self.feat_stride, self.anchor_scales, self.anchor_ratios, cfg=c)
        self.rpn_loss_cls = 0
        self.rpn_loss_box = 0

    @staticmethod
    def reshape(x, d):
        s = x.size()
        return x.view(s[0], int(d), int(float(s[1] * s[2]) / float(d)), s[3])

    def head(self, base_feat, conv1=None):
        """The convolutional part: (cls_score, cls_score_r, cls_prob, bbox_pred).  conv1: relu(RPN_Conv(base_feat)) when
        the fused trunk has already computed it (dtt.fuse.FusedTrunkNHWC)."""
        if conv1 is None:
            conv1 = F.relu(self.RPN_Conv(base_feat), inplace=True)
        cls_score = self.RPN_cls_score(conv1).contiguous()   # (a channels-last input gives a channels-last map: the views below need NCHW)
        cls_score_r = self.reshape(cls_score, 2)
        cls_prob = self.reshape(F.softmax(cls_score_r, dim=1), self.nc_score_out)
        return cls_score, cls_score_r, cls_prob, self.RPN_bbox_pred(conv1).contiguous()

    def head_scores(self, base_feat, conv1=None):
        """The score half of `head`: (conv1, cls_prob) -- the box-delta convolution is the caller's (inference overlap)."""
        if conv1 is None:
            conv1 = F.relu(self.RPN_Conv(base_feat), inplace=True)
        cls_score_r = self.reshape(self.RPN_cls_score(conv1).contiguous(), 2)
        return conv1, self.reshape(F.softmax(cls_score_r, dim=1), self.nc_score_out)

    def proposals(self, cls_prob, bbox_pred, im_info):
        return self.RPN_proposal((cls_prob.detach(), bbox_pred.detach(), im_info, "TRAIN" if self.training else "TEST"))

    def forward(self, base_feat, im_info, gt_boxes, num_boxes):
        B = base_feat.size(0)
        cls_score, cls_score_r, cls_prob, bbox_pred = self.head(base_feat)
        rois = self.proposals(cls_prob, bbox_pred, im_info)
        self.rpn_loss_cls = 0
        self.rpn_loss_box = 0
        if self.training:
            assert gt_boxes is not None
            labels, tgt, w_in, w_out = self.RPN_anchor_target((cls_score.detach(), gt_boxes[:, :, :5], im_info, num_boxes))
            score = cls_score_r.permute(0, 2, 3, 1).contiguous().view(B, -1, 2)
            label = labels.view(B, -1)
            keep = label.view(-1).ne(-1).nonzero().view(-1)
            score = torch.index_select(score.view(-1, 2), 0, keep)
            label = torch.index_select(label.view(-1), 0, keep).long()
            self.rpn_loss_cls = F.cross_entropy(score, label)
            self.rpn_loss_box = _smooth_l1_loss(bbox_pred, tgt, w_in, w_out, sigma=3, dim=[1, 2, 3])
        return rois, self.rpn_loss_cls, self.rpn_loss_box


# ------------------------------------------------------------------------------------------ RFCN
class _RFCN(nn.Module):
    """faster_rcnn/rfcn.py:22-272."""

    def __init__(self, classes, class_agnostic, cfg=None):
        super().__init__()
        self._cfg = cfg or _global_cfg
        c = self._cfg
        self.classes = classes
        self.n_classes = len(classes)
        self.n_reg_classes = 1 if class_agnostic else len(classes)
        self.class_agnostic = class_agnostic
        self.RFCN_loss_cls = 0
        self.RFCN_loss_bbox = 0
        P = c.POOLING_SIZE
        self.RFCN_rpn = _RPN(self.dout_base_model, cfg=c)
        self.RFCN_proposal_target = _ProposalTargetLayer(self.n_classes, cfg=c)
        self.RFCN_tracking_proposal_target = _TrackingProposalTargetLayer(self.n_classes, cfg=c)
        self.RFCN_psroi_cls_pool = _PSRoIPooling(P, P, spatial_scale=1.0 / 16.0, group_size=7, output_dim=self.n_classes)
        self.RFCN_psroi_loc_pool = _PSRoIPooling(P, P, spatial_scale=1.0 / 16.0, group_size=7,
                                                 output_dim=4 * self.n_reg_classes)
        self.grid_size = P * 2 if c.CROP_RESIZE_WITH_MAX_POOL else P
        self.RFCN_cls_net = nn.Conv2d(512, self.n_classes * 7 * 7, [1, 1], padding=0, stride=1)
        nn.init.normal_(self.RFCN_cls_net.weight, 0.0, 0.01)
        self.RFCN_bbox_net = nn.Conv2d(512, 4 * self.n_reg_classes * 7 * 7, [1, 1], padding=0, stride=1)
        nn.init.normal_(self.RFCN_bbox_net.weight, 0.0, 0.01)
        d = int(getattr(c, "CORR_MAX_DISPLACEMENT", 8))  # 8 in the reference (rfcn.py:58-60); 16 = BASELINE config 5
        self.conv3_corr_layer = Correlation(pad_size=d, kernel_size=1, max_displacement=d, stride1=2, stride2=2)
        self.conv4_corr_layer = Correlation(pad_size=d, kernel_size=1, max_displacement=d, stride1=1, stride2=1)
        self.conv5_corr_layer = Correlation(pad_size=d, kernel_size=1, max_displacement=d, stride1=1, stride2=1)
        # legacy-head RoI pooling of `top` (faster_rcnn.py:33-37, 72-83), only run when cfg.RFCN_ROI_FEATURES asks for it
        self.RFCN_roi_align = RoIAlignAvg(P, P, 1.0 / 16.0)
        self.RFCN_roi_pool = _RoIPooling(P, P, 1.0 / 16.0)
        self.roi_feat = None
        self.RFCN_cls_score = nn.AvgPool2d((7, 7), stride=(7, 7))
        self.RFCN_bbox_pred = nn.AvgPool2d((7, 7), stride=(7, 7))
        self.RFCN_tracking_pred = nn.AvgPool2d((7, 7), stride=(7, 7))

    # -- pooling + vote: autograd path = op + AvgPool2d (reference graph); inference = fused kernel pair
    def _pool_vote(self, pool, vote, feat, rois):
        if torch.is_grad_enabled() and feat.requires_grad:
            return vote(pool(feat, rois)).squeeze(3).squeeze(2)
        return psroi_vote(feat, rois, pool.pooled_height, pool.pooled_width, pool.spatial_scale, pool.group_size,
                          pool.output_dim)

    def _roi_features(self, top, flat_rois):
        """cfg.RFCN_ROI_FEATURES: pool the 512-channel `top` map for the RoIs being scored with the legacy head's op
        (faster_rcnn.py:72-83) -> (R_total, 512, P, P), kept in `self.roi_feat`.  BASELINE config 5's "RoI-Align path"."""
        mode = getattr(self._cfg, "RFCN_ROI_FEATURES", "")
        if not mode:
            self.roi_feat = None
            return None
        rois = flat_rois.detach().contiguous()
        top = top if top.is_contiguous() else top.contiguous()   # the pooling kernels read NCHW planes
        if mode == "align":
            self.roi_feat = self.RFCN_roi_align(top, rois)
        elif mode == "pool":
            self.roi_feat = self.RFCN_roi_pool(top, rois)
        elif mode == "crop":
            self.roi_feat = roi_crop_pool(top, rois, self._cfg.POOLING_SIZE, self._cfg.CROP_RESIZE_WITH_MAX_POOL)
        else:
            raise ValueError("cfg.RFCN_ROI_FEATURES must be '', 'align', 'pool' or 'crop' (got %r)" % (mode,))
        return self.roi_feat

    def _tracking_features(self, rfcn_bbox, conv3, conv4, conv5, whole=None):
        """cat([bbox_t, bbox_t+tau, corr3, corr4, corr5], 1) (rfcn.py:166-174).  Without autograd the
        correlations write directly into their channel slices of the concat buffer.  `whole`: the three un-sliced
        (n_legs * B, C, H, W) maps the per-leg lists were cut from -- a channels-last training trunk's maps stay whole under
        autograd (dtt.ops.Correlation.pair)."""
        layers = (self.conv3_corr_layer, self.conv4_corr_layer, self.conv5_corr_layer)
        feats = (conv3, conv4, conv5)
        need_grad = torch.is_grad_enabled() and any(f.requires_grad for pair in feats for f in pair)
        if need_grad or len(rfcn_bbox) != 2:
            out = list(rfcn_bbox)
            n = len(rfcn_bbox)
            for i in range(n - 1):
                for j in range(i + 1, n):
                    if whole is not None:
                        out += [l.pair(m, conv3[0].size(0), i, j) for l, m in zip(layers, whole)]
                    else:
                        out += [l(f[i], f[j]) for l, f in zip(layers, feats)]
            return torch.cat(out, dim=1)
        B, cb, H, W = rfcn_bbox[0].shape
        chans = [correlation_output_shape(f[0].size(1), f[0].size(2), f[0].size(3), l.pad_size, l.kernel_size,
                                          l.max_displacement, l.stride1, l.stride2)[0] for l, f in zip(layers, feats)]
        buf = torch.empty((B, 2 * cb + sum(chans), H, W), dtype=torch.float32, device=rfcn_bbox[0].device)
        buf[:, :cb] = rfcn_bbox[0]
        buf[:, cb:2 * cb] = rfcn_bbox[1]
        off = 2 * cb
        for l, f, ch in zip(layers, feats, chans):
            correlation_forward_into(buf[:, off:off + ch], f[0].contiguous(), f[1].contiguous(), l.pad_size,
                                     l.kernel_size, l.max_displacement, l.stride1, l.stride2, l.corr_multiply)
            off += ch
        return buf

    def _launch_correlations(self, pm, maps, which, B, dev, budget=0):
        """Correlations `which` (indices into conv3 / conv4 / conv5) of the frame pair, written as columns of the tracking
        head's input rows (`pm.tracking_rows`), on the current stream."""
        from .ops import correlation_forward_nhwc, correlation_forward_rows
        layers = (self.conv3_corr_layer, self.conv4_corr_layer, self.conv5_corr_layer)
        hw = maps[2].size(2) * maps[2].size(3)
        rows = pm.tracking_rows(B * hw, dev)
        col, jobs = 2 * pm.n_box, []
        for l, f in zip(layers, maps):
            oc = correlation_output_shape(f.size(1), f.size(2), f.size(3), l.pad_size, l.kernel_size, l.max_displacement,
                                          l.stride1, l.stride2)[0]
            jobs.append((l, f, col))
            col += oc
        assert col == pm.trk_in, "tracking feature width %d != corr_bbox_net input %d" % (col, pm.trk_in)
        for i in which:
            l, f, c0 = jobs[i]
            if f.is_contiguous(memory_format=torch.channels_last) and not f.is_contiguous():
                # channels-last trunk maps: the single-launch window-split kernel, no transposes.  `budget` > 0: the proposal
                # layer's kernels are resident on a few CUs beside this launch -- plan for fewer CUs (more, shorter
                # workgroups in two rounds) instead of exactly one workgroup per CU, which would leave a few workgroups
                # waiting for a whole second round
                correlation_forward_nhwc(f[:B], f[B:2 * B], l.pad_size, l.kernel_size, l.max_displacement, l.stride1,
                                         l.stride2, rows=rows, col=c0, max_workgroups=budget)
            else:
                correlation_forward_rows(rows, c0, f[:B].contiguous(), f[B:2 * B].contiguous(), l.pad_size, l.kernel_size,
                                         l.max_displacement, l.stride1, l.stride2, l.corr_multiply)
        return rows

    def _inference_tail_pm(self, pm, fused, c3, c4, c5, all_rois, side, n_legs, B, dev, top=None, corr_done=()):
        """rfcn.py:133-140, 166-196 at inference on the position-major layout: one MFMA GEMM for the class + box heads of
        every image (`dtt_head_gemm`), lanes = classes PSRoI pooling + vote (`dtt_psroi_pm_forward`), the tracking
        head's input rows assembled in place (box-delta columns copied, correlations written by their reduce kernels)."""
        from .heads import gather_column_blocks, head_gemm, psroi_pm
        top_rows, (H, W) = fused.top_rows, fused.top_hw
        fused.top_rows = None
        cur = torch.cuda.current_stream(dev)
        single_frame = n_legs == 1
        trk = rows = None
        hw = H * W
        if not single_frame:
            # The correlations only need the trunk maps.  conv5 (117 us, the largest) has been issued ahead of the RPN's 1x1
            # heads (forward()), i.e. before the side stream had anything to run: it is dispatched onto an empty chip.  The
            # kernels that DO run beside the proposal layer are the short ones -- conv3 (24 us) and conv4 (77 us) -- whose
            # chain with the tracking head (≈ 155 us) is as long as the side stream's (select / sort, decode, NMS ≈ 150 us),
            # so nothing is lost by taking conv5 out of the overlap.  A one-workgroup-per-CU kernel dispatched while a
            # foreign workgroup sits on one of "its" shader engines has, in some steps, one workgroup parked until a CU of
            # that engine frees up (tools/wg_trace.py, tools/probes/wg_placement.hip): that now costs conv4 ≈ 25 us in
            # some steps instead of conv5 20 - 60.  (env DTT_CORR_ORDER: developer A/B over the order of what is left.)
            idx = [int(c) for c in os.environ.get("DTT_CORR_ORDER", "021") if int(c) not in corr_done]
            rows = self._launch_correlations(pm, (c3, c4, c5), idx, B, dev, budget=int(os.environ.get("DTT_CORR_BUDGET", "240")))
        det, fused.det_rows = getattr(fused, "det_rows", None), None    # (n_legs*B*H*W, stride): issued by the fused trunk ...
        if det is None:
            det = head_gemm(top_rows, pm.det)                           # ... or here
        if not single_frame:
            gather_column_blocks(rows, 0, det, pm.loc_head["offset"], B * hw, n_legs, pm.n_box)   # box deltas of both legs
            trk = head_gemm(rows, pm.trk)                           # (B*H*W, stride)
        R = all_rois.size(1)
        # The poolings need the RoIs as the NMS epilogue wrote them (image index inside the n_legs * B batch): they start as soon
        # as the proposal layer is done.  The per-leg copy the caller gets back (batch index within the leg) is made on the side
        # stream BESIDE them -- it used to sit, with its two small launches and a stream hop, between the NMS and the first pooling.
        rois_ready = torch.cuda.Event()
        rois_ready.record(side)
        with torch.cuda.stream(side):
            leg_rois = all_rois.view(n_legs, B, R, 5).clone()
            for i in range(1, n_legs):
                leg_rois[i, :, :, 0] -= i * B  # batch index within the leg
        cur.wait_event(rois_ready)
        all_rois.record_stream(cur); leg_rois.record_stream(cur)
        flat_rois = all_rois.view(-1, 5)
        scale = self.RFCN_psroi_cls_pool.spatial_scale
        if top is not None:
            self._roi_features(top, flat_rois)
        score = psroi_pm(det, pm.cls_head, n_legs * B, H, W, flat_rois, scale)
        prob = F.softmax(score, dim=1).view(n_legs, B, R, -1)
        pred = psroi_pm(det, pm.loc_head, n_legs * B, H, W, flat_rois, scale).view(n_legs, B, R, -1)
        zeros = torch.zeros(n_legs, 1, device=dev)
        tracking_pred = torch.zeros(0, 4, device=dev)
        if trk is not None:
            # frame-t RoIs (rfcn.py:192): leg 0 of all_rois -- its batch indices are already leg-local
            tracking_pred = psroi_pm(trk, pm.trk_head, B, H, W, all_rois[:B].reshape(-1, 5), scale)
        cur.wait_stream(side)   # leg_rois
        return leg_rois, prob, pred, tracking_pred, zeros, zeros, zeros, zeros, [], zeros[0]

    def forward(self, im_data, im_info, gt_boxes, num_boxes):
        B, n_legs = im_data.size(0), im_data.size(1)
        dev = im_data.device
        # Both legs of the siamese net go through the trunk and the 1x1 heads as ONE batch of n_legs*B images (the
        # reference loops over the legs, rfcn.py:95): BatchNorm is frozen and every op is per-image, so the
        # result is the same, with half the launches and better-filled kernels.
        chw = im_data.shape[2:]
        if getattr(getattr(self, "_fused_trunk", None), "pm_heads", False) and not self.training and im_data.is_cuda:
            # channels-last trunk: leg-major order and channels-last memory in ONE strided copy (instead of two passes)
            flat = torch.empty((n_legs * B, *chw), dtype=im_data.dtype, device=dev, memory_format=torch.channels_last)
            flat.view(n_legs, B, *chw).copy_(im_data.permute(1, 0, 2, 3, 4))
        else:
            flat = im_data.permute(1, 0, 2, 3, 4).contiguous().view(n_legs * B, *chw)  # (n_legs * B, C, H, W)
        im_info = im_info.permute(1, 0, 2).contiguous().detach()
        if self.training:   # (inference never looks at the ground truth: two small copy kernels less per step)
            gt_boxes = gt_boxes.permute(1, 0, 2, 3).contiguous().detach()
            num_boxes = num_boxes.permute(1, 0, 2).contiguous().detach()
        c3, c4, c5, top = self._im_to_head(flat)
        side = None
        if not self.training and top.is_cuda and not torch.is_grad_enabled():
            # The proposal layer (select / sort, decode, NMS mask + sweep) is a handful of small kernels that leave most
            # CUs idle; it runs on a side stream underneath the correlations and the tracking head, which do not depend
            # on it.  Its first kernel -- select / sort: one 1024-thread, 72 KB-LDS workgroup per image, 84 us -- needs
            # the scores only and is started as soon as the softmax is done, while the RPN's box-delta convolution
            # still runs here: when the one-workgroup-per-CU correlation kernels are dispatched it has long been placed.
            # (Dispatched in the same microseconds, the two race for CUs and the loser's workgroups stay parked on a
            # full shader engine until one of ITS CUs frees up: +75 us on a correlation or +90 us on the sort;
            # tools/wg_trace.py, tools/probes/wg_placement.hip.)  The RPN's own convolutions stay on the main stream:
            # beside the correlation kernels they are starved of CUs (50 -> 220 us).
            cur = torch.cuda.current_stream(dev)
            side = getattr(self, "_side_stream", None)
            if side is None or side.device != dev:
                side = self._side_stream = torch.cuda.Stream(device=dev)
            fused = getattr(self, "_fused_trunk", None)
            conv1 = getattr(fused, "rpn_conv1", None)   # set by the channels-last fused trunk during _im_to_head above
            if conv1 is not None:
                fused.rpn_conv1 = None
            rpn = self.RFCN_rpn
            pm_early = getattr(self, "_pm_tail", None)
            corr_done = ()
            if (pm_early is not None and n_legs == 2 and getattr(fused, "top_rows", None) is not None and
                    os.environ.get("DTT_CORR5_EARLY", "1") != "0"):
                self._launch_correlations(pm_early, (c3, c4, c5), (2,), B, dev)   # conv5, on an otherwise empty chip
                corr_done = (2,)
            rpn_rows = getattr(fused, "rpn_rows", None)
            if rpn_rows is not None and pm_early is not None and pm_early.rpn is not None:
                # both 1x1 heads + the pairwise softmax in ONE hand-written launch over the channels-last rows
                # (dtt_rpn_head_gemm: no transpose, no library GEMMs, no bias / softmax kernels), straight into the
                # (B, 2A, H, W) / (B, 4A, H, W) tensors the proposal layer reads
                from .heads import rpn_head_gemm
                fused.rpn_rows = None
                rpn_prob, rpn_bbox = rpn_head_gemm(rpn_rows, pm_early.rpn, n_legs * B, top.size(2), top.size(3))
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    # scores and box deltas arrive together: one dtt_proposal_forward (the ranking kernel decodes the boxes)
                    all_rois = rpn.RPN_proposal((rpn_prob, rpn_bbox, im_info.view(n_legs * B, -1), "TEST"))
            else:
                conv1, rpn_prob = rpn.head_scores(top, conv1)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    selection = rpn.RPN_proposal.select(rpn_prob.detach(), "TEST")
                rpn_bbox = rpn.RPN_bbox_pred(conv1)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    all_rois = rpn.RPN_proposal.finish(selection, rpn_bbox.detach(), im_info.view(n_legs * B, -1), "TEST")
            rpn_prob.record_stream(side); rpn_bbox.record_stream(side)
        leg = lambda t, i: t[i * B:(i + 1) * B]
        pm = getattr(self, "_pm_tail", None)
        if pm is not None and side is not None and n_legs <= 2 and getattr(fused, "top_rows", None) is not None:
            # hand-written heads + position-major pooling (dtt.heads): no NCHW score maps at all
            return self._inference_tail_pm(pm, fused, c3, c4, c5, all_rois, side, n_legs, B, dev, top=top, corr_done=corr_done)
        fused_inf = getattr(self, "_fused_trunk", None)
        if fused_inf is not None and not self.training:
            fused_inf.top_rows = fused_inf.det_rows = fused_inf.rpn_rows = None   # (not consumed: more than two legs take the NCHW graph below)
        train_pm = (self.training and getattr(self, "_train_pm", False) and top.is_cuda and torch.is_grad_enabled()
                    and top.is_contiguous(memory_format=torch.channels_last) and not top.is_contiguous())
        det = det_heads = None
        if train_pm:
            # training on the hand-written heads: one exact-fp32 MFMA GEMM over the channels-last `top` rows
            # for RFCN_cls_net + RFCN_bbox_net of every image (forward, dX and dW all on dtt_head_gemm: dtt.heads.HeadGemmFn),
            # position-major score map, lanes = classes PSRoI pooling with a map-stationary backward (PsroiPmFn)
            from .heads import HeadGemmFn, pack_heads_differentiable, pm_to_nchw
            rows = top.permute(0, 2, 3, 1).reshape(-1, top.size(1))
            w_pk, b_pk, det_heads, n_store, stride = pack_heads_differentiable([self.RFCN_cls_net, self.RFCN_bbox_net])
            det = HeadGemmFn.apply(rows, w_pk, b_pk, n_store, stride)
            cls_maps = None
            bbox_maps = pm_to_nchw(det, det_heads[1], n_legs * B, top.size(2), top.size(3))   # the tracking branch's NCHW concat
        else:
            if top.is_cuda and not top.is_contiguous():
                top = top.contiguous()
            cls_maps = self.RFCN_cls_net(top)
            bbox_maps = self.RFCN_bbox_net(top)
        conv3 = [leg(c3, i) for i in range(n_legs)]
        conv4 = [leg(c4, i) for i in range(n_legs)]
        conv5 = [leg(c5, i) for i in range(n_legs)]
        rfcn_bbox = [leg(bbox_maps, i) for i in range(n_legs)]
        rois, rois_label = [], []
        rpn_loss_cls, rpn_loss_bbox, cls_prob, bbox_pred = [], [], [], []
        loss_cls, loss_bbox = [], []
        tracking_reg = None
        single_frame = n_legs == 1   # BASELINE configs 1-2: plain R-FCN on one frame, no tracking branch
        if not self.training:
            # inference: RPN, proposal layer and PSRoI pooling also run once for all n_legs*B images
            if side is not None:
                if not single_frame:
                    tracking_reg = self.corr_bbox_net(self._tracking_features(rfcn_bbox, conv3, conv4, conv5, whole=(c3, c4, c5)))
                torch.cuda.current_stream(dev).wait_stream(side)
                all_rois.record_stream(torch.cuda.current_stream(dev))
            else:
                all_rois, _, _ = self.RFCN_rpn(top, im_info.view(n_legs * B, -1), None, None)
            R = all_rois.size(1)
            flat_rois = all_rois.view(-1, 5)
            self._roi_features(top, flat_rois)
            score = self._pool_vote(self.RFCN_psroi_cls_pool, self.RFCN_cls_score, cls_maps, flat_rois)
            prob = F.softmax(score, dim=1).view(n_legs, B, R, -1)
            pred = self._pool_vote(self.RFCN_psroi_loc_pool, self.RFCN_bbox_pred, bbox_maps, flat_rois)
            pred = pred.view(n_legs, B, R, -1)
            leg_rois = all_rois.view(n_legs, B, R, 5).clone()
            for i in range(1, n_legs):
                leg_rois[i, :, :, 0] -= i * B  # batch index within the leg
            # everything is already laid out (n_legs, B, R, .): hand the tensors over instead of re-stacking slices
            zeros = torch.zeros(n_legs, 1, device=dev)
            zero = zeros[0]
            tracking_pred = torch.zeros(0, 4, device=dev)
            if not single_frame:
                if tracking_reg is None:
                    tracking_reg = self.corr_bbox_net(self._tracking_features(rfcn_bbox, conv3, conv4, conv5, whole=(c3, c4, c5)))
                # tracking RoIs = frame-t RoIs (rfcn.py:192)
                tracking_pred = self._pool_vote(self.RFCN_psroi_loc_pool, self.RFCN_tracking_pred, tracking_reg,
                                                leg_rois[0].view(-1, 5))
            return leg_rois, prob, pred, tracking_pred, zeros, zeros, zeros, zeros, [], zero
        for i in range(n_legs if self.training else 0):
            # training keeps the reference's per-leg order: anchor-target and RoI sampling draw from numpy's RNG
            top_i, cls_map, bbox_map = leg(top, i), (leg(cls_maps, i) if cls_maps is not None else None), rfcn_bbox[i]
            leg_rois, l_cls, l_box = self.RFCN_rpn(top_i, im_info[i], gt_boxes[i][:, :, :5], num_boxes[i])
            leg_rois, label, target, w_in, w_out = self.RFCN_proposal_target(leg_rois, gt_boxes[i][:, :, :5],
                                                                              num_boxes[i])
            label = label.view(-1).long()
            target = target.view(-1, target.size(2))
            w_in = w_in.view(-1, w_in.size(2))
            w_out = w_out.view(-1, w_out.size(2))
            rois_label.append(label)
            rois.append(leg_rois)
            rpn_loss_cls.append(l_cls.view(1)); rpn_loss_bbox.append(l_box.view(1))
            flat_rois = leg_rois.view(-1, 5)
            if getattr(self._cfg, "RFCN_ROI_FEATURES", ""):
                feats = (feats if i else []) + [self._roi_features(top_i.detach(), flat_rois)]
                self.roi_feat = feats
            if det is not None:
                from .heads import PsroiPmFn
                rois_all = flat_rois.detach().clone()
                rois_all[:, 0] += i * B                     # batch index inside the (n_legs * B)-image score map
                score, pred = PsroiPmFn.apply(det, rois_all, n_legs * B, top.size(2), top.size(3),
                                              self.RFCN_psroi_cls_pool.spatial_scale, det_heads)
            else:
                score = self._pool_vote(self.RFCN_psroi_cls_pool, self.RFCN_cls_score, cls_map, flat_rois)
                pred = self._pool_vote(self.RFCN_psroi_loc_pool, self.RFCN_bbox_pred, bbox_map, flat_rois)
            prob = F.softmax(score, dim=1)
            if not self.class_agnostic:
                pv = pred.view(pred.size(0), int(pred.size(1) / 4), 4)
                pred = torch.gather(pv, 1, label.view(-1, 1, 1).expand(label.size(0), 1, 4)).squeeze(1)
            loss_cls.append(F.cross_entropy(score, label).view(1))
            loss_bbox.append(_smooth_l1_loss(pred, target, w_in, w_out).view(1))
            cls_prob.append(prob.view(B, leg_rois.size(1), -1))
            bbox_pred.append(pred.view(B, leg_rois.size(1), -1))

        if single_frame:
            zero = torch.zeros(1, device=dev)
            rois = torch.stack(rois, 0)
            if rois_label:
                rois_label = torch.stack(rois_label, 0).view(n_legs, B, -1)
            return (rois, torch.stack(cls_prob, 0), torch.stack(bbox_pred, 0), torch.zeros(0, 4, device=dev),
                    torch.stack(rpn_loss_cls, 0), torch.stack(rpn_loss_bbox, 0), torch.stack(loss_cls, 0),
                    torch.stack(loss_bbox, 0), rois_label, zero)
        if tracking_reg is None:
            tracking_reg = self.corr_bbox_net(self._tracking_features(rfcn_bbox, conv3, conv4, conv5, whole=(c3, c4, c5)))
        if self.training:
            trk_rois, trk_label, trk_target, trk_in, trk_out = self.RFCN_tracking_proposal_target(gt_boxes, num_boxes)
            trk_target = trk_target.view(-1, trk_target.size(2))
            trk_in = trk_in.view(-1, trk_in.size(2))
            trk_out = trk_out.view(-1, trk_out.size(2))
        else:
            trk_rois = rois[0].clone()  # tracking RoIs = frame-t RoIs (rfcn.py:192)
        tracking_pred = self._pool_vote(self.RFCN_psroi_loc_pool, self.RFCN_tracking_pred, tracking_reg,
                                        trk_rois.contiguous().view(-1, 5))
        if self.training:
            tracking_loss = _smooth_l1_loss(tracking_pred, trk_target, trk_in, trk_out)
        else:
            tracking_loss = torch.zeros(1, device=dev)
        rois = torch.stack(rois, 0)
        cls_prob = torch.stack(cls_prob, 0)
        bbox_pred = torch.stack(bbox_pred, 0)
        if rois_label:
            rois_label = torch.stack(rois_label, 0).view(n_legs, B, -1)
        return (rois, cls_prob, bbox_pred, tracking_pred, torch.stack(rpn_loss_cls, 0), torch.stack(rpn_loss_bbox, 0),
                torch.stack(loss_cls, 0), torch.stack(loss_bbox, 0), rois_label, tracking_loss)

    def _init_weights(self):
        if not getattr(self, "pretrained_rfcn", False):
            for m in (self.RFCN_rpn.RPN_Conv, self.RFCN_rpn.RPN_cls_score, self.RFCN_rpn.RPN_bbox_pred):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()

    def create_architecture(self):
        self._init_modules()
        self._init_weights()


class resnet(_RFCN):
    """faster_rcnn/resnet.py:247-345."""

    def __init__(self, classes, num_layers=101, pretrained=False, pretrained_rfcn=False, class_agnostic=False, cfg=None):
        self.model_path = "data/pretrained_model/res101.pth"
        self.model_rfcn_path = "data/pretrained_model/rfcn_detect.pth"
        self.dout_base_model = 512
        self.num_layers = num_layers
        self.pretrained = pretrained
        self.pretrained_rfcn = pretrained_rfcn
        super().__init__(classes, class_agnostic, cfg=cfg)

    def _init_modules(self):
        base = _trunk(self.num_layers)
        if self.pretrained:
            sd = torch.load(self.model_path, map_location="cpu")
            names = ["conv1", "bn1", None, None, "layer1", "layer2", "layer3", "layer4"]
            own = base.state_dict()
            remap = {}
            for k, v in sd.items():
                head, _, rest = k.partition(".")
                if head in names:
                    nk = "%d.%s" % (names.index(head), rest)
                    if nk in own:
                        remap[nk] = v
            base.load_state_dict(remap, strict=False)
        self.RFCN_base = base
        for idx in (0, 1):
            for p in self.RFCN_base[idx].parameters():
                p.requires_grad = False
        fixed = self._cfg.RESNET.FIXED_BLOCKS
        assert 0 <= fixed < 4
        for blk, idx in ((3, 6), (2, 5), (1, 4)):
            if fixed >= blk:
                for p in self.RFCN_base[idx].parameters():
                    p.requires_grad = False
        for m in self.RFCN_base.modules():
            if isinstance(m, nn.BatchNorm2d):
                for p in m.parameters():
                    p.requires_grad = False
        # position-sensitive feature conv: 3x3, dilation 6 (resnet.py:296-301); registered under both names
        self.RFCN_net = nn.Conv2d(2048, 512, kernel_size=3, padding=6, stride=1, dilation=6)
        self.RFCN_base.add_module("RFCN_net", self.RFCN_net)
        self.RFCN_base.add_module("resnet", nn.ReLU(inplace=True))
        nn.init.kaiming_normal_(self.RFCN_net.weight)
        if self.pretrained_rfcn:
            sd = torch.load(self.model_rfcn_path, map_location="cpu")["model"]
            own = self.state_dict()
            self.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
        d = int(getattr(self._cfg, "CORR_MAX_DISPLACEMENT", 8))
        d3, d45 = (2 * (d // 2) + 1) ** 2, (2 * d + 1) ** 2
        tracking_in = 2 * 4 * self.n_reg_classes * 49 + d3 + 2 * d45  # 392 + 81 + 289 + 289 = 1051 (resnet.py:311)
        self.corr_bbox_net = nn.Conv2d(tracking_in, 4 * self.n_reg_classes * 7 * 7, [1, 1], padding=0, stride=1)
        nn.init.normal_(self.corr_bbox_net.weight, 0.0, 0.01)

    def train(self, mode=True):
        nn.Module.train(self, mode)
        if mode:
            self.RFCN_base.eval()
            for idx in (5, 6, 7, 8):
                self.RFCN_base[idx].train()
            for m in self.RFCN_base.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self

    def _im_to_head(self, x):
        fused = getattr(self, "_fused_trunk", None)
        if fused is not None and not self.training and not torch.is_grad_enabled() and x.is_cuda:
            return fused(x)  # dtt.fuse: BatchNorm folded, bias + residual + ReLU in one HIP pass
        fused_train = getattr(self, "_fused_train_trunk", None)
        if fused_train is not None and self.training and torch.is_grad_enabled() and x.is_cuda:
            return fused_train(x)
        b = self.RFCN_base
        x = b[3](b[2](b[1](b[0](x))))
        conv3 = b[5](b[4](x))
        conv4 = b[6](conv3)
        conv5 = b[7](conv4)
        top = b[9](b[8](conv5))
        return conv3, conv4, conv5, top


IMAGENET_VID_CLASSES = ["__background__"] + ["class_%d" % i for i in range(1, 31)]
