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
            score = cls_score_r.permute(0, 2, 3, 1).contiguous().view(-1, 2)
            # rpn.py:90-97 keeps the sampled anchors with nonzero() + index_select: a host read of their number in the middle of the
            # forward.  The mean over the anchors whose label is not -1 is the same number (ignore_index), without it.
            self.rpn_loss_cls = F.cross_entropy(score, labels.view(-1).long(), ignore_index=-1)
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

    def _inference_tail_pm(self, pm, ex, c3, c4, c5, all_rois, side, n_legs, B, dev, top=None, corr_done=()):
        """rfcn.py:133-140, 166-196 at inference on the position-major layout: one MFMA GEMM for the class + box heads of
        every image (`dtt_head_gemm`), lanes = classes PSRoI pooling + vote (`dtt_psroi_pm_forward`), the tracking
        head's input rows assembled in place (box-delta columns copied, correlations written by their kernels).
        ex: the fused trunk's TrunkExtras (channels-last `top` rows, the early head GEMM's output).
        The four loss outputs and the tracking loss of the returned 10-tuple are views of ONE cached zero tensor (inference has no
        losses; allocating and filling them cost two launches per step): read-only -- a caller that wants to accumulate into them
        must clone first."""
        from .heads import gather_column_blocks, head_gemm, psroi_pm
        top_rows, (H, W) = ex.top_rows, ex.top_hw
        cur = torch.cuda.current_stream(dev)
        single_frame = n_legs == 1
        trk = rows = None
        hw = H * W
        if not single_frame:
            # The correlations only need the trunk maps.  conv5 (the largest) has been issued ahead of the RPN's 1x1
            # heads (_infer_proposals), i.e. before the side stream had anything to run: it is dispatched onto an empty chip.  The
            # kernels that DO run beside the proposal layer are the short ones -- conv3 and conv4 -- whose
            # chain with the tracking head is as long as the side stream's (selection, decode, NMS),
            # so nothing is lost by taking conv5 out of the overlap.  A one-workgroup-per-CU kernel dispatched while a
            # foreign workgroup sits on one of "its" shader engines has, in some steps, one workgroup parked until a CU of
            # that engine frees up (tools/wg_trace.py, tools/probes/wg_placement.hip): that now costs conv4 ≈ 25 us in
            # some steps instead of conv5 20 - 60.  (env DTT_CORR_ORDER: developer A/B over the order of what is left.)
            idx = [int(c) for c in os.environ.get("DTT_CORR_ORDER", "021") if int(c) not in corr_done]
            rows = self._launch_correlations(pm, (c3, c4, c5), idx, B, dev, budget=int(os.environ.get("DTT_CORR_BUDGET", "240")))
        det = ex.det_rows                                               # (n_legs*B*H*W, stride): issued by the fused trunk ...
        if det is None:
            det = head_gemm(top_rows, pm.det)                           # ... or here
        R = all_rois.size(1)
        scale = self.RFCN_psroi_cls_pool.spatial_scale
        fused_det = (pm.cls_head["cp"] == 32 and pm.loc_head["cp"] == 4 and os.environ.get("DTT_PSROI_DET_FUSED", "1") != "0")
        det_on_side = fused_det and not single_frame and os.environ.get("DTT_PSROI_DET_SIDE", "0") == "1"   # developer A/B (measured: no gain, below)
        prob = pred = None
        if det_on_side:
            # The detection pooling needs the score map and the RoIs, not the tracking head: on the SIDE stream it could run beside
            # the tracking head's GEMM (32 us, 40 CUs idle) instead of behind it.  Measured (round 4, rocprofv3 of the bench step):
            # the pooling still starts 8 us after the tracking head ends and 22 us after the second sweep, and the nodes behind it
            # pay 7 - 14 us each -- 238.9 - 239.2 frame-pairs/s against 238.3 - 239.8 on the same box.  Off by default: the tail
            # below stays on one stream after the proposal layer has joined.
            from .heads import psroi_pm_det
            det_ready = torch.cuda.Event()
            det_ready.record(cur)
            side.wait_event(det_ready)
            with torch.cuda.stream(side):
                prob, pred = psroi_pm_det(det, pm.cls_head, pm.loc_head, n_legs * B, H, W, all_rois.view(-1, 5), scale)
            det.record_stream(side); prob.record_stream(cur); pred.record_stream(cur)
            prob, pred = prob.view(n_legs, B, R, -1), pred.view(n_legs, B, R, -1)
        if not single_frame:
            gather_column_blocks(rows, 0, det, pm.loc_head["offset"], B * hw, n_legs, pm.n_box)   # box deltas of both legs
            trk = head_gemm(rows, pm.trk)                           # (B*H*W, stride)
        # The poolings need the RoIs as the NMS epilogue wrote them (image index inside the n_legs * B batch): they start as soon
        # as the proposal layer is done.  The per-leg copy the caller gets back (batch index within the leg) is ONE elementwise
        # launch behind the poolings (it was a clone -- a runtime copy launch -- and an in-place subtract on the side
        # stream in front of them).  What remains between the last of {second sweep, tracking head} and the detection pooling is
        # the join of the two queues itself: 20 - 28 us in every trace of the round (profiles/r04_bench_step_sequence.txt), with
        # the copy in front, behind, or on the other stream.
        rois_ready = torch.cuda.Event()
        rois_ready.record(side)
        cur.wait_event(rois_ready)
        all_rois.record_stream(cur)
        flat_rois = all_rois.view(-1, 5)
        if top is not None:
            self._roi_features(top, flat_rois)
        if prob is not None:
            pass                                                        # (issued on the side stream above)
        elif fused_det:
            # class scores + box deltas of a RoI in ONE launch, the class softmax in its epilogue (was: pooling 17 + softmax 3.6 +
            # pooling 10.3 us in a row, the second pooling re-reading the rows and RoIs the first had just read)
            from .heads import psroi_pm_det
            prob, pred = psroi_pm_det(det, pm.cls_head, pm.loc_head, n_legs * B, H, W, flat_rois, scale)
            prob, pred = prob.view(n_legs, B, R, -1), pred.view(n_legs, B, R, -1)
        else:
            score = psroi_pm(det, pm.cls_head, n_legs * B, H, W, flat_rois, scale)
            prob = F.softmax(score, dim=1).view(n_legs, B, R, -1)
            pred = psroi_pm(det, pm.loc_head, n_legs * B, H, W, flat_rois, scale).view(n_legs, B, R, -1)
        # (the zero "losses" of an inference step, rfcn.py:125-126: one cached read-only tensor instead of a fill launch per step)
        zeros = self._leg_offsets(n_legs, 0, dev)[:, 0, 0, :1]             # (B = 0: all zeros) -> (n_legs, 1)
        tracking_pred = torch.zeros(0, 4, device=dev)
        if trk is not None:
            # frame-t RoIs (rfcn.py:192): leg 0 of all_rois -- its batch indices are already leg-local
            tracking_pred = psroi_pm(trk, pm.trk_head, B, H, W, all_rois[:B].reshape(-1, 5), scale)
        leg_rois = all_rois.view(n_legs, B, R, 5) - self._leg_offsets(n_legs, B, dev)   # batch index within the leg
        cur.wait_stream(side)   # (the side-stream pooling of the A/B switch)
        return leg_rois, prob, pred, tracking_pred, zeros, zeros, zeros, zeros, [], zeros[0]

    def _leg_offsets(self, n_legs, B, dev):
        """(n_legs, 1, 1, 5) with i * B in column 0 of leg i: all_rois' image index -> the index within the leg (cached: no fill
        launches per step)."""
        key = (n_legs, B, str(dev))
        cache = self.__dict__.setdefault("_leg_offsets_cache", {})
        if key not in cache:
            offs = torch.zeros(n_legs, 1, 1, 5)
            offs[:, 0, 0, 0] = torch.arange(n_legs, dtype=torch.float32) * B
            cache[key] = offs.to(dev)
        return cache[key]

    # ------------------------------------------------------------------------------------------------ forward: four graphs
    # `forward` flattens the legs, runs the trunk and picks one of four graph builders (rfcn.py:66-250 is the contract of all):
    #   _infer_proposals + _inference_tail_pm    inference, channels-last fused trunk, hand-written heads, position-major pooling
    #   _infer_proposals + _inference_tail_nchw  inference on NCHW maps (plain / NCHW-fused trunk, more than two legs, CPU)
    #   _forward_train_pm                        training on the hand-written heads (channels-last fused training trunk)
    #   _forward_train_nchw                      training, the reference's graph on library convolutions + the NCHW operators
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
        c3, c4, c5, top, ex = self._im_to_head_ex(flat)
        if self.training:
            # (the hand-written RPN heads pack (bg, fg) pairs two to a 4-row group: an even anchor count.  The default 9-anchor
            #  configurations of the non-imagenet datasets train on the library graph, as their inference falls back in PositionMajorTail.)
            train_pm = (getattr(self, "_train_pm", False) and top.is_cuda and torch.is_grad_enabled() and n_legs <= 2
                        and top.is_contiguous(memory_format=torch.channels_last) and not top.is_contiguous()
                        and self.RFCN_rpn.RPN_cls_score.weight.shape[0] % 4 == 0)
            build = self._forward_train_pm if train_pm else self._forward_train_nchw
            return build(c3, c4, c5, top, im_info, gt_boxes, num_boxes, n_legs, B, dev)
        side = all_rois = None
        corr_done = ()
        if top.is_cuda and not torch.is_grad_enabled():
            all_rois, side, corr_done = self._infer_proposals(ex, c3, c4, c5, top, im_info, n_legs, B, dev)
        pm = getattr(self, "_pm_tail", None)
        if pm is not None and side is not None and n_legs <= 2 and ex is not None and ex.top_rows is not None:
            # hand-written heads + position-major pooling (dtt.heads): no NCHW score maps at all
            return self._inference_tail_pm(pm, ex, c3, c4, c5, all_rois, side, n_legs, B, dev, top=top, corr_done=corr_done)
        return self._inference_tail_nchw(c3, c4, c5, top, all_rois, side, im_info, n_legs, B, dev)

    def _infer_proposals(self, ex, c3, c4, c5, top, im_info, n_legs, B, dev):
        """The RPN heads on the main stream and the proposal layer on a side stream -> (all_rois, side stream, correlations
        already issued).  The proposal layer (selection, decode, NMS mask + sweep) is a handful of small kernels that leave most
        CUs idle; it runs underneath the correlations and the tracking head, which do not depend on it.  (Dispatched in the
        same microseconds as a one-workgroup-per-CU kernel, the two race for CUs and the loser's workgroups stay parked on a
        full shader engine until one of ITS CUs frees up: tools/wg_trace.py, tools/probes/wg_placement.hip.)  The RPN's own
        convolutions stay on the main stream: beside the correlation kernels they are starved of CUs."""
        cur = torch.cuda.current_stream(dev)
        side = getattr(self, "_side_stream", None)
        if side is None or side.device != dev:
            side = self._side_stream = torch.cuda.Stream(device=dev)
        rpn = self.RFCN_rpn
        pm = getattr(self, "_pm_tail", None)
        conv1 = ex.rpn_conv1 if ex is not None else None          # relu(RPN_Conv(top)) when the fused trunk has computed it
        rpn_rows = ex.rpn_rows if ex is not None else None
        corr_done = ()
        if (pm is not None and n_legs == 2 and ex is not None and ex.top_rows is not None and
                os.environ.get("DTT_CORR5_EARLY", "1") != "0"):
            self._launch_correlations(pm, (c3, c4, c5), (2,), B, dev)   # conv5, on an otherwise empty chip
            corr_done = (2,)
        if rpn_rows is not None and pm is not None and pm.rpn is not None:
            # both 1x1 heads + the pairwise softmax in ONE hand-written launch over the channels-last rows
            # (dtt_rpn_head_gemm: no transpose, no library GEMMs, no bias / softmax kernels), straight into the
            # (B, 2A, H, W) / (B, 4A, H, W) tensors the proposal layer reads
            from .heads import rpn_head_gemm
            rpn_prob, rpn_bbox = rpn_head_gemm(rpn_rows, pm.rpn, n_legs * B, top.size(2), top.size(3))
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                # scores and box deltas arrive together: one dtt_proposal_forward (the ranking kernel decodes the boxes)
                all_rois = rpn.RPN_proposal((rpn_prob, rpn_bbox, im_info.view(n_legs * B, -1), "TEST"))
        else:
            # library convolutions: the selection needs the scores only and starts under the box-delta convolution
            conv1, rpn_prob = rpn.head_scores(top, conv1)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                selection = rpn.RPN_proposal.select(rpn_prob.detach(), "TEST")
            rpn_bbox = rpn.RPN_bbox_pred(conv1)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                all_rois = rpn.RPN_proposal.finish(selection, rpn_bbox.detach(), im_info.view(n_legs * B, -1), "TEST")
        rpn_prob.record_stream(side); rpn_bbox.record_stream(side)
        return all_rois, side, corr_done

    def _inference_tail_nchw(self, c3, c4, c5, top, all_rois, side, im_info, n_legs, B, dev):
        """Inference on NCHW score maps: library 1x1 heads, `psroi_vote`, the tracking concat written by the correlations'
        reduce kernels; RPN, proposal layer and PSRoI pooling run once for all n_legs * B images."""
        leg = lambda t, i: t[i * B:(i + 1) * B]
        if top.is_cuda and not top.is_contiguous():
            top = top.contiguous()
        cls_maps = self.RFCN_cls_net(top)
        bbox_maps = self.RFCN_bbox_net(top)
        conv3, conv4, conv5 = ([leg(c, i) for i in range(n_legs)] for c in (c3, c4, c5))
        rfcn_bbox = [leg(bbox_maps, i) for i in range(n_legs)]
        single_frame = n_legs == 1   # BASELINE configs 1-2: plain R-FCN on one frame, no tracking branch
        tracking_reg = None
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
        tracking_pred = torch.zeros(0, 4, device=dev)
        if not single_frame:
            if tracking_reg is None:
                tracking_reg = self.corr_bbox_net(self._tracking_features(rfcn_bbox, conv3, conv4, conv5, whole=(c3, c4, c5)))
            # tracking RoIs = frame-t RoIs (rfcn.py:192)
            tracking_pred = self._pool_vote(self.RFCN_psroi_loc_pool, self.RFCN_tracking_pred, tracking_reg,
                                            leg_rois[0].view(-1, 5))
        return leg_rois, prob, pred, tracking_pred, zeros, zeros, zeros, zeros, [], zeros[0]

    def _leg_losses(self, i, B, score, pred, label, target, w_in, w_out, out):
        """rfcn.py:142-160 for one leg: class gather (class-specific boxes), the two R-FCN losses, the per-leg outputs."""
        prob = F.softmax(score, dim=1)
        if not self.class_agnostic:
            pv = pred.view(pred.size(0), int(pred.size(1) / 4), 4)
            pred = torch.gather(pv, 1, label.view(-1, 1, 1).expand(label.size(0), 1, 4)).squeeze(1)
        out["loss_cls"].append(F.cross_entropy(score, label).view(1))
        out["loss_bbox"].append(_smooth_l1_loss(pred, target, w_in, w_out).view(1))
        out["cls_prob"].append(prob.view(B, -1, prob.size(1)))
        out["bbox_pred"].append(pred.view(B, -1, pred.size(1)))

    def _train_outputs(self, out, n_legs, B, tracking_pred, tracking_loss):
        rois = torch.stack(out["rois"], 0)
        rois_label = torch.stack(out["rois_label"], 0).view(n_legs, B, -1) if out["rois_label"] else []
        return (rois, torch.stack(out["cls_prob"], 0), torch.stack(out["bbox_pred"], 0), tracking_pred,
                torch.stack(out["rpn_loss_cls"], 0), torch.stack(out["rpn_loss_bbox"], 0), torch.stack(out["loss_cls"], 0),
                torch.stack(out["loss_bbox"], 0), rois_label, tracking_loss)

    @staticmethod
    def _new_out():
        return {k: [] for k in ("rois", "rois_label", "rpn_loss_cls", "rpn_loss_bbox", "cls_prob", "bbox_pred", "loss_cls", "loss_bbox")}

    def _forward_train_pm(self, c3, c4, c5, top, im_info, gt_boxes, num_boxes, n_legs, B, dev):
        """Training on the hand-written heads (rfcn.py:95-250, rpn/rpn.py:58-107).  Every 1x1 head of the
        graph is the exact-fp32 MFMA GEMM with its own backward (dtt.heads: HeadGemmFn = forward + dX on dtt_head_gemm, dW on
        dtt_head_gemm_dw): RFCN_cls_net + RFCN_bbox_net of all legs in one launch, the RPN's two heads + pairwise softmax in one
        launch (RpnHeadFn), corr_bbox_net over tracking rows that the correlations write in place (TrackingRowsFn: no 1051-channel
        concat, no NCHW copies; the correlation gradient kernels read their columns of the rows' gradient).  The proposal layer
        runs once for both legs; anchor-target and RoI sampling keep the reference's per-leg order (they draw from numpy's RNG);
        PSRoI pooling of both legs is one position-major launch pair with a map-stationary backward (PsroiPmFn)."""
        from .heads import (HeadGemmFn, PsroiPmFn, RpnHeadFn, RpnLossFn, TrackingRowsFn, pack_heads_differentiable,
                            pack_rpn_heads_differentiable)
        rpn = self.RFCN_rpn
        H, W = top.size(2), top.size(3)
        scale = self.RFCN_psroi_cls_pool.spatial_scale
        rows = top.permute(0, 2, 3, 1).reshape(-1, top.size(1))           # (a view of the channels-last map)
        w_pk, b_pk, det_heads, n_store, stride = pack_heads_differentiable([self.RFCN_cls_net, self.RFCN_bbox_net])
        det = HeadGemmFn.apply(rows, w_pk, b_pk, n_store, stride)
        # ---- RPN: 3x3 convolution and both heads once for all legs
        conv1 = F.relu(rpn.RPN_Conv(top), inplace=True)
        if not (conv1.is_contiguous(memory_format=torch.channels_last) and not conv1.is_contiguous()):
            conv1 = conv1.contiguous(memory_format=torch.channels_last)
        w_rpn, b_rpn, A = pack_rpn_heads_differentiable(rpn.RPN_cls_score, rpn.RPN_bbox_pred)
        # (the class loss's gradient comes back with respect to the score logits: RpnLossFn below is cls_prob's only differentiable consumer)
        rpn_prob, rpn_bbox = RpnHeadFn.apply(conv1.permute(0, 2, 3, 1).reshape(-1, conv1.size(1)), w_rpn, b_rpn, A, n_legs * B, H, W, True)
        if os.environ.get("DTT_TRAIN_PROPOSALS_MERGED", "1") != "0":   # (env: developer A/B switch)
            all_rois = rpn.proposals(rpn_prob, rpn_bbox, im_info.view(n_legs * B, -1))   # (n_legs * B, post, 5), image index in column 0
        else:
            all_rois = torch.cat([rpn.proposals(rpn_prob[i * B:(i + 1) * B], rpn_bbox[i * B:(i + 1) * B], im_info[i]) for i in range(n_legs)], 0)
            for i in range(1, n_legs):
                all_rois[i * B:(i + 1) * B, :, 0] += i * B
        out = self._new_out()
        sampled = []
        # the anchor-target layer's outputs of all legs in four tensors (each leg's call writes its slice: the reference's per-leg order
        # of RNG draws stays), read by ONE launch of the hand-written RPN losses behind the loop
        f32 = dict(dtype=torch.float32, device=dev)
        at_all = [torch.empty((n_legs * B, 1, A * H, W), **f32)] + [torch.empty((n_legs * B, 4 * A, H, W), **f32) for _ in range(3)]
        for i in range(n_legs):
            sl = slice(i * B, (i + 1) * B)
            rpn.RPN_anchor_target((rpn_prob[sl].detach(), gt_boxes[i][:, :, :5], im_info[i], num_boxes[i]), out=[t[sl] for t in at_all])
            leg_rois = all_rois[sl].clone()
            leg_rois[:, :, 0] -= i * B                                    # batch index within the leg, as the reference's per-leg RPN
            leg_rois, label, target, w_in, w_out = self.RFCN_proposal_target(leg_rois, gt_boxes[i][:, :, :5], num_boxes[i])
            label = label.view(-1).long()
            sampled.append((label, target.view(-1, target.size(2)), w_in.view(-1, w_in.size(2)), w_out.view(-1, w_out.size(2))))
            out["rois_label"].append(label)
            out["rois"].append(leg_rois)
            if getattr(self._cfg, "RFCN_ROI_FEATURES", ""):
                feats = (feats if i else []) + [self._roi_features(top[sl].detach(), leg_rois.view(-1, 5))]
                self.roi_feat = feats
        rpn_losses = RpnLossFn.apply(rpn_prob, rpn_bbox, *at_all, n_legs, 3.0)      # rpn.py:86-105 (sigma = 3)
        for i in range(n_legs):
            out["rpn_loss_cls"].append(rpn_losses[i:i + 1]); out["rpn_loss_bbox"].append(rpn_losses[n_legs + i:n_legs + i + 1])
        rpn.rpn_loss_cls, rpn.rpn_loss_box = rpn_losses[n_legs - 1], rpn_losses[2 * n_legs - 1]   # (the module attributes: the last leg's, as the reference's)
        # ---- PSRoI pooling + vote of both legs over the one position-major map (one gradient map comes back)
        rois_all = torch.cat([r.detach().reshape(-1, 5) for r in out["rois"]], 0).clone()
        n_per = out["rois"][0].size(0) * out["rois"][0].size(1)
        for i in range(1, n_legs):
            rois_all[i * n_per:(i + 1) * n_per, 0] += i * B               # batch index inside the (n_legs * B)-image map
        single_frame = n_legs == 1
        loc = det_heads[1]
        n_box = loc["group"] * loc["group"] * loc["cp"]
        pooled = PsroiPmFn.apply(det, rois_all, n_legs * B, H, W, scale, det_heads, None if single_frame else (loc["offset"], n_box))
        score_all, pred_all = pooled[0], pooled[1]
        for i in range(n_legs):
            label, target, w_in, w_out = sampled[i]
            self._leg_losses(i, B, score_all[i * n_per:(i + 1) * n_per], pred_all[i * n_per:(i + 1) * n_per], label, target, w_in, w_out, out)
        if single_frame:
            return self._train_outputs(out, n_legs, B, torch.zeros(0, 4, device=dev), torch.zeros(1, device=dev))
        # ---- tracking branch (rfcn.py:166-196) on position-major rows
        layers = (self.conv3_corr_layer, self.conv4_corr_layer, self.conv5_corr_layer)
        geoms = tuple((l.pad_size, l.kernel_size, l.max_displacement, l.stride1, l.stride2) for l in layers)
        K_in = self.corr_bbox_net.weight.shape[1]
        k_pad = -(-K_in // 32) * 32
        od, G = loc["od"], loc["group"]
        def tracking_perm():   # rows column leg*n_box + bin*od + k  <-  reference channel leg*n_box + k*G*G + bin
            perm = torch.arange(K_in)
            bb, kk = torch.meshgrid(torch.arange(G * G), torch.arange(od), indexing="ij")
            for l in range(2):
                perm[l * n_box:(l + 1) * n_box] = (l * n_box + kk * G * G + bb).reshape(-1)
            return perm.to(dev)
        from .heads import _device_constant
        perm = _device_constant(("tracking_perm", K_in, G, od, n_box, str(dev)), tracking_perm)   # (uploaded once, not per step)
        trk_rows = TrackingRowsFn.apply(pooled[2], c3, c4, c5, B, geoms, k_pad)
        w_trk, b_trk, trk_heads, n_store_t, stride_t = pack_heads_differentiable([self.corr_bbox_net], k_pad=k_pad, in_perm=perm)
        trk = HeadGemmFn.apply(trk_rows, w_trk, b_trk, n_store_t, stride_t)
        trk_rois, trk_label, trk_target, trk_in, trk_out = self.RFCN_tracking_proposal_target(gt_boxes, num_boxes)
        (tracking_pred,) = PsroiPmFn.apply(trk, trk_rois.contiguous().view(-1, 5), B, H, W, scale, trk_heads, None)
        tracking_loss = _smooth_l1_loss(tracking_pred, trk_target.view(-1, trk_target.size(2)), trk_in.view(-1, trk_in.size(2)),
                                        trk_out.view(-1, trk_out.size(2)))
        return self._train_outputs(out, n_legs, B, tracking_pred, tracking_loss)

    def _forward_train_nchw(self, c3, c4, c5, top, im_info, gt_boxes, num_boxes, n_legs, B, dev):
        """The reference's training graph (rfcn.py:95-250) on library 1x1 convolutions and the NCHW operators with autograd
        (PSRoIPoolFunction + AvgPool2d, torch.cat of the tracking features): any trunk, any device the operators run on."""
        leg = lambda t, i: t[i * B:(i + 1) * B]
        if top.is_cuda and not top.is_contiguous():
            top = top.contiguous()
        cls_maps = self.RFCN_cls_net(top)
        bbox_maps = self.RFCN_bbox_net(top)
        conv3, conv4, conv5 = ([leg(c, i) for i in range(n_legs)] for c in (c3, c4, c5))
        rfcn_bbox = [leg(bbox_maps, i) for i in range(n_legs)]
        out = self._new_out()
        for i in range(n_legs):
            # the reference's per-leg order: anchor-target and RoI sampling draw from numpy's RNG
            leg_rois, l_cls, l_box = self.RFCN_rpn(leg(top, i), im_info[i], gt_boxes[i][:, :, :5], num_boxes[i])
            leg_rois, label, target, w_in, w_out = self.RFCN_proposal_target(leg_rois, gt_boxes[i][:, :, :5], num_boxes[i])
            label = label.view(-1).long()
            out["rois_label"].append(label)
            out["rois"].append(leg_rois)
            out["rpn_loss_cls"].append(l_cls.view(1)); out["rpn_loss_bbox"].append(l_box.view(1))
            flat_rois = leg_rois.view(-1, 5)
            if getattr(self._cfg, "RFCN_ROI_FEATURES", ""):
                feats = (feats if i else []) + [self._roi_features(leg(top, i).detach(), flat_rois)]
                self.roi_feat = feats
            score = self._pool_vote(self.RFCN_psroi_cls_pool, self.RFCN_cls_score, leg(cls_maps, i), flat_rois)
            pred = self._pool_vote(self.RFCN_psroi_loc_pool, self.RFCN_bbox_pred, rfcn_bbox[i], flat_rois)
            self._leg_losses(i, B, score, pred, label, target.view(-1, target.size(2)), w_in.view(-1, w_in.size(2)),
                             w_out.view(-1, w_out.size(2)), out)
        if n_legs == 1:
            return self._train_outputs(out, n_legs, B, torch.zeros(0, 4, device=dev), torch.zeros(1, device=dev))
        tracking_reg = self.corr_bbox_net(self._tracking_features(rfcn_bbox, conv3, conv4, conv5, whole=(c3, c4, c5)))
        trk_rois, trk_label, trk_target, trk_in, trk_out = self.RFCN_tracking_proposal_target(gt_boxes, num_boxes)
        tracking_pred = self._pool_vote(self.RFCN_psroi_loc_pool, self.RFCN_tracking_pred, tracking_reg,
                                        trk_rois.contiguous().view(-1, 5))
        tracking_loss = _smooth_l1_loss(tracking_pred, trk_target.view(-1, trk_target.size(2)), trk_in.view(-1, trk_in.size(2)),
                                        trk_out.view(-1, trk_out.size(2)))
        return self._train_outputs(out, n_legs, B, tracking_pred, tracking_loss)

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
        """(conv3, conv4, conv5, top) of resnet.py:334-345 -- on the fused trunk when one is attached (dtt.fuse)."""
        return self._im_to_head_ex(x)[:4]

    def _im_to_head_ex(self, x):
        """_im_to_head plus what the fused inference trunk computes on the way for the tail (dtt.fuse.TrunkExtras: the
        channels-last rows of `top` and of relu(RPN_Conv(top)), the early head GEMM's output) -- None on the other trunks."""
        fused = getattr(self, "_fused_trunk", None)
        if fused is not None and not self.training and not torch.is_grad_enabled() and x.is_cuda:
            res = fused(x)  # dtt.fuse: BatchNorm folded, bias + residual + ReLU in one HIP pass
            return res if len(res) == 5 else (*res, None)
        fused_train = getattr(self, "_fused_train_trunk", None)
        if fused_train is not None and self.training and torch.is_grad_enabled() and x.is_cuda:
            return (*fused_train(x), None)
        b = self.RFCN_base
        x = b[3](b[2](b[1](b[0](x))))
        conv3 = b[5](b[4](x))
        conv4 = b[6](conv3)
        conv5 = b[7](conv4)
        top = b[9](b[8](conv5))
        return conv3, conv4, conv5, top, None


IMAGENET_VID_CLASSES = ["__background__"] + ["class_%d" % i for i in range(1, 31)]
