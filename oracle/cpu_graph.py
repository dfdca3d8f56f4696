"""CPU restatement of the D&T test-time forward (reference faster_rcnn/rfcn.py:66-250, TEST branch), built from
stock PyTorch CPU convolutions for the trunk / heads and the oracle (dtt_oracle.c, rpn_oracle.py) for every
hot-path op.  TEST INFRASTRUCTURE ONLY: used by tests/ (end-to-end parity of the GPU graph) and by bench.py's
cpu_baseline leg ("the reference's pure-CPU PyTorch path" -- the reference itself has no CPU implementation
of these ops).

`model` is a dtt.model.resnet instance living on the CPU (its nn.Conv2d / BatchNorm modules are reused as
plain PyTorch layers; none of its dtt ops are called).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle_lib as O
from . import rpn_oracle as ro


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@torch.no_grad()
def rfcn_forward_test(model, im_data, im_info, cfg):
    """im_data (B,2,3,H,W), im_info (B,2,3) CPU tensors -> dict with the reference's test-time outputs:
    rois (2,B,R,5), cls_prob (2,B,R,ncls), bbox_pred (2,B,R,4), tracking_pred (B*R,4)."""
    assert not model.training
    im = im_data.permute(1, 0, 2, 3, 4).contiguous()
    info = im_info.permute(1, 0, 2).contiguous().numpy()
    n_legs, B = im.shape[0], im.shape[1]
    rpn = model.RFCN_rpn
    base = ro.generate_anchors(scales=np.array(cfg.ANCHOR_SCALES), ratios=np.array(cfg.ANCHOR_RATIOS))
    A = base.shape[0]
    P = cfg.POOLING_SIZE
    od_cls, od_loc = model.n_classes, 4 * model.n_reg_classes
    conv3, conv4, conv5, bbox_maps, rois, cls_prob, bbox_pred = [], [], [], [], [], [], []
    for leg in range(n_legs):
        c3, c4, c5, top = model._im_to_head(im[leg])
        conv3.append(c3); conv4.append(c4); conv5.append(c5)
        cls_map = model.RFCN_cls_net(top)
        bbox_map = model.RFCN_bbox_net(top)
        bbox_maps.append(bbox_map)
        x = F.relu(rpn.RPN_Conv(top))
        score = rpn.RPN_cls_score(x)
        prob = rpn.reshape(F.softmax(rpn.reshape(score, 2), dim=1), 2 * A)
        deltas = rpn.RPN_bbox_pred(x)
        T = cfg.TEST
        r, _ = ro.proposal_layer(prob.numpy(), deltas.numpy(), info[leg], base, cfg.FEAT_STRIDE[0],
                                 T.RPN_PRE_NMS_TOP_N, T.RPN_POST_NMS_TOP_N, T.RPN_NMS_THRESH, O.nms)
        rois.append(r)
        flat = r.reshape(-1, 5)
        pc, _ = O.psroi_pool_forward(cls_map.numpy(), flat, P, P, 1.0 / 16.0, 7, od_cls)
        pl, _ = O.psroi_pool_forward(bbox_map.numpy(), flat, P, P, 1.0 / 16.0, 7, od_loc)
        s = F.avg_pool2d(_t(pc), (7, 7), stride=(7, 7)).squeeze(3).squeeze(2)
        cls_prob.append(F.softmax(s, dim=1).view(B, r.shape[1], -1))
        bbox_pred.append(F.avg_pool2d(_t(pl), (7, 7), stride=(7, 7)).squeeze(3).squeeze(2).view(B, r.shape[1], -1))
    if n_legs == 1:   # single-frame R-FCN (BASELINE configs 1-2): no tracking branch
        return dict(rois=_t(np.stack(rois, 0)), cls_prob=torch.stack(cls_prob, 0), bbox_pred=torch.stack(bbox_pred, 0),
                    tracking_pred=torch.zeros(0, 4))
    feats = list(bbox_maps)
    for (f, pad, k, d, s1, s2) in ((conv3, 8, 1, 8, 2, 2), (conv4, 8, 1, 8, 1, 1), (conv5, 8, 1, 8, 1, 1)):
        feats.append(_t(O.correlation_forward(f[0].numpy(), f[1].numpy(), pad, k, d, s1, s2)))
    tracking_reg = model.corr_bbox_net(torch.cat(feats, 1))
    pt, _ = O.psroi_pool_forward(tracking_reg.numpy(), rois[0].reshape(-1, 5), P, P, 1.0 / 16.0, 7, od_loc)
    tracking_pred = F.avg_pool2d(_t(pt), (7, 7), stride=(7, 7)).squeeze(3).squeeze(2)
    return dict(rois=_t(np.stack(rois, 0)), cls_prob=torch.stack(cls_prob, 0), bbox_pred=torch.stack(bbox_pred, 0),
                tracking_pred=tracking_pred)
