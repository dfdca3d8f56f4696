"""End-to-end: the GPU D&T graph (dtt.model, HIP ops) against the CPU graph (torch CPU convs + oracle ops) on
the same random weights and synthetic frame pair.  Convolution outputs differ in their last bits between
MIOpen and the CPU, so proposals are matched with a tolerance rather than bit for bit (the ops themselves
are bit-checked in test_gpu_ops.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rfcn_forward_contract_and_parity():
    from dtt.config import apply_dataset_defaults, cfg
    from dtt.synth import build_model, calibrate_batchnorm_, make_batch
    from oracle import cpu_graph
    apply_dataset_defaults("imagenet_vid")
    dev = torch.device("cuda:0")
    B, H, W = 2, 224, 320
    model = build_model(50, cfg=cfg).eval()
    im, info, gt, nb = make_batch(B, H, W, seed=5)
    calibrate_batchnorm_(model, im[:, 0])
    ref = cpu_graph.rfcn_forward_test(model, im, info, cfg)
    model = model.to(dev)
    with torch.no_grad():
        out = model(im.to(dev), info.to(dev), gt.to(dev), nb.to(dev))
    rois, cls_prob, bbox_pred, tracking_pred = (o.cpu() for o in out[:4])
    R = cfg.TEST.RPN_POST_NMS_TOP_N
    assert tuple(rois.shape) == (2, B, R, 5) and tuple(cls_prob.shape) == (2, B, R, 31)
    assert tuple(bbox_pred.shape) == (2, B, R, 4) and tuple(tracking_pred.shape) == (B * R, 4)
    for t in out[4:8]:
        assert tuple(t.shape) == (2, 1)
    assert torch.isfinite(cls_prob).all() and torch.isfinite(tracking_pred).all()
    # row-by-row agreement of proposals (same order unless two scores are within conv round-off)
    same = (rois - ref["rois"]).abs().amax(dim=3) < 0.05
    assert same.float().mean() > 0.9, "only %.3f of RoI rows agree" % same.float().mean()
    d_cls = (cls_prob - ref["cls_prob"]).abs().amax(dim=3)[same]
    d_box = (bbox_pred - ref["bbox_pred"]).abs().amax(dim=3)[same]
    assert d_cls.max() < 1e-3 and d_box.max() < 1e-2 * max(1.0, float(ref["bbox_pred"].abs().max()))
    same0 = same[0].reshape(-1)
    d_trk = (tracking_pred - ref["tracking_pred"]).abs().amax(dim=1)[same0]
    assert d_trk.max() < 1e-2 * max(1.0, float(ref["tracking_pred"].abs().max()))


def test_state_dict_layout_matches_reference_checkpoints():
    from dtt.config import cfg
    from dtt.synth import build_model
    sd = build_model(101, cfg=cfg).state_dict()
    for k in ("RFCN_base.0.weight", "RFCN_base.1.running_mean", "RFCN_base.4.0.conv1.weight",
              "RFCN_base.6.22.bn3.weight", "RFCN_base.7.2.conv3.weight", "RFCN_base.7.0.downsample.0.weight",
              "RFCN_base.RFCN_net.weight", "RFCN_net.bias", "RFCN_rpn.RPN_Conv.weight",
              "RFCN_rpn.RPN_cls_score.bias", "RFCN_rpn.RPN_bbox_pred.weight", "RFCN_cls_net.weight",
              "RFCN_bbox_net.bias", "corr_bbox_net.weight"):
        assert k in sd, k
    assert tuple(sd["corr_bbox_net.weight"].shape) == (196, 1051, 1, 1)
    assert tuple(sd["RFCN_cls_net.weight"].shape) == (31 * 49, 512, 1, 1)
    assert tuple(sd["RFCN_rpn.RPN_cls_score.weight"].shape) == (24, 512, 1, 1)
