"""Synthetic 2-frame batches and random-init weights of the D&T shape: there is no
network access for datasets or checkpoints, so benchmarks and end-to-end tests run on these."""
import numpy as np
import torch
from torch import nn


def make_batch(batch, height=600, width=1067, seed=3, max_gt=30, device="cpu", scale=600.0 / 720.0):
    """im_data (B,2,3,H,W) ~ N(0, 50^2) with frame t+tau = frame t rolled by a few pixels + N(0, 5^2) noise;
    im_info (B,2,3) = [H, W, scale]; gt_boxes (B,2,max_gt,6) [x1,y1,x2,y2,cls,track_id]; num_boxes (B,2,1)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    rng = np.random.RandomState(seed)
    f0 = torch.randn(batch, 3, height, width, generator=g) * 50.0
    im = torch.empty(batch, 2, 3, height, width)
    gt = torch.zeros(batch, 2, max_gt, 6)
    nb = torch.zeros(batch, 2, 1, dtype=torch.long)
    for b in range(batch):
        dy, dx = int(rng.randint(-8, 9)), int(rng.randint(-8, 9))
        im[b, 0] = f0[b]
        im[b, 1] = torch.roll(f0[b], (dy, dx), (1, 2)) + torch.randn(3, height, width, generator=g) * 5.0
        n = int(rng.randint(1, 6))
        for i in range(n):
            w, h = rng.uniform(64, min(400, width / 2)), rng.uniform(64, min(400, height / 2))
            x1, y1 = rng.uniform(0, width - w - 1), rng.uniform(0, height - h - 1)
            cls = float(rng.randint(1, 31))
            box0 = [x1, y1, x1 + w, y1 + h]
            box1 = [min(max(v + d, 0), lim) for v, d, lim in zip(box0, (dx, dy, dx, dy),
                                                                 (width - 1, height - 1, width - 1, height - 1))]
            gt[b, 0, i] = torch.tensor(box0 + [cls, float(i + 1)])
            gt[b, 1, i] = torch.tensor(box1 + [cls, float(i + 1)])
        nb[b, :, 0] = n
    info = torch.tensor([float(height), float(width), float(scale)]).view(1, 1, 3).expand(batch, 2, 3).contiguous()
    return im.to(device), info.to(device), gt.to(device), nb.to(device)


@torch.no_grad()
def calibrate_batchnorm_(model, images):
    """Random-init trunks have identity BatchNorm statistics and their activations explode over 100 layers.
    One pass with BN in training mode (cumulative average) gives every BN layer the statistics of this
    synthetic input, so the frozen-BN inference graph sees O(1) activations like a trained checkpoint would."""
    bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    saved = [(m.training, m.momentum) for m in bns]
    for m in bns:
        m.reset_running_stats()
        m.momentum = None
        m.train()
    model._im_to_head(images)
    for m, (tr, mom) in zip(bns, saved):
        m.train(tr)
        m.momentum = mom
    return model


def build_model(num_layers=101, n_classes=31, class_agnostic=True, seed=3, cfg=None, pretrained=False,
                pretrained_rfcn=False):
    """pretrained / pretrained_rfcn: start from data/pretrained_model/res101.pth / rfcn_detect.pth as the reference
    driver does for real datasets (trainval_net.py:264-269); the synthetic benchmarks use random weights."""
    from .model import resnet
    torch.manual_seed(seed)
    classes = ["__background__"] + ["c%d" % i for i in range(1, n_classes)]
    m = resnet(classes, num_layers, pretrained=pretrained, pretrained_rfcn=pretrained_rfcn, class_agnostic=class_agnostic,
               cfg=cfg)
    m.create_architecture()
    return m
