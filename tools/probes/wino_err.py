import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/pytorch-detect-to-track_amd"]
import torch
from dtt.config import cfg
from dtt.fuse import fuse_for_inference, unfuse
from dtt.synth import build_model, calibrate_batchnorm_, make_batch
dev = torch.device("cuda:0")
for layers, hw in ((50, (224, 320)), (101, (600, 1067))):
    model = build_model(layers, cfg=cfg).to(dev).eval()
    im, _, _, _ = make_batch(2, hw[0], hw[1], seed=9, device=dev)
    calibrate_batchnorm_(model, im[:, 0])
    x = im[:, 0].contiguous()
    with torch.no_grad():
        ref = model._im_to_head(x)
        for wino, f4 in (("0", "0"), ("1", "0"), ("1", "1")):
            os.environ["DTT_WINOGRAD"], os.environ["DTT_WINOGRAD_F4"] = wino, f4
            fuse_for_inference(model, channels_last=True)
            got = model._im_to_head(x)
            picks = [c.pick for st in model._fused_trunk.stages for b in st for c in (b.c2,) if c.u is not None]
            unfuse(model)
            errs = [float((a - b).abs().max()) / max(1.0, float(b.abs().max())) for a, b in zip(got, ref)]
            print(layers, hw, "winograd", wino, "f4", f4, "rel err per map", ["%.1e" % e for e in errs],
                  "picks", sorted(set(v for p in picks for v in p.values())))
