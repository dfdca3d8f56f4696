#!/usr/bin/env python3
"""Where / when the workgroups of the proposal layer's select / sort ran during real inference steps (the correlation has its
own phase trace: tools/build_ws_trace.sh + tools/ws_trace.py) (developer tool; needs the -DDTT_WG_TRACE build: DTT_HIP_LIBRARY=tools/_variants/wgtrace.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import torch
from dtt import _lib
from dtt.config import apply_dataset_defaults, cfg
from dtt.fuse import fuse_for_inference
from dtt.synth import build_model, calibrate_batchnorm_, make_batch
dev = torch.device("cuda:0")
apply_dataset_defaults("imagenet_vid")
torch.backends.cudnn.benchmark = True
model = build_model(101, cfg=cfg).to(dev)
im, info, gt, nb = make_batch(2, 600, 1067, seed=3, device=dev)
calibrate_batchnorm_(model, im[:, 0])
model.eval()
fuse_for_inference(model, channels_last=True)
with torch.no_grad():
    for _ in range(16 + 5):
        model(im, info, gt, nb)
torch.cuda.synchronize()
L = _lib.lib()
st = (ctypes.c_ulonglong * (16 * 8 * 8))()
assert L.dtt_sort_trace_read(st, len(st))
loc = lambda h: "xcc%d/se%d/cu%d" % (h >> 16, (h >> 8) & 7, h & 15)
for s in range(0, 16, 4):
    t0 = min(int(st[(s * 8 + b) * 8 + 0]) for b in range(4))
    ph = [[(int(st[(s * 8 + b) * 8 + k]) - int(st[(s * 8 + b) * 8 + 0])) / 100.0 for k in (2, 3, 4, 5, 7)] for b in range(4)]
    print("select / sort launch %2d: workgroups on %s" % (s, "  ".join("%s start %+.1f" % (loc(st[(s * 8 + b) * 8 + 1]), (int(st[(s * 8 + b) * 8 + 0]) - t0) / 100.0) for b in range(4))))
    print("    phases (us since start: keys loaded, threshold found, compacted, sorted, decoded):", ph)
