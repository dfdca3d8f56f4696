#!/usr/bin/env python3
"""Where / when the workgroups of the conv5 correlation and of the proposal layer's select / sort ran during real inference
steps (developer tool; needs the -DDTT_WG_TRACE build: DTT_HIP_LIBRARY=tools/_variants/wgtrace.so)."""
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
nt = (ctypes.c_ulonglong * (16 * 256 * 4))()
st = (ctypes.c_ulonglong * (16 * 8 * 8))()
assert L.dtt_nhwc_trace_read(nt, len(nt)) and L.dtt_sort_trace_read(st, len(st))
for slot in range(0, 16, 4):
    wg = [(nt[(slot * 256 + i) * 4 + 0], nt[(slot * 256 + i) * 4 + 1], nt[(slot * 256 + i) * 4 + 2], nt[(slot * 256 + i) * 4 + 3]) for i in range(240)]
    t0 = min(w[0] for w in wg)
    end = max(max(w[2], w[3]) for w in wg)
    # the select / sort launch that overlaps: nearest start
    best = min(range(16), key=lambda s: abs(int(st[(s * 8) * 8 + 0]) - int(t0)))
    srt = [(st[(best * 8 + b) * 8 + 0], st[(best * 8 + b) * 8 + 1], st[(best * 8 + b) * 8 + 7]) for b in range(4)]
    ph = [[(int(st[(best * 8 + b) * 8 + k]) - int(st[(best * 8 + b) * 8 + 0])) / 100.0 for k in (2, 3, 4, 5, 7)] for b in range(4)]
    print("    sort phases (us since start: keys loaded, threshold found, compacted, sorted, decoded):", ph)
    late = [(i, w) for i, w in enumerate(wg) if (w[0] - t0) > 1000]
    loc = lambda h: "xcc%d/se%d/cu%d" % (h >> 16, (h >> 8) & 7, h & 15)
    print("conv5 launch %2d: %.1f us; %d workgroups started > 10 us late" % (slot, (end - t0) / 100.0, len(late)))
    print("    sort wgs: " + "  ".join("%s start %+.1f end %+.1f" % (loc(h), (int(a) - int(t0)) / 100.0, (int(e) - int(t0)) / 100.0) for a, h, e in srt))
    per = [0] * 8
    for w in wg:
        per[(w[1] >> 16) & 7] += 1
    first_done = {}
    for w in wg:
        x = (w[1] >> 16) & 7
        first_done[x] = min(first_done.get(x, 1e9), (max(w[2], w[3]) - t0) / 100.0)
    print("    conv5 workgroups per XCC:", per, " distinct CUs on xcc0:", len({w[1] for w in wg if (w[1] >> 16) == 0}),
          " first workgroup done per XCC (us):", [round(first_done.get(x, -1), 1) for x in range(8)])
    for i, w in late[:8]:
        print("    late wg %3d on %s: start +%.1f us, loop done +%.1f us" % (i, loc(w[1]), (w[0] - t0) / 100.0, (w[2] - t0) / 100.0))
    if late:
        cus = {}
        for i, w in enumerate(wg):
            cus.setdefault(w[1], []).append(i)
        shared = {loc(h): v for h, v in cus.items() if len(v) > 1}
        print("    CUs that ran two workgroups of this launch:", shared)
