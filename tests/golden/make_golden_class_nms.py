#!/usr/bin/env python3
"""Generate tests/golden/class_nms.npz by RUNNING the reference's test-time per-class NMS loop (build container only).

The loop is not a function: it is lines 274-301 of `/root/reference/test_net.py`, inside the script's `__main__` block (per
class: score threshold, `torch.sort`, `nms`, then the `max_per_image` cut over all classes).  This script reads exactly those
lines from the reference where it lies, dedents them, rewrites `xrange` -> `range` (Python 2), and `exec`s them with the names
they use bound to seeded CPU tensors: `scores` / `pred_boxes` in the (frame sample, leg, roi, ...) layout the script has at that
point, `imdb.num_classes`, `thresh`, `cfg.TEST.NMS`, `args.class_agnostic`, `max_per_image`, `all_boxes`, `empty_array`, `i`, and
`nms` = the CPU oracle NMS on the tensor's numpy view (the reference's `nms` is its CUDA kernel; the oracle NMS is pinned to that
kernel bit for bit by tests/test_gpu_ref_kernels.py).  Nothing of the reference is stored: only the inputs and the `all_boxes`
arrays the executed lines produce.

    python tests/golden/make_golden_class_nms.py
"""
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT]
from oracle import oracle_lib as O  # noqa: E402

REF = "/root/reference/test_net.py"


def reference_loop_source():
    lines = open(REF).read().split("\n")
    first = next(k for k, l in enumerate(lines) if "for j in xrange(1, imdb.num_classes):" in l)
    last = next(k for k, l in enumerate(lines) if "all_boxes[j][i] = all_boxes[j][i][keep, :]" in l)
    assert 250 < first < last < 320, (first, last)
    src = "\n".join(l.expandtabs(8) for l in lines[first:last + 1])
    return textwrap.dedent(src).replace("xrange", "range")


def run_reference(scores_rc, boxes, thresh, nms_thresh, max_per_image, class_agnostic):
    """scores_rc (R, ncls), boxes (R, 4) or (R, 4*ncls) -> list over classes of (n, 5) float32 arrays."""
    ncls = scores_rc.shape[1]
    env = dict(
        torch=torch, np=np,
        scores=torch.from_numpy(scores_rc)[None, None],           # (frame sample, leg, roi, class)
        pred_boxes=torch.from_numpy(boxes)[None, None],
        imdb=types.SimpleNamespace(num_classes=ncls),
        args=types.SimpleNamespace(class_agnostic=class_agnostic),
        cfg=types.SimpleNamespace(TEST=types.SimpleNamespace(NMS=nms_thresh)),
        thresh=thresh, max_per_image=max_per_image, i=0,
        empty_array=np.transpose(np.array([[], [], [], [], []]), (1, 0)),
        all_boxes=[[[] for _ in range(1)] for _ in range(ncls)],
        nms=lambda dets, t: torch.from_numpy(np.asarray(O.nms(dets.numpy().astype(np.float32), t), dtype=np.int64).reshape(-1, 1)),
    )
    exec(compile(reference_loop_source(), REF + ":274-301", "exec"), env)
    return [np.zeros((0, 5), np.float32)] + [np.asarray(env["all_boxes"][j][0], dtype=np.float32).reshape(-1, 5) for j in range(1, ncls)]


def main():
    out = {}
    rng = np.random.RandomState(2024)
    cases = [("agnostic_300x31", 300, 31, True, 0.05, 0.3, 100), ("perclass_120x7", 120, 7, False, 0.05, 0.3, 40),
             ("nocut_80x5", 80, 5, True, 0.3, 0.5, 0), ("sparse_60x31", 60, 31, True, 0.6, 0.3, 100)]
    for name, R, ncls, agn, thresh, nms_t, mpi in cases:
        logits = rng.normal(0, 2.0, size=(R, ncls)).astype(np.float32)
        scores = torch.softmax(torch.from_numpy(logits), 1).numpy()
        # distinct scores per class (torch.sort's order among equal scores is unspecified in the reference)
        for j in range(ncls):
            assert len(np.unique(scores[:, j])) == R
        c = rng.uniform(20, 500, size=(R, 1, 2)); wh = rng.uniform(10, 200, size=(R, 1 if agn else ncls, 2))
        b = np.concatenate([c - wh / 2, c + wh / 2], 2).astype(np.float32).reshape(R, -1)
        res = run_reference(scores, b, thresh, nms_t, mpi, agn)
        out[name + "/scores"] = scores; out[name + "/boxes"] = b
        out[name + "/params"] = np.array([thresh, nms_t, mpi, int(agn)], dtype=np.float64)
        out[name + "/counts"] = np.array([len(r) for r in res], dtype=np.int64)
        out[name + "/dets"] = np.concatenate(res, 0) if sum(len(r) for r in res) else np.zeros((0, 5), np.float32)
        print(name, "kept per class:", out[name + "/counts"].tolist())
    np.savez_compressed(os.path.join(HERE, "class_nms.npz"), **out)


if __name__ == "__main__":
    main()
