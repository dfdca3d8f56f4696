#!/usr/bin/env python3
"""Generate tests/golden/tubes.npz by RUNNING the reference's zero-jump Viterbi tube linker (build container only).

`lib/model/utils/tracking_utils.py` is Python-2 era, CUDA-only code: it mixes tabs and spaces (py3 TabError), builds its
tensors with `torch.cuda.FloatTensor`, uses `np.object` and cv2.  It is executed here unmodified except for what the
harness supplies around it:
  * the source text is tab-expanded (Python 2's rule: tab stops every 8 columns) before `compile()`;
  * `torch.cuda.FloatTensor / LongTensor` and `Tensor.cuda()` are pointed at CPU tensors, `np.object = object`;
  * `model.nms.nms_wrapper.nms` is the CPU oracle NMS (as in make_golden.py), `model.utils.blob` an empty stand-in;
  * `cv2` is a stand-in module with only `filter2D` (1-D correlation, BORDER_REFLECT_101 -- OpenCV's default); that one
    call produces `smooth_scores`, which is therefore pinned to this restatement of OpenCV, not to OpenCV itself.
Only inputs and the outputs the reference computes are stored.

    python tests/golden/make_golden_tubes.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402


def filter2d_reflect101(src, ddepth, kernel):
    a = np.asarray(src)
    k = np.asarray(kernel, dtype=np.float64).ravel()
    r = len(k) // 2
    v = a.reshape(-1).astype(np.float64)
    n = len(v)
    idx = np.arange(-r, n + r)
    if n == 1:
        idx = np.zeros_like(idx)
    else:
        period = 2 * (n - 1)
        idx = np.mod(idx, period)
        idx = np.where(idx >= n, period - idx, idx)
    p = v[idx]
    out = np.array([np.dot(p[i:i + len(k)], k) for i in range(n)])
    return out.astype(a.dtype).reshape(a.shape)


def load_reference_module():
    make_golden.import_reference()
    import torch
    np.object = object
    cv2 = types.ModuleType("cv2")
    cv2.filter2D = filter2d_reflect101
    sys.modules["cv2"] = cv2
    blob = types.ModuleType("model.utils.blob")
    blob.prep_im_for_blob = blob.im_list_to_blob = None
    sys.modules["model.utils.blob"] = blob

    def _ft(*a):
        if len(a) == 1 and isinstance(a[0], (list, tuple)):
            return torch.tensor(a[0], dtype=torch.float32)
        return torch.empty(*a, dtype=torch.float32)

    def _lt(*a):
        if len(a) == 1 and isinstance(a[0], (list, tuple)):
            return torch.tensor(a[0], dtype=torch.int64)
        return torch.empty(*a, dtype=torch.int64)

    torch.cuda.FloatTensor, torch.cuda.LongTensor = _ft, _lt
    torch.Tensor.cuda = lambda self, *a, **k: self
    path = os.path.join(make_golden.REF, "lib", "model", "utils", "tracking_utils.py")
    src = open(path).read().expandtabs(8)
    mod = types.ModuleType("ref_tracking_utils")
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod


def make_case(rng, F, n_per_frame, M, spread, with_tracks=True, missing=(), ties=False):
    """Detections drift slowly from frame to frame (so tracklets of frame f overlap boxes of f and f+1)."""
    nmax = max(n_per_frame)
    centers = rng.uniform(40, spread, size=(nmax, 2))
    sizes = rng.uniform(30, 120, size=(nmax, 2))
    dets, trks = [], []
    for f in range(F):
        n = n_per_frame[f]
        c = centers[:n] + 4.0 * f + rng.normal(0, 2.5, size=(n, 2))
        wh = sizes[:n] + rng.normal(0, 2, size=(n, 2))
        b = np.concatenate([c - wh / 2, c + wh / 2], 1)
        s = rng.uniform(0.05, 1.0, size=n)
        if ties:
            s = np.round(s * 4) / 4 + 0.125
        # a few near-duplicates so the per-frame NMS has work to do
        dup = rng.randint(0, n, size=max(1, n // 4))
        b2 = b[dup] + rng.normal(0, 1.5, size=(len(dup), 4))
        s2 = s[dup] * rng.uniform(0.5, 0.95, size=len(dup))
        b, s = np.concatenate([b, b2]), np.concatenate([s, s2])
        order = np.argsort(-s, kind="stable")
        d = np.concatenate([b[order], s[order, None], 1 - s[order, None]], 1).astype(np.float32)
        dets.append(d)
        if with_tracks and f not in missing and M > 0:
            pick = rng.randint(0, n, size=M)
            t0 = np.concatenate([c[pick] - wh[pick] / 2, c[pick] + wh[pick] / 2], 1) + rng.normal(0, 3, size=(M, 4))
            t1 = t0 + 4.0 + rng.normal(0, 3, size=(M, 4))
            trks.append((t0.astype(np.float32), t1.astype(np.float32)))
        else:
            trks.append(None)
    return dets, trks


def pack(dets, trks):
    F = len(dets)
    nmax = max(len(d) for d in dets)
    mmax = max([len(t[0]) for t in trks if t is not None] + [1])
    D = np.zeros((F, nmax, 6), np.float32)
    n = np.zeros(F, np.int32)
    T = np.zeros((F, 2, mmax, 4), np.float32)
    m = np.full(F, -1, np.int32)
    for f in range(F):
        D[f, :len(dets[f])] = dets[f]
        n[f] = len(dets[f])
        if trks[f] is not None:
            m[f] = len(trks[f][0])
            T[f, 0, :m[f]], T[f, 1, :m[f]] = trks[f]
    return D, n, T, m


def main():
    mod = load_reference_module()
    import torch
    vp = object.__new__(mod.VideoPostProcessor)
    out = {}
    specs = {
        "tracks": dict(seed=1, F=8, n=[12] * 8, M=30, spread=420),
        "no_tracks": dict(seed=2, F=6, n=[10] * 6, M=0, spread=420, with_tracks=False),
        "ragged_cut": dict(seed=3, F=7, n=[40, 9, 33, 12, 28, 45, 30], M=60, spread=900),
        "missing_tracks": dict(seed=4, F=9, n=[14] * 9, M=25, spread=420, missing=(2, 3, 7)),
        "ties": dict(seed=5, F=6, n=[8] * 6, M=20, spread=300, ties=True),
        "two_frames": dict(seed=6, F=2, n=[6, 6], M=10, spread=300),
        "three_frames": dict(seed=7, F=3, n=[7, 5, 9], M=10, spread=300),
        "long": dict(seed=8, F=40, n=[16] * 40, M=40, spread=500),
    }
    for name, sp in specs.items():
        rng = np.random.RandomState(sp["seed"])
        dets, trks = make_case(rng, sp["F"], sp["n"], sp["M"], sp["spread"], sp.get("with_tracks", True),
                               sp.get("missing", ()), sp.get("ties", False))
        frame_boxes = np.ndarray((len(dets),), dtype=object)
        tb = np.ndarray((len(dets), 2), dtype=object)
        for f, d in enumerate(dets):
            frame_boxes[f] = torch.from_numpy(d.copy())
            if trks[f] is not None:
                tb[f, 0], tb[f, 1] = torch.from_numpy(trks[f][0].copy()), torch.from_numpy(trks[f][1].copy())
        tracks_cell = [tb if sp.get("with_tracks", True) else None, None, 1]
        paths = vp._make_tubes(frame_boxes, 25, False, tracks_cell)
        D, n, T, m = pack(dets, trks)
        out[name + "/dets"], out[name + "/n"], out[name + "/trk"], out[name + "/m"] = D, n, T, m
        out[name + "/has_tracks"] = np.array([int(sp.get("with_tracks", True))], np.int32)
        for k in ("total_score", "boxes", "idx", "smooth_scores", "scores"):
            out[name + "/" + k] = paths[k].detach().cpu().numpy()
        print(name, {k: tuple(paths[k].shape) for k in paths})
    # ---- the whole VideoPostProcessor: __init__ (_process_frame_pairs, _keep_top_k) + build_class_paths
    rng = np.random.RandomState(11)
    P, R, NC, nobj = 6, 300, 5, 14
    classes = ["bg"] + ["c%d" % i for i in range(1, NC)]
    cent = rng.uniform(60, 900, (nobj, 2)); size = rng.uniform(40, 200, (nobj, 2)); ocls = rng.randint(1, NC, nobj)
    pb = np.zeros((P, 2, R, 4), np.float32); sc = np.zeros((P, 2, R, NC), np.float32); trk = np.zeros((P, R, 4), np.float32)
    for p in range(P):
        for l in range(2):
            which = rng.randint(0, nobj, R)
            c = cent[which] + 5.0 * (p + l) + rng.normal(0, 6, (R, 2)); wh = size[which] * rng.uniform(0.8, 1.2, (R, 2))
            pb[p, l] = np.concatenate([c - wh / 2, c + wh / 2], 1)
            logits = rng.normal(0, 1.0, (R, NC)); logits[np.arange(R), ocls[which]] += rng.uniform(0, 4, R)
            e = np.exp(logits); sc[p, l] = (e / e.sum(1, keepdims=True)).astype(np.float32)
        trk[p] = pb[p, 0] + 5.0 + rng.normal(0, 3, (R, 4))
    sc[2, 0, :, 1:] *= 0.005            # a frame without tracklets (every class score <= 0.01)
    vp = mod.VideoPostProcessor(torch.from_numpy(pb), torch.from_numpy(sc), torch.from_numpy(trk), classes)
    paths = vp.build_class_paths()
    out["video/pred_boxes"], out["video/scores"], out["video/pred_trk_boxes"] = pb, sc, trk
    out["video/conf_thresh"] = vp.CONF_THRESH.numpy()
    for c in range(1, NC):
        out["video/n_kept_c%d" % c] = np.array([0 if b is None else b.shape[0] for b in vp._aboxes[c]], np.int32)
        for k in ("total_score", "boxes", "idx", "smooth_scores", "scores"):
            out["video/c%d/%s" % (c, k)] = paths[c][k].detach().cpu().numpy()
        print("video class", c, {k: tuple(paths[c][k].shape) for k in paths[c]}, out["video/n_kept_c%d" % c])
    np.savez_compressed(os.path.join(HERE, "tubes.npz"), **out)


if __name__ == "__main__":
    main()
