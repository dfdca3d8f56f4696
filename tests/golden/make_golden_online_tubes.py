#!/usr/bin/env python3
"""Golden vectors for the online tube linker, produced by RUNNING the reference's lib/model/utils/online_tubes.py
(`VideoPostProcessor.class_paths`, build container only):

    python tests/golden/make_golden_online_tubes.py     # rewrites tests/golden/online_tubes.npz

The module is PyTorch-0.3 / CUDA-only code.  What the harness supplies so that it runs on CPU tensors under torch 2.x:
  * tab expansion; `np.object`; `torch.cuda.FloatTensor / LongTensor` and `.cuda()` pointed at CPU tensors;
    `model.nms.nms_wrapper.nms` = the CPU oracle NMS; `cv2`, `model.utils.blob` empty stand-ins (video decoding only);
  * torch 0.3 had no 0-dim tensors: `x.max(0)` returned 1-element tensors that the module then uses as indices and
    concatenates.  The one call that depends on it (`box_to_lp_score.max(0)`, online_tubes.py:380) is given
    `keepdim=True` in the source text before compiling -- the 0.3 result shape, nothing else;
  * the module collects per-path tensors with `np.array(list_of_tensors, dtype=np.object)`; with CUDA tensors numpy cannot
    look inside them and builds a 1-D object array, with CPU tensors it would unpack them.  The module's `np.array` is
    wrapped to build that 1-D object array.
Quirk worth knowing when generating inputs: the labelling indexes its (classes - 1)-row score table with 1-based class
ids (online_tubes.py:121-127), so a path that is ever labelled with the LAST class raises IndexError in the reference; the
inputs below keep the last class's scores lowest so that never happens.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402


def make_video(seed, P=14, R=36, C=6, n_obj=4):
    """Softmax-like scores of n_obj moving objects (each with a dominant class that may change mid-video) + clutter."""
    rng = np.random.RandomState(seed)
    boxes = np.zeros((P, 2, R, 4), np.float32)
    scores = np.zeros((P, 2, R, C), np.float32)
    centers = rng.uniform(60, 400, size=(n_obj, 2)); sizes = rng.uniform(40, 120, size=(n_obj, 2))
    cls_a = rng.randint(1, C - 1, size=n_obj); cls_b = rng.randint(1, C - 1, size=n_obj)
    switch = rng.randint(P // 3, P, size=n_obj)
    gone = {(1, t) for t in range(4, 7)} | {(2, t) for t in range(5, P)}          # a gap and a disappearance
    for t in range(P):
        for leg in range(2):
            k = 0
            for o in range(n_obj):
                if (o, t) in gone:
                    continue
                c = centers[o] + 5.0 * (t + leg) + rng.normal(0, 2, 2)
                for _ in range(3):                                               # near-duplicate RoIs of the object
                    wh = sizes[o] + rng.normal(0, 3, 2); cc = c + rng.normal(0, 3, 2)
                    boxes[t, leg, k] = [cc[0] - wh[0] / 2, cc[1] - wh[1] / 2, cc[0] + wh[0] / 2, cc[1] + wh[1] / 2]
                    logit = rng.normal(0, 0.5, C); logit[cls_a[o] if t < switch[o] else cls_b[o]] += rng.uniform(2.5, 4.0)
                    logit[C - 1] -= 6.0
                    scores[t, leg, k] = np.exp(logit) / np.exp(logit).sum()
                    k += 1
            while k < R:                                                         # clutter
                xy = rng.uniform(0, 500, 2); wh = rng.uniform(20, 150, 2)
                boxes[t, leg, k] = [xy[0], xy[1], xy[0] + wh[0], xy[1] + wh[1]]
                logit = rng.normal(0, 1.0, C); logit[0] += 2.0; logit[C - 1] -= 6.0
                scores[t, leg, k] = np.exp(logit) / np.exp(logit).sum()
                k += 1
    return boxes, scores


class _NumpyProxy(object):
    """numpy, except that array(list of tensors, dtype=object) gives a 1-D object array (what CUDA tensors gave)."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def array(obj, dtype=None, **kw):
        import torch
        if dtype is object and isinstance(obj, (list, tuple)) and (len(obj) == 0 or torch.is_tensor(obj[0]) or
                                                                   not hasattr(obj[0], "__len__")):
            out = np.empty((len(obj),), dtype=object)
            for i, v in enumerate(obj):
                out[i] = v
            return out
        return np.array(obj, dtype=dtype, **kw)


def load_reference_module():
    make_golden.import_reference()
    import torch
    from collections import deque
    np.object = object
    for name in ("cv2", "model.utils.blob"):
        m = types.ModuleType(name)
        m.prep_im_for_blob = m.im_list_to_blob = None
        sys.modules[name] = m

    def _mk(dtype):
        def f(*a):
            if len(a) == 1 and isinstance(a[0], (list, tuple, deque)):
                return torch.tensor([float(v) if dtype is torch.float32 else int(v) for v in a[0]], dtype=dtype)
            return torch.empty(*a, dtype=dtype)
        return f
    torch.cuda.FloatTensor, torch.cuda.LongTensor = _mk(torch.float32), _mk(torch.int64)
    real_ft = torch.FloatTensor
    torch.FloatTensor = lambda *a: (_mk(torch.float32)(*a) if len(a) == 1 and isinstance(a[0], deque) else real_ft(*a))
    torch.Tensor.cuda = lambda self, *a, **k: self
    path = os.path.join(make_golden.REF, "lib", "model", "utils", "online_tubes.py")
    src = open(path).read().expandtabs(8)
    needle = "m_score, max_ind = box_to_lp_score.max(0)"
    assert src.count(needle) == 1
    src = src.replace(needle, "m_score, max_ind = box_to_lp_score.max(0, keepdim=True)")
    mod = types.ModuleType("ref_online_tubes")
    exec(compile(src, path, "exec"), mod.__dict__)
    mod.np = _NumpyProxy()
    return mod


def flatten(vp, tubes, tag, out):
    """all_paths per class + the final tubes -> flat dict of arrays."""
    import torch
    t2n = lambda t: (t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t))
    for c in range(1, vp.num_classes):
        paths = vp.all_paths[c]
        out["%s_c%d_n" % (tag, c)] = np.array(0 if paths is None else len(paths["count"]))
        if paths is None:
            continue
        for i in range(len(paths["count"])):
            for k in ("start", "end", "boxes", "scores", "all_scores", "path_score", "found_at"):
                out["%s_c%d_p%d_%s" % (tag, c, i, k)] = t2n(paths[k][i]).astype(np.float64)
            out["%s_c%d_p%d_count" % (tag, c, i)] = np.array(int(np.asarray(paths["count"][i]).reshape(-1)[0]))
            out["%s_c%d_p%d_last_found" % (tag, c, i)] = np.array(int(np.asarray(paths["last_found"][i]).reshape(-1)[0]))
    for k in ("starts", "ends", "ts", "te", "dpActionScore", "label", "dpPathScore", "path_total_score"):
        out["%s_tubes_%s" % (tag, k)] = t2n(tubes[k]).astype(np.float64).reshape(-1)
    out["%s_tubes_n" % tag] = np.array(len(tubes["path_boxes"]))
    for i in range(len(tubes["path_boxes"])):
        out["%s_tubes_boxes%d" % (tag, i)] = t2n(tubes["path_boxes"][i]).astype(np.float64)
        out["%s_tubes_scores%d" % (tag, i)] = t2n(tubes["path_scores"][i]).astype(np.float64).reshape(-1)
    out["%s_kept_labels" % tag] = t2n(vp.path_labels).astype(np.float64).reshape(-1)
    out["%s_kept_starts" % tag] = t2n(vp.path_starts).astype(np.float64).reshape(-1)
    out["%s_kept_n" % tag] = np.array(len(vp.path_boxes))


CASES = [("a", 11, dict()), ("b", 12, dict(P=20, R=30, C=5, n_obj=3)), ("c", 13, dict(P=9, R=24, C=4, n_obj=2))]


def main():
    import torch
    ref = load_reference_module()
    out = {}
    for tag, seed, kw in CASES:
        boxes, scores = make_video(seed, **kw)
        C = scores.shape[-1]
        vp = ref.VideoPostProcessor(torch.from_numpy(boxes), torch.from_numpy(scores), torch.zeros(1),
                                    ["__background__"] + ["class_%d" % j for j in range(1, C)], "vid_" + tag)
        tubes = vp.class_paths(path_score_thresh=0.5)
        flatten(vp, tubes, tag, out)
        print(tag, "tubes:", len(tubes["path_boxes"]), "kept:", len(vp.path_boxes),
              "paths per class:", [0 if p is None else len(p["count"]) for p in vp.all_paths[1:]])
    path = os.path.join(HERE, "online_tubes.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d arrays, %d bytes" % (path, len(out), os.path.getsize(path)))


if __name__ == "__main__":
    main()
