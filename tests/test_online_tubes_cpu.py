"""dtt.online_tubes (incremental tube linking + temporal labelling of the video demo) against golden vectors produced by
RUNNING the reference's online_tubes.py (tests/golden/make_golden_online_tubes.py): the paths of every class (boxes,
scores, frames found, gap filling, ranking) and the labelled tubes must agree.  Pure tensor code: CPU only here."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
G = os.path.join(os.path.dirname(__file__), "golden", "online_tubes.npz")


@pytest.mark.parametrize("case", [0, 1, 2])
def test_online_tubes_match_reference_run(case):
    import make_golden_online_tubes as mk
    from dtt.online_tubes import VideoPostProcessor
    gold = np.load(G)
    tag, seed, kw = mk.CASES[case]
    boxes, scores = mk.make_video(seed, **kw)
    C = scores.shape[-1]
    vp = VideoPostProcessor(torch.from_numpy(boxes), torch.from_numpy(scores), torch.zeros(1),
                            ["__background__"] + ["class_%d" % j for j in range(1, C)], "vid_" + tag)
    tubes = vp.class_paths(path_score_thresh=0.5)
    out = {}
    mk.flatten(vp, tubes, tag, out)
    keys = [k for k in gold.files if k.startswith(tag + "_")]
    assert keys and set(keys) == set(out), sorted(set(keys) ^ set(out))[:6]
    for k in keys:
        assert out[k].shape == gold[k].shape, k
        np.testing.assert_allclose(out[k], gold[k], rtol=1e-6, atol=1e-6, err_msg=k)


def test_candidate_selection_is_the_per_frame_nms():
    """select_candidates (batched: stable sort, top 50, greedy NMS at 0.3, first 10) against the straightforward
    per-(pair, class) loop with the CPU oracle NMS."""
    from dtt.online_tubes import select_candidates
    from oracle import oracle_lib as O
    rng = np.random.RandomState(3)
    P, R, C = 4, 70, 5
    xy = rng.uniform(0, 200, size=(P, R, 2)); wh = rng.uniform(10, 120, size=(P, R, 2))
    boxes = np.concatenate([xy, xy + wh], 2).astype(np.float32)
    scores = rng.uniform(0, 1, size=(P, R, C)).astype(np.float32)
    scores[0, :, 2] = 0.0; scores[0, 5, 2] = 0.3            # a class with a single candidate
    idx, count = select_candidates(torch.from_numpy(boxes), torch.from_numpy(scores))
    for p in range(P):
        for c in range(C):
            s = scores[p, :, c]
            pick = np.nonzero(s > 0)[0]
            order = pick[np.argsort(-s[pick], kind="stable")][:50]
            keep = O.nms(np.concatenate([boxes[p][order], s[order, None]], 1), 0.3).reshape(-1)[:10]
            assert int(count[p, c]) == len(keep)
            np.testing.assert_array_equal(idx[p, c, :len(keep)].numpy(), order[keep])
