"""Tube linking on the GPU (dtt.tubes -> dtt_tube_link) against the reference's own outputs (tests/golden/tubes.npz) and
against the oracle restatement on larger random videos.  Paths (box indices per frame) and boxes are bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import tubes_oracle as to

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tubes.npz"))
CASES = sorted({k.split("/")[0] for k in G.files} - {"video"})


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from dtt import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _check(got, want, T):
    np.testing.assert_array_equal(got["idx"].cpu().numpy(), want["idx"])
    np.testing.assert_array_equal(got["boxes"].cpu().numpy(), want["boxes"])
    np.testing.assert_allclose(got["total_score"].cpu().numpy(), want["total_score"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got["scores"].cpu().numpy(), want["scores"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(got["smooth_scores"].cpu().numpy(), want["smooth_scores"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("case", CASES)
def test_make_tubes_vs_reference_golden(dev, case):
    from dtt.tubes import make_tubes
    g = lambda k: G[case + "/" + k]
    D, n, T_, m = g("dets"), g("n"), g("trk"), g("m")
    F = len(n)
    frame_dets = [torch.from_numpy(D[f, :n[f]]).to(dev) for f in range(F)]
    tracks = None
    if int(g("has_tracks")[0]):
        tracks = [None if m[f] < 0 else (torch.from_numpy(T_[f, 0, :m[f]]).to(dev), torch.from_numpy(T_[f, 1, :m[f]]).to(dev))
                  for f in range(F)]
    got = make_tubes(frame_dets, tracks)
    _check(got, {k: g(k) for k in ("idx", "boxes", "total_score", "scores", "smooth_scores")}, F - 1)


def _random_video(rs, F, nlo, nhi, M, spread, p_missing=0.0):
    nmax = nhi
    centers = rs.uniform(40, spread, size=(nmax, 2))
    sizes = rs.uniform(30, 140, size=(nmax, 2))
    dets, n, trks, m = [], [], [], []
    for f in range(F):
        k = rs.randint(nlo, nhi + 1)
        c = centers[:k] + 3.0 * f + rs.normal(0, 3, size=(k, 2))
        wh = sizes[:k] + rs.normal(0, 3, size=(k, 2))
        b = np.concatenate([c - wh / 2, c + wh / 2], 1)
        s = rs.uniform(0.02, 1.0, size=k)
        if rs.rand() < 0.3:
            s = np.round(s * 8) / 8 + 0.0625                       # exact ties
        dup = rs.randint(0, k, size=k // 3 + 1)
        b = np.concatenate([b, b[dup] + rs.normal(0, 2, size=(len(dup), 4))])
        s = np.concatenate([s, s[dup] * rs.uniform(0.4, 1.0, size=len(dup))])
        order = np.argsort(-s, kind="stable")
        dets.append(np.concatenate([b[order], s[order, None], 1 - s[order, None]], 1).astype(np.float32))
        n.append(len(order))
        if M and rs.rand() >= p_missing:
            pick = rs.randint(0, k, size=M)
            t0 = np.concatenate([c[pick] - wh[pick] / 2, c[pick] + wh[pick] / 2], 1) + rs.normal(0, 4, size=(M, 4))
            trks.append((t0.astype(np.float32), (t0 + 3 + rs.normal(0, 4, size=(M, 4))).astype(np.float32)))
            m.append(M)
        else:
            trks.append(None)
            m.append(-1)
    nm = max(n)
    D = np.zeros((F, nm, 6), np.float32)
    Tk = np.zeros((F, 2, max(M, 1), 4), np.float32)
    for f in range(F):
        D[f, :n[f]] = dets[f]
        if trks[f] is not None:
            Tk[f, 0], Tk[f, 1] = trks[f]
    return D, np.array(n, np.int32), Tk, np.array(m, np.int32)


@pytest.mark.parametrize("F,nlo,nhi,M,spread,miss,seed", [
    (30, 5, 20, 40, 500, 0.0, 0), (120, 10, 60, 300, 1200, 0.1, 1), (12, 30, 200, 120, 2500, 0.0, 2),
    (300, 8, 30, 100, 700, 0.05, 3), (5, 1, 3, 10, 300, 0.5, 4)])
def test_link_tubes_batched_vs_oracle(dev, F, nlo, nhi, M, spread, miss, seed):
    """Several classes of one video in a single call (the production shape: 30 classes x a few hundred frames)."""
    from dtt.tubes import link_tubes, paths_from_link
    rs = np.random.RandomState(seed)
    P = 4
    vids = [_random_video(rs, F, nlo, nhi, M if p != 2 else 0, spread, miss) for p in range(P)]   # problem 2: no tracklets at all
    nm = max(v[0].shape[1] for v in vids)
    mm = max(v[2].shape[2] for v in vids)
    D = np.zeros((P, F, nm, 6), np.float32); N = np.zeros((P, F), np.int32)
    Tk = np.zeros((P, F, 2, mm, 4), np.float32); Mc = np.full((P, F), -1, np.int32)
    for p, (d, n, t, m) in enumerate(vids):
        D[p, :, :d.shape[1]] = d; N[p] = n; Tk[p, :, :, :t.shape[2]] = t; Mc[p] = m
    kb, ks, kn, pidx, ptot, npaths = link_tubes(torch.from_numpy(D).to(dev), torch.from_numpy(N), torch.from_numpy(Tk).to(dev),
                                                torch.from_numpy(Mc))
    for p, (d, n, t, m) in enumerate(vids):
        want = to.make_tubes(d, n, t, m)
        k = int(npaths[p])
        assert k == want["idx"].shape[0]
        got = paths_from_link(kb[p], ks[p], pidx[p], ptot[p], k)
        _check(got, want, F - 1)


def test_empty_frame_raises_like_the_reference(dev):
    from dtt.tubes import make_tubes
    d = torch.tensor([[0, 0, 10, 10, 0.9, 0.1]], device=dev)
    with pytest.raises(RuntimeError, match="empty box"):
        make_tubes([d, torch.zeros(0, 6, device=dev), d, d])
    with pytest.raises(RuntimeError):
        make_tubes([d.cpu(), d.cpu()])                               # no CPU fallback


def test_video_post_processor_vs_reference_golden(dev):
    """dtt.tubes.VideoPostProcessor (frame-pair bookkeeping, class thresholds, tracklets, linking -- all on the device)
    against the reference object's own output on the same predictions."""
    from dtt.tubes import VideoPostProcessor
    pb, sc, trk = (torch.from_numpy(G["video/" + k]).to(dev) for k in ("pred_boxes", "scores", "pred_trk_boxes"))
    C = sc.shape[3]
    vp = VideoPostProcessor(pb, sc, trk, ["bg"] + ["c%d" % i for i in range(1, C)])
    np.testing.assert_array_equal(vp.CONF_THRESH.cpu().numpy(), G["video/conf_thresh"])
    paths = vp.build_class_paths()
    assert paths[0] is None
    for c in range(1, C):
        np.testing.assert_array_equal(vp._n[c - 1].cpu().numpy(), G["video/n_kept_c%d" % c])
        _check(paths[c], {k: G["video/c%d/%s" % (c, k)] for k in ("idx", "boxes", "total_score", "scores", "smooth_scores")}, 0)
    with pytest.raises(IndexError):                                   # fewer detections than 160 per frame: as the reference
        VideoPostProcessor(pb[:, :, :50], sc[:, :, :50], trk[:, :50], ["bg"] + ["c%d" % i for i in range(1, C)])


@pytest.mark.parametrize("case", [0, 1])
def test_online_tubes_on_device_tensors_match_reference_run(case):
    """dtt.online_tubes with the detections on the GPU (batched candidate selection on the device, linking on the host)
    against the goldens of the reference run (tests/golden/make_golden_online_tubes.py)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_online_tubes as mk
    from dtt.online_tubes import VideoPostProcessor
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "online_tubes.npz"))
    tag, seed, kw = mk.CASES[case]
    boxes, scores = mk.make_video(seed, **kw)
    dev = torch.device("cuda:0")
    C = scores.shape[-1]
    vp = VideoPostProcessor(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), torch.zeros(1),
                            ["__background__"] + ["class_%d" % j for j in range(1, C)], "vid_" + tag)
    tubes = vp.class_paths(path_score_thresh=0.5)
    out = {}
    mk.flatten(vp, tubes, tag, out)
    for k in [k for k in gold.files if k.startswith(tag + "_")]:
        np.testing.assert_allclose(out[k], gold[k], rtol=1e-6, atol=1e-6, err_msg=k)
