"""The training-time RoI / tracking target samplers as HIP kernels (csrc/targets.hip through dtt.targets) against
  * fixtures produced by RUNNING the reference's own _ProposalTargetLayer / _TrackingProposalTargetLayer
    (tests/golden/targets.npz, tests/golden/make_golden.py) in the reference's RNG order, and
  * the torch restatement (oracle/targets_oracle.py, itself pinned against the same fixtures in the CPU suite) on random
    cases, including the device-side selection rule (`SAMPLER_RNG = "device"`: nothing is read back from the GPU).
Needs an MI355X."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from dtt import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _cfg(mode):
    from dtt.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(ROOT, "cfgs", "res101.yml"))
    c = copy.deepcopy(cfg)
    c.TRAIN.SAMPLER_RNG = mode
    return c


def test_reference_rng_mode_matches_reference_golden(dev):
    from dtt.targets import _ProposalTargetLayer, _TrackingProposalTargetLayer
    c = _cfg("reference")
    g = np.load(os.path.join(ROOT, "tests", "golden", "targets.npz"))
    gt = torch.from_numpy(g["gt_boxes"]).to(dev)
    nb = torch.from_numpy(g["num_boxes"]).to(dev)
    np.random.seed(int(g["rng_seed"][0]))
    out = _ProposalTargetLayer(31, cfg=c)(torch.from_numpy(g["pt/in_rois"]).to(dev), gt[0][:, :, :5], nb[0])
    # the SAME RoIs and labels as the reference drew (bit for bit); targets to log ulp
    assert np.array_equal(out[0].cpu().numpy(), g["pt/rois"]) and np.array_equal(out[1].cpu().numpy(), g["pt/labels"])
    for name, t in zip(("targets", "inside", "outside"), out[2:]):
        np.testing.assert_allclose(t.cpu().numpy(), g["pt/" + name], rtol=1e-6, atol=1e-6, err_msg="pt/" + name)
    out = _TrackingProposalTargetLayer(31, cfg=c)(gt, nb)
    assert np.array_equal(out[0].cpu().numpy(), g["tt/rois"]) and np.array_equal(out[1].cpu().numpy(), g["tt/labels"])
    for name, t in zip(("targets", "inside", "outside"), out[2:]):
        np.testing.assert_allclose(t.cpu().numpy(), g["tt/" + name], rtol=1e-6, atol=1e-6, err_msg="tt/" + name)


def _random_case(rs, B, R, G, im_w=1067, im_h=600):
    gt = np.zeros((B, G, 6), np.float32)
    nb = np.zeros((B,), np.int64)
    for b in range(B):
        n = rs.randint(0 if b else 1, min(G, 9) + 1)
        nb[b] = n
        for i in range(n):
            w, h = rs.uniform(20, 400), rs.uniform(20, 300)
            x1, y1 = rs.uniform(0, im_w - w - 1), rs.uniform(0, im_h - h - 1)
            gt[b, i] = [x1, y1, x1 + w, y1 + h, rs.randint(1, 31), rs.randint(1, 6)]
    rois = np.zeros((B, R, 5), np.float32)
    for b in range(B):
        rois[b, :, 0] = b
        for r in range(R):
            if nb[b] and rs.rand() < 0.4:                     # jittered copies of a ground-truth box: foreground candidates
                g = gt[b, rs.randint(0, nb[b]), :4] + rs.normal(0, 12, 4)
                rois[b, r, 1:] = [min(g[0], g[2]), min(g[1], g[3]), max(g[0], g[2]), max(g[1], g[3])]
            else:
                w, h = rs.uniform(1, 500), rs.uniform(1, 400)
                x1, y1 = rs.uniform(0, im_w - 2), rs.uniform(0, im_h - 2)
                rois[b, r, 1:] = [x1, y1, min(x1 + w, im_w - 1), min(y1 + h, im_h - 1)]
        rois[b, 0, 1:] = [5, 5, 5, 5]                          # zero-area candidate (overlap -1)
    return rois, gt, nb


@pytest.mark.parametrize("B,R,G,seed", [(2, 300, 30, 0), (4, 2000, 30, 1), (1, 17, 5, 2), (3, 1100, 20, 3)])
def test_reference_rng_mode_matches_oracle_on_random_cases(dev, B, R, G, seed):
    from dtt.targets import _ProposalTargetLayer
    from oracle.targets_oracle import _ProposalTargetLayer as Ref
    c = _cfg("reference")
    rs = np.random.RandomState(seed)
    rois, gt, nb = _random_case(rs, B, R, G)
    np.random.seed(11 + seed)
    ref = Ref(31, cfg=c)(torch.from_numpy(rois), torch.from_numpy(gt[:, :, :5].copy()), torch.from_numpy(nb))
    np.random.seed(11 + seed)
    got = _ProposalTargetLayer(31, cfg=c)(torch.from_numpy(rois).to(dev), torch.from_numpy(gt).to(dev)[:, :, :5], torch.from_numpy(nb).to(dev))
    assert np.array_equal(got[0].cpu().numpy(), ref[0].numpy()) and np.array_equal(got[1].cpu().numpy(), ref[1].numpy())
    for a, b in zip(got[2:], ref[2:]):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-5, atol=1e-6)
    assert np.random.rand() == np.random.rand() or True      # (both consumed the generator; the streams are compared by the outputs)


@pytest.mark.parametrize("B,R,G,seed", [(2, 300, 30, 5), (4, 2000, 30, 6), (2, 40, 8, 7)])
def test_device_mode_follows_its_selection_rule(dev, B, R, G, seed):
    """SAMPLER_RNG = "device": the kernel's choice is a deterministic function of the IoUs and the uniforms the host drew
    blind; restated in numpy (oracle.targets_oracle.device_rule_sample) on the oracle's IoU matrix.  Also: the foreground
    subset has no repeats, counts follow the reference's quota, labels / targets are consistent with the chosen RoIs."""
    from dtt.rpn import bbox_overlaps_batch, bbox_transform_batch
    from dtt.targets import _ProposalTargetLayer
    from oracle.targets_oracle import device_rule_sample
    c = _cfg("device")
    T = c.TRAIN
    rs = np.random.RandomState(seed)
    rois, gt, nb = _random_case(rs, B, R, G)
    n_out = int(T.BATCH_SIZE)
    fg_per = int(np.round(T.FG_FRACTION * n_out)) or 1
    np.random.seed(100 + seed)
    u_fg, u_bg = np.random.rand(B, R + G), np.random.rand(B, n_out)          # what the layer will draw
    np.random.seed(100 + seed)
    got = [t.cpu().numpy() for t in _ProposalTargetLayer(31, cfg=c)(torch.from_numpy(rois).to(dev), torch.from_numpy(gt).to(dev)[:, :, :5],
                                                                      torch.from_numpy(nb).to(dev))]
    cand = np.concatenate([rois, np.concatenate([np.zeros((B, G, 1), np.float32), gt[:, :, :4]], 2)], 1)
    ov = bbox_overlaps_batch(torch.from_numpy(cand), torch.from_numpy(gt[:, :, :5].copy()))
    max_ov, assign = ov.max(2)
    for b in range(B):
        idx, fg_n = device_rule_sample(max_ov[b].numpy(), T.FG_THRESH, T.BG_THRESH_HI, T.BG_THRESH_LO, u_fg[b], u_bg[b], n_out, fg_per)
        exp_rois = cand[b, idx].copy()
        exp_rois[:, 0] = b
        assert np.array_equal(got[0][b], exp_rois), b
        lab = gt[b, assign[b].numpy()[idx], 4].copy()
        lab[fg_n:] = 0
        assert np.array_equal(got[1][b], lab)
        assert len(set(idx[:fg_n].tolist())) == fg_n                         # without replacement
        tgt = bbox_transform_batch(torch.from_numpy(exp_rois[None, :, 1:5]), torch.from_numpy(gt[b, assign[b].numpy()[idx], :4][None]))[0]
        tgt = ((tgt - torch.tensor(T.BBOX_NORMALIZE_MEANS)) / torch.tensor(T.BBOX_NORMALIZE_STDS)).numpy() * (lab > 0)[:, None]
        np.testing.assert_allclose(got[2][b], tgt, rtol=1e-5, atol=1e-6)
        assert np.array_equal(got[3][b] > 0, np.repeat((lab > 0)[:, None], 4, 1)) and np.array_equal(got[4][b], (got[3][b] > 0).astype(np.float32))


def test_tracking_targets_match_oracle_on_random_tracks(dev):
    from dtt.targets import _TrackingProposalTargetLayer
    from oracle.targets_oracle import _TrackingProposalTargetLayer as Ref
    c = _cfg("device")
    rs = np.random.RandomState(4)
    for trial in range(20):
        B, G = rs.randint(1, 5), rs.choice([5, 20, 30])
        gt = np.zeros((2, B, G, 6), np.float32)
        nb = np.zeros((2, B, 1), np.int64)
        for f in range(2):
            for b in range(B):
                n = rs.randint(0, min(G, 8) + 1)
                nb[f, b, 0] = n
                ids = rs.permutation(10)[:n] + 1 if trial % 3 else rs.randint(1, 4, n)      # every third trial: repeated track ids
                for i in range(n):
                    w, h = rs.uniform(20, 300, 2)
                    x1, y1 = rs.uniform(0, 600, 2)
                    gt[f, b, i] = [x1, y1, x1 + w, y1 + h, rs.randint(1, 31), ids[i]]
        ref = Ref(31, cfg=c)(torch.from_numpy(gt), torch.from_numpy(nb))
        got = _TrackingProposalTargetLayer(31, cfg=c)(torch.from_numpy(gt).to(dev), torch.from_numpy(nb).to(dev))
        assert np.array_equal(got[0].cpu().numpy(), ref[0].numpy()) and np.array_equal(got[1].cpu().numpy(), ref[1].numpy()), trial
        for a, b in zip(got[2:], ref[2:]):
            np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-5, atol=1e-6)
