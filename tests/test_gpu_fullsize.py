"""The hot-path ops at BASELINE.json's full sizes (600 x 1067 -> 38 x 67 stride-16 maps, 2048 / 1024 / 512 channels,
d = 8, 6000 / 12000 pre-NMS boxes, 7x7x31 position-sensitive maps), where the scalar oracle would take minutes.
Checked through size-independent properties that pin the result without it:

  * correlation forward: every output is one dot product -> a few thousand entries recomputed in float64; the
    zero-displacement channel against an elementwise product; bilinearity; the swap identity
    corr(a, b)[d, p] = corr(b, a)[-d, p + d];
  * correlation backward: the op is bilinear, so <corr(u, x2), g> = <u, grad1(g)> and <corr(x1, v), g> = <v, grad2(g)>
    for arbitrary u, v (adjoint identity);
  * greedy NMS: the keep list is the unique set with (i) no kept pair above the threshold and (ii) every suppressed
    box overlapped above the threshold by an earlier kept one -- both verified on the full IoU matrix; idempotence;
  * PSRoI pooling: constant maps pool to the constant, linearity in the features, adjoint identity for the backward,
    and a float64 recomputation of sampled bins.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from dtt import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _feat(g, shape, dev):
    return torch.relu(torch.randn(*shape, generator=g, device=dev))


CORR_FULL = [(2, 2048, 38, 67, 8, 1, 8, 1, 1),    # conv5
             (2, 1024, 38, 67, 8, 1, 8, 1, 1),    # conv4
             (2, 512, 75, 134, 8, 1, 8, 2, 2),    # conv3 (stride 2)
             (4, 2048, 38, 67, 8, 1, 8, 1, 1),    # both legs' worth of images in one call
             (1, 2048, 36, 63, 16, 1, 16, 1, 1),  # config 5 (563 x 1000, d = 16): conv5, 33 x 33 displacements
             (1, 512, 71, 125, 16, 1, 16, 2, 2)]  # config 5 conv3 (stride 2, R = 8)


@pytest.mark.parametrize("case", CORR_FULL)
def test_correlation_forward_full_size_properties(dev, case):
    from dtt.ops import Correlation
    B, C, H, W, pad, k, d, s1, s2 = case
    g = torch.Generator(device=dev).manual_seed(sum(case))
    x1, x2, y1 = _feat(g, (B, C, H, W), dev), _feat(g, (B, C, H, W), dev), _feat(g, (B, C, H, W), dev)
    corr = Correlation(pad, k, d, s1, s2)
    out = corr(x1, x2)
    R = d // s2
    D = 2 * R + 1
    oh, ow = out.shape[2], out.shape[3]
    assert out.shape == (B, D * D, oh, ow) and torch.isfinite(out).all()
    # (1) sampled entries in float64 (including image corners / borders and out-of-image displacements)
    rs = np.random.RandomState(1)
    n = 3000
    bb, tj, ti = rs.randint(0, B, n), rs.randint(0, D, n), rs.randint(0, D, n)
    yy, xx = rs.randint(0, oh, n), rs.randint(0, ow, n)
    yy[:200] = rs.choice([0, 1, oh - 2, oh - 1], 200); xx[:200] = rs.choice([0, 1, ow - 2, ow - 1], 200)
    x1d, x2d = x1.double(), x2.double()
    py, px = yy * s1 + d - pad, xx * s1 + d - pad          # centre pixel of output (yy, xx) in unpadded coordinates
    qy, qx = py + (tj - R) * s2, px + (ti - R) * s2
    inside = (qy >= 0) & (qy < H) & (qx >= 0) & (qx < W) & (py >= 0) & (py < H) & (px >= 0) & (px < W)
    t = lambda a: torch.from_numpy(a).to(dev)
    a = x1d[t(bb), :, t(np.clip(py, 0, H - 1)), t(np.clip(px, 0, W - 1))]
    b = x2d[t(bb), :, t(np.clip(qy, 0, H - 1)), t(np.clip(qx, 0, W - 1))]
    want = (a * b).sum(1) / C * t(inside.astype(np.float64))
    got = out[t(bb), t(tj * D + ti), t(yy), t(xx)].double()
    assert float((got - want).abs().max()) <= 1e-4          # north-star tolerance
    assert inside.sum() > n // 2 and (~inside).sum() > 50
    # (2) zero displacement = mean over channels of the elementwise product on the stride lattice
    c0 = (x1d * x2d).mean(1)[:, d - pad::s1, d - pad::s1][:, :oh, :ow]
    assert float((out[:, R * D + R].double() - c0).abs().max()) <= 1e-4
    # (3) bilinearity in the first argument
    lin = corr(2.5 * x1 + y1, x2)
    assert float((lin - (2.5 * out + corr(y1, x2))).abs().max()) <= 2e-4
    # (4) swap identity on the pixels where both sides are defined (stride-1 lattices)
    if s1 == 1 and s2 == 1 and pad == d:
        sw = corr(x2, x1)
        for (dy, dx) in ((-R, 3), (5, -7), (0, R), (-1, -1)):
            ch, chs = (dy + R) * D + (dx + R), (-dy + R) * D + (-dx + R)
            ys = slice(max(0, -dy), min(oh, oh - dy)); xs = slice(max(0, -dx), min(ow, ow - dx))
            yt = slice(max(0, dy), min(oh, oh + dy)); xt = slice(max(0, dx), min(ow, ow + dx))
            assert float((out[:, ch, ys, xs] - sw[:, chs, yt, xt]).abs().max()) <= 1e-5


@pytest.mark.parametrize("case", CORR_FULL[:3] + CORR_FULL[4:])
def test_correlation_backward_full_size_adjoint(dev, case):
    from dtt.ops import Correlation
    B, C, H, W, pad, k, d, s1, s2 = case
    g = torch.Generator(device=dev).manual_seed(7 + sum(case))
    x1 = _feat(g, (B, C, H, W), dev).requires_grad_(True)
    x2 = _feat(g, (B, C, H, W), dev).requires_grad_(True)
    corr = Correlation(pad, k, d, s1, s2)
    out = corr(x1, x2)
    go = torch.randn(out.shape, generator=g, device=dev)
    out.backward(go)
    u, v = _feat(g, (B, C, H, W), dev), _feat(g, (B, C, H, W), dev)
    with torch.no_grad():
        lhs1 = float((corr(u, x2.detach()).double() * go.double()).sum())
        rhs1 = float((u.double() * x1.grad.double()).sum())
        lhs2 = float((corr(x1.detach(), v).double() * go.double()).sum())
        rhs2 = float((v.double() * x2.grad.double()).sum())
    assert abs(lhs1 - rhs1) <= 1e-4 * max(1.0, abs(lhs1)), (lhs1, rhs1)
    assert abs(lhs2 - rhs2) <= 1e-4 * max(1.0, abs(lhs2)), (lhs2, rhs2)
    # sampled gradient entries in float64: dL/dx1[b,c,p] = (1/C) sum_d go[b,d,p] * x2[b,c,p+d]
    rs = np.random.RandomState(3)
    R, D = d // s2, 2 * (d // s2) + 1
    oh, ow = out.shape[2], out.shape[3]
    for _ in range(40):
        b, c, y, x = rs.randint(B), rs.randint(C), rs.randint(oh), rs.randint(ow)
        py, px = y * s1 + d - pad, x * s1 + d - pad
        acc = 0.0
        win = torch.zeros(D, D, dtype=torch.float64, device=dev)
        y0, y1_ = max(0, py - R * s2), min(H - 1, py + R * s2)
        ys = [(py + (j - R) * s2) for j in range(D)]
        xs = [(px + (i - R) * s2) for i in range(D)]
        for j, qy in enumerate(ys):
            for i, qx in enumerate(xs):
                if 0 <= qy < H and 0 <= qx < W:
                    win[j, i] = x2[b, c, qy, qx].double()
        want = float((go[b, :, y, x].double().view(D, D) * win).sum().item() / C)
        assert abs(float(x1.grad[b, c, py, px]) - want) <= 1e-5 * max(1.0, abs(want))


@pytest.mark.parametrize("n,thresh,seed", [(6000, 0.7, 0), (12000, 0.7, 1), (12000, 0.3, 2), (20000, 0.5, 3)])
def test_nms_full_size_certificate(dev, n, thresh, seed):
    from dtt.ops import nms
    rs = np.random.RandomState(seed)
    centers = rs.uniform(0, 1000, size=(n // 20, 2))
    c = centers[rs.randint(0, len(centers), size=n)] + rs.normal(0, 12, size=(n, 2))
    wh = rs.uniform(16, 200, size=(n, 2))
    scores = np.sort(rs.uniform(0, 1, size=n))[::-1]
    dets = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2, scores[:, None]], 1).astype(np.float32)).to(dev)
    keep = nms(dets, thresh).view(-1).long()
    assert keep.numel() > 0 and bool((keep[1:] > keep[:-1]).all())      # sorted, unique (input is score-sorted)
    # IoU exactly as the reference kernel computes it (nms_cuda_kernel.cu:31-42), float32, +1 widths
    bx = dets[:, :4]
    area = (bx[:, 2] - bx[:, 0] + 1) * (bx[:, 3] - bx[:, 1] + 1)
    kept = torch.zeros(n, dtype=torch.bool, device=dev); kept[keep] = True
    kb = bx[keep]
    ka = area[keep]
    covered = torch.zeros(n, dtype=torch.bool, device=dev)
    worst_kept = 0.0
    for s in range(0, n, 2048):                                          # row blocks of the (n x |keep|) IoU matrix
        q = bx[s:s + 2048]
        w = (torch.min(q[:, None, 2], kb[None, :, 2]) - torch.max(q[:, None, 0], kb[None, :, 0]) + 1).clamp(min=0)
        h = (torch.min(q[:, None, 3], kb[None, :, 3]) - torch.max(q[:, None, 1], kb[None, :, 1]) + 1).clamp(min=0)
        inter = w * h
        iou = inter / (area[s:s + 2048, None] + ka[None, :] - inter)
        earlier = keep[None, :] < torch.arange(s, min(n, s + 2048), device=dev)[:, None]
        hit = ((iou > thresh) & earlier).any(1)
        covered[s:s + 2048] = hit
    assert not bool((covered & kept).any()), "a kept box is suppressed by an earlier kept box"
    assert bool((covered | kept).all()), "a dropped box has no earlier kept box above the threshold"
    again = nms(dets[keep].contiguous(), thresh).view(-1).long()        # idempotence
    assert again.numel() == keep.numel() and bool((again == torch.arange(keep.numel(), device=dev)).all())


def test_psroi_full_size_properties(dev):
    from dtt.ops import _PSRoIPooling
    B, od, gsz, H, W = 4, 31, 7, 38, 67
    C = od * gsz * gsz
    g = torch.Generator(device=dev).manual_seed(5)
    rs = np.random.RandomState(5)
    R = 1200
    x1 = rs.uniform(-30, 1040, R); y1 = rs.uniform(-30, 580, R)
    w = rs.uniform(8, 700, R); h = rs.uniform(8, 500, R)
    rois = torch.from_numpy(np.stack([rs.randint(0, B, R), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)).to(dev)
    pool = _PSRoIPooling(gsz, gsz, 1 / 16.0, gsz, od)
    f = torch.randn(B, C, H, W, generator=g, device=dev).requires_grad_(True)
    out = pool(f, rois)
    assert out.shape == (R, od, gsz, gsz)
    # constant maps pool to the constant wherever the bin is not empty, and to 0 where it is
    const = pool(torch.full((B, C, H, W), 3.25, device=dev), rois)
    assert bool(((const == 3.25) | (const == 0)).all()) and float((const == 3.25).float().mean()) > 0.5
    # linearity
    f2 = torch.randn(B, C, H, W, generator=g, device=dev)
    assert float((pool(1.5 * f.detach() + f2, rois) - (1.5 * out.detach() + pool(f2, rois))).abs().max()) <= 1e-4
    # adjoint identity for the backward
    go = torch.randn(out.shape, generator=g, device=dev)
    out.backward(go)
    lhs = float((pool(f2, rois).double() * go.double()).sum())
    rhs = float((f2.double() * f.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    # sampled bins recomputed in float64 with the reference's bin arithmetic (psroi_pooling_kernel.cu:15-80)
    fd = f.detach().double().cpu().numpy(); rn = rois.cpu().numpy(); on = out.detach().cpu().numpy()
    r32 = np.float32
    for _ in range(300):
        n, ct, ph, pw = rs.randint(R), rs.randint(od), rs.randint(gsz), rs.randint(gsz)
        b = int(rn[n, 0])
        sw, sh = r32(np.round(rn[n, 1])) * r32(0.0625), r32(np.round(rn[n, 2])) * r32(0.0625)
        ew, eh = r32(np.round(rn[n, 3]) + 1.0) * r32(0.0625), r32(np.round(rn[n, 4]) + 1.0) * r32(0.0625)
        rw, rh = max(r32(ew - sw), 0.1), max(r32(eh - sh), 0.1)
        bh, bw = r32(rh / r32(gsz)), r32(rw / r32(gsz))
        hs = int(min(max(np.floor(r32(r32(ph) * bh + sh)), 0), H)); he = int(min(max(np.ceil(r32(r32(ph + 1) * bh + sh)), 0), H))
        ws = int(min(max(np.floor(r32(r32(pw) * bw + sw)), 0), W)); we = int(min(max(np.ceil(r32(r32(pw + 1) * bw + sw)), 0), W))
        c = (ct * gsz + ph) * gsz + pw
        want = 0.0 if (he <= hs or we <= ws) else fd[b, c, hs:he, ws:we].mean()
        assert abs(on[n, ct, ph, pw] - want) <= 1e-5 * max(1.0, abs(want)), (n, ct, ph, pw)


def test_correlation_race_screen(dev):
    """The forward is deterministic (fixed-order split-K reduction), so any run-to-run difference is a synchronisation bug
    in the LDS-DMA pipeline (DMA landing vs operand reads vs buffer reuse).  Repeated under traffic from a second stream."""
    from dtt.ops import Correlation
    side = torch.cuda.Stream()
    junk = torch.randn(32 << 20, device=dev)
    for (B, C, H, W, d) in [(2, 2048, 38, 67, 8), (1, 2048, 36, 63, 16), (3, 64, 38, 67, 8)]:
        g = torch.Generator(device=dev).manual_seed(C + d)
        x1, x2 = _feat(g, (B, C, H, W), dev), _feat(g, (B, C, H, W), dev)
        corr = Correlation(d, 1, d, 1, 1)
        ref = corr(x1, x2).clone()
        for _ in range(60):
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
            assert torch.equal(corr(x1, x2), ref)
    torch.cuda.synchronize()


def test_correlation_channels_last_stress_three_streams(dev):
    """The window-split channels-last kernel (csrc/correlation_wsplit.hip) keeps nothing in memory between workgroups, so
    a result can depend neither on the partition nor on what runs beside it.  Stress: conv5 at full size (2 x 2048 x 38 x 67,
    256 workgroups), conv4 planned for 240 CUs (426 workgroups, two rounds) and the native d = 16 window, 150 iterations
    each, while a second stream runs the proposal layer (select / sort + decode + two-phase NMS: few large and many tiny
    workgroups that take CUs away) and a third stream runs another correlation into its own output: every result
    bit-identical to the first quiet run, the neighbour's output untouched, and sampled entries equal to a float64
    recomputation (the test also catches LDS ring reuse bugs: DMA landing vs operand reads vs slot recycling under
    uneven load)."""
    from dtt.ops import correlation_forward_nhwc
    from dtt.rpn import _ProposalLayer
    from dtt.config import cfg
    s2, s3 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.Generator(device=dev).manual_seed(11)
    prop = _ProposalLayer(16, cfg.ANCHOR_SCALES, cfg.ANCHOR_RATIOS).to(dev)
    A = len(cfg.ANCHOR_SCALES) * len(cfg.ANCHOR_RATIOS)
    scores = torch.rand(4, 2 * A, 38, 67, device=dev, generator=g)
    deltas = 0.1 * torch.randn(4, 4 * A, 38, 67, device=dev, generator=g)
    info = torch.tensor([[600.0, 1067.0, 1.0]] * 4, device=dev)
    rois0 = prop((scores, deltas, info, "TEST")).clone()
    o1, o2 = _feat(g, (2, 1024, 38, 67), dev), _feat(g, (2, 1024, 38, 67), dev)
    o1, o2 = o1.contiguous(memory_format=torch.channels_last), o2.contiguous(memory_format=torch.channels_last)
    other0 = correlation_forward_nhwc(o1, o2, 8, 1, 8, 1, 1).clone()
    for (B, C, H, W, d, budget) in [(2, 2048, 38, 67, 8, 0), (2, 1024, 38, 67, 8, 240), (1, 512, 36, 63, 16, 0)]:
        x1, x2 = _feat(g, (B, C, H, W), dev), _feat(g, (B, C, H, W), dev)
        c1, c2 = x1.contiguous(memory_format=torch.channels_last), x2.contiguous(memory_format=torch.channels_last)
        ref = correlation_forward_nhwc(c1, c2, d, 1, d, 1, 1, max_workgroups=budget).clone()
        torch.cuda.synchronize()
        # float64 spot check of the quiet run (1/C * sum_c f1[p] f2[p + disp], zero outside the map)
        D = 2 * d + 1
        rs = np.random.RandomState(C)
        a64, b64 = x1.double().cpu().numpy(), x2.double().cpu().numpy()
        rf = ref.cpu().numpy()
        for _ in range(200):
            n, y, x, dy, dx = rs.randint(B), rs.randint(H), rs.randint(W), rs.randint(-d, d + 1), rs.randint(-d, d + 1)
            yy, xx = y + dy, x + dx
            want = 0.0 if not (0 <= yy < H and 0 <= xx < W) else float(a64[n, :, y, x] @ b64[n, :, yy, xx]) / C
            assert abs(rf[n, (dy + d) * D + dx + d, y, x] - want) <= 1e-5 * max(1.0, abs(want))
        s2.wait_stream(torch.cuda.current_stream()); s3.wait_stream(torch.cuda.current_stream())
        for it in range(150):
            with torch.cuda.stream(s2):
                rois = prop((scores, deltas, info, "TEST"))
            with torch.cuda.stream(s3):
                other = correlation_forward_nhwc(o1, o2, 8, 1, 8, 1, 1, max_workgroups=128 if it & 1 else 0)
            out = correlation_forward_nhwc(c1, c2, d, 1, d, 1, 1, max_workgroups=budget)
            if it % 10 == 9 or it < 3:
                torch.cuda.synchronize()
                assert torch.equal(out, ref), (C, d, it)
                assert torch.equal(other, other0) and torch.equal(rois, rois0), (C, d, it)
    torch.cuda.synchronize()


def _cfg5_rois(rs, R, B, im_h, im_w):
    x1 = rs.uniform(-20, im_w - 30, R); y1 = rs.uniform(-20, im_h - 30, R)
    w = rs.uniform(4, im_w * 0.8, R); h = rs.uniform(4, im_h * 0.8, R)
    return np.stack([rs.randint(0, B, R), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)


def test_roi_align_pool_crop_config5_size_vs_oracle(dev):
    """The legacy-head RoI ops at BASELINE config 5's shape (512-channel 36 x 63 map of a 563 x 1000 frame, 300 RoIs per
    image): the scalar C oracle still finishes in seconds here (OpenMP), so the forwards are compared bit for bit and the
    backwards through the oracle and the adjoint identity."""
    from oracle import oracle_lib as O
    from dtt.ops import RoIAlign, RoIAlignAvg, RoIPoolFunction, _RoICrop
    rs = np.random.RandomState(9)
    B, C, H, W, R = 2, 512, 36, 63, 600
    feat = rs.normal(size=(B, C, H, W)).astype(np.float32)
    rois = _cfg5_rois(rs, R, B, 563, 1000)
    ft = torch.from_numpy(feat).to(dev).requires_grad_(True)
    rt = torch.from_numpy(rois).to(dev)
    # RoI Align (8 x 8 samples, then the 2 x 2 / stride-1 average of RoIAlignAvg)
    ref8 = O.roi_align_forward(feat, rois, 8, 8, 1 / 16.0)
    out = RoIAlign(8, 8, 1 / 16.0)(ft, rt)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref8)
    go = torch.randn(out.shape, device=dev)
    out.backward(go)
    v = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        lhs = float((RoIAlign(8, 8, 1 / 16.0)(v, rt).double() * go.double()).sum())
        rhs = float((v.double() * ft.grad.double()).sum())
        avg = RoIAlignAvg(7, 7, 1 / 16.0)(ft.detach(), rt)
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    np.testing.assert_allclose(avg.cpu().numpy(), torch.nn.functional.avg_pool2d(torch.from_numpy(ref8), 2, 1).numpy(),
                               rtol=1e-6, atol=1e-6)
    # RoI max pooling: values and argmax exact, backward = scatter through argmax
    ref, ref_arg = O.roi_pool_forward(feat, rois, 7, 7, 1 / 16.0)
    ft2 = torch.from_numpy(feat).to(dev).requires_grad_(True)
    out2, arg = RoIPoolFunction.apply(ft2, rt, 7, 7, 1 / 16.0)
    np.testing.assert_array_equal(out2.detach().cpu().numpy(), ref)
    np.testing.assert_array_equal(arg.cpu().numpy(), ref_arg)
    go2 = torch.randn(out2.shape, device=dev)
    out2.backward(go2)
    gref = O.roi_pool_backward(go2.cpu().numpy(), rois, ref_arg, feat.shape, 7, 7, 1 / 16.0)
    np.testing.assert_allclose(ft2.grad.cpu().numpy(), gref, rtol=1e-5, atol=1e-5)
    # RoI crop (bilinear grid sampler, 14 x 14 grids as with CROP_RESIZE_WITH_MAX_POOL)
    grid = rs.uniform(-1.2, 1.2, size=(64, 14, 14, 2)).astype(np.float32)
    img = feat[:1].repeat(64, 0)[:, :128].copy()
    it = torch.from_numpy(img).to(dev).requires_grad_(True)
    oc = _RoICrop()(it, torch.from_numpy(grid).to(dev))
    np.testing.assert_array_equal(oc.detach().cpu().numpy(), O.roi_crop_forward(img, grid))
    go3 = torch.randn(oc.shape, device=dev)
    oc.backward(go3)
    np.testing.assert_allclose(it.grad.cpu().numpy(), O.roi_crop_backward(img, grid, go3.cpu().numpy()), rtol=1e-5, atol=1e-5)


def test_config4_per_rank_training_step_full_size():
    """BASELINE configs[3]'s per-rank workload: ONE Res-101 D&T training step at 600 x 1067 with 2 frame pairs (4 images)
    through the fused channels-last training trunk -- anchor targets, RoI / tracking target sampling, the five losses,
    backward through PSRoI pooling and the three correlations.  Checks: contract shapes, finite losses and gradients for
    every trainable parameter, and a directional derivative: the loss change along a random direction in the R-FCN / tracking
    head weights (the parameters fed by the PSRoI and correlation backward kernels) matches <grad, direction>."""
    from dtt.config import apply_dataset_defaults, cfg, cfg_from_file
    from dtt.dist import prepare_replica
    from dtt.synth import build_model, calibrate_batchnorm_, make_batch
    apply_dataset_defaults("imagenet_vid")
    cfg_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfgs", "res101.yml"))
    dev = torch.device("cuda:0")
    B, H, W = 2, 600, 1067
    model = build_model(101, cfg=cfg).to(dev)
    im, info, gt, nb = make_batch(B, H, W, seed=3, device=dev)
    calibrate_batchnorm_(model, im[:, 0])
    model.train()
    runner = prepare_replica(model, 1, channels_last=True)

    def loss_of():
        np.random.seed(cfg.RNG_SEED)   # same anchor / RoI samples on every evaluation
        out = runner(im, info, gt, nb)
        return out, out[4].mean() + out[5].mean() + out[6].mean() + out[7].mean() + out[9].mean()

    # one throw-away step first: on a fresh box the first pass makes MIOpen / hipBLASLt search and cache their kernels, and the
    # passes after it may run other algorithms than the first did -- last-bit differences that can swap two near-tied proposals
    # between the evaluations the directional derivative below compares
    loss_of()[1].backward()
    torch.cuda.synchronize()
    runner.zero_grad(set_to_none=True)
    out, loss = loss_of()
    loss.backward()
    runner.finish_gradients()
    torch.cuda.synchronize()
    N = cfg.TRAIN.BATCH_SIZE
    assert tuple(out[0].shape) == (2, B, N, 5) and tuple(out[1].shape) == (2, B, N, 31) and tuple(out[8].shape) == (2, B, N)
    assert tuple(out[3].shape) == (B * gt.size(2), 4)
    assert all(bool(torch.isfinite(out[i]).all()) for i in (4, 5, 6, 7, 9)) and np.isfinite(float(loss))
    n_grad = 0
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
            n_grad += 1
    assert n_grad > 100
    # (these heads do not feed the RPN: the proposals, hence the sampled RoIs, stay the same under the perturbation)
    heads = [model.RFCN_cls_net.weight, model.RFCN_bbox_net.weight, model.corr_bbox_net.weight]
    g = torch.Generator().manual_seed(1)
    dirs = [torch.randn(p.shape, generator=g).to(dev) * float(p.detach().abs().mean()) for p in heads]
    predicted = sum(float((p.grad * d).sum()) for p, d in zip(heads, dirs))
    eps = 0.02
    vals = []
    for sgn in (1.0, -1.0):
        with torch.no_grad():
            for p, d in zip(heads, dirs):
                p.add_(d, alpha=sgn * eps)
        vals.append(float(loss_of()[1].detach()))   # same (training) graph as the step above
        with torch.no_grad():
            for p, d in zip(heads, dirs):
                p.add_(d, alpha=-sgn * eps)
    measured = (vals[0] - vals[1]) / (2 * eps)
    assert abs(measured - predicted) <= 0.05 * max(abs(predicted), abs(measured)) + 1e-3, (measured, predicted)


def test_config5_graph_roi_align_path_full_size(dev):
    """BASELINE configs[4] end to end at its frame size: Res-101 D&T on a 563 x 1000 pair (36 x 63 maps), correlation
    d = 16 (33 x 33 displacements, 2859-channel tracking head) and the RoI-Align path (cfg.RFCN_ROI_FEATURES = "align":
    RoIAlignAvg(7, 7, 1/16) of the 512-channel `top` map for the 300 RoIs of each image, faster_rcnn.py:72-83 /
    roi_align/modules/roi_align.py:18-29) next to the PSRoI heads.  The model's RoI-Align features are checked against
    the oracle on the model's own `top` map and RoIs: 8 x 8 one-tap bilinear samples bit for bit, then the 2 x 2 average."""
    import copy
    from oracle import oracle_lib as O
    from dtt.config import cfg
    from dtt.fuse import fuse_for_inference
    from dtt.ops import RoIAlign
    from dtt.synth import build_model, calibrate_batchnorm_, make_batch
    c = copy.deepcopy(cfg)
    c.CORR_MAX_DISPLACEMENT = 16
    c.RFCN_ROI_FEATURES = "align"
    model = build_model(101, class_agnostic=True, cfg=c).to(dev).eval()
    assert model.corr_bbox_net.in_channels == 2859
    im, info, gt, nb = make_batch(1, 563, 1000, seed=11, device=dev)
    calibrate_batchnorm_(model, im[:, 0])
    fuse_for_inference(model)
    captured = {}
    orig = model._roi_features

    def spy(top, flat_rois):
        captured["top"], captured["rois"] = top.detach().clone(), flat_rois.detach().clone()
        return orig(top, flat_rois)
    model._roi_features = spy
    with torch.no_grad():
        out = model(im, info, gt, nb)
    torch.cuda.synchronize()
    rois, cls_prob, bbox_pred, tracking_pred = out[:4]
    R = rois.shape[2]
    assert tuple(rois.shape) == (2, 1, R, 5) and tuple(cls_prob.shape) == (2, 1, R, 31) and R == c.TEST.RPN_POST_NMS_TOP_N
    assert tuple(tracking_pred.shape) == (R, 4) and bool(torch.isfinite(tracking_pred).all())
    feat = model.roi_feat
    top = captured["top"].contiguous()
    assert tuple(top.shape) == (2, 512, 36, 63) and tuple(feat.shape) == (2 * R, 512, 7, 7)
    assert torch.equal(captured["rois"].view(2, 1, R, 5)[0], rois[0])        # the RoIs that were scored (leg 0 as is)
    ref8 = O.roi_align_forward(top.cpu().numpy(), captured["rois"].cpu().numpy(), 8, 8, 1 / 16.0)
    got8 = RoIAlign(8, 8, 1 / 16.0)(top, captured["rois"])
    np.testing.assert_array_equal(got8.cpu().numpy(), ref8)
    ref = torch.nn.functional.avg_pool2d(torch.from_numpy(ref8), 2, 1).numpy()
    np.testing.assert_allclose(feat.cpu().numpy(), ref, rtol=1e-6, atol=1e-6 * max(1.0, float(np.abs(ref).max())))
