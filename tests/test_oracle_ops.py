"""The C oracle (oracle/dtt_oracle.c) against INDEPENDENT formulations: shifted-product correlation + float64
autograd, brute-force pooling, O(N^2) greedy NMS, F.grid_sample for the bilinear ops.  CPU only.  (The reference
ships no tests or golden vectors for its CUDA ops; the oracle is pinned bit for bit against the reference's own
kernels in tests/test_oracle_ref_golden.py / tests/test_gpu_ref_kernels.py -- these second formulations are the
cross-check that does not depend on the reference at all.)"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle_lib as O


def corr_shifted(x1, x2, pad, d, s1, s2):
    """out[n, tj*D+ti, y, x] = mean_c x1p[n,c,y*s1+d, x*s1+d] * x2p[n,c,y*s1+d+tj*s2, x*s1+d+ti*s2] (k=1)."""
    B, C, H, W = x1.shape
    r = d // s2
    big = pad + r * s2 + 2
    p1 = F.pad(x1, (pad, pad, pad, pad))
    p2 = F.pad(x2, (big, big, big, big))  # extra margin so every shifted window exists
    pH, pW = H + 2 * pad, W + 2 * pad
    oh = int(np.ceil((pH - 2 * d) / s1))
    ow = int(np.ceil((pW - 2 * d) / s1))
    ys = torch.arange(oh) * s1 + d
    xs = torch.arange(ow) * s1 + d
    a = p1[:, :, ys][:, :, :, xs]
    outs = []
    for tj in range(-r, r + 1):
        for ti in range(-r, r + 1):
            b = p2[:, :, ys + tj * s2 + (big - pad)][:, :, :, xs + ti * s2 + (big - pad)]
            outs.append((a * b).mean(1))
    return torch.stack(outs, 1)


@pytest.mark.parametrize("B,C,H,W,pad,d,s1,s2", [(2, 12, 10, 13, 4, 4, 1, 1), (1, 8, 15, 18, 8, 8, 2, 2),
                                                 (1, 6, 9, 9, 3, 4, 1, 2), (1, 5, 12, 11, 6, 4, 1, 1)])
def test_correlation_forward_backward(B, C, H, W, pad, d, s1, s2):
    g = torch.Generator().manual_seed(B * 100 + C)
    x1 = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    x2 = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    ref = corr_shifted(x1, x2, pad, d, s1, s2)
    out = O.correlation_forward(x1.detach().float().numpy(), x2.detach().float().numpy(), pad, 1, d, s1, s2)
    assert out.shape == tuple(ref.shape)
    np.testing.assert_allclose(out, ref.detach().numpy(), atol=2e-6)
    gout = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(gout)
    g1, g2 = O.correlation_backward(gout.float().numpy(), x1.detach().float().numpy(), x2.detach().float().numpy(),
                                    pad, 1, d, s1, s2)
    np.testing.assert_allclose(g1, x1.grad.numpy(), atol=5e-6)
    np.testing.assert_allclose(g2, x2.grad.numpy(), atol=5e-6)


def test_correlation_self_peak():
    """A map correlated with itself peaks at zero displacement with value mean(x^2)."""
    rng = np.random.RandomState(0)
    x = rng.normal(size=(1, 16, 12, 14)).astype(np.float32)
    out = O.correlation_forward(x, x, 4, 1, 4, 1, 1)
    centre = out[0, 4 * 9 + 4]
    np.testing.assert_allclose(centre, (x[0] ** 2).mean(0), rtol=1e-5)
    assert (out[0].max(0) <= centre + 1e-6).mean() > 0.9


def psroi_brute(feat, rois, P, scale, od):
    R = rois.shape[0]
    H, W = feat.shape[2:]
    out = np.zeros((R, od, P, P), dtype=np.float64)
    for n in range(R):
        b = int(rois[n, 0])
        rnd = lambda v: np.floor(abs(v) + 0.5) * np.sign(v)  # half away from zero
        sw, sh = rnd(rois[n, 1]) * scale, rnd(rois[n, 2]) * scale
        ew, eh = (rnd(rois[n, 3]) + 1) * scale, (rnd(rois[n, 4]) + 1) * scale
        bw, bh = max(ew - sw, 0.1) / P, max(eh - sh, 0.1) / P
        for ph in range(P):
            for pw in range(P):
                h0 = min(max(int(np.floor(ph * bh + sh)), 0), H); h1 = min(max(int(np.ceil((ph + 1) * bh + sh)), 0), H)
                w0 = min(max(int(np.floor(pw * bw + sw)), 0), W); w1 = min(max(int(np.ceil((pw + 1) * bw + sw)), 0), W)
                if h1 <= h0 or w1 <= w0:
                    continue
                for c in range(od):
                    out[n, c, ph, pw] = feat[b, (c * P + ph) * P + pw, h0:h1, w0:w1].mean()
    return out


def test_psroi_pool_forward_backward():
    rng = np.random.RandomState(1)
    B, od, P, H, W = 2, 3, 7, 14, 17
    feat = rng.normal(size=(B, od * P * P, H, W)).astype(np.float32)
    x1 = rng.uniform(0, 200, size=20); y1 = rng.uniform(0, 150, size=20)
    rois = np.stack([rng.randint(0, B, 20), x1, y1, x1 + rng.uniform(10, 120, 20), y1 + rng.uniform(10, 120, 20)], 1)
    rois = np.round(rois * 4) / 4  # quarter-pixel coordinates: bin edges exactly representable in binary32
    rois = rois.astype(np.float32)
    out, mapping = O.psroi_pool_forward(feat, rois, P, P, 1 / 16.0, P, od)
    np.testing.assert_allclose(out, psroi_brute(feat.astype(np.float64), rois.astype(np.float64), P, 1 / 16.0, od), atol=2e-6)
    assert mapping[0, 1, 2, 3] == (1 * P + 2) * P + 3
    const = np.full_like(feat, 2.5)
    np.testing.assert_allclose(O.psroi_pool_forward(const, rois, P, P, 1 / 16.0, P, od)[0][:, :, 1:-1, 1:-1], 2.5)
    # backward = adjoint of forward: <pool(x), g> == <x, pool^T(g)>
    gout = rng.normal(size=out.shape).astype(np.float32)
    gin = O.psroi_pool_backward(gout, rois, feat.shape, P, P, 1 / 16.0, P, od, mapping)
    np.testing.assert_allclose((out.astype(np.float64) * gout).sum(), (feat.astype(np.float64) * gin).sum(), rtol=1e-4)


def nms_greedy(dets, thresh):
    keep, alive = [], np.ones(len(dets), bool)
    b = dets[:, :4].astype(np.float32)
    for i in range(len(dets)):
        if not alive[i]:
            continue
        keep.append(i)
        for j in range(i + 1, len(dets)):
            if not alive[j]:
                continue
            w = max(np.float32(min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0]) + np.float32(1)), np.float32(0))
            h = max(np.float32(min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1]) + np.float32(1)), np.float32(0))
            inter = np.float32(w * h)
            sa = np.float32((b[i, 2] - b[i, 0] + np.float32(1)) * (b[i, 3] - b[i, 1] + np.float32(1)))
            sb = np.float32((b[j, 2] - b[j, 0] + np.float32(1)) * (b[j, 3] - b[j, 1] + np.float32(1)))
            if np.float32(inter / np.float32(np.float32(sa + sb) - inter)) > np.float32(thresh):
                alive[j] = False
    return np.array(keep)


@pytest.mark.parametrize("n,thresh", [(1, 0.7), (70, 0.7), (200, 0.3), (333, 0.5)])
def test_nms_equals_quadratic_greedy(n, thresh):
    rng = np.random.RandomState(n)
    c = rng.uniform(0, 150, size=(n, 2))
    wh = rng.uniform(10, 80, size=(n, 2))
    dets = np.concatenate([c, c + wh, np.sort(rng.uniform(size=(n, 1)), 0)[::-1]], 1).astype(np.float32)
    keep, mask = O.nms(dets, thresh, return_mask=True)
    np.testing.assert_array_equal(keep, nms_greedy(dets, thresh))
    assert np.all(np.diff(keep) > 0)
    np.testing.assert_array_equal(O.nms(dets[keep], thresh), np.arange(len(keep)))  # idempotent
    assert O.nms(np.zeros((0, 5), np.float32), thresh).size == 0


def test_roi_crop_equals_grid_sample():
    rng = np.random.RandomState(2)
    img = rng.normal(size=(2, 5, 11, 13)).astype(np.float32)
    grid = rng.uniform(-1.2, 1.2, size=(6, 7, 7, 2)).astype(np.float32)  # (y, x)
    out = O.roi_crop_forward(img, grid)
    gxy = torch.from_numpy(grid[..., ::-1].copy())  # grid_sample wants (x, y)
    ref = F.grid_sample(torch.from_numpy(img).repeat_interleave(3, 0), gxy, mode="bilinear", padding_mode="zeros",
                        align_corners=True)
    np.testing.assert_allclose(out, ref.numpy(), atol=1e-5)
    gout = rng.normal(size=out.shape).astype(np.float32)
    x = torch.from_numpy(img).double().requires_grad_(True)
    F.grid_sample(x.repeat_interleave(3, 0), gxy.double(), mode="bilinear", padding_mode="zeros",
                  align_corners=True).backward(torch.from_numpy(gout).double())
    np.testing.assert_allclose(O.roi_crop_backward(img, grid, gout), x.grad.numpy(), atol=1e-4)


def test_roi_align_equals_grid_sample_inside_the_map():
    rng = np.random.RandomState(3)
    B, C, H, W = 1, 4, 12, 15
    feat = rng.normal(size=(B, C, H, W)).astype(np.float32)
    rois = np.array([[0, 16, 24, 150, 120], [0, 40, 8, 170, 100]], dtype=np.float32)  # samples stay < H-1, W-1
    out = O.roi_align_forward(feat, rois, 8, 8, 1 / 16.0)
    for n in range(2):
        x1, y1, x2, y2 = rois[n, 1:] / 16.0
        bw, bh = (x2 - x1 + 1) / 7, (y2 - y1 + 1) / 7
        xs = x1 + np.arange(8) * bw
        ys = y1 + np.arange(8) * bh
        gx = torch.tensor(xs / (W - 1) * 2 - 1).float()
        gy = torch.tensor(ys / (H - 1) * 2 - 1).float()
        grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], -1).unsqueeze(0)
        ref = F.grid_sample(torch.from_numpy(feat), grid, mode="bilinear", align_corners=True)
        np.testing.assert_allclose(out[n], ref[0].numpy(), atol=1e-4)
    gout = rng.normal(size=out.shape).astype(np.float32)
    gin = O.roi_align_backward(gout, rois, feat.shape, 8, 8, 1 / 16.0)
    np.testing.assert_allclose((out.astype(np.float64) * gout).sum(), (feat.astype(np.float64) * gin).sum(), rtol=1e-4)


def test_roi_pool_full_image_is_adaptive_max_pool():
    rng = np.random.RandomState(4)
    feat = rng.normal(size=(1, 3, 14, 21)).astype(np.float32)
    rois = np.array([[0, 0, 0, 21 * 16 - 16, 14 * 16 - 16]], dtype=np.float32)  # rounds to the whole 14x21 map
    out, arg = O.roi_pool_forward(feat, rois, 7, 7, 1 / 16.0)
    ref = F.adaptive_max_pool2d(torch.from_numpy(feat), (7, 7))
    np.testing.assert_array_equal(out[0], ref[0].numpy())
    np.testing.assert_array_equal(feat.reshape(-1)[arg.reshape(-1)], out.reshape(-1))
    gout = rng.normal(size=out.shape).astype(np.float32)
    gin = O.roi_pool_backward(gout, rois, arg, feat.shape, 7, 7, 1 / 16.0)
    expect = np.zeros(feat.size, dtype=np.float32)
    np.add.at(expect, arg.reshape(-1), gout.reshape(-1))
    np.testing.assert_allclose(gin.reshape(-1), expect, atol=1e-6)
