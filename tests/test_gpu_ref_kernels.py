"""The reference's OWN operator kernels (oracle/_ref, built by oracle/build_ref.sh with ROCm's hipify-perl from the
sources under /root/reference) run on the MI355X as a second checker:

  * they pin the CPU oracle (oracle/dtt_oracle.c) -- same inputs, the restatement must reproduce the reference
    kernels bit for bit where no atomics are involved;
  * libdtt_hip.so is compared with them directly, including at BASELINE.json's full sizes, where the CPU oracle is
    too slow to be the checker.

Skipped (loudly) when oracle/_ref was not built; nothing here reads /root/reference at run time."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_lib as O
from oracle import ref_kernels as RK
from test_gpu_ops import cu, random_rois

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not RK.available(), reason="oracle/_ref/libdtt_ref_kernels.so not built "
                                                            "(oracle/build_ref.sh needs /root/reference + hipify-perl)")]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from dtt import _lib
    _lib.lib()
    return torch.device("cuda:0")


CORR_CASES = [  # B, C, H, W, pad, k, d, s1, s2
    (2, 37, 13, 17, 4, 1, 4, 1, 1),
    (1, 64, 19, 23, 8, 1, 8, 1, 2),
    (2, 16, 12, 15, 8, 1, 8, 1, 1),
    (1, 33, 21, 18, 2, 1, 2, 2, 2),
    (1, 8, 14, 14, 3, 3, 2, 1, 1),
    (1, 40, 9, 31, 0, 1, 2, 1, 1),     # pad < displacement: shrinking output
]


@pytest.mark.parametrize("case", CORR_CASES)
def test_correlation_oracle_reproduces_reference_kernels(dev, case):
    B, C, H, W, pad, k, d, s1, s2 = case
    rng = np.random.RandomState(sum(case))
    x1 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    x2 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    ref = RK.correlation_forward(x1, x2, pad, k, d, s1, s2)
    orc = O.correlation_forward(x1, x2, pad, k, d, s1, s2)
    assert ref.shape == orc.shape
    np.testing.assert_array_equal(orc, ref)  # same 32 strided partials + serial sum, no contraction
    # hipcc's default contraction (the nvcc -fmad=true analogue) stays inside the parity tolerance
    np.testing.assert_allclose(RK.correlation_forward(x1, x2, pad, k, d, s1, s2, fma=True), ref, rtol=0, atol=1e-5)
    from dtt.ops import Correlation
    t1, t2 = cu(x1, dev).requires_grad_(True), cu(x2, dev).requires_grad_(True)
    out = Correlation(pad, k, d, s1, s2, 1)(t1, t2)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    # backward: compared with the reference only where the reference is right -- for stride1 > 1 it indexes out of bounds and
    # for kernel_size > 1 its backward is not the adjoint of its forward; the product handles both as
    # the exact adjoint (autograd-checked in tests/test_gpu_ops.py)
    if s1 == 1 and k == 1:
        g = rng.normal(size=ref.shape).astype(np.float32)
        r1, r2 = RK.correlation_backward(g, x1, x2, pad, k, d, s1, s2)
        o1, o2 = O.correlation_backward(g, x1, x2, pad, k, d, s1, s2)
        np.testing.assert_allclose(o1, r1, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(o2, r2, rtol=1e-5, atol=1e-5)
        out.backward(cu(g, dev))
        np.testing.assert_allclose(t1.grad.cpu().numpy(), r1, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(t2.grad.cpu().numpy(), r2, rtol=1e-4, atol=1e-5)


def test_psroi_oracle_reproduces_reference_kernels(dev):
    from dtt.ops import _PSRoIPooling
    rng = np.random.RandomState(11)
    for (B, od, g, H, W, n) in [(2, 5, 7, 24, 31, 40), (1, 31, 7, 38, 67, 64), (3, 4, 3, 10, 12, 17), (1, 8, 1, 9, 9, 9)]:
        C = od * g * g
        feat = rng.normal(size=(B, C, H, W)).astype(np.float32)
        rois = random_rois(rng, n, B, H * 16, W * 16)
        ref, ref_map = RK.psroi_pool_forward(feat, rois, g, g, 1 / 16.0, g, od)
        orc, orc_map = O.psroi_pool_forward(feat, rois, g, g, 1 / 16.0, g, od)
        np.testing.assert_array_equal(orc_map, ref_map)
        np.testing.assert_array_equal(orc, ref)
        ft = cu(feat, dev).requires_grad_(True)
        out = _PSRoIPooling(g, g, 1 / 16.0, g, od)(ft, cu(rois, dev))
        np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
        top = rng.normal(size=ref.shape).astype(np.float32)
        rb = RK.psroi_pool_backward(top, rois, feat.shape, g, g, 1 / 16.0, g, od, ref_map)  # atomics: order free
        np.testing.assert_allclose(O.psroi_pool_backward(top, rois, feat.shape, g, g, 1 / 16.0, g, od), rb, rtol=1e-5,
                                   atol=1e-5)
        out.backward(cu(top, dev))
        np.testing.assert_allclose(ft.grad.cpu().numpy(), rb, rtol=1e-5, atol=1e-5)


def test_roi_align_pool_crop_oracle_reproduces_reference_kernels(dev):
    from dtt.ops import RoIAlign, RoIPoolFunction, _RoICrop
    rng = np.random.RandomState(12)
    for (B, C, H, W, n) in [(2, 9, 20, 27, 33), (1, 32, 38, 67, 50), (3, 3, 7, 9, 12)]:
        feat = rng.normal(size=(B, C, H, W)).astype(np.float32)
        rois = random_rois(rng, n, B, H * 16, W * 16)
        # RoI align
        ref = RK.roi_align_forward(feat, rois, 7, 7, 1 / 16.0)
        np.testing.assert_array_equal(O.roi_align_forward(feat, rois, 7, 7, 1 / 16.0), ref)
        ft = cu(feat, dev).requires_grad_(True)
        out = RoIAlign(7, 7, 1 / 16.0)(ft, cu(rois, dev))
        np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
        top = rng.normal(size=ref.shape).astype(np.float32)
        rb = RK.roi_align_backward(top, rois, feat.shape, 7, 7, 1 / 16.0)
        np.testing.assert_allclose(O.roi_align_backward(top, rois, feat.shape, 7, 7, 1 / 16.0), rb, rtol=1e-5, atol=1e-5)
        out.backward(cu(top, dev))
        np.testing.assert_allclose(ft.grad.cpu().numpy(), rb, rtol=1e-5, atol=1e-5)
        # RoI max pooling
        ref, ref_arg = RK.roi_pool_forward(feat, rois, 7, 7, 1 / 16.0)
        orc, orc_arg = O.roi_pool_forward(feat, rois, 7, 7, 1 / 16.0)
        np.testing.assert_array_equal(orc, ref)
        np.testing.assert_array_equal(orc_arg, ref_arg)
        ft = cu(feat, dev).requires_grad_(True)
        out, arg = RoIPoolFunction.apply(ft, cu(rois, dev), 7, 7, 1 / 16.0)
        np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
        np.testing.assert_array_equal(arg.cpu().numpy(), ref_arg)
        rb = RK.roi_pool_backward(top, rois, ref_arg, feat.shape, 7, 7, 1 / 16.0)
        np.testing.assert_allclose(O.roi_pool_backward(top, rois, ref_arg, feat.shape, 7, 7, 1 / 16.0), rb, rtol=1e-5,
                                   atol=1e-5)
        out.backward(cu(top, dev))
        np.testing.assert_allclose(ft.grad.cpu().numpy(), rb, rtol=1e-5, atol=1e-5)
        # RoI crop (bilinear sampler); B * rois_per_image grids
        grid = rng.uniform(-1.3, 1.3, size=(B * 4, 7, 7, 2)).astype(np.float32)
        ref = RK.roi_crop_forward(feat, grid)
        np.testing.assert_array_equal(O.roi_crop_forward(feat, grid), ref)
        it = cu(feat, dev).requires_grad_(True)
        out = _RoICrop()(it, cu(grid, dev))
        np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
        go = rng.normal(size=ref.shape).astype(np.float32)
        rb = RK.roi_crop_backward(feat, grid, go)
        np.testing.assert_allclose(O.roi_crop_backward(feat, grid, go), rb, rtol=1e-5, atol=1e-5)
        out.backward(cu(go, dev))
        np.testing.assert_allclose(it.grad.cpu().numpy(), rb, rtol=1e-5, atol=1e-5)


def _boxes(rng, n, span=600.0, ties=False):
    x1 = rng.uniform(0, span, n); y1 = rng.uniform(0, span * 0.6, n)
    w = rng.uniform(4, span * 0.4, n); h = rng.uniform(4, span * 0.3, n)
    s = np.sort(rng.uniform(0, 1, n))[::-1]
    if ties:
        s = np.round(s, 2)
    d = np.stack([x1, y1, x1 + w, y1 + h, s], 1).astype(np.float32)
    d[n // 2] = d[n // 3]; d[n // 2, 4] = d[n // 2 - 1, 4]  # an exact duplicate box
    return d


def test_nms_oracle_reproduces_reference_kernel(dev):
    from dtt.ops import nms
    rng = np.random.RandomState(13)
    for n, thresh, ties in [(1, 0.7, False), (63, 0.7, False), (64, 0.5, True), (65, 0.3, False), (1000, 0.7, False),
                            (6000, 0.7, True)]:
        dets = _boxes(rng, n, ties=ties) if n > 3 else np.array([[0, 0, 10, 10, 0.9]] * n, np.float32)
        ref = RK.nms(dets, thresh)
        np.testing.assert_array_equal(O.nms(dets, thresh), ref)
        keep = nms(cu(dets, dev), thresh).view(-1).cpu().numpy()
        np.testing.assert_array_equal(keep, ref)
        np.testing.assert_array_equal(RK.nms(dets, thresh, fma=True), ref)


def test_full_size_against_reference_kernels(dev):
    """BASELINE.json configs[1] shapes (600 x 1067 -> 38 x 67 at stride 16): conv5 correlation (2048 ch, d = 8),
    R-FCN class PSRoI pooling (31 x 7 x 7 score maps, 300 RoIs), proposal-sized NMS (12000 boxes): the product
    against the reference kernels themselves, where the CPU oracle would take minutes."""
    from dtt.ops import Correlation, _PSRoIPooling, nms
    rng = np.random.RandomState(14)
    B, C, H, W, d = 1, 2048, 38, 67, 8
    x1 = np.maximum(rng.normal(size=(B, C, H, W)), 0).astype(np.float32)  # post-ReLU features
    x2 = np.maximum(rng.normal(size=(B, C, H, W)), 0).astype(np.float32)
    ref = RK.correlation_forward(x1, x2, d, 1, d, 1, 1)
    t1, t2 = cu(x1, dev).requires_grad_(True), cu(x2, dev).requires_grad_(True)
    out = Correlation(d, 1, d, 1, 1, 1)(t1, t2)
    assert ref.shape == (B, (2 * d + 1) ** 2, H, W)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-6)
    g = rng.normal(size=ref.shape).astype(np.float32)
    r1, r2 = RK.correlation_backward(g, x1, x2, d, 1, d, 1, 1)
    out.backward(cu(g, dev))
    np.testing.assert_allclose(t1.grad.cpu().numpy(), r1, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(t2.grad.cpu().numpy(), r2, rtol=1e-4, atol=1e-5)

    od, gs, n = 31, 7, 300
    feat = rng.normal(size=(2, od * gs * gs, H, W)).astype(np.float32)
    rois = random_rois(rng, 2 * n, 2, 600, 1067)
    ref, ref_map = RK.psroi_pool_forward(feat, rois, gs, gs, 1 / 16.0, gs, od)
    ft = cu(feat, dev).requires_grad_(True)
    out = _PSRoIPooling(gs, gs, 1 / 16.0, gs, od)(ft, cu(rois, dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
    top = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(cu(top, dev))
    np.testing.assert_allclose(ft.grad.cpu().numpy(),
                               RK.psroi_pool_backward(top, rois, feat.shape, gs, gs, 1 / 16.0, gs, od, ref_map), rtol=1e-5,
                               atol=1e-5)

    dets = _boxes(rng, 12000, span=1000.0, ties=True)
    np.testing.assert_array_equal(nms(cu(dets, dev), 0.7).view(-1).cpu().numpy(), RK.nms(dets, 0.7))


def _time_gpu(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us


def test_speed_against_reference_kernels(dev):
    """The reference's own kernels and libdtt_hip.so timed side by side on the same MI355X at the 600 x 1067 D&T shapes
    (B = 2 frame pairs: what bench.py runs per step).  Reference side = the launcher calls only (its per-call
    allocations and fills are left out, in its favour).  The table goes to gpurun_out/ref_kernel_timing.txt."""
    import ctypes
    import time
    from dtt.ops import Correlation, _PSRoIPooling, nms
    L = RK.lib()
    rng = np.random.RandomState(15)
    rows = []
    # conv5 correlation, 2048 ch, d = 8
    B, C, H, W, d = 2, 2048, 38, 67, 8
    a = torch.relu(torch.randn(B, C, H, W, device=dev)); b = torch.relu(torch.randn(B, C, H, W, device=dev))
    r1 = torch.zeros(B, H + 2 * d, W + 2 * d, C, device=dev); r2 = torch.zeros_like(r1)
    out = torch.zeros(B, (2 * d + 1) ** 2, H, W, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    ref_fwd = lambda: L.Correlation_forward_cuda_kernel(P(out), *out.shape, *out.stride(), P(a), C, H, W, *a.stride(), P(b), C,
                                                        *b.stride(), P(r1), P(r2), d, 1, d, 1, 1, 1, None)
    corr = Correlation(d, 1, d, 1, 1, 1)
    mine = corr(a, b)
    ref_fwd(); torch.cuda.synchronize()
    np.testing.assert_allclose(mine.cpu().numpy(), out.cpu().numpy(), rtol=1e-4, atol=1e-6)
    rows.append(("correlation fwd conv5 (2x2048x38x67, d=8)", _time_gpu(ref_fwd, 3), _time_gpu(lambda: corr(a, b), 20)))
    g = torch.randn_like(out); g1 = torch.zeros_like(a); g2 = torch.zeros_like(b)
    ref_bwd = lambda: L.Correlation_backward_cuda_kernel(P(g), *g.shape, *g.stride(), P(a), C, H, W, *a.stride(), P(b),
                                                         *b.stride(), P(g1), *g1.stride(), P(g2), C, *g2.stride(), P(r1), P(r2),
                                                         d, 1, d, 1, 1, 1, None)
    a_, b_ = a.clone().requires_grad_(True), b.clone().requires_grad_(True)

    def mine_bwd():
        a_.grad = b_.grad = None
        corr(a_, b_).backward(g)
    rows.append(("correlation fwd+bwd conv5", _time_gpu(lambda: (ref_fwd(), ref_bwd()), 2), _time_gpu(mine_bwd, 10)))
    # R-FCN class PSRoI pooling: 31 x 7 x 7 score maps, 300 RoIs per image, 4 images
    n_img, od, gs = 4, 31, 7
    feat = torch.randn(n_img, od * gs * gs, H, W, device=dev)
    rois = cu(random_rois(rng, 300 * n_img, n_img, 600, 1067), dev)
    rois[:, 0] = torch.arange(n_img, device=dev).repeat_interleave(300).float()
    top = torch.zeros(300 * n_img, od, gs, gs, device=dev); mp = torch.zeros_like(top, dtype=torch.int32)
    ref_ps = lambda: L.PSROIPoolForwardLauncher(P(feat), 1 / 16.0, rois.shape[0], H, W, feat.shape[1], gs, gs, P(rois), gs, od,
                                                P(top), P(mp), None)
    ps = _PSRoIPooling(gs, gs, 1 / 16.0, gs, od)
    ref_ps(); torch.cuda.synchronize()
    assert torch.equal(ps(feat, rois), top)
    rows.append(("PSRoI fwd (4x1519x38x67, 1200 RoIs)", _time_gpu(ref_ps, 10), _time_gpu(lambda: ps(feat, rois), 20)))
    # NMS over 12000 sorted boxes (RPN train-time pre-NMS size): the reference copies the mask to the host and sweeps there
    dets = _boxes(rng, 12000, span=1000.0)
    dd = cu(dets, dev)
    t0 = time.perf_counter(); ref_keep = RK.nms(dets, 0.7); t_ref = (time.perf_counter() - t0) * 1e6
    nms(dd, 0.7); torch.cuda.synchronize()
    t0 = time.perf_counter(); keep = nms(dd, 0.7); torch.cuda.synchronize(); t_mine = (time.perf_counter() - t0) * 1e6
    np.testing.assert_array_equal(keep.view(-1).cpu().numpy(), ref_keep)
    rows.append(("NMS 12000 boxes (wall, incl. host sync)", t_ref, t_mine))
    lines = ["%-46s %14s %14s %8s" % ("op (same MI355X, same inputs, same results)", "reference us", "libdtt_hip us", "ratio")]
    lines += ["%-46s %14.1f %14.1f %7.1fx" % (n, r, m, r / m) for n, r, m in rows]
    text = "\n".join(lines)
    print("\n" + text)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "ref_kernel_timing.txt"), "w") as f:
            f.write(text + "\n")
    for n, r, m in rows:
        assert m < r, "%s: libdtt_hip (%.1f us) is not faster than the reference kernel (%.1f us)" % (n, m, r)
