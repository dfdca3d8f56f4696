"""Parity of the HIP ops (through the C ABI of libdtt_hip.so) against the CPU oracle.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_lib as O
from oracle import rpn_oracle as ro

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from dtt import _lib
    _lib.lib()  # must load: no fallback
    return torch.device("cuda:0")


def cu(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def random_rois(rng, n, batch, im_h, im_w, extra=True):
    x1 = rng.uniform(-20, im_w - 10, size=n)
    y1 = rng.uniform(-20, im_h - 10, size=n)
    w = rng.uniform(1, im_w * 0.7, size=n)
    h = rng.uniform(1, im_h * 0.7, size=n)
    rois = np.stack([rng.randint(0, batch, size=n), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    if extra and n >= 8:
        rois[0, 1:] = [0, 0, im_w - 1, im_h - 1]          # full image
        rois[1, 1:] = [5, 5, 5, 5]                        # 1 px
        rois[2, 1:] = [im_w + 50, im_h + 50, im_w + 90, im_h + 90]  # fully outside
        rois[3, 1:] = [16, 32, 16 + 7 * 16 - 1, 32 + 7 * 16 - 1]   # bin edges on exact pixel boundaries
        rois[4, 1:] = [100.5, 50.5, 30.5, 20.5]           # inverted
        rois[5, 1:] = [-40, -40, 30, 30]                  # straddles the border
        rois[6, 1:] = np.round(rois[6, 1:])               # integer coordinates
        rois[7, 1:] = rois[7, 1:] + 0.5                   # .5 coordinates (round half away from zero)
    return rois


# ----------------------------------------------------------------------------------------------- NMS
def clustered_dets(rng, n, spread=400.0):
    centers = rng.uniform(0, spread, size=(max(n // 12, 1), 2))
    c = centers[rng.randint(0, len(centers), size=n)] + rng.normal(0, 6, size=(n, 2))
    wh = rng.uniform(20, 120, size=(n, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1)
    scores = np.sort(rng.uniform(0, 1, size=n))[::-1]
    return np.concatenate([boxes, scores[:, None]], 1).astype(np.float32)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 300, 1000, 1024, 1025, 2049, 6000])
@pytest.mark.parametrize("thresh", [0.7, 0.3])
def test_nms_bit_exact(dev, n, thresh):
    from dtt.ops import nms
    rng = np.random.RandomState(n * 7 + int(thresh * 10))
    dets = clustered_dets(rng, n)
    if n >= 64:
        dets[10, :4] = dets[3, :4]  # exact duplicates
        dets[40, :4] = dets[3, :4]
    ref = O.nms(dets, thresh)
    got = nms(cu(dets, dev), thresh)
    assert got.dtype == torch.int32 and got.dim() == 2 and got.size(1) == 1
    np.testing.assert_array_equal(got.cpu().numpy().ravel(), ref)


@pytest.mark.parametrize("n,clusters,mk", [(12000, 2600, 2000),    # the training proposal layer: 12000 -> 2000, reached in the second phase
                                          (12000, 1500, 2000),    # heavy overlap: fewer clusters than max_keep, every super-chunk is walked
                                          (12000, 5000, 2000),    # spread out: max_keep inside the first phase
                                          (20000, 3000, 2000), (20000, 900, 0), (12345, 700, 0)])   # 20 super-chunks; keep all; a ragged last block
def test_nms_training_size_sweep_pipeline(dev, n, clusters, mk):
    """The sweep as a software pipeline (round 5: the next diagonal super-block is loaded under the serial walk, the far columns of a
    super-chunk's kept rows are OR-ed beside the NEXT walk by the other waves, the keep list leaves after the walk): keep lists at
    the training proposal layer's sizes and beyond, single- and two-phase, against the oracle (nms_cuda_kernel.cu:131-144)."""
    from dtt.ops import nms
    rng = np.random.RandomState(n + clusters + mk)
    ctr = rng.uniform(0, 1000, size=(clusters, 2))
    wh = rng.uniform(30, 200, size=(clusters, 2))
    which = rng.randint(0, clusters, size=n)
    jitter = rng.normal(0, 2.5, size=(n, 4))
    boxes = np.concatenate([ctr[which] - wh[which] / 2, ctr[which] + wh[which] / 2], 1) + jitter
    dets = np.concatenate([boxes, np.sort(rng.uniform(0, 1, n))[::-1][:, None]], 1).astype(np.float32)
    ref = O.nms(dets, 0.7)
    got = nms(cu(dets, dev), 0.7, max_keep=mk).cpu().numpy().ravel()
    np.testing.assert_array_equal(got, ref[:mk] if mk else ref)
    again = nms(cu(dets, dev), 0.7, max_keep=mk).cpu().numpy().ravel()
    np.testing.assert_array_equal(got, again)


def test_nms_largest_box_count_and_the_error_beyond_it(dev):
    """include/dtt_hip.h: boxes_num <= 65408 (the pipelined sweep keeps its removal words beside 156 KB of staging area in one CU's
    LDS).  At the bound the keep list still equals the oracle's (well separated boxes + a cluster: the oracle's O(n * kept)
    loop stays cheap); one more box is a clean error through the C ABI, not a launch failure."""
    from dtt.ops import nms
    n = 65408
    rng = np.random.RandomState(7)
    gx, gy = np.meshgrid(np.arange(256), np.arange(256))
    ctr = np.stack([gx.ravel(), gy.ravel()], 1)[:n].astype(np.float64) * 40.0          # 40 px apart, 30 px boxes: no overlap
    boxes = np.concatenate([ctr, ctr + 30.0], 1)
    boxes[1000:1400] = boxes[1000] + rng.normal(0, 1.0, size=(400, 4))                  # one cluster of 400 heavily overlapping boxes
    dets = np.concatenate([boxes, np.sort(rng.uniform(0, 1, n))[::-1][:, None]], 1).astype(np.float32)
    keep = nms(cu(dets, dev), 0.7, max_keep=300).cpu().numpy().ravel()
    ref = O.nms(dets[:4000], 0.7)          # (the first 300 survivors lie among the first 4000 boxes: everything outside the cluster survives)
    np.testing.assert_array_equal(keep, ref[:300])
    more = np.concatenate([dets, dets[:1]], 0)
    with pytest.raises(RuntimeError, match="65408|boxes"):
        nms(cu(more, dev), 0.7, max_keep=300)


@pytest.mark.parametrize("C,H,W,d,s", [(1024, 38, 67, 8, 1), (512, 75, 134, 8, 2), (128, 36, 63, 16, 1)])
def test_correlation_backward_in_two_phases_on_two_streams(dev, C, H, W, d, s):
    """dtt_correlation_backward_nhwc_phase (round 6): the band words laid out on a SECOND stream (phase 1, reads the rows' gradient
    only), the gradients from that workspace on the main stream behind an event (phase 2) -- bit-identical to the one-call op
    (phase 3), planes and rows layouts, radius 8 and 16; this is how dtt.heads.TrackingRowsFn.backward overlaps the bands of conv4 /
    conv5 with conv3's gradient op."""
    from dtt.ops import correlation_backward_nhwc, correlation_output_shape
    rng = np.random.RandomState(C + d)
    B = 2
    x1 = torch.from_numpy(rng.normal(size=(B, C, H, W)).astype(np.float32)).to(dev).contiguous(memory_format=torch.channels_last)
    x2 = torch.from_numpy(rng.normal(size=(B, C, H, W)).astype(np.float32)).to(dev).contiguous(memory_format=torch.channels_last)
    oc, oh, ow = correlation_output_shape(C, H, W, d, 1, d, s, s)
    ld, col = oc + 40, 16
    rows = torch.from_numpy(rng.normal(size=(B * oh * ow, ld)).astype(np.float32)).to(dev)
    a1, a2 = torch.full_like(x1, float("nan")), torch.full_like(x2, float("nan"))
    correlation_backward_nhwc(None, x1, x2, a1, a2, d, 1, d, s, s, rows=rows, col=col)
    b1, b2 = torch.full_like(x1, float("nan")), torch.full_like(x2, float("nan"))
    main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)
    ready = torch.cuda.Event(); ready.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ready)
        ws = correlation_backward_nhwc(None, x1, x2, None, None, d, 1, d, s, s, rows=rows, col=col, phase=1)
        laid = torch.cuda.Event(); laid.record(side)
    main.wait_event(laid)
    assert correlation_backward_nhwc(None, x1, x2, b1, b2, d, 1, d, s, s, rows=rows, col=col, phase=2, workspace=ws) is None
    torch.cuda.synchronize(dev)
    assert torch.equal(a1, b1) and torch.equal(a2, b2) and not bool(torch.isnan(a1).any())
    with pytest.raises(ValueError):
        correlation_backward_nhwc(None, x1, x2, b1, b2, d, 1, d, s, s, rows=rows, col=col, phase=2)      # no workspace


def test_nms_empty_and_max_keep(dev):
    from dtt.ops import nms
    assert nms(torch.zeros((0, 5), device=dev), 0.7) == []
    rng = np.random.RandomState(5)
    dets = clustered_dets(rng, 3000)
    ref = O.nms(dets, 0.7)
    for mk in (1, 17, 300, len(ref), len(ref) + 5):
        got = nms(cu(dets, dev), 0.7, max_keep=mk).cpu().numpy().ravel()
        np.testing.assert_array_equal(got, ref[:mk])


@pytest.mark.parametrize("n,clusters,mk", [(6000, 200, 300), (6000, 450, 300), (6000, 3000, 300), (12000, 700, 500),
                                          (2500, 90, 100), (6000, 200, 2000)])
def test_nms_two_phase_paths(dev, n, clusters, mk):
    """max_keep << n makes dtt_nms compute / sweep the first 1024-box super-chunks first and the rest only if the keep
    list is still short (nms.hip: two phases).  Heavily duplicated boxes force the second phase (`clusters` < mk: it never
    reaches max_keep; a few more clusters than mk: it gets there in phase 2), spread-out boxes finish in the first; one
    configuration is above the split rule and stays single-phase.  Same keep list as the oracle in every case."""
    from dtt.ops import nms
    rng = np.random.RandomState(n + clusters)
    ctr = rng.uniform(100, 3000, size=(clusters, 2))
    wh = rng.uniform(40, 120, size=(clusters, 2))
    which = rng.randint(0, clusters, size=n)
    which[:clusters // 3] = np.arange(clusters // 3)                      # some clusters show up early, most late
    jitter = rng.normal(0, 1.5, size=(n, 4))
    boxes = np.concatenate([ctr[which] - wh[which] / 2, ctr[which] + wh[which] / 2], 1) + jitter
    dets = np.concatenate([boxes, np.sort(rng.uniform(0, 1, n))[::-1][:, None]], 1).astype(np.float32)
    ref = O.nms(dets, 0.7)
    got = nms(cu(dets, dev), 0.7, max_keep=mk).cpu().numpy().ravel()
    np.testing.assert_array_equal(got, ref[:mk])
    assert len(ref) >= 50


def test_nms_threshold_boundary(dev):
    """IoU within an ulp of the threshold: strict '>' and the exact fp32 op order decide."""
    from dtt.ops import nms
    rng = np.random.RandomState(9)
    n = 512
    base = np.array([10, 10, 109, 109], dtype=np.float32)
    dets = np.zeros((n, 5), dtype=np.float32)
    for i in range(n):
        # boxes overlapping `base` with IoU spread tightly around 0.7
        shift = rng.uniform(16.5, 18.5)
        dets[i, :4] = base + np.float32([shift * (i % 2), shift * ((i + 1) % 2), shift * (i % 2), shift * ((i + 1) % 2)])
    dets[0, :4] = base
    dets[:, 4] = np.linspace(1, 0, n)
    np.testing.assert_array_equal(nms(cu(dets, dev), 0.7).cpu().numpy().ravel(), O.nms(dets, 0.7))


@pytest.mark.parametrize("case", ["exact_ratio", "one_ulp_below", "degenerate"])
def test_nms_division_shortcut_edges(dev, case):
    """nms_mask_kernel decides IoU > thresh from inter against thresh * union and only divides when a lane of the wave is within
    1e-6 of the threshold (nms.hip: iou_over).  Pairs whose IoU is EXACTLY the threshold (integer areas 100 / 200 -> 0.5, 150 / 200
    -> 0.75, ...: strict '>' keeps both), the same pairs with the threshold one ulp lower (now suppressed), and boxes the shortcut
    must not touch -- negative and zero areas, areas that overflow to inf -- all inside ordinary clusters so that a wave holds
    decided and undecided lanes at once.  Keep list identical to the oracle's (nms_cuda_kernel.cu:31-39, 101)."""
    from dtt.ops import nms
    rng = np.random.RandomState(31)
    dets = clustered_dets(rng, 700)
    if case == "degenerate":
        dets[5, :4] = [50, 50, 20, 90]            # x2 < x1 - 1: negative width, negative area
        dets[17, :4] = [60, 60, 59, 200]          # x2 == x1 - 1: zero area
        dets[33, :4] = [-3e19, -3e19, 3e19, 3e19] # area overflows to inf
        dets[34, :4] = [-2e19, -2e19, 3e19, 3e19]
        dets[70, :4] = [100, 100, 100, 100]       # a single pixel
        dets[71, :4] = [100, 100, 100, 100]
        threshes = [0.7, 0.3]
    else:
        # nested integer boxes: IoU = area(inner) / area(outer) exactly
        pairs = [((0, 0, 9, 9), (0, 0, 9, 19)), ((300, 300, 314, 309), (300, 300, 319, 309)), ((40, 500, 46, 506), (40, 500, 53, 513))]
        for k, (a, b) in enumerate(pairs):
            dets[100 + 64 * k, :4] = np.float32(a) + 1000 * (k + 1)      # far from the clusters, in different 64-box tiles
            dets[131 + 64 * k, :4] = np.float32(b) + 1000 * (k + 1)
        threshes = [np.float32(0.5), np.float32(0.75), np.float32(0.25)]
        if case == "one_ulp_below":
            threshes = [np.nextafter(t, np.float32(0)) for t in threshes]
    for t in threshes:
        ref = O.nms(dets, float(t))
        got = nms(cu(dets, dev), float(t)).cpu().numpy().ravel()
        np.testing.assert_array_equal(got, ref)
    if case != "degenerate":
        kept = set(O.nms(dets, float(threshes[0])).tolist())
        assert (131 in kept) == (case == "exact_ratio")                  # IoU == 0.5: kept under '>', dropped one ulp below


# --------------------------------------------------------------------------------------------- PSRoI
@pytest.mark.parametrize("B,od,H,W,R", [(2, 4, 38, 67, 300), (1, 31, 19, 32, 64), (3, 5, 24, 40, 37)])
def test_psroi_forward_bit_exact_and_backward(dev, B, od, H, W, R):
    from dtt.ops import _PSRoIPooling
    rng = np.random.RandomState(od * 100 + R)
    G7 = 7
    feat = rng.normal(size=(B, od * G7 * G7, H, W)).astype(np.float32)
    rois = random_rois(rng, R, B, H * 16, W * 16)
    ref, ref_map = O.psroi_pool_forward(feat, rois, G7, G7, 1.0 / 16, G7, od)
    m = _PSRoIPooling(G7, G7, 1.0 / 16.0, G7, od)
    ft = cu(feat, dev).requires_grad_(True)
    out = m(ft, cu(rois, dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)  # same summation order -> identical bits
    gout = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(cu(gout, dev))
    gref = O.psroi_pool_backward(gout, rois, feat.shape, G7, G7, 1.0 / 16, G7, od, ref_map)
    np.testing.assert_allclose(ft.grad.cpu().numpy(), gref, rtol=1e-5, atol=1e-5)


def test_psroi_vote_and_empty(dev):
    from dtt.ops import _PSRoIPooling, psroi_pool_vote
    rng = np.random.RandomState(1)
    feat = rng.normal(size=(2, 4 * 49, 20, 30)).astype(np.float32)
    rois = random_rois(rng, 50, 2, 320, 480)
    ref, _ = O.psroi_pool_forward(feat, rois, 7, 7, 1 / 16.0, 7, 4)
    pooled, vote = psroi_pool_vote(cu(feat, dev), cu(rois, dev), 7, 7, 1 / 16.0, 7, 4)
    np.testing.assert_array_equal(pooled.cpu().numpy(), ref)
    np.testing.assert_allclose(vote.cpu().numpy(), ref.reshape(50, 4, 49).mean(2), rtol=1e-5, atol=1e-6)
    from dtt.ops import psroi_vote
    assert torch.equal(psroi_vote(cu(feat, dev), cu(rois, dev), 7, 7, 1 / 16.0, 7, 4), vote)   # channel-major scratch path
    big = rng.normal(size=(4, 31 * 49, 38, 67)).astype(np.float32)
    brois = random_rois(rng, 1200, 4, 600, 1067)
    v1 = psroi_pool_vote(cu(big, dev), cu(brois, dev), 7, 7, 1 / 16.0, 7, 31)[1]
    assert torch.equal(psroi_vote(cu(big, dev), cu(brois, dev), 7, 7, 1 / 16.0, 7, 31), v1)
    out = _PSRoIPooling(7, 7, 1 / 16.0, 7, 4)(cu(feat, dev), torch.zeros((0, 5), device=dev))
    assert tuple(out.shape) == (0, 4, 7, 7)
    with pytest.raises(ValueError):
        _PSRoIPooling(7, 7, 1 / 16.0, 7, 4)(cu(feat, dev), torch.zeros((3, 4), device=dev))


# --------------------------------------------------------------------------------------- correlation
CORR_CASES = [
    # B, C, H, W, pad, k, d, s1, s2
    (2, 64, 20, 27, 8, 1, 8, 1, 1),     # conv4/conv5 geometry (R = 8)
    (1, 48, 37, 45, 8, 1, 8, 2, 2),     # conv3 geometry (stride 2, R = 4)
    (1, 20, 18, 22, 16, 1, 16, 1, 1),   # config-5 geometry (R = 16)
    (2, 19, 9, 11, 4, 1, 4, 1, 1),      # C not a multiple of the channel chunk, tiny map
    (1, 33, 21, 17, 3, 1, 3, 1, 1),     # R = 3 (padded up to the R <= 4 instantiation)
    (1, 16, 16, 16, 6, 1, 4, 1, 1),     # pad > displacement
    (1, 8, 12, 13, 4, 3, 4, 1, 2),      # kernel_size 3, stride2 != stride1 -> generic path
    (1, 16, 13, 19, 4, 1, 4, 1, 1),     # LDS-DMA kernel, R = 4
    (1, 8, 9, 12, 16, 1, 16, 1, 1),     # LDS-DMA kernel, R = 16, one chunk
    (2, 16, 16, 16, 8, 1, 4, 1, 1),     # LDS-DMA kernel, pad > displacement (output pixels inside the padding)
    (1, 8, 5, 4, 8, 1, 8, 1, 1),        # LDS-DMA kernel, map narrower than one piece row
    (1, 24, 38, 67, 8, 1, 8, 1, 1),     # LDS-DMA kernel, the 600 px map, odd number of chunks
    (3, 8, 8, 8, 8, 1, 8, 1, 1),        # LDS-DMA kernel, single chunk, batch 3
    (2, 16, 20, 30, 12, 1, 12, 1, 1),   # R = 12: four R = 8 sub-windows of the LDS-DMA kernel
    (1, 24, 37, 50, 16, 1, 16, 1, 1),   # R = 16 the same way, three chunks
]


@pytest.mark.parametrize("case", CORR_CASES)
def test_correlation_forward(dev, case):
    from dtt.ops import Correlation
    B, C, H, W, pad, k, d, s1, s2 = case
    rng = np.random.RandomState(sum(case))
    x1 = np.maximum(rng.normal(size=(B, C, H, W)), 0).astype(np.float32)
    x2 = np.maximum(np.roll(x1, (1, -2), (2, 3)) + 0.3 * rng.normal(size=x1.shape), 0).astype(np.float32)
    ref = O.correlation_forward(x1, x2, pad, k, d, s1, s2)
    out = Correlation(pad, k, d, s1, s2)(cu(x1, dev), cu(x2, dev)).cpu().numpy()
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-4)  # north-star tolerance: 1e-4 fp32
    assert np.abs(out - ref).max() < 2e-6 * max(1.0, np.abs(ref).max() * C ** 0.5)


@pytest.mark.parametrize("case", CORR_CASES[:6])
def test_correlation_backward(dev, case):
    from dtt.ops import Correlation
    B, C, H, W, pad, k, d, s1, s2 = case
    rng = np.random.RandomState(sum(case) + 1)
    x1 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    x2 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    t1, t2 = cu(x1, dev).requires_grad_(True), cu(x2, dev).requires_grad_(True)
    out = Correlation(pad, k, d, s1, s2)(t1, t2)
    gout = rng.normal(size=tuple(out.shape)).astype(np.float32)
    out.backward(cu(gout, dev))
    g1, g2 = O.correlation_backward(gout, x1, x2, pad, k, d, s1, s2)
    np.testing.assert_allclose(t1.grad.cpu().numpy(), g1, rtol=0, atol=1e-4)
    np.testing.assert_allclose(t2.grad.cpu().numpy(), g2, rtol=0, atol=1e-4)


CORR_CL_CASES = [
    # B, C, H, W, pad, k, d, s1, s2 -- geometries of the channels-last training path (kernel 1, stride1 == stride2, R <= 8, C % 16 == 0)
    (2, 64, 20, 27, 8, 1, 8, 1, 1),     # conv4 / conv5 geometry
    (1, 48, 37, 45, 8, 1, 8, 2, 2),     # conv3 geometry: lattice stride 2, R = 4
    (1, 16, 16, 16, 6, 1, 4, 1, 1),     # pad > displacement
    (1, 32, 21, 17, 3, 1, 3, 1, 1),     # R = 3
    (2, 16, 9, 11, 4, 1, 4, 1, 1),      # tiny map, one chunk
    (1, 80, 38, 67, 8, 1, 8, 1, 1),     # the 600 px map, five chunks
    (3, 16, 5, 4, 8, 1, 8, 1, 1),       # map narrower than the window
    # channels % 64 == 0: the band-stationary streamed gradient kernels (csrc/correlation_bwd.hip)
    (1, 128, 37, 45, 8, 1, 8, 2, 2),    # conv3 geometry: lattice stride 2, R = 4, two channel groups
    (1, 64, 16, 16, 6, 1, 4, 1, 1),     # pad > displacement: outputs beyond the map
    (1, 64, 30, 30, 2, 1, 4, 1, 1),     # pad < displacement: the border pixels are no targets of gradInput1
    (1, 64, 21, 17, 3, 1, 3, 1, 1),     # R = 3, odd block row, one odd block column (4 x 1 / 1 x 1 tiles)
    (2, 64, 9, 11, 4, 1, 4, 1, 1),      # tiny map: 2 x 3 and 1 x 3 tiles only
    (1, 128, 38, 67, 8, 1, 8, 1, 1),    # the 600 px map: 2 x 4 tiles + the 17th block column as 4 x 1 / 2 x 1
    (1, 192, 24, 40, 8, 1, 8, 1, 1),    # two odd block columns (4 x 2 / 2 x 2 tiles), three channel groups
    (3, 64, 5, 4, 8, 1, 8, 1, 1),       # map narrower than the window
    (2, 320, 17, 29, 8, 1, 8, 1, 1),    # five groups, odd rows and three odd columns (2 x 3 tiles)
]


@pytest.mark.parametrize("case", CORR_CL_CASES)
def test_correlation_channels_last_autograd(dev, case):
    """dtt.ops.Correlation on channels-last maps (CorrelationNHWCFunction: window-split forward, channels-last matrix-core
    backward): output and both gradients against the oracle at the north-star tolerance, gradients channels-last, and
    bit-identical to the NCHW functions' (same MFMA sequence)."""
    from dtt.ops import Correlation, CorrelationNHWCFunction
    B, C, H, W, pad, k, d, s1, s2 = case
    rng = np.random.RandomState(sum(case) + 7)
    x1 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    x2 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    layer = Correlation(pad, k, d, s1, s2)
    t1 = cu(x1, dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    t2 = cu(x2, dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert CorrelationNHWCFunction.supports(t1, t2, k, d, s1, s2)
    out = layer(t1, t2)
    assert out.is_contiguous()
    ref = O.correlation_forward(x1, x2, pad, k, d, s1, s2)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=0, atol=1e-4)
    gout = rng.normal(size=tuple(out.shape)).astype(np.float32)
    out.backward(cu(gout, dev))
    for t in (t1, t2):
        assert t.grad.is_contiguous(memory_format=torch.channels_last) and t.grad.shape == t.shape
    g1, g2 = O.correlation_backward(gout, x1, x2, pad, k, d, s1, s2)
    np.testing.assert_allclose(t1.grad.cpu().numpy(), g1, rtol=0, atol=1e-4)
    np.testing.assert_allclose(t2.grad.cpu().numpy(), g2, rtol=0, atol=1e-4)
    n1, n2 = cu(x1, dev).requires_grad_(True), cu(x2, dev).requires_grad_(True)
    layer(n1, n2).backward(cu(gout, dev))
    if C % 64:
        # round 1's kernel in its channels-last instantiation: the NCHW instantiation's MFMA sequence
        assert torch.equal(n1.grad, t1.grad) and torch.equal(n2.grad, t2.grad)
    else:
        # the streamed kernels sum the window in another (fixed) order: same values to rounding, identical from run to run
        assert float((n1.grad - t1.grad).abs().max()) < 1e-5 and float((n2.grad - t2.grad).abs().max()) < 1e-5
        r1 = t1.detach().clone().requires_grad_(True)
        r2 = t2.detach().clone().requires_grad_(True)
        layer(r1, r2).backward(cu(gout, dev))
        assert torch.equal(r1.grad, t1.grad) and torch.equal(r2.grad, t2.grad)


@pytest.mark.parametrize("chunk", [99, 2])
@pytest.mark.parametrize("case", [(1, 256, 38, 67, 8, 1, 8, 1, 1), (2, 320, 17, 29, 8, 1, 8, 1, 1), (1, 192, 24, 40, 8, 1, 8, 1, 1),
                                  (1, 192, 37, 45, 8, 1, 8, 2, 2), (1, 256, 21, 17, 3, 1, 3, 1, 1)])
def test_correlation_streamed_backward_many_groups_per_work_item(dev, monkeypatch, case, chunk):
    """The streamed gradient kernels with SEVERAL channel groups per work item (the plan of the full-size training step: 6 of
    conv5's 32 groups per workgroup; small maps plan one group per item, so the switch DTT_CORR_BWD_CHUNK forces it): the ring
    runs across group boundaries -- where the rows of a tall tile (the 4 x 1 tiles of an odd block column) all step into fresh
    positions at once -- accumulators are stored and reset per group.  Both gradients against the oracle."""
    from dtt.ops import correlation_backward_nhwc, correlation_output_shape
    monkeypatch.setenv("DTT_CORR_BWD_CHUNK", str(chunk))
    B, C, H, W, pad, k, d, s1, s2 = case
    rng = np.random.RandomState(sum(case) + chunk)
    x1 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    x2 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    cl = lambda a: cu(a, dev).contiguous(memory_format=torch.channels_last)
    t1, t2 = cl(x1), cl(x2)
    oc, oh, ow = correlation_output_shape(C, H, W, pad, k, d, s1, s2)
    gout = rng.normal(size=(B, oc, oh, ow)).astype(np.float32)
    ga, gb = torch.full_like(t1, float("nan")), torch.full_like(t2, float("nan"))
    correlation_backward_nhwc(cu(gout, dev), t1, t2, ga, gb, pad, k, d, s1, s2)
    g1, g2 = O.correlation_backward(gout, x1, x2, pad, k, d, s1, s2)
    np.testing.assert_allclose(ga.cpu().numpy(), g1, rtol=0, atol=1e-4)
    np.testing.assert_allclose(gb.cpu().numpy(), g2, rtol=0, atol=1e-4)
    ga2, gb2 = torch.empty_like(t1), torch.empty_like(t2)
    correlation_backward_nhwc(cu(gout, dev), t1, t2, ga2, gb2, pad, k, d, s1, s2)
    assert torch.equal(ga, ga2) and torch.equal(gb, gb2)     # run-to-run identical


@pytest.mark.parametrize("case", [(1, 64, 20, 27, 16, 1, 16, 1, 1),     # BASELINE configs[4]'s d = 16: 33 x 33 window, four quarters
                                  (2, 128, 13, 30, 12, 1, 12, 1, 1),    # R = 12: the last quarter holds two of its five block rows
                                  (1, 64, 36, 63, 16, 1, 16, 1, 1),     # the configs[4] map size
                                  (1, 64, 41, 37, 20, 1, 20, 2, 2),     # stride-2 lattice, R = 10
                                  (1, 64, 17, 19, 9, 1, 9, 1, 1),       # R = 9: the second quarter is almost empty
                                  (1, 64, 24, 24, 10, 1, 16, 1, 1)])    # pad < displacement at R = 16
def test_correlation_streamed_backward_window_radius_above_8(dev, case):
    """Window radius 9 .. 16 on channels-last maps (the training step of BASELINE configs[4] stays channels-last): the
    window is covered in four quarters of 5 x 5 blocks, the quarters after the first add to the gradient in a fixed order.
    dtt.ops.Correlation under autograd: forward and both gradients against the oracle, gradients channels-last, run-to-run
    identical."""
    from dtt.ops import Correlation, CorrelationNHWCFunction
    B, C, H, W, pad, k, d, s1, s2 = case
    rng = np.random.RandomState(sum(case) + 3)
    x1 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    x2 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    layer = Correlation(pad, k, d, s1, s2)
    t1 = cu(x1, dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    t2 = cu(x2, dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert CorrelationNHWCFunction.supports(t1, t2, k, d, s1, s2, pad)
    out = layer(t1, t2)
    np.testing.assert_allclose(out.detach().cpu().numpy(), O.correlation_forward(x1, x2, pad, k, d, s1, s2), rtol=0, atol=1e-4)
    gout = rng.normal(size=tuple(out.shape)).astype(np.float32)
    out.backward(cu(gout, dev))
    for t in (t1, t2):
        assert t.grad.is_contiguous(memory_format=torch.channels_last) and t.grad.shape == t.shape
    g1, g2 = O.correlation_backward(gout, x1, x2, pad, k, d, s1, s2)
    np.testing.assert_allclose(t1.grad.cpu().numpy(), g1, rtol=0, atol=1e-4)
    np.testing.assert_allclose(t2.grad.cpu().numpy(), g2, rtol=0, atol=1e-4)
    r1, r2 = t1.detach().clone().requires_grad_(True), t2.detach().clone().requires_grad_(True)
    layer(r1, r2).backward(cu(gout, dev))
    assert torch.equal(r1.grad, t1.grad) and torch.equal(r2.grad, t2.grad)


@pytest.mark.parametrize("case", [c for c in CORR_CL_CASES if c[1] % 64 == 0])
def test_correlation_streamed_backward_rows_layout_and_single_gradients(dev, case):
    """dtt_correlation_backward_nhwc_strided reading gradOut as columns of position-major rows (the gradient of the tracking
    head's input rows, read where it lies) and computing one gradient at a time: bit-identical to the planes layout / both."""
    from dtt.ops import correlation_backward_nhwc, correlation_output_shape
    B, C, H, W, pad, k, d, s1, s2 = case
    rng = np.random.RandomState(sum(case) + 11)
    cl = lambda a: cu(a, dev).contiguous(memory_format=torch.channels_last)
    t1, t2 = cl(rng.normal(size=(B, C, H, W)).astype(np.float32)), cl(rng.normal(size=(B, C, H, W)).astype(np.float32))
    oc, oh, ow = correlation_output_shape(C, H, W, pad, k, d, s1, s2)
    gout = cu(rng.normal(size=(B, oc, oh, ow)).astype(np.float32), dev)
    ga, gb = torch.empty_like(t1), torch.empty_like(t2)
    correlation_backward_nhwc(gout, t1, t2, ga, gb, pad, k, d, s1, s2)
    ld, col = oc + 37, 13
    rows = torch.full((B * oh * ow, ld), float("nan"), device=dev)
    rows[:, col:col + oc] = gout.permute(0, 2, 3, 1).reshape(-1, oc)
    ra, rb = torch.full_like(t1, float("nan")), torch.full_like(t2, float("nan"))
    correlation_backward_nhwc(None, t1, t2, ra, None, pad, k, d, s1, s2, rows=rows, col=col)
    correlation_backward_nhwc(None, t1, t2, None, rb, pad, k, d, s1, s2, rows=rows, col=col)
    assert torch.equal(ra, ga) and torch.equal(rb, gb)


def test_correlation_pair_keeps_a_channels_last_batch_whole(dev):
    """Correlation.pair(maps, B, 0, 1) on the (2B, C, H, W) channels-last batch of both legs: same output as slicing, and the
    gradient of `maps` comes back channels-last in one piece (legs written in place), equal to the sliced path's."""
    from dtt.ops import Correlation
    rng = np.random.RandomState(21)
    B, C, H, W = 2, 32, 19, 23
    x = rng.normal(size=(2 * B, C, H, W)).astype(np.float32)
    for (pad, k, d, s1, s2) in [(8, 1, 8, 1, 1), (8, 1, 8, 2, 2)]:
        layer = Correlation(pad, k, d, s1, s2)
        m = cu(x, dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        out = layer.pair(m, B, 0, 1)
        gout = torch.from_numpy(rng.normal(size=tuple(out.shape)).astype(np.float32)).to(dev)
        out.backward(gout)
        assert m.grad.is_contiguous(memory_format=torch.channels_last)
        a = cu(x[:B], dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        b = cu(x[B:], dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ref = layer(a, b)
        ref.backward(gout)
        assert torch.equal(out, ref)
        assert torch.equal(m.grad[:B], a.grad) and torch.equal(m.grad[B:], b.grad)


def test_correlation_channels_last_falls_back_outside_its_geometry(dev):
    """R = 16 and C % 16 != 0 on channels-last maps: the module converts to NCHW and takes the reference-layout functions."""
    from dtt.ops import Correlation, CorrelationNHWCFunction
    rng = np.random.RandomState(5)
    for (B, C, H, W, pad, k, d, s1, s2) in [(1, 16, 18, 22, 16, 1, 16, 1, 1), (1, 20, 12, 13, 4, 1, 4, 1, 1)]:
        x1 = rng.normal(size=(B, C, H, W)).astype(np.float32)
        x2 = rng.normal(size=(B, C, H, W)).astype(np.float32)
        t1 = cu(x1, dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        t2 = cu(x2, dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        assert not CorrelationNHWCFunction.supports(t1, t2, k, d, s1, s2)
        out = Correlation(pad, k, d, s1, s2)(t1, t2)
        gout = rng.normal(size=tuple(out.shape)).astype(np.float32)
        out.backward(cu(gout, dev))
        g1, g2 = O.correlation_backward(gout, x1, x2, pad, k, d, s1, s2)
        np.testing.assert_allclose(t1.grad.cpu().numpy(), g1, rtol=0, atol=1e-4)
        np.testing.assert_allclose(t2.grad.cpu().numpy(), g2, rtol=0, atol=1e-4)


def _corr_autograd_f64(x1, x2, pad, k, d, s1, s2):
    """Independent formulation of the forward (correlation_cuda_kernel.cu:34-106: k x k patches anchored at their top-left
    corner (oy*s1 + d, ox*s1 + d) in padded coordinates, displaced by multiples of s2, mean over k*k*C) in float64 torch
    ops, so that autograd supplies both gradients."""
    import torch.nn.functional as F
    B, C, H, W = x1.shape
    r = d // s2
    p1, p2 = F.pad(x1, (pad,) * 4), F.pad(F.pad(x2, (pad,) * 4), (d,) * 4)   # second pad: every displaced window exists
    pH, pW = H + 2 * pad, W + 2 * pad
    krad = (k - 1) // 2
    oh = int(np.ceil((pH - 2 * (krad + d)) / s1)); ow = int(np.ceil((pW - 2 * (krad + d)) / s1))
    ys = torch.arange(oh) * s1 + d; xs = torch.arange(ow) * s1 + d
    out = []
    for tj in range(-r, r + 1):
        for ti in range(-r, r + 1):
            acc = 0
            for j in range(k):
                for i in range(k):
                    a = p1[:, :, (ys + j)[:, None], (xs + i)[None, :]]
                    b = p2[:, :, (ys + j + tj * s2 + d)[:, None], (xs + i + ti * s2 + d)[None, :]]
                    acc = acc + (a * b).sum(1)
            out.append(acc / (k * k * C))
    return torch.stack(out, 1)


@pytest.mark.parametrize("case", [(1, 6, 12, 13, 4, 3, 4, 1, 2), (2, 5, 11, 9, 3, 3, 2, 1, 1), (1, 4, 14, 15, 6, 5, 4, 2, 2),
                                  (1, 7, 10, 12, 2, 1, 2, 2, 1), (1, 3, 9, 9, 4, 3, 4, 2, 1)])
def test_correlation_backward_any_kernel_size_is_the_exact_adjoint(dev, case):
    """kernel_size > 1 and stride1 > 1 (where the reference's own backward departs from its forward, see include/dtt_hip.h):
    the HIP forward equals the independent float64 formulation and both HIP gradients equal its autograd gradients."""
    from dtt.ops import Correlation
    B, C, H, W, pad, k, d, s1, s2 = case
    rng = np.random.RandomState(sum(case) + 7)
    x1 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    x2 = rng.normal(size=(B, C, H, W)).astype(np.float32)
    a1, a2 = torch.from_numpy(x1).double().requires_grad_(True), torch.from_numpy(x2).double().requires_grad_(True)
    ref = _corr_autograd_f64(a1, a2, pad, k, d, s1, s2)
    t1, t2 = cu(x1, dev).requires_grad_(True), cu(x2, dev).requires_grad_(True)
    out = Correlation(pad, k, d, s1, s2)(t1, t2)
    assert tuple(out.shape) == tuple(ref.shape)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=1e-5)
    gout = rng.normal(size=tuple(out.shape)).astype(np.float32)
    out.backward(cu(gout, dev))
    ref.backward(torch.from_numpy(gout).double())
    np.testing.assert_allclose(t1.grad.cpu().numpy(), a1.grad.numpy(), atol=1e-5)
    np.testing.assert_allclose(t2.grad.cpu().numpy(), a2.grad.numpy(), atol=1e-5)


NHWC_CASES = [
    (2, 64, 38, 67, 8, 1, 8, 1, 1),     # conv4 / conv5 geometry at the 600 px map size (fewer channels): 10 x 17 pixel blocks
    (2, 2048, 38, 67, 8, 1, 8, 1, 1),   # conv5 at full size: 3 window parts x (42 full tiles + one 2 x 1) x 2 images = 256 workgroups
    (2, 512, 75, 134, 8, 1, 8, 2, 2),   # conv3 at full size: stride 2 = the stride-1 problem on the even lattice
    (1, 48, 37, 45, 8, 1, 8, 2, 2),     # odd map sizes under stride 2
    (1, 32, 36, 63, 16, 1, 16, 1, 1),   # config 5: R = 16 natively (9 x 9 window blocks)
    (1, 32, 18, 22, 12, 1, 12, 1, 1),   # R = 12 (7 x 7 window blocks)
    (3, 16, 9, 11, 4, 1, 4, 1, 1),      # R = 4 (3 x 3 window blocks), tiny map, one chunk
    (1, 48, 21, 17, 3, 1, 3, 1, 1),     # odd radii: R = 3 ...
    (1, 32, 20, 27, 5, 1, 5, 1, 1),     # ... 5 ...
    (1, 16, 19, 23, 7, 1, 7, 1, 1),     # ... 7 ...
    (2, 16, 14, 30, 13, 1, 13, 1, 1),   # ... 13
    (1, 32, 20, 27, 6, 1, 6, 1, 1),     # R = 6
    (1, 16, 12, 9, 1, 1, 1, 1, 1),      # R = 1 and R = 2: two window blocks per axis
    (2, 16, 10, 13, 2, 1, 2, 1, 1),
    (1, 16, 16, 16, 6, 1, 4, 1, 1),     # pad > displacement: output pixels inside the padding
    (2, 32, 13, 19, 2, 1, 4, 1, 1),     # pad < displacement: the output is smaller than the map
    (1, 16, 5, 4, 8, 1, 8, 1, 1),       # map smaller than one tile (2 x 1 pixel blocks)
    (1, 16, 3, 3, 8, 1, 8, 1, 1),       # a single pixel block
    (1, 16, 20, 4, 8, 1, 8, 1, 1),      # one block column: 4 x 1 tiles + remainder
    (1, 16, 4, 33, 8, 1, 8, 1, 1),      # one block row, odd count: 1 x 4 tiles + a 1 x 1
    (1, 16, 14, 34, 16, 1, 16, 1, 1),   # R = 16 with a 4 x 1 edge tile: 40 KB ring slots, only three fit (found by tests/fuzz_ops.py)
    (2, 16, 22, 26, 8, 1, 8, 1, 1),     # odd x odd block grid (6 x 7 ... 5.5 -> 6 rows, 6.5 -> 7 cols): every segment kind
]


@pytest.mark.parametrize("case", NHWC_CASES)
def test_correlation_nhwc_forward(dev, case):
    """The channels-last kernel (dtt_correlation_forward_nhwc: window-split, csrc/correlation_wsplit.hip) against the oracle on
    the same values, in both output layouts (NCHW tensor; columns of a position-major matrix), run-to-run identical, and
    bit-identical under every plan (all CUs / a CU budget that forces more, shorter workgroups / one CU): each output is one
    wave's fma chain over the channels in order, whatever the partition."""
    from dtt.ops import correlation_forward_nhwc
    B, C, H, W, pad, k, d, s1, s2 = case
    rng = np.random.RandomState(sum(case))
    x1 = np.maximum(rng.normal(size=(B, C, H, W)), 0).astype(np.float32)
    x2 = np.maximum(np.roll(x1, (1, -2), (2, 3)) + 0.3 * rng.normal(size=x1.shape), 0).astype(np.float32)
    ref = O.correlation_forward(x1, x2, pad, k, d, s1, s2)
    t1 = cu(x1, dev).contiguous(memory_format=torch.channels_last)
    t2 = cu(x2, dev).contiguous(memory_format=torch.channels_last)
    out = correlation_forward_nhwc(t1, t2, pad, k, d, s1, s2)
    assert tuple(out.shape) == ref.shape
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=1e-4)  # north-star tolerance: 1e-4 fp32
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-6 * max(1.0, np.abs(ref).max() * C ** 0.5)
    oc, oh, ow = ref.shape[1:]
    rows = torch.full((B * oh * ow, oc + 11), 7.0, device=dev)
    correlation_forward_nhwc(t1, t2, pad, k, d, s1, s2, rows=rows, col=5)
    assert torch.equal(rows[:, 5:5 + oc].reshape(B, oh, ow, oc).permute(0, 3, 1, 2), out)
    assert bool((rows[:, :5] == 7).all()) and bool((rows[:, 5 + oc:] == 7).all())
    for _ in range(3):
        assert torch.equal(correlation_forward_nhwc(t1, t2, pad, k, d, s1, s2), out)
    for budget in (240, 100, 17, 1):
        assert torch.equal(correlation_forward_nhwc(t1, t2, pad, k, d, s1, s2, max_workgroups=budget), out), budget


def test_correlation_nhwc_plans():
    """dtt_correlation_nhwc_plan at the shapes of the benchmark step (pure host code)."""
    import ctypes
    from dtt import _lib
    L = _lib.lib()

    def plan(*a):
        v = [ctypes.c_int() for _ in range(4)]
        assert L.dtt_correlation_nhwc_plan(*a, *[ctypes.byref(x) for x in v]) == 1, a
        return tuple(x.value for x in v)
    parts, nacc, wgs, slots = plan(2, 38, 67, 8, 256)
    assert (parts, nacc, wgs) == (3, 9, 256) and slots >= 4      # 2 x (40 + 2 full tiles x 3 + one 2 x 1 tile x 2)
    parts, nacc, wgs, slots = plan(2, 38, 67, 8, 240)            # CUs left free: more, shorter workgroups
    assert parts == 5 and nacc == 5 and wgs == 426
    assert plan(2, 38, 67, 4, 256)[:3] == (3, 3, 256)            # conv3
    parts, nacc, wgs, _ = plan(1, 36, 63, 16, 256)               # BASELINE configs[4]: 81 window blocks
    assert parts * nacc >= 81 and wgs >= 36
    assert L.dtt_correlation_nhwc_plan(1, 8, 8, 17, 0, None, None, None, None) == 0


def test_correlation_into_concat_slice(dev):
    from dtt.ops import correlation_forward_into
    rng = np.random.RandomState(4)
    x1 = rng.normal(size=(2, 32, 14, 18)).astype(np.float32)
    x2 = rng.normal(size=(2, 32, 14, 18)).astype(np.float32)
    ref = O.correlation_forward(x1, x2, 8, 1, 8, 1, 1)
    buf = torch.full((2, 10 + 289 + 7, 14, 18), 5.0, device=dev)
    correlation_forward_into(buf[:, 10:299], cu(x1, dev), cu(x2, dev), 8, 1, 8, 1, 1)
    np.testing.assert_allclose(buf[:, 10:299].cpu().numpy(), ref, atol=1e-4)
    assert (buf[:, :10] == 5).all() and (buf[:, 299:] == 5).all()


# ------------------------------------------------------------------------------ RoI align/pool/crop
def test_roi_align(dev):
    from dtt.ops import RoIAlign, RoIAlignAvg, RoIAlignMax
    rng = np.random.RandomState(2)
    B, C, H, W, R = 2, 16, 23, 31, 40
    feat = rng.normal(size=(B, C, H, W)).astype(np.float32)
    rois = random_rois(rng, R, B, H * 16, W * 16)
    ref8 = O.roi_align_forward(feat, rois, 8, 8, 1 / 16.0)
    ft = cu(feat, dev).requires_grad_(True)
    out = RoIAlign(8, 8, 1 / 16.0)(ft, cu(rois, dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref8)
    gout = rng.normal(size=ref8.shape).astype(np.float32)
    out.backward(cu(gout, dev))
    np.testing.assert_allclose(ft.grad.cpu().numpy(), O.roi_align_backward(gout, rois, feat.shape, 8, 8, 1 / 16.0),
                               rtol=1e-5, atol=1e-5)
    t8 = torch.from_numpy(ref8)
    with torch.no_grad():
        avg = RoIAlignAvg(7, 7, 1 / 16.0)(cu(feat, dev), cu(rois, dev)).cpu()
        mx = RoIAlignMax(7, 7, 1 / 16.0)(cu(feat, dev), cu(rois, dev)).cpu()
    np.testing.assert_allclose(avg.numpy(), torch.nn.functional.avg_pool2d(t8, 2, 1).numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(mx.numpy(), torch.nn.functional.max_pool2d(t8, 2, 1).numpy())
    # autograd path of RoIAlignAvg = unfused sampling + avg_pool2d
    ft2 = cu(feat, dev).requires_grad_(True)
    avg2 = RoIAlignAvg(7, 7, 1 / 16.0)(ft2, cu(rois, dev))
    np.testing.assert_allclose(avg2.detach().cpu().numpy(), avg.numpy(), rtol=1e-6, atol=1e-6)
    avg2.sum().backward()
    assert torch.isfinite(ft2.grad).all()
    # a map too large for the LDS-resident kernel takes the thread-per-output path: same results
    big = rng.normal(size=(1, 3, 120, 130)).astype(np.float32)
    brois = random_rois(rng, 30, 1, 120 * 16, 130 * 16)
    outb = RoIAlign(8, 8, 1 / 16.0)(cu(big, dev), cu(brois, dev))
    np.testing.assert_array_equal(outb.cpu().numpy(), O.roi_align_forward(big, brois, 8, 8, 1 / 16.0))
    avgb = RoIAlignAvg(7, 7, 1 / 16.0)(cu(big, dev), cu(brois, dev)).cpu()
    np.testing.assert_allclose(avgb.numpy(), torch.nn.functional.avg_pool2d(outb.cpu(), 2, 1).numpy(), rtol=1e-6, atol=1e-6)


def test_roi_pool(dev):
    from dtt.ops import RoIPoolFunction, _RoIPooling
    rng = np.random.RandomState(3)
    B, C, H, W, R = 2, 12, 20, 26, 48
    feat = rng.normal(size=(B, C, H, W)).astype(np.float32)
    rois = random_rois(rng, R, B, H * 16, W * 16)
    ref, ref_arg = O.roi_pool_forward(feat, rois, 7, 7, 1 / 16.0)
    ft = cu(feat, dev).requires_grad_(True)
    out, arg = RoIPoolFunction.apply(ft, cu(rois, dev), 7, 7, 1 / 16.0)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
    np.testing.assert_array_equal(arg.cpu().numpy(), ref_arg)
    gout = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(cu(gout, dev))
    gref = O.roi_pool_backward(gout, rois, ref_arg, feat.shape, 7, 7, 1 / 16.0)
    np.testing.assert_allclose(ft.grad.cpu().numpy(), gref, rtol=1e-5, atol=1e-5)
    assert tuple(_RoIPooling(7, 7, 1 / 16.0)(cu(feat, dev), cu(rois, dev)).shape) == (R, C, 7, 7)


def test_roi_crop(dev):
    from dtt.ops import _RoICrop
    rng = np.random.RandomState(6)
    B, C, H, W, RPI, Gs = 2, 8, 15, 19, 5, 7
    img = rng.normal(size=(B, C, H, W)).astype(np.float32)
    grid = rng.uniform(-1.3, 1.3, size=(B * RPI, Gs, Gs, 2)).astype(np.float32)
    ref = O.roi_crop_forward(img, grid)
    it = cu(img, dev).requires_grad_(True)
    out = _RoICrop()(it, cu(grid, dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
    gout = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(cu(gout, dev))
    np.testing.assert_allclose(it.grad.cpu().numpy(), O.roi_crop_backward(img, grid, gout), rtol=1e-5, atol=1e-5)


def test_roi_crop_pool_matches_grid_sample(dev):
    """faster_rcnn.py:73-80 ('crop' pooling mode): grids from RoIs -> bilinear crop -> 2x2 max pool."""
    import torch.nn.functional as F
    from dtt.ops import affine_grid_gen, roi_crop_pool
    rng = np.random.RandomState(16)
    B, C, H, W, RPI, P = 2, 12, 19, 31, 6, 7
    feat = rng.normal(size=(B, C, H, W)).astype(np.float32)
    x1 = rng.uniform(0, 300, B * RPI); y1 = rng.uniform(0, 200, B * RPI)
    rois = np.stack([np.repeat(np.arange(B), RPI), x1, y1, x1 + rng.uniform(8, 180, B * RPI),
                     y1 + rng.uniform(8, 90, B * RPI)], 1).astype(np.float32)
    for max_pool in (True, False):
        ft = cu(feat, dev).requires_grad_(True)
        out = roi_crop_pool(ft, cu(rois, dev).view(B, RPI, 5), P, max_pool)
        G = 2 * P if max_pool else P
        grid = affine_grid_gen(torch.from_numpy(rois), (H, W), G)
        x = torch.from_numpy(feat).requires_grad_(True)
        ref = F.grid_sample(x.repeat_interleave(RPI, 0), grid, mode="bilinear", padding_mode="zeros",
                            align_corners=True)
        if max_pool:
            ref = F.max_pool2d(ref, 2, 2)
        assert tuple(out.shape) == (B * RPI, C, P, P)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=1e-4)
        gout = rng.normal(size=tuple(out.shape)).astype(np.float32)
        out.backward(cu(gout, dev))
        ref.backward(torch.from_numpy(gout))
        np.testing.assert_allclose(ft.grad.cpu().numpy(), x.grad.numpy(), atol=1e-4)


def test_layout_transposes(dev):
    """dtt_transpose_batched behind nhwc_to_nchw / nchw_to_nhwc: exact copies, all four vector / scalar variants."""
    from dtt.fuse import _ToNCHWFn, nchw_to_nhwc, nhwc_to_nchw
    g = torch.Generator().manual_seed(5)
    for (n, c, h, w) in [(2, 64, 38, 67), (1, 130, 5, 7), (3, 4, 1, 9), (2, 7, 8, 8), (1, 3, 13, 11), (2, 256, 2, 2)]:
        x = torch.randn(n, c, h, w, generator=g).to(dev)
        xl = x.contiguous(memory_format=torch.channels_last)
        if c > 1 and h * w > 1:
            y = nhwc_to_nchw(xl)
            assert y.is_contiguous() and torch.equal(y, x)
        z = nchw_to_nhwc(x)
        assert z.is_contiguous(memory_format=torch.channels_last) and torch.equal(z, x)
    xl = torch.randn(2, 32, 9, 10, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = _ToNCHWFn.apply(xl)
    gy = torch.randn(2, 32, 9, 10, generator=g).to(dev)
    y.backward(gy)
    assert torch.equal(xl.grad, gy)


def test_scale_rows_batch_and_fold_scales(dev):
    """dtt_scale_rows_batch (one launch per 48 tensors) == a PyTorch multiply per tensor, bit for bit: NCHW and channels-last
    filters, a row length that is not a multiple of 4, more tensors than one launch holds; _FoldScalesFn's gradient."""
    from dtt.fuse import _FoldScalesFn, scale_rows_batch
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 64, 1, 1), (64, 64, 3, 3), (256, 64, 1, 1), (5, 3, 7, 7), (512, 128, 3, 3), (7, 1, 1, 1), (33, 5, 3, 3)]
    ws, ss = [], []
    for i in range(60):
        k, c, kh, kw = shapes[i % len(shapes)]
        w = torch.randn(k, c, kh, kw, generator=g).to(dev)
        if i % 2:
            w = w.contiguous(memory_format=torch.channels_last)
        ws.append(w)
        ss.append((torch.rand(k, generator=g) + 0.5).view(-1, 1, 1, 1).to(dev))
    outs = scale_rows_batch(ws, ss)
    for w, s_, o in zip(ws, ss, outs):
        assert o.stride() == w.stride()
        assert torch.equal(o, w * s_)
    params = [w.clone().requires_grad_(True) for w in ws[:9]]
    folded = _FoldScalesFn.apply(*params, *ss[:9])
    gs = [torch.randn(p.shape, generator=g).to(dev) for p in params]
    torch.autograd.backward(folded, gs)
    for p, s_, g_ in zip(params, ss, gs):
        assert torch.equal(p.grad, g_ * s_)


# ------------------------------------------------------------------------------------ proposal layer
def _proposal_inputs(rng, B, A, H, W):
    logits = rng.normal(0, 2, size=(B, 2, A * H, W)).astype(np.float32)
    prob = torch.softmax(torch.from_numpy(logits), 1).view(B, 2 * A, H, W).numpy()
    bbox = rng.normal(0, 0.4, size=(B, 4 * A, H, W)).astype(np.float32)
    info = np.tile(np.array([[H * 16.0, W * 16.0, 1.0]], dtype=np.float32), (B, 1))
    info[-1, :2] -= 7
    return prob, bbox, info


@pytest.mark.parametrize("B,H,W,pre,post", [(2, 19, 32, 6000, 300), (2, 38, 67, 6000, 300), (1, 38, 67, 12000, 2000),
                                           (2, 6, 8, 6000, 300), (3, 10, 12, 500, 50)])
def test_proposal_layer_vs_oracle(dev, B, H, W, pre, post):
    from dtt.rpn import generate_anchors, proposal_forward
    rng = np.random.RandomState(B * 1000 + H)
    base = generate_anchors(scales=(4, 8, 16, 32))
    prob, bbox, info = _proposal_inputs(rng, B, base.shape[0], H, W)
    prob[0, base.shape[0]:, 0, :5] = prob[0, base.shape[0], 0, 0]  # inject exact score ties
    ref, nref = ro.proposal_layer(prob, bbox, info, base, 16, pre, post, 0.7, O.nms)
    rois, num = proposal_forward(cu(prob, dev), cu(bbox, dev), cu(info, dev),
                                 torch.from_numpy(base).float(), 16, pre, post, 0.7)
    np.testing.assert_array_equal(num.cpu().numpy(), nref)
    np.testing.assert_array_equal(rois.cpu().numpy(), ref)  # bit-exact rows, order and zero padding


@pytest.mark.parametrize("H,W,pre,post,quant", [
    (16, 16, 6000, 300, 16),     # n = 3072 = 3 full runs of 1024, scores on a 1/16 grid: ties within and across runs
    (2, 43, 700, 100, 8),        # n = 1032: the second run holds 8 keys + padding
    (38, 67, 6000, 300, 64),     # the benchmark map, ~470 anchors per distinct score
    (38, 67, 12000, 2000, 0),    # quant 0: every score equal -- the order is the anchor index alone
    (5, 7, 6000, 300, 4),        # n = 420 < one run, pre_nms_topN > n
    (60, 60, 6000, 300, 32),     # n = 43200 > 38 runs: the one-workgroup-per-image selection
])
def test_proposal_selection_ties_and_run_edges(dev, H, W, pre, post, quant):
    """The selection's total order is (score descending, flattened anchor index ascending).  The many-workgroup path gets
    it from sorted runs of 1024 consecutive anchor indices ranked against each other (<= against earlier runs, < against
    later ones): scores with massive ties across runs, runs that end exactly at n / hold a few keys, a single short run and a
    map past the LDS capacity must all give the oracle's RoIs bit for bit."""
    from dtt.rpn import generate_anchors, proposal_forward
    rng = np.random.RandomState(H * 100 + W + quant)
    base = generate_anchors(scales=(4, 8, 16, 32))
    A = base.shape[0]
    prob, bbox, info = _proposal_inputs(rng, 2, A, H, W)
    prob = (np.round(prob * quant) / quant if quant else np.full_like(prob, 0.5)).astype(np.float32)
    ref, nref = ro.proposal_layer(prob, bbox, info, base, 16, pre, post, 0.7, O.nms)
    rois, num = proposal_forward(cu(prob, dev), cu(bbox, dev), cu(info, dev), torch.from_numpy(base).float(), 16, pre, post, 0.7)
    np.testing.assert_array_equal(num.cpu().numpy(), nref)
    np.testing.assert_array_equal(rois.cpu().numpy(), ref)


def test_proposal_layer_two_phases_across_streams(dev):
    """_ProposalLayer.select (scores only, side stream) + .finish (box deltas, after a stream dependency) -- the split
    dtt/model.py uses to start the sort under the RPN's box-delta convolution -- is the one-call layer, bit for bit."""
    from dtt.config import cfg
    from dtt.rpn import _ProposalLayer
    rng = np.random.RandomState(77)
    layer = _ProposalLayer(16, cfg.ANCHOR_SCALES, cfg.ANCHOR_RATIOS, cfg=cfg).to(dev)
    A = layer._num_anchors
    prob, bbox, info = _proposal_inputs(rng, 2, A, 38, 67)
    prob, bbox, info = cu(prob, dev), cu(bbox, dev), cu(info, dev)
    ref = layer((prob, bbox, info, "TEST"))
    side, cur = torch.cuda.Stream(device=dev), torch.cuda.current_stream(dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        handle = layer.select(prob, "TEST")
    bbox2 = bbox.clone()                       # "still being computed" on the main stream
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        rois = layer.finish(handle, bbox2, info, "TEST")
    cur.wait_stream(side)
    assert torch.equal(rois, ref) and float(ref.abs().sum()) > 0
    with pytest.raises(ValueError):
        layer.finish(handle, bbox2[:, :4], info, "TEST")


@pytest.mark.parametrize("case", ["test_19x32", "train_19x32", "test_6x8", "test_small_pre"])
def test_proposal_layer_vs_reference_golden(dev, case):
    from dtt.rpn import generate_anchors, proposal_forward
    g = np.load(os.path.join(G, "proposal.npz"))
    base = generate_anchors(scales=g["scales"], ratios=g["ratios"])
    stride, pre, post = (int(v) for v in g[case + "/params"])
    rois, _ = proposal_forward(cu(g[case + "/cls_prob"], dev), cu(g[case + "/bbox_pred"], dev),
                               cu(g[case + "/im_info"], dev), torch.from_numpy(base).float(), stride, pre, post,
                               float(g[case + "/nms_thresh"][0]))
    np.testing.assert_allclose(rois.cpu().numpy(), g[case + "/rois"], rtol=1e-6, atol=2e-4)


# ------------------------------------------------------------------------------ anchor target layer
@pytest.mark.parametrize("case", ["b2_19x32", "b3_12x20", "b2_38x67"])
def test_anchor_target_vs_reference_golden(dev, case):
    from dtt.rpn import anchor_target_forward, generate_anchors
    g = np.load(os.path.join(G, "anchor_target.npz"))
    base = generate_anchors(scales=g["scales"], ratios=g["ratios"])
    H, W = (int(v) for v in g[case + "/hw"])
    np.random.seed(int(g["rng_seed"][0]))
    lab, tgt, inw, outw = anchor_target_forward(cu(g[case + "/gt_boxes"], dev), torch.from_numpy(g[case + "/im_info"]),
                                                torch.from_numpy(base).float(), H, W, 16)
    np.testing.assert_array_equal(lab.cpu().numpy(), g[case + "/labels"])
    np.testing.assert_array_equal(inw.cpu().numpy(), g[case + "/inside"])
    np.testing.assert_array_equal(outw.cpu().numpy(), g[case + "/outside"])
    np.testing.assert_allclose(tgt.cpu().numpy(), g[case + "/bbox_targets"], rtol=1e-6, atol=1e-6)


def test_anchor_target_vs_oracle_bit_exact(dev):
    from dtt.rpn import anchor_target_forward, generate_anchors
    g = np.load(os.path.join(G, "anchor_target.npz"))
    base = generate_anchors(scales=g["scales"], ratios=g["ratios"])
    case = "b2_38x67"
    H, W = (int(v) for v in g[case + "/hw"])
    np.random.seed(11)
    ref = ro.anchor_target_layer(g[case + "/gt_boxes"], g[case + "/im_info"], base, H, W, 16)
    np.random.seed(11)
    got = anchor_target_forward(cu(g[case + "/gt_boxes"], dev), torch.from_numpy(g[case + "/im_info"]),
                                torch.from_numpy(base).float(), H, W, 16)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a.cpu().numpy(), b)


def test_anchor_target_positive_weight(dev):
    """cfg.TRAIN.RPN_POSITIVE_WEIGHT in (0, 1) (anchor_target_layer.py:148-152; see dtt.rpn.anchor_target_forward for how
    the reference's unfinished branch is completed): labels / targets / inside weights as with uniform weighting, outside
    weights p / #positives on positives and (1 - p) / #negatives on negatives, counted on the last image."""
    from dtt.rpn import anchor_target_forward, generate_anchors
    rng = np.random.RandomState(3)
    B, H, W = 2, 19, 32
    base = torch.from_numpy(generate_anchors(scales=(4, 8, 16, 32))).float()
    A = base.shape[0]
    gt = np.zeros((B, 20, 5), np.float32)
    for b in range(B):
        for i in range(3 + b):
            x1, y1 = rng.uniform(0, 350), rng.uniform(0, 180)
            gt[b, i] = [x1, y1, x1 + rng.uniform(40, 150), y1 + rng.uniform(40, 110), 1 + i]
    info = torch.tensor([[H * 16.0, W * 16.0, 1.0]] * B)
    np.random.seed(5)
    uni = anchor_target_forward(cu(gt, dev), info, base, H, W, 16)
    np.random.seed(5)
    p = 0.3
    wtd = anchor_target_forward(cu(gt, dev), info, base, H, W, 16, positive_weight=p)
    for a, b in zip(uni[:3], wtd[:3]):
        assert torch.equal(a, b)
    labels = wtd[0].view(B, A, H, W)                                   # (B, 1, A*H, W) -> anchor-major
    n_pos, n_neg = int((labels[B - 1] == 1).sum()), int((labels[B - 1] == 0).sum())
    assert n_pos > 0 and n_neg > 0
    out_w = wtd[3].view(B, A, 4, H, W)
    pos = (labels == 1).unsqueeze(2).expand_as(out_w)
    neg = (labels == 0).unsqueeze(2).expand_as(out_w)
    assert torch.all(out_w[pos] == np.float32(p) / np.float32(n_pos)) and torch.all(out_w[neg] == np.float32(1 - p) / np.float32(n_neg))
    assert torch.all(out_w[~(pos | neg)] == 0)


@pytest.mark.parametrize("case,batchsize,fg_fraction,tie", [("b2_38x67", 256, 0.5, False), ("b3_12x20", 256, 0.5, False),
                                                          ("b2_38x67", 16, 0.25, False), ("b2_19x32", 2, 1.0, False),
                                                          ("b2_38x67", 256, 0.5, True), ("b2_38x67", 256, 0.5, "coarse"),
                                                          ("b2_38x67", 64, 0.5, "coarse")])
@pytest.mark.parametrize("slow", [False, True])
def test_anchor_target_device_mode_follows_its_selection_rule(dev, monkeypatch, case, batchsize, fg_fraction, tie, slow):
    """cfg.TRAIN.SAMPLER_RNG = "device" (`dtt_anchor_target_device`: no host read anywhere in the layer): of a class over its quota
    the anchors with the smallest (key, anchor index) stay -- restated in numpy on the labels BEFORE subsampling (the reference-mode
    kernels with a quota nothing exceeds; they are pinned bit-exact to the reference above).  Labels exact, quotas as
    anchor_target_layer.py:118-141 (num_fg foreground, batch - foreground-before-subsampling background, every background anchor
    when that is <= 0), targets untouched by the subsampling, outside weights 1 / num_examples of the LAST image.  `tie`: all keys
    equal -- the k-th smallest key is shared by every candidate and the anchor index decides (more candidates in one histogram bin
    than the one-pass selection lists: it hands over to the radix select); "coarse": only the keys' top 12 bits vary -- a dozen
    candidates per bin, all ties, ranked by anchor index inside the one-pass selection.  `slow`: DTT_AT_SUBSAMPLE_SLOW, the radix
    select alone -- both paths must give the rule's subset."""
    from dtt.rpn import anchor_target_forward, generate_anchors
    if slow:
        monkeypatch.setenv("DTT_AT_SUBSAMPLE_SLOW", "1")
    g = np.load(os.path.join(G, "anchor_target.npz"))
    base = torch.from_numpy(generate_anchors(scales=g["scales"], ratios=g["ratios"])).float()
    H, W = (int(v) for v in g[case + "/hw"])
    gt, info = cu(g[case + "/gt_boxes"], dev), torch.from_numpy(g[case + "/im_info"])
    B, A, K = gt.shape[0], base.shape[0], H * W
    n = A * K
    pre = anchor_target_forward(gt, info, base, H, W, 16, rpn_batchsize=10 ** 7)            # nothing over quota: labels before subsampling
    rs = np.random.RandomState(B * 1000 + batchsize)
    keys = np.zeros((B, n), np.int32) + 12345 if tie is True else rs.randint(0, 2 ** 31 - 1, size=(B, n)).astype(np.int32)
    if tie == "coarse":
        keys &= ~np.int32((1 << 20) - 1)
    if not tie:
        m = keys[:, 3::7].shape[1]
        keys[:, 0:7 * m:7] = keys[:, 3::7]                                                 # plenty of equal keys among the candidates
    got = anchor_target_forward(gt, info.to(dev), base, H, W, 16, rpn_batchsize=batchsize, fg_fraction=fg_fraction, mode="device",
                                keys=torch.from_numpy(keys).to(dev))
    to_t = lambda x: x.view(B, A, K).permute(0, 2, 1).reshape(B, n)                         # (B, 1, A*H, W) -> anchor index t = k * A + a
    lab_pre = to_t(pre[0]).cpu().numpy()
    num_fg = int(fg_fraction * batchsize)
    want = lab_pre.copy()
    t = np.arange(n)
    for b in range(B):
        fg, bg = np.nonzero(lab_pre[b] == 1)[0], np.nonzero(lab_pre[b] == 0)[0]
        if fg.size > num_fg:
            order = fg[np.lexsort((t[fg], keys[b, fg]))]
            want[b, order[num_fg:]] = -1
        num_bg = batchsize - fg.size
        if bg.size > num_bg:
            order = bg[np.lexsort((t[bg], keys[b, bg]))]
            want[b, order[max(num_bg, 0):]] = -1
        assert (want[b] == 1).sum() == min(fg.size, num_fg) and (want[b] == 0).sum() == min(bg.size, max(num_bg, 0))
    lab = to_t(got[0]).cpu().numpy()
    np.testing.assert_array_equal(lab, want)
    assert torch.equal(got[1], pre[1])                                                       # regression targets: every inside anchor
    lab4 = got[0].view(B, A, 1, H, W).expand(B, A, 4, H, W).reshape(B, 4 * A, H, W)
    assert torch.equal(got[2], (lab4 == 1).float())                                          # inside weights (1.0) on the kept positives
    n_ex = int((want[B - 1] >= 0).sum())
    w = np.float32(1.0) / np.float32(n_ex) if n_ex else np.float32(np.inf)
    assert torch.equal(got[3], (lab4 >= 0).float() * float(w)) if n_ex else True
    if batchsize == 256 and not tie:
        assert (lab_pre != want).any()                                                       # the case does subsample


def test_anchor_target_device_mode_draws_uniform_subsets(dev):
    """The device-mode subset is uniform: over 300 seeded calls with a quota of 56 of ~250 background anchors per image, every
    candidate is kept about 300 * 56 / m times (binomial, +- 5 sigma) and every call keeps exactly the quota.  (The draws are the
    device generator's, seeded from numpy: the counts are the same on every box.)"""
    from dtt.rpn import anchor_target_forward, generate_anchors
    g = np.load(os.path.join(G, "anchor_target.npz"))
    base = torch.from_numpy(generate_anchors(scales=g["scales"], ratios=g["ratios"])).float()
    case = "b3_12x20"
    H, W = (int(v) for v in g[case + "/hw"])
    gt, info = cu(g[case + "/gt_boxes"], dev), cu(g[case + "/im_info"], dev)
    pre = anchor_target_forward(gt, info.cpu(), base, H, W, 16, rpn_batchsize=10 ** 7)[0]
    cand = (pre == 0)
    trials, batchsize = 300, 64
    kept = torch.zeros_like(pre)
    np.random.seed(77)
    for _ in range(trials):
        lab = anchor_target_forward(gt, info, base, H, W, 16, rpn_batchsize=batchsize, mode="device")[0]
        assert bool(((lab == 0) <= cand).all())                                      # only candidates are kept
        kept += (lab == 0).float()
    for b in range(gt.shape[0]):
        m = int(cand[b].sum())
        k = batchsize - int((pre[b] == 1).sum())                                       # the reference's background quota
        assert 0 < k < m
        cnt = kept[b][cand[b]].cpu().numpy()
        assert cnt.sum() == trials * k                                                 # exactly the quota, every call
        p = k / m
        sigma = np.sqrt(trials * p * (1 - p))
        assert np.abs(cnt - trials * p).max() < 5 * sigma, (b, cnt.min(), cnt.max(), trials * p, sigma)
        assert abs(cnt.std() - sigma) < 0.25 * sigma                                   # spread of a binomial, not of a biased rule


def test_anchor_target_device_mode_is_seeded_by_numpy_and_reads_nothing_back(dev):
    """The layer in device mode: the keys come from the device generator, seeded by ONE integer drawn from numpy's global generator
    per call -- the same numpy seed gives the same sample, another seed another one -- and the call makes no synchronising
    operation (torch.cuda.set_sync_debug_mode("error") would raise on a device-to-host copy or a blocking upload)."""
    from dtt.config import cfg
    from dtt.rpn import _AnchorTargetLayer
    g = np.load(os.path.join(G, "anchor_target.npz"))
    case = "b2_38x67"
    H, W = (int(v) for v in g[case + "/hw"])
    gt, info = cu(g[case + "/gt_boxes"], dev), cu(g[case + "/im_info"], dev)
    import copy
    c = copy.deepcopy(cfg)
    c.TRAIN.SAMPLER_RNG = "device"
    layer = _AnchorTargetLayer(16, [int(v) for v in g["scales"]], [float(v) for v in g["ratios"]], cfg=c).to(dev)
    score = torch.zeros(gt.shape[0], 2 * layer._num_anchors, H, W, device=dev)
    nb = torch.zeros(gt.shape[0], 1, device=dev)
    np.random.seed(21); a = layer((score, gt, info, nb))
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        np.random.seed(21); b = layer((score, gt, info, nb))
        np.random.seed(22); d = layer((score, gt, info, nb))
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[0], d[0]) and torch.equal(a[1], d[1])
    assert int((a[0] == 1).sum()) + int((a[0] == 0).sum()) <= gt.shape[0] * c.TRAIN.RPN_BATCHSIZE


# ------------------------------------------------------------------------ test-time per-class NMS
@pytest.mark.parametrize("R,ncls,agnostic,mpi", [(300, 31, True, 100), (300, 31, False, 100), (1000, 31, True, 100),
                                                  (77, 5, True, 0), (300, 31, True, 5)])
def test_class_nms_vs_oracle(dev, R, ncls, agnostic, mpi):
    from dtt.postprocess import class_nms, to_all_boxes
    rng = np.random.RandomState(R + ncls + mpi)
    I = 2
    scores = rng.dirichlet(np.ones(ncls) * 0.3, size=(I, R)).astype(np.float32)
    scores[0, :7, 3] = scores[0, 0, 3]  # exact ties
    ctr = rng.uniform(50, 600, size=(I, R, 2))
    m = min(ctr[:, ::3].shape[1], ctr[:, 1::3].shape[1])
    ctr[:, ::3][:, :m] = ctr[:, 1::3][:, :m] + rng.normal(0, 5, size=(I, m, 2))  # clusters of near-duplicates
    wh = rng.uniform(30, 200, size=(I, R, 2))
    base = np.concatenate([ctr - wh / 2, ctr + wh / 2], 2).astype(np.float32)
    if agnostic:
        boxes = base
    else:
        boxes = (base[:, :, None, :] + rng.normal(0, 3, size=(I, R, ncls, 4))).reshape(I, R, 4 * ncls).astype(np.float32)
    dets, counts = class_nms(cu(scores, dev), cu(boxes, dev), 0.05, 0.3, mpi, agnostic)
    got = to_all_boxes(dets, counts)
    total = 0
    for i in range(I):
        ref = ro.class_nms(scores[i], boxes[i], O.nms, 0.05, 0.3, mpi, agnostic)
        for j in range(ncls):
            np.testing.assert_array_equal(got[i][j], ref[j], err_msg="image %d class %d" % (i, j))
            total += len(ref[j])
    assert total > 0


@pytest.mark.parametrize("case", ["agnostic_300x31", "perclass_120x7", "nocut_80x5", "sparse_60x31"])
def test_class_nms_vs_reference_loop_golden(dev, case):
    """dtt_class_nms against the detections the reference's own test_net.py loop (lines 274-301, executed by
    tests/golden/make_golden_class_nms.py) kept on the same inputs: bit for bit, order included."""
    from dtt.postprocess import class_nms, to_all_boxes
    g = np.load(os.path.join(G, "class_nms.npz"))
    thresh, nms_t, mpi, agn = g[case + "/params"]
    counts = g[case + "/counts"]
    want = np.split(g[case + "/dets"], np.cumsum(counts)[:-1])
    dets, cnt = class_nms(cu(g[case + "/scores"][None], dev), cu(g[case + "/boxes"][None], dev), float(thresh), float(nms_t),
                          int(mpi), bool(agn))
    got = to_all_boxes(dets, cnt)[0]
    for j in range(len(want)):
        np.testing.assert_array_equal(got[j], want[j].reshape(-1, 5), err_msg="class %d" % j)


@pytest.mark.parametrize("n,c,h,w", [(4, 64, 300, 534), (1, 64, 33, 47), (2, 8, 7, 9), (1, 4, 3, 3), (2, 64, 282, 500)])
def test_stem_maxpool_bias_relu_fused(dev, n, c, h, w):
    """dtt_maxpool3s2_bias_relu_nhwc == the reference's stem order relu(x + b) -> MaxPool2d(3, 2, 0, ceil_mode=True)
    (resnet.py:110-117 with the BatchNorm folded), bit for bit: shift and clamp commute with the window maximum."""
    import torch.nn.functional as F
    from dtt.fuse import maxpool3s2_bias_relu_nhwc
    g = torch.Generator().manual_seed(h * 7 + w)
    x = torch.randn(n, c, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    b = torch.randn(c, generator=g).to(dev)
    want = F.max_pool2d(torch.relu(x + b.view(1, -1, 1, 1)), 3, 2, 0, ceil_mode=True)
    got = maxpool3s2_bias_relu_nhwc(x, b)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)
