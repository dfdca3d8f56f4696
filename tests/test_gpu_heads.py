"""Hand-written R-FCN heads (exact-fp32 MFMA GEMM, position-major output) and the lanes = classes PSRoI pooling
(csrc/heads.hip) through the C ABI: head output against F.conv2d in fp32 (1e-4), pooled bins and votes bit-identical
to the CPU oracle on the same map.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from dtt import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _convs(dev, K, ods, seed):
    g = torch.Generator().manual_seed(seed)
    convs = []
    for od in ods:
        c = torch.nn.Conv2d(K, od * 49, 1)
        c.weight.data = torch.randn(c.weight.shape, generator=g) * 0.05
        c.bias.data = torch.randn(c.bias.shape, generator=g) * 0.1
        convs.append(c.to(dev))
    return convs


def _rois(rng, n, batch, H, W):
    from test_gpu_ops import random_rois
    return random_rois(rng, n, batch, H * 16, W * 16)


@pytest.mark.parametrize("B,H,W,K,ods,passes", [
    (4, 38, 67, 512, (31, 4), 0),      # BASELINE configs[2]: both legs of two frame pairs, class + box heads in one GEMM
    (4, 38, 67, 512, (31, 4), 1),
    (4, 38, 67, 512, (31, 4), 4),
    (2, 36, 63, 512, (31, 4), 0),      # configs[4] map size, one pair
    (1, 19, 32, 512, (31,), 0),        # configs[0] map size, class head alone
    (2, 38, 67, 1056, (4,), 0),        # tracking head: 1051 input channels padded to 1056, narrow configuration
    (1, 5, 7, 64, (4,), 0),
    (3, 9, 11, 96, (31, 4), 0),
])
def test_head_gemm_matches_conv2d(dev, B, H, W, K, ods, passes):
    from dtt.heads import PackedHeads, head_gemm, pm_to_nchw
    convs = _convs(dev, K, ods, seed=K + H)
    packed = PackedHeads(convs)
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(B, K, H, W, generator=g)).to(dev)
    rows = x.permute(0, 2, 3, 1).reshape(-1, K).contiguous()
    out = head_gemm(rows, packed, passes=passes)
    assert out.shape == (B * H * W, packed.stride)
    for conv, head in zip(convs, packed.heads):
        ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double())
        got = pm_to_nchw(out, head, B, H, W)
        err = float((got.double() - ref).abs().max())
        assert err < 1e-4, err
        # exact-fp32 MFMA: the error is fp32 round-off of a K-term sum, far inside the tolerance
        assert err < 2e-5 * max(1.0, float(ref.abs().max()))
        # padding classes of a bin are exact zeros (zero weight rows, zero bias)
        if head["cp"] > head["od"]:
            pad = out[:, head["offset"]:head["offset"] + 49 * head["cp"]].reshape(-1, 49, head["cp"])[..., head["od"]:]
            assert float(pad.abs().max()) == 0.0


def test_head_gemm_k_padding(dev):
    """1051 tracking channels (rfcn.py:166-174) padded with zero columns to the kernel's K granularity."""
    from dtt.heads import PackedHeads, head_gemm, pm_to_nchw
    convs = _convs(dev, 1051, (4,), seed=3)
    packed = PackedHeads(convs, k_pad=1056)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 1051, 10, 13, generator=g).to(dev)
    rows = torch.zeros(2 * 10 * 13, 1056, device=dev)
    rows[:, :1051] = x.permute(0, 2, 3, 1).reshape(-1, 1051)
    got = pm_to_nchw(head_gemm(rows, packed), packed.heads[0], 2, 10, 13)
    ref = F.conv2d(x.double(), convs[0].weight.double(), convs[0].bias.double())
    assert float((got.double() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("B,H,W,ods,R", [(4, 38, 67, (31, 4), 1200), (2, 36, 63, (31, 4), 600), (1, 19, 32, (31,), 300),
                                         (2, 7, 9, (4,), 40), (3, 50, 21, (31, 4), 77)])
def test_psroi_pm_bit_identical_to_oracle(dev, B, H, W, ods, R):
    from dtt.heads import PackedHeads, pm_to_nchw, psroi_pm
    convs = _convs(dev, 32, ods, seed=5)
    packed = PackedHeads(convs)
    rng = np.random.RandomState(B * 100 + H)
    pm = torch.from_numpy(rng.normal(size=(B * H * W, packed.stride)).astype(np.float32)).to(dev)
    rois = _rois(rng, R, B, H, W)
    rois_d = torch.from_numpy(rois).to(dev)
    for head in packed.heads:
        nchw = pm_to_nchw(pm, head, B, H, W).cpu().numpy()
        ref_pooled, _ = O.psroi_pool_forward(nchw, rois, 7, 7, 1.0 / 16, 7, head["od"])
        # the vote of rfcn.py:62-64: row-major sum of the 49 bins, one division (same order as csrc/psroi.hip `psroi_vote`)
        s = np.zeros(ref_pooled.shape[:2], np.float32)
        for k in range(49):
            s = (s + ref_pooled.reshape(R, head["od"], 49)[:, :, k]).astype(np.float32)
        ref_vote = (s / np.float32(49)).astype(np.float32)
        vote, pooled = psroi_pm(pm, head, B, H, W, rois_d, 1.0 / 16, want_pooled=True)
        assert np.array_equal(pooled.cpu().numpy(), ref_pooled)
        assert np.array_equal(vote.cpu().numpy(), ref_vote)
        assert torch.equal(psroi_pm(pm, head, B, H, W, rois_d, 1.0 / 16), vote)


def test_psroi_pm_matches_plane_kernel_on_head_output(dev):
    """End to end on the same weights: head GEMM -> position-major pooling  ==  plane-stationary PSRoI kernel on the
    NCHW view of that map (bit for bit), and within 1e-4 of F.conv2d -> reference-layout pooling."""
    from dtt.heads import PackedHeads, head_gemm, pm_to_nchw, psroi_pm
    from dtt.ops import psroi_pool_vote
    B, H, W, K = 2, 38, 67, 512
    convs = _convs(dev, K, (31, 4), seed=9)
    packed = PackedHeads(convs)
    g = torch.Generator().manual_seed(4)
    x = torch.relu(torch.randn(B, K, H, W, generator=g)).to(dev)
    pm = head_gemm(x.permute(0, 2, 3, 1).reshape(-1, K).contiguous(), packed)
    rng = np.random.RandomState(11)
    rois = torch.from_numpy(_rois(rng, 600, B, H, W)).to(dev)
    for conv, head in zip(convs, packed.heads):
        vote = psroi_pm(pm, head, B, H, W, rois, 1.0 / 16)
        _, v2 = psroi_pool_vote(pm_to_nchw(pm, head, B, H, W), rois, 7, 7, 1.0 / 16, 7, head["od"])
        assert torch.equal(vote, v2)
        _, v3 = psroi_pool_vote(F.conv2d(x, conv.weight, conv.bias).contiguous(), rois, 7, 7, 1.0 / 16, 7, head["od"])
        assert float((vote - v3).abs().max()) < 1e-4


def test_gather_column_blocks_matches_slicing():
    """dtt_gather_column_blocks == the two sliced copies it replaces (box-delta columns of both legs -> tracking rows)."""
    from dtt.heads import gather_column_blocks
    g = torch.Generator().manual_seed(5)
    src = torch.randn(2 * 611, 1744, generator=g).cuda()
    dst = torch.full((611, 1056), -7.0).cuda()
    want = dst.clone()
    want[:, 0:196] = src[:611, 1536:1732]
    want[:, 196:392] = src[611:, 1536:1732]
    gather_column_blocks(dst, 0, src, 1536, 611, 2, 196)
    assert torch.equal(dst, want)
    with pytest.raises(ValueError):
        gather_column_blocks(dst, 900, src, 1536, 611, 2, 196)


@pytest.mark.parametrize("B,H,W,K,A", [
    (4, 38, 67, 512, 12),     # the benchmark step: both legs of two frame pairs, 12 anchors (rpn.py:63-71)
    (1, 19, 32, 512, 12),     # one small image
    (2, 7, 5, 64, 2),         # tiny: two anchors, one chunk pair, rows not a multiple of anything
    (3, 24, 33, 96, 6),
])
def test_rpn_heads_one_launch(dev, B, H, W, K, A):
    """dtt_rpn_head_gemm = RPN_cls_score + reshape(2) / softmax / reshape(2A) + RPN_bbox_pred of rpn.py:63-71 in one launch over
    channels-last rows, against F.conv2d / F.softmax in float64 (1e-4 absolute: an fp32 fma chain + expf against the
    library's GEMM and softmax); and the proposal layer fed by it returns exactly the RoIs it returns for torch's tensors
    whenever the kernel's scores rank the anchors the same way (checked through the score order it was given)."""
    from dtt.heads import PackedRPNHeads, rpn_head_gemm
    g = torch.Generator().manual_seed(B * 100 + A)
    cls = torch.nn.Conv2d(K, 2 * A, 1)
    box = torch.nn.Conv2d(K, 4 * A, 1)
    for c, sc in ((cls, 0.08), (box, 0.03)):
        c.weight.data = torch.randn(c.weight.shape, generator=g) * sc
        c.bias.data = torch.randn(c.bias.shape, generator=g) * 0.1
    cls, box = cls.to(dev), box.to(dev)
    x = torch.relu(torch.randn(B, K, H, W, generator=g)).to(dev)
    rows = x.permute(0, 2, 3, 1).reshape(B * H * W, K).contiguous()
    with torch.no_grad():
        prob, bbox = rpn_head_gemm(rows, PackedRPNHeads(cls, box), B, H, W)
    x64 = x.double()
    cls.requires_grad_(False); box.requires_grad_(False)
    score = F.conv2d(x64, cls.weight.double(), cls.bias.double())
    want_prob = F.softmax(score.view(B, 2, A * H, W), dim=1).view(B, 2 * A, H, W)
    want_bbox = F.conv2d(x64, box.weight.double(), box.bias.double())
    assert prob.shape == want_prob.shape and bbox.shape == want_bbox.shape
    assert float((prob.double() - want_prob).abs().max()) < 1e-4
    assert float((bbox.double() - want_bbox).abs().max()) < 1e-4
    assert float((prob[:, :A] + prob[:, A:] - 1).abs().max()) < 1e-6       # the pair sums to one
    if A == 12:
        # fed to the proposal layer: same RoIs as from the float32 library path when no two of the top scores swap
        from dtt.rpn import _ProposalLayer
        prop = _ProposalLayer(16, [4, 8, 16, 32], [0.5, 1, 2]).to(dev)          # 12 anchors (ImageNet VID: trainval_net.py:162-172)
        info = torch.tensor([[H * 16.0, W * 16.0, 1.0]] * B, device=dev)
        lib_prob = F.softmax(F.conv2d(x, cls.weight, cls.bias).view(B, 2, A * H, W), dim=1).view(B, 2 * A, H, W).contiguous()
        a = prop((prob, bbox, info, "TEST"))
        b = prop((prob, F.conv2d(x, box.weight, box.bias).contiguous(), info, "TEST"))
        assert float((a - b).abs().max()) < 1e-2          # same selection (scores identical), box deltas within rounding
        assert prop((lib_prob, bbox, info, "TEST")).shape == a.shape


@pytest.mark.parametrize("B,H,W,K,ods", [
    (4, 38, 67, 512, (31, 4)),     # the training step's shape: class + box heads of both legs of two frame pairs
    (2, 13, 17, 64, (31, 4)),
    (1, 9, 11, 96, (4,)),
])
def test_head_gemm_autograd_matches_conv2d_in_float64(dev, B, H, W, K, ods):
    """HeadGemmFn (forward, dX, dW, dBias all on dtt_head_gemm) over the differentiable packing of the LIVE conv parameters,
    against F.conv2d autograd in float64: 1e-4 relative to each tensor's largest entry (rfcn.py:49-53 in the training graph)."""
    from dtt.heads import HeadGemmFn, pack_heads_differentiable, pm_to_nchw
    convs = _convs(dev, K, ods, seed=B + K)
    for c in convs:
        c.weight.requires_grad_(True); c.bias.requires_grad_(True)
    g = torch.Generator().manual_seed(7)
    x = torch.relu(torch.randn(B, K, H, W, generator=g)).to(dev)
    rows = x.permute(0, 2, 3, 1).reshape(B * H * W, K).contiguous().requires_grad_(True)
    w, b, heads, n_store, stride = pack_heads_differentiable(convs)
    out = HeadGemmFn.apply(rows, w, b, n_store, stride)
    gouts = [torch.randn(B, od * 49, H, W, generator=g).to(dev) for od in ods]
    loss = sum((pm_to_nchw(out, h, B, H, W) * go).sum() for h, go in zip(heads, gouts))
    loss.backward()
    x64 = x.double().requires_grad_(True)
    ref_params = [(c.weight.detach().double().requires_grad_(True), c.bias.detach().double().requires_grad_(True)) for c in convs]
    ref_loss = sum((F.conv2d(x64, wr, br) * go.double()).sum() for (wr, br), go in zip(ref_params, gouts))
    ref_loss.backward()
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max().clamp_min(1e-30))
    for h, (wr, br), go in zip(heads, ref_params, gouts):
        assert rel(pm_to_nchw(out.detach(), h, B, H, W), F.conv2d(x64.detach(), wr.detach(), br.detach())) < 1e-4
    assert rel(rows.grad.view(B, H, W, K).permute(0, 3, 1, 2), x64.grad) < 1e-4
    for c, (wr, br) in zip(convs, ref_params):
        assert rel(c.weight.grad, wr.grad) < 1e-4, "dW"
        assert rel(c.bias.grad, br.grad) < 1e-4, "dBias"


@pytest.mark.parametrize("B,H,W,R", [(4, 38, 67, 512), (2, 20, 30, 77), (1, 6, 9, 300), (3, 12, 9, 0)])
def test_psroi_pm_backward_matches_oracle(dev, B, H, W, R):
    """dtt_psroi_pm_backward (map-stationary, no atomics) for the class and box heads of one map in one autograd node, against
    the oracle's PSROIPoolBackward (psroi_pooling_kernel.cu:109-170) fed the AvgPool2d gradient (rfcn.py:62-64): 1e-5; the
    padding columns of the gradient rows are zero; run-to-run identical (RoI order, not atomics)."""
    from dtt.heads import PsroiPmFn, pm_to_nchw
    rng = np.random.RandomState(B * 7 + R)
    heads = [dict(offset=0, cp=32, od=31, group=7), dict(offset=49 * 32, cp=4, od=4, group=7)]
    stride = 1792
    pm = torch.from_numpy(rng.normal(size=(B * H * W, stride)).astype(np.float32)).to(dev).requires_grad_(True)
    rois = _rois(rng, R, B, H, W) if R else np.zeros((0, 5), np.float32)
    rt = torch.from_numpy(rois).to(dev)
    votes = PsroiPmFn.apply(pm, rt, B, H, W, 1 / 16.0, heads)
    gv = [rng.normal(size=tuple(v.shape)).astype(np.float32) for v in votes]
    torch.autograd.backward(votes, [torch.from_numpy(g).to(dev) for g in gv])
    gm = pm.grad
    assert bool((gm[:, 49 * 36:] == 0).all())
    for h, g in zip(heads, gv):
        od = h["od"]
        top_diff = np.repeat((g / np.float32(49.0)).reshape(R, od, 1, 1), 49, axis=2).reshape(R, od, 7, 7).astype(np.float32)
        want = O.psroi_pool_backward(top_diff, rois, (B, od * 49, H, W), 7, 7, 1 / 16.0, 7, od) if R else np.zeros((B, od * 49, H, W), np.float32)
        got = pm_to_nchw(gm, h, B, H, W).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(want).max())))
        pad = gm[:, h["offset"]:h["offset"] + 49 * h["cp"]].reshape(-1, 49, h["cp"])[:, :, od:]
        assert bool((pad == 0).all())
    pm.grad = None
    votes = PsroiPmFn.apply(pm, rt, B, H, W, 1 / 16.0, heads)
    torch.autograd.backward(votes, [torch.from_numpy(g).to(dev) for g in gv])
    assert torch.equal(pm.grad, gm)


def _pm_bwd_heads(dev, gvs, heads, rois, B, H, W, stride, row_floats, add=None, add_first=0):
    """dtt_psroi_pm_backward_heads through the C ABI into a NaN-filled map (every column it owns must be written)."""
    import ctypes
    from dtt import _lib
    from dtt._lib import check, ptr, stream_ptr
    L = _lib.lib()
    R = rois.shape[0]
    gmap = torch.full((B * H * W, stride), float("nan"), dtype=torch.float32, device=dev)
    h0, h1 = heads[0], heads[1] if len(heads) == 2 else None
    with torch.cuda.device(dev):
        check(L.dtt_psroi_pm_backward_heads(ptr(gvs[0]), h0["od"], h0["cp"], ptr(gvs[1]) if h1 else None, h1["od"] if h1 else 0,
                                            h1["cp"] if h1 else 0, ptr(rois), R, B, H, W, 7, 1 / 16.0, stride, row_floats,
                                            ptr(add) if add is not None else None, add_first, add.shape[1] if add is not None else 0,
                                            ptr(gmap), stream_ptr(dev)), "psroi_pm backward (heads)")
    return gmap


def _pm_bwd_single(dev, gv, head, rois, B, H, W, stride, old):
    """dtt_psroi_pm_backward for one head; old = the one-workgroup-per-pixel kernel of rounds 4 - 5 (DTT_PSROI_BWD_OLD=1)."""
    import ctypes
    from dtt import _lib
    from dtt._lib import check, ptr, stream_ptr
    L = _lib.lib()
    R = rois.shape[0]
    gmap = torch.full((B * H * W, stride), float("nan"), dtype=torch.float32, device=dev)
    edges = torch.empty((max(R, 1) * 29 + 2 * B,), dtype=torch.int32, device=dev)
    prev = os.environ.get("DTT_PSROI_BWD_OLD")
    os.environ["DTT_PSROI_BWD_OLD"] = "1" if old else "0"
    try:
        with torch.cuda.device(dev):
            check(L.dtt_psroi_pm_backward(ptr(gv), ptr(rois), R, B, H, W, 7, 1 / 16.0, head["od"], head["cp"], stride,
                                          ctypes.c_void_p(gmap.data_ptr() + 4 * head["offset"]), ptr(edges), stream_ptr(dev)), "psroi_pm backward")
        torch.cuda.synchronize(dev)
    finally:
        if prev is None:
            del os.environ["DTT_PSROI_BWD_OLD"]
        else:
            os.environ["DTT_PSROI_BWD_OLD"] = prev
    return gmap


@pytest.mark.parametrize("B,H,W,R,kind", [
    (4, 38, 67, 512, "train"),        # the training step: 128 RoIs per image, one staged chunk, two rounds of 64
    (2, 38, 67, 16, "few"),           # the tracking head's handful of RoIs
    (1, 38, 67, 700, "chunks"),       # more RoIs per image than one chunk holds (3 chunks of 256): workgroup-synchronised restaging
    (2, 10, 12, 400, "tiny"),         # tiny RoIs: all 49 bins of each cover the same pixels -> the list overflows, serial walk
    (3, 12, 9, 200, "unsorted"),      # RoIs of the images interleaved (an image's run contains other images' RoIs)
    (2, 9, 11, 0, "none"),
])
def test_psroi_pm_backward_heads_one_launch(dev, B, H, W, R, kind):
    """dtt_psroi_pm_backward_heads (csrc/psroi_bwd.hip: one wave per pixel, class + box heads in one launch, padding columns and the
    added compact gradient in the same pass) against (a) the oracle's PSROIPoolBackward (psroi_pooling_kernel.cu:109-170) fed the
    AvgPool2d gradient (rfcn.py:62-64) at 1e-5, (b) the one-workgroup-per-pixel kernel of rounds 4 - 5 head by head, BIT for bit (same
    RoI order of the adds), (c) itself run twice.  Cases: one chunk / several chunks of staged RoIs, list overflow, interleaved images,
    zero-gradient RoIs (skipped), a row length that is not a multiple of 4 (scalar write-out), no RoIs."""
    from dtt.heads import pm_to_nchw
    rng = np.random.RandomState(R + H)
    heads = [dict(offset=0, cp=32, od=31, group=7), dict(offset=49 * 32, cp=4, od=4, group=7)]
    stride = 1792
    if R:
        rois = _rois(rng, R, B, H, W)
        if kind == "tiny":
            cx, cy = rng.randint(0, W * 16 - 8, size=R), rng.randint(0, H * 16 - 8, size=R)
            rois[:, 1], rois[:, 2] = cx, cy
            rois[:, 3], rois[:, 4] = cx + rng.randint(0, 3, size=R), cy + rng.randint(0, 3, size=R)
            rois[: R // 2, 1:] = np.array([33, 17, 34, 18], np.float32)          # 200 RoIs on one pixel: 9800 list entries for it
        if kind == "unsorted":
            rois[:, 0] = rng.randint(0, B, size=R)
        else:
            rois[:, 0] = np.sort(rois[:, 0])
    else:
        rois = np.zeros((0, 5), np.float32)
    gv = [rng.normal(size=(R, h["od"])).astype(np.float32) for h in heads]
    if R:
        gv[1][rng.rand(R) < 0.75] = 0            # background RoIs: no box gradient
        dead = rng.rand(R) < 0.1
        gv[0][dead] = 0; gv[1][dead] = 0         # rows that are zero in both heads: given to no image
    rt = torch.from_numpy(rois).to(dev)
    gvt = [torch.from_numpy(g).to(dev) for g in gv]
    add = torch.from_numpy(rng.normal(size=(B * H * W, 196)).astype(np.float32)).to(dev)
    gm = _pm_bwd_heads(dev, gvt, heads, rt, B, H, W, stride, stride, add, 49 * 32)
    assert bool((gm[:, 49 * 36:] == 0).all()), "padding columns behind the heads"
    assert torch.equal(gm, _pm_bwd_heads(dev, gvt, heads, rt, B, H, W, stride, stride, add, 49 * 32)), "run to run"
    plain = _pm_bwd_heads(dev, gvt, heads, rt, B, H, W, stride, 49 * 36)
    assert bool(torch.isnan(plain[:, 49 * 36:]).all()), "columns past row_floats must not be written"
    want_add = plain[:, :49 * 36].clone()
    want_add[:, 49 * 32:] += add
    assert torch.equal(gm[:, :49 * 36], want_add), "added compact gradient"
    for h, g, gt in zip(heads, gv, gvt):
        od = h["od"]
        top_diff = np.repeat((g / np.float32(49.0)).reshape(R, od, 1, 1), 49, axis=2).reshape(R, od, 7, 7).astype(np.float32)
        want = O.psroi_pool_backward(top_diff, rois, (B, od * 49, H, W), 7, 7, 1 / 16.0, 7, od) if R else np.zeros((B, od * 49, H, W), np.float32)
        got = pm_to_nchw(plain, h, B, H, W).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(want).max())))
        cols = slice(h["offset"], h["offset"] + 49 * h["cp"])
        old = _pm_bwd_single(dev, gt, h, rt, B, H, W, stride, old=True)
        assert torch.equal(plain[:, cols], old[:, cols]), "one-launch kernel vs the per-pixel-workgroup kernel, head od=%d" % od
        new1 = _pm_bwd_single(dev, gt, h, rt, B, H, W, stride, old=False)
        assert torch.equal(new1[:, cols], old[:, cols]), "single-head entry on the wave-per-pixel kernel"
    # a row that is not a multiple of four floats (scalar write-out), one head
    h = dict(offset=0, cp=4, od=3, group=7)
    g3 = torch.from_numpy(rng.normal(size=(R, 3)).astype(np.float32)).to(dev)
    odd = _pm_bwd_heads(dev, [g3], [h], rt, B, H, W, 203, 197)
    ref = _pm_bwd_single(dev, g3, h, rt, B, H, W, 203, old=True)
    assert torch.equal(odd[:, :196], ref[:, :196]) and bool((odd[:, 196] == 0).all()) and bool(torch.isnan(odd[:, 197:]).all())


@pytest.mark.parametrize("M,N,K,g_cols", [(10184, 1776, 512, 1792), (1000, 100, 36, 128), (33, 17, 4, 20), (4097, 300, 132, 320),
                                          (31, 256, 128, 256)])
def test_head_gemm_dw_kernel(M, N, K, g_cols):
    """dtt_head_gemm_dw: dW = gOut[:, :N].T @ x in exact f32 on the matrix cores, operands read as they lie, pixel rows split
    over workgroups and added in a fixed order.  Against a float64 product at 1e-4 of the largest entry; bit-identical from run
    to run; padding columns of the gradient (finite garbage) do not leak into the stored rows; ragged M / N / K."""
    from dtt import _lib
    from dtt._lib import check, ptr, stream_ptr
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + N)
    gout = torch.randn(M, g_cols, generator=g).to(dev)
    gout[:, N:] = 1e3                       # padding columns: finite, large
    x = torch.randn(M, K, generator=g).to(dev)
    L = _lib.lib()
    nb = L.dtt_head_gemm_dw_workspace_bytes(M, N, K)
    outs = []
    for _ in range(2):
        ws = torch.full((nb,), 255, dtype=torch.uint8, device=dev)     # NaN-filled workspace: every word read must have been written
        dw = torch.empty(N, K, device=dev)
        with torch.cuda.device(dev):
            check(L.dtt_head_gemm_dw(ptr(gout), g_cols, g_cols, ptr(x), K, M, N, K, ptr(dw), ptr(ws), nb, stream_ptr(dev)), "dw")
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    ref = gout[:, :N].double().t() @ x.double()
    err = (outs[0].double() - ref).abs().max().item()
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("B,H,W,K,A", [
    (4, 38, 67, 512, 12),     # the training step: both legs of two frame pairs
    (2, 13, 17, 64, 2),
    (1, 9, 11, 96, 6),
])
def test_rpn_head_autograd_matches_conv2d_softmax_in_float64(dev, B, H, W, K, A):
    """RpnHeadFn (rpn.py:63-71 under autograd: one GEMM launch forward; backward = dtt_rpn_head_grad_rows with the pairwise
    softmax's adjoint, dX on dtt_head_gemm, dW on dtt_head_gemm_dw) over the differentiable packing of the live parameters,
    against F.conv2d + F.softmax autograd in float64: outputs and all five gradients to 1e-4 of each tensor's largest entry."""
    from dtt.heads import RpnHeadFn, pack_rpn_heads_differentiable
    g = torch.Generator().manual_seed(B * 10 + A)
    cls, box = torch.nn.Conv2d(K, 2 * A, 1), torch.nn.Conv2d(K, 4 * A, 1)
    for c, sc in ((cls, 0.08), (box, 0.03)):
        c.weight.data = torch.randn(c.weight.shape, generator=g) * sc
        c.bias.data = torch.randn(c.bias.shape, generator=g) * 0.1
    cls, box = cls.to(dev), box.to(dev)
    x = torch.relu(torch.randn(B, K, H, W, generator=g)).to(dev)
    rows = x.permute(0, 2, 3, 1).reshape(B * H * W, K).contiguous().requires_grad_(True)
    w, b, A2 = pack_rpn_heads_differentiable(cls, box)
    assert A2 == A
    prob, bbox = RpnHeadFn.apply(rows, w, b, A, B, H, W)
    gp = torch.randn(prob.shape, generator=g).to(dev)
    gb = torch.randn(bbox.shape, generator=g).to(dev)
    ((prob * gp).sum() + (bbox * gb).sum()).backward()
    x64 = x.double().requires_grad_(True)
    ref = [t.detach().double().requires_grad_(True) for t in (cls.weight, cls.bias, box.weight, box.bias)]
    score = F.conv2d(x64, ref[0], ref[1])
    want_prob = F.softmax(score.view(B, 2, A * H, W), dim=1).view(B, 2 * A, H, W)
    want_bbox = F.conv2d(x64, ref[2], ref[3])
    ((want_prob * gp.double()).sum() + (want_bbox * gb.double()).sum()).backward()
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max().clamp_min(1e-30))
    assert rel(prob.detach(), want_prob.detach()) < 1e-4 and rel(bbox.detach(), want_bbox.detach()) < 1e-4
    assert rel(rows.grad.view(B, H, W, K).permute(0, 3, 1, 2), x64.grad) < 1e-4, "dX"
    for name, p, r in (("cls dW", cls.weight, ref[0]), ("cls dBias", cls.bias, ref[1]), ("box dW", box.weight, ref[2]), ("box dBias", box.bias, ref[3])):
        assert rel(p.grad, r.grad) < 1e-4, name
    # only one of the two outputs carries a gradient (the other arrives as None / zeros)
    rows2 = rows.detach().clone().requires_grad_(True)
    prob2, bbox2 = RpnHeadFn.apply(rows2, w.detach(), b.detach(), A, B, H, W)
    (bbox2 * gb).sum().backward()
    x64b = x.double().requires_grad_(True)
    (F.conv2d(x64b, ref[2].detach(), ref[3].detach()) * gb.double()).sum().backward()
    assert rel(rows2.grad.view(B, H, W, K).permute(0, 3, 1, 2), x64b.grad) < 1e-4


@pytest.mark.parametrize("B,legs,H,W,K,A", [(4, 2, 38, 67, 512, 12), (2, 1, 13, 17, 64, 2), (6, 3, 9, 11, 96, 6)])
def test_rpn_losses_autograd_match_cross_entropy_and_smooth_l1_in_float64(dev, B, legs, H, W, K, A):
    """RpnHeadFn(logit_grads=True) + RpnLossFn (rpn.py:63-71, 86-105: the two RPN losses of all legs in one launch, the class loss's
    gradient taken with respect to the score logits) against the reference's graph in float64 -- conv2d -> view(B, 2, A*H, W) ->
    permute -> F.cross_entropy over the labelled anchors (nonzero() + index_select, rpn.py:90-97) and _smooth_l1_loss(sigma = 3,
    dim = [1, 2, 3]) -- per leg: loss values to 1e-5, the gradients with respect to the input rows, both weights and both biases
    to 1e-4 of each tensor's largest entry.  Run twice: bit-identical losses (deterministic reduction)."""
    from dtt.heads import RpnHeadFn, RpnLossFn, pack_rpn_heads_differentiable
    g = torch.Generator().manual_seed(B * 100 + A)
    cls, box = torch.nn.Conv2d(K, 2 * A, 1), torch.nn.Conv2d(K, 4 * A, 1)
    for c, sc in ((cls, 0.08), (box, 0.03)):
        c.weight.data = torch.randn(c.weight.shape, generator=g) * sc
        c.bias.data = torch.randn(c.bias.shape, generator=g) * 0.1
    cls, box = cls.to(dev), box.to(dev)
    x = torch.relu(torch.randn(B, K, H, W, generator=g)).to(dev)
    # anchor-target outputs: ~2 % labelled anchors, targets / weights as the layer lays them out
    labels = torch.full((B, 1, A * H, W), -1.0)
    r = torch.rand(labels.shape, generator=g)
    labels[r < 0.015] = 0.0
    labels[r < 0.005] = 1.0
    tgt = torch.randn(B, 4 * A, H, W, generator=g) * 0.5
    fg = (labels.view(B, A, H, W) == 1).repeat_interleave(4, dim=1)
    lab_any = (labels.view(B, A, H, W) >= 0).repeat_interleave(4, dim=1)
    w_in = fg.float()
    w_out = lab_any.float() / 256.0
    labels, tgt, w_in, w_out = (t.to(dev) for t in (labels, tgt, w_in, w_out))
    wsum = torch.tensor([1.0 + 0.5 * i for i in range(2 * legs)], device=dev)        # a different weight on every loss scalar

    def ours():
        for p in (cls.weight, cls.bias, box.weight, box.bias):
            p.grad = None
        rows = x.permute(0, 2, 3, 1).reshape(B * H * W, K).contiguous().requires_grad_(True)
        w, b, _ = pack_rpn_heads_differentiable(cls, box)
        prob, bbox = RpnHeadFn.apply(rows, w, b, A, B, H, W, True)
        losses = RpnLossFn.apply(prob, bbox, labels, tgt, w_in, w_out, legs, 3.0)
        (losses * wsum).sum().backward()
        return losses.detach().clone(), rows.grad.clone(), [p.grad.clone() for p in (cls.weight, cls.bias, box.weight, box.bias)]
    losses, gx, gp = ours()
    losses_b, gx_b, _ = ours()
    assert torch.equal(losses, losses_b) and torch.equal(gx, gx_b), "the reduction is not deterministic"
    x64 = x.double().requires_grad_(True)
    ref = [t.detach().double().requires_grad_(True) for t in (cls.weight, cls.bias, box.weight, box.bias)]
    per = B // legs
    total = 0.0
    want = torch.zeros(2 * legs, dtype=torch.float64)
    for leg in range(legs):
        sl = slice(leg * per, (leg + 1) * per)
        score = F.conv2d(x64[sl], ref[0], ref[1])
        score_r = score.view(per, 2, A * H, W).permute(0, 2, 3, 1).contiguous().view(-1, 2)
        lab = labels[sl].view(-1)
        keep = (lab != -1).nonzero().view(-1)
        l_cls = F.cross_entropy(score_r.index_select(0, keep), lab.index_select(0, keep).long())
        pred = F.conv2d(x64[sl], ref[2], ref[3])
        d = w_in[sl].double() * (pred - tgt[sl].double())
        ad = d.abs()
        quad = (ad < 1.0 / 9.0).double()
        l_box = (w_out[sl].double() * (d * d * 4.5 * quad + (ad - 0.5 / 9.0) * (1.0 - quad))).sum((1, 2, 3)).mean()
        want[leg], want[legs + leg] = float(l_cls), float(l_box)
        total = total + float(wsum[leg]) * l_cls + float(wsum[legs + leg]) * l_box
    total.backward()
    assert np.allclose(losses.cpu().double().numpy(), want.numpy(), rtol=1e-5, atol=1e-7), (losses, want)
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max().clamp_min(1e-30))
    assert rel(gx.view(B, H, W, K).permute(0, 3, 1, 2), x64.grad) < 1e-4, "dX"
    for name, a, r in zip(("cls dW", "cls dBias", "box dW", "box dBias"), gp, ref):
        assert rel(a, r.grad) < 1e-4, name


def test_rpn_class_loss_gradient_survives_an_underflowing_probability(dev):
    """Where the label's probability underflows in fp32 (a logit gap above ~88) the reference's cross_entropy on the logits still has
    the gradient p - y = -1 / count for that anchor (rpn.py:97); a loss taken as -log(p) through the softmax's adjoint multiplies by p
    and returns 0.  dtt_rpn_loss_backward writes the logit gradient itself: -1 / count on the label's score, +1 / count on the other."""
    from dtt import _lib
    from dtt._lib import check, ptr, stream_ptr
    L = _lib.lib()
    A, H, W = 2, 1, 4
    prob = torch.zeros(1, 2 * A, H, W, device=dev)
    prob[0, :A] = 1.0                                   # background certain, foreground probability exactly 0
    labels = torch.full((1, 1, A * H, W), -1.0, device=dev)
    labels.view(1, A, H, W)[0, 0, 0, 1] = 1.0           # a foreground label on an anchor whose p_fg = 0
    labels.view(1, A, H, W)[0, 1, 0, 2] = 0.0           # and an easy background anchor
    z = torch.zeros(1, 4 * A, H, W, device=dev)
    g_loss = torch.tensor([1.0, 0.0], device=dev)
    count = torch.tensor([2.0], device=dev)
    g_logits, g_bbox = torch.full_like(prob, float("nan")), torch.full_like(z, float("nan"))
    with torch.cuda.device(dev):
        check(L.dtt_rpn_loss_backward(ptr(prob), ptr(z), ptr(labels), ptr(z), ptr(z), ptr(z), ptr(g_loss), ptr(count), 1, 1, A, H * W, 3.0,
                                      ptr(g_logits), ptr(g_bbox), stream_ptr(dev)), "rpn_loss backward")
    gl = g_logits.cpu().view(2, A, H, W)
    assert float(gl[1, 0, 0, 1]) == -0.5 and float(gl[0, 0, 0, 1]) == 0.5          # (p - y) / count with p_fg = 0, p_bg = 1
    assert float(gl[0, 1, 0, 2]) == 0.0 and float(gl[1, 1, 0, 2]) == 0.0            # the easy anchor: p - y = 0
    gl[1, 0, 0, 1] = gl[0, 0, 0, 1] = 0.0
    assert float(gl.abs().max()) == 0.0 and float(g_bbox.abs().max()) == 0.0       # everything else written as zero


@pytest.mark.parametrize("B,H,W,C345", [(2, 38, 67, (128, 192, 256)), (1, 20, 31, (64, 64, 128))])
def test_tracking_rows_autograd_matches_the_reference_concat(dev, B, H, W, C345):
    """TrackingRowsFn + HeadGemmFn over it = corr_bbox_net(torch.cat([bbox_t, bbox_t+tau, corr3, corr4, corr5], 1)) of
    rfcn.py:166-174 under autograd: the tracking head's output and the gradients with respect to the box-delta columns, the
    three trunk maps (both legs) and the head's weight / bias, against the reference graph on NCHW tensors (library 1x1
    convolution in float64 over the NCHW correlation functions' outputs) to 1e-4 of each tensor's largest entry."""
    from dtt.heads import HeadGemmFn, TrackingRowsFn, pack_heads_differentiable, pm_to_nchw
    from dtt.ops import Correlation
    g = torch.Generator().manual_seed(B + H)
    od, G = 4, 7
    n_box = od * G * G
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    c3 = cl(torch.relu(torch.randn(2 * B, C345[0], 2 * H - 1, 2 * W, generator=g))).requires_grad_(True)
    c4 = cl(torch.relu(torch.randn(2 * B, C345[1], H, W, generator=g))).requires_grad_(True)
    c5 = cl(torch.relu(torch.randn(2 * B, C345[2], H, W, generator=g))).requires_grad_(True)
    loc_cols = (0.3 * torch.randn(2 * B * H * W, n_box, generator=g)).to(dev).requires_grad_(True)   # position-major order: bin * od + k
    layers = (Correlation(8, 1, 8, 2, 2), Correlation(8, 1, 8, 1, 1), Correlation(8, 1, 8, 1, 1))
    geoms = tuple((l.pad_size, l.kernel_size, l.max_displacement, l.stride1, l.stride2) for l in layers)
    K_in = 2 * n_box + 81 + 289 + 289
    conv = torch.nn.Conv2d(K_in, n_box, 1).to(dev)
    conv.weight.data.normal_(0, 0.02, generator=None)
    k_pad = -(-K_in // 32) * 32
    perm = torch.arange(K_in)
    bb, kk = torch.meshgrid(torch.arange(G * G), torch.arange(od), indexing="ij")
    for l in range(2):
        perm[l * n_box:(l + 1) * n_box] = (l * n_box + kk * G * G + bb).reshape(-1)
    rows = TrackingRowsFn.apply(loc_cols, c3, c4, c5, B, geoms, k_pad)
    w, b, heads, n_store, stride = pack_heads_differentiable([conv], k_pad=k_pad, in_perm=perm)
    out = HeadGemmFn.apply(rows, w, b, n_store, stride)
    got = pm_to_nchw(out, heads[0], B, H, W)
    gout = torch.randn(got.shape, generator=g).to(dev)
    (got * gout).sum().backward()
    # ---- the reference graph: NCHW box-delta maps in the reference's channel order k * G * G + bin, NCHW correlations, torch.cat
    n3, n4, n5 = (t.detach().contiguous().requires_grad_(True) for t in (c3, c4, c5))
    loc_ref = loc_cols.detach().clone().requires_grad_(True)
    bbox = loc_ref.view(2, B, H, W, G * G, od).permute(0, 1, 5, 4, 2, 3).reshape(2, B, n_box, H, W)
    feats = [bbox[0], bbox[1]] + [l(m[:B], m[B:]) for l, m in zip(layers, (n3, n4, n5))]
    x = torch.cat(feats, 1)
    wr, br = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    want = F.conv2d(x.double(), wr, br)
    (want * gout.double()).sum().backward()
    rel = lambda a, r: float((a.double() - r.double()).abs().max() / r.double().abs().max().clamp_min(1e-30))
    assert rel(got.detach(), want.detach()) < 1e-4
    assert rel(loc_cols.grad, loc_ref.grad) < 1e-4, "box-delta columns"
    for name, a, r in (("conv3", c3, n3), ("conv4", c4, n4), ("conv5", c5, n5)):
        assert a.grad.is_contiguous(memory_format=torch.channels_last)
        assert rel(a.grad, r.grad) < 1e-4, name
    assert rel(conv.weight.grad, wr.grad) < 1e-4 and rel(conv.bias.grad, br.grad) < 1e-4


@pytest.mark.parametrize("B,H,W,R", [(4, 38, 67, 1200), (2, 20, 30, 77), (1, 6, 9, 300)])
def test_psroi_pm_det_one_launch_matches_the_two_poolings(dev, B, H, W, R):
    """dtt_psroi_pm_det_forward (class scores + box deltas of a RoI pooled and voted in one launch, class softmax in the epilogue)
    against the two dtt_psroi_pm_forward launches + F.softmax it replaces: votes bit-identical (shared pooling code, same vote
    order), probabilities to 1e-6 (expf + a class-order sum against torch's softmax)."""
    from dtt.heads import PackedHeads, head_gemm, psroi_pm, psroi_pm_det
    convs = _convs(dev, 64, (31, 4), seed=R)
    g = torch.Generator().manual_seed(R)
    x = torch.relu(torch.randn(B, 64, H, W, generator=g)).to(dev)
    packed = PackedHeads(convs)
    pm = head_gemm(x.permute(0, 2, 3, 1).reshape(-1, 64).contiguous(), packed)
    rois = torch.from_numpy(_rois(np.random.RandomState(R), R, B, H, W)).to(dev)
    cls_head, loc_head = packed.heads
    prob, pred, score = psroi_pm_det(pm, cls_head, loc_head, B, H, W, rois, 1.0 / 16, want_scores=True)
    want_score = psroi_pm(pm, cls_head, B, H, W, rois, 1.0 / 16)
    want_pred = psroi_pm(pm, loc_head, B, H, W, rois, 1.0 / 16)
    assert torch.equal(score, want_score) and torch.equal(pred, want_pred)
    assert float((prob - F.softmax(want_score, dim=1)).abs().max()) < 1e-6
    assert float((prob.sum(1) - 1).abs().max()) < 1e-5
