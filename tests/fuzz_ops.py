#!/usr/bin/env python3
"""Test helper (also runnable on its own on the GPU box): randomized shape sweep of the hot-path ops against the oracle (GPU box).  Complements the fixed cases
of tests/test_gpu_ops.py: map sizes around tile / piece boundaries, channel counts around chunk boundaries, batch 1-3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np, torch
from oracle import oracle_lib as O
from dtt.ops import Correlation, _PSRoIPooling, nms


def run(N=150, seed=0):
    """Returns (correlation mismatches, psroi mismatches, nms mismatches) over N / N//3 / N//3 random cases."""
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(seed)
    bad = 0
    for it in range(N):
        B = rs.randint(1, 4); C = int(rs.choice([8, 16, 24, 40, 64, 12, 20])); H = rs.randint(2, 41); W = rs.randint(4, 70)
        d = int(rs.choice([4, 8, 8, 8, 3, 16])); s = int(rs.choice([1, 1, 1, 2])); pad = d if rs.rand() < 0.8 else d + 4 * rs.randint(0, 2)
        if (H + 2 * pad - 2 * d + s - 1) // s < 1 or (W + 2 * pad - 2 * d + s - 1) // s < 1:
            continue
        if it < int(os.environ.get("START", 0)) or it > int(os.environ.get("STOP", 10**9)):
            continue
        if os.environ.get("VERBOSE"):
            print("corr case", it, (B, C, H, W, pad, d, s), flush=True)
        x1 = np.maximum(rs.normal(size=(B, C, H, W)), 0).astype(np.float32)
        x2 = np.maximum(rs.normal(size=(B, C, H, W)), 0).astype(np.float32)
        ref = O.correlation_forward(x1, x2, pad, 1, d, s, s)
        t1 = torch.from_numpy(x1).to(dev).requires_grad_(True); t2 = torch.from_numpy(x2).to(dev).requires_grad_(True)
        out = Correlation(pad, 1, d, s, s)(t1, t2)
        torch.cuda.synchronize()
        err = float(np.abs(out.detach().cpu().numpy() - ref).max())
        go = rs.normal(size=ref.shape).astype(np.float32)
        e1 = e2 = 0.0
        if not os.environ.get("FWD_ONLY"):
            out.backward(torch.from_numpy(go).to(dev))
            torch.cuda.synchronize()
            g1, g2 = O.correlation_backward(go, x1, x2, pad, 1, d, s, s)
            e1 = float(np.abs(t1.grad.cpu().numpy() - g1).max()); e2 = float(np.abs(t2.grad.cpu().numpy() - g2).max())
        if err > 1e-5 or e1 > 1e-4 or e2 > 1e-4:
            bad += 1
            print("CORR MISMATCH", (B, C, H, W, pad, d, s), err, e1, e2, flush=True)
    print("correlation: %d cases, %d bad" % (N, bad), flush=True)
    badp = 0
    for it in range(N // 3):
        B = rs.randint(1, 4); od = rs.randint(1, 6); gs = int(rs.choice([3, 7])); H = rs.randint(3, 45); W = rs.randint(3, 70); R = rs.randint(1, 400)
        feat = rs.normal(size=(B, od * gs * gs, H, W)).astype(np.float32)
        x1 = rs.uniform(-30, W * 16, R); y1 = rs.uniform(-30, H * 16, R)
        rois = np.stack([rs.randint(0, B, R), x1, y1, x1 + rs.uniform(0, W * 12, R), y1 + rs.uniform(0, H * 12, R)], 1).astype(np.float32)
        ref, refmap = O.psroi_pool_forward(feat, rois, gs, gs, 1 / 16.0, gs, od)
        ft = torch.from_numpy(feat).to(dev).requires_grad_(True)
        out = _PSRoIPooling(gs, gs, 1 / 16.0, gs, od)(ft, torch.from_numpy(rois).to(dev))
        gop = rs.normal(size=ref.shape).astype(np.float32)
        out.backward(torch.from_numpy(gop).to(dev))
        gref = O.psroi_pool_backward(gop, rois, feat.shape, gs, gs, 1 / 16.0, gs, od, refmap)
        if not np.array_equal(out.detach().cpu().numpy(), ref) or float(np.abs(ft.grad.cpu().numpy() - gref).max()) > 1e-4:
            badp += 1
            print("PSROI MISMATCH", (B, od, gs, H, W, R), flush=True)
    print("psroi: %d cases, %d bad" % (N // 3, badp), flush=True)
    badn = 0
    for it in range(N // 3):
        n = rs.randint(1, 3000); thr = float(rs.choice([0.3, 0.5, 0.7]))
        c = rs.uniform(0, 300, size=(n, 2)); wh = rs.uniform(5, 150, size=(n, 2))
        dets = np.concatenate([c - wh / 2, c + wh / 2, np.sort(rs.uniform(0, 1, n))[::-1][:, None]], 1).astype(np.float32)
        if rs.rand() < 0.3:
            dets[:, :4] = np.round(dets[:, :4])            # integer boxes: exact-threshold IoUs
        keep = nms(torch.from_numpy(dets).to(dev), thr).view(-1).cpu().numpy()
        if not np.array_equal(keep, O.nms(dets, thr).reshape(-1)):
            badn += 1
            print("NMS MISMATCH", n, thr, flush=True)
    print("nms: %d cases, %d bad" % (N // 3, badn), flush=True)
    return bad, badp, badn


def run_more(N=60, seed=0):
    """Second sweep: proposal layer, anchor-target layer, RoI Align / Pool / Crop, per-class NMS, tube linking.
    Returns a dict op -> number of mismatching cases."""
    from oracle import rpn_oracle as ro
    from oracle import tubes_oracle as to
    from dtt.ops import RoIAlign, RoIAlignAvg, RoIPoolFunction, _RoICrop
    from dtt.postprocess import class_nms, to_all_boxes
    from dtt.rpn import anchor_target_forward, generate_anchors, proposal_forward
    from dtt.tubes import make_tubes
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(seed)
    bad = dict(proposal=0, anchor_target=0, roi_align=0, roi_pool=0, roi_crop=0, class_nms=0, tubes=0)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for it in range(N):
        # ---- proposal layer
        scales = (4, 8, 16, 32) if rs.rand() < 0.7 else (8, 16, 32)
        base = generate_anchors(scales=scales); A = base.shape[0]
        B = rs.randint(1, 4); H = rs.randint(1, 40); W = rs.randint(1, 70)
        pre = int(rs.choice([100, 300, 6000, 12000])); post = int(rs.choice([20, 300, 2000])); thr = float(rs.choice([0.5, 0.7]))
        logits = rs.normal(0, 2, size=(B, 2, A * H, W)).astype(np.float32)
        prob = torch.softmax(torch.from_numpy(logits), 1).view(B, 2 * A, H, W).numpy()
        if rs.rand() < 0.3:
            prob = (np.round(prob * 16) / 16).astype(np.float32)          # many exact score ties
        bbox = rs.normal(0, 0.5, size=(B, 4 * A, H, W)).astype(np.float32)
        info = np.tile(np.array([[H * 16.0, W * 16.0, 1.0]], dtype=np.float32), (B, 1)); info[-1, :2] -= rs.randint(0, 9)
        ref, nref = ro.proposal_layer(prob, bbox, info, base, 16, pre, post, thr, O.nms)
        rois, num = proposal_forward(cu(prob), cu(bbox), cu(info), torch.from_numpy(base).float(), 16, pre, post, thr)
        if not (np.array_equal(num.cpu().numpy(), nref) and np.array_equal(rois.cpu().numpy(), ref)):
            bad["proposal"] += 1; print("PROPOSAL MISMATCH", (B, A, H, W, pre, post, thr), flush=True)
        # ---- anchor-target layer
        G_ = rs.randint(1, 31)
        gt = np.zeros((B, G_, 5), np.float32)
        for b in range(B):
            k = rs.randint(1, G_ + 1)
            xy = rs.uniform(0, [W * 16 * 0.8 + 1, H * 16 * 0.8 + 1], size=(k, 2)); wh = rs.uniform(8, [W * 8 + 9, H * 8 + 9], size=(k, 2))
            gt[b, :k, :2] = xy; gt[b, :k, 2:4] = xy + wh; gt[b, :k, 4] = rs.randint(1, 31, k)
        np.random.seed(100 + it)
        try:
            aref = ro.anchor_target_layer(gt, info, base, H, W, 16)
        except ValueError:          # no anchor inside the image (tiny maps): the reference fails the same way
            aref = None
        if aref is not None:
            np.random.seed(100 + it)
            agot = anchor_target_forward(cu(gt), torch.from_numpy(info), torch.from_numpy(base).float(), H, W, 16)
            if not all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(agot, aref)):
                bad["anchor_target"] += 1; print("ANCHOR TARGET MISMATCH", (B, A, H, W, G_), flush=True)
        # ---- RoI ops
        C = rs.randint(1, 20); Hh = rs.randint(2, 40); Ww = rs.randint(2, 60); R = rs.randint(1, 200)
        feat = rs.normal(size=(B, C, Hh, Ww)).astype(np.float32)
        x1 = rs.uniform(-40, Ww * 16, R); y1 = rs.uniform(-40, Hh * 16, R)
        rr = np.stack([rs.randint(0, B, R), x1, y1, x1 + rs.uniform(-10, Ww * 14, R), y1 + rs.uniform(-10, Hh * 14, R)], 1).astype(np.float32)
        fa = cu(feat).requires_grad_(True)
        oa = RoIAlign(8, 8, 1 / 16.0)(fa, cu(rr))
        ga = rs.normal(size=tuple(oa.shape)).astype(np.float32)
        oa.backward(cu(ga))
        if not np.array_equal(oa.detach().cpu().numpy(), O.roi_align_forward(feat, rr, 8, 8, 1 / 16.0)) or \
                float(np.abs(fa.grad.cpu().numpy() - O.roi_align_backward(ga, rr, feat.shape, 8, 8, 1 / 16.0)).max()) > 1e-4:
            bad["roi_align"] += 1; print("ROI ALIGN MISMATCH", (B, C, Hh, Ww, R), flush=True)
        a7 = RoIAlignAvg(7, 7, 1 / 16.0)(cu(feat), cu(rr)).cpu()
        if not np.allclose(a7.numpy(), torch.nn.functional.avg_pool2d(torch.from_numpy(O.roi_align_forward(feat, rr, 8, 8, 1 / 16.0)), 2, 1).numpy(), rtol=1e-6, atol=1e-6):
            bad["roi_align"] += 1; print("ROI ALIGN AVG MISMATCH", (B, C, Hh, Ww, R), flush=True)
        pref, parg = O.roi_pool_forward(feat, rr, 7, 7, 1 / 16.0)
        fp = cu(feat).requires_grad_(True)
        pout, pa = RoIPoolFunction.apply(fp, cu(rr), 7, 7, 1 / 16.0)
        gp = rs.normal(size=pref.shape).astype(np.float32)
        pout.backward(cu(gp))
        if not (np.array_equal(pout.detach().cpu().numpy(), pref) and np.array_equal(pa.cpu().numpy(), parg)) or \
                float(np.abs(fp.grad.cpu().numpy() - O.roi_pool_backward(gp, rr, parg, feat.shape, 7, 7, 1 / 16.0)).max()) > 1e-4:
            bad["roi_pool"] += 1; print("ROI POOL MISMATCH", (B, C, Hh, Ww, R), flush=True)
        Gs = int(rs.choice([7, 14])); nro = rs.randint(1, 5)
        grid = rs.uniform(-1.4, 1.4, size=(B * nro, Gs, Gs, 2)).astype(np.float32)
        imgc = np.repeat(feat, nro, axis=0)                            # one image per grid, as the op expects
        fc = cu(imgc).requires_grad_(True)
        oc_ = _RoICrop()(fc, cu(grid))
        gc = rs.normal(size=tuple(oc_.shape)).astype(np.float32)
        oc_.backward(cu(gc))
        if not np.array_equal(oc_.detach().cpu().numpy(), O.roi_crop_forward(imgc, grid)) or \
                float(np.abs(fc.grad.cpu().numpy() - O.roi_crop_backward(imgc, grid, gc)).max()) > 1e-4:
            bad["roi_crop"] += 1; print("ROI CROP MISMATCH", (B, C, Hh, Ww, nro, Gs), flush=True)
        # ---- per-class NMS
        Rn = rs.randint(1, 400); ncls = rs.randint(2, 32); agn = bool(rs.rand() < 0.6); mpi = int(rs.choice([0, 5, 100]))
        sc = rs.dirichlet(np.ones(ncls) * 0.3, size=(B, Rn)).astype(np.float32)
        ctr = rs.uniform(50, 400, size=(B, Rn, 2)); whh = rs.uniform(20, 200, size=(B, Rn, 2))
        bb = np.concatenate([ctr - whh / 2, ctr + whh / 2], 2).astype(np.float32)
        boxes = bb if agn else (bb[:, :, None, :] + rs.normal(0, 3, size=(B, Rn, ncls, 4))).reshape(B, Rn, 4 * ncls).astype(np.float32)
        dets, counts = class_nms(cu(sc), cu(boxes), 0.05, 0.3, mpi, agn)
        got = to_all_boxes(dets, counts)
        okc = True
        for i in range(B):
            refc = ro.class_nms(sc[i], boxes[i], O.nms, 0.05, 0.3, mpi, agn)
            okc &= all(np.array_equal(got[i][j], refc[j]) for j in range(ncls))
        if not okc:
            bad["class_nms"] += 1; print("CLASS NMS MISMATCH", (B, Rn, ncls, agn, mpi), flush=True)
        # ---- tube linking
        F_ = rs.randint(2, 25); M = int(rs.choice([0, 5, 60]))
        dets_l, trk_l = [], []
        cen = rs.uniform(40, 500, size=(60, 2)); siz = rs.uniform(30, 140, size=(60, 2))
        for f in range(F_):
            k = rs.randint(1, 40)
            c = cen[:k] + 3.0 * f + rs.normal(0, 3, size=(k, 2)); wh2 = siz[:k] + rs.normal(0, 3, size=(k, 2))
            b2 = np.concatenate([c - wh2 / 2, c + wh2 / 2], 1); s2 = np.sort(rs.uniform(0.02, 1, k))[::-1]
            dets_l.append(np.concatenate([b2, s2[:, None], 1 - s2[:, None]], 1).astype(np.float32))
            if M and rs.rand() < 0.85:
                pk = rs.randint(0, k, M); t0 = np.concatenate([c[pk] - wh2[pk] / 2, c[pk] + wh2[pk] / 2], 1) + rs.normal(0, 4, size=(M, 4))
                trk_l.append((t0.astype(np.float32), (t0 + 3).astype(np.float32)))
            else:
                trk_l.append(None)
        nmx = max(len(d) for d in dets_l)
        Dp = np.zeros((F_, nmx, 6), np.float32); npd = np.array([len(d) for d in dets_l], np.int32)
        Tp = np.zeros((F_, 2, max(M, 1), 4), np.float32); mp = np.full(F_, -1, np.int32)
        for f in range(F_):
            Dp[f, :npd[f]] = dets_l[f]
            if trk_l[f] is not None:
                Tp[f, 0], Tp[f, 1] = trk_l[f]; mp[f] = M
        want = to.make_tubes(Dp, npd, Tp if M else None, mp)
        gott = make_tubes([cu(d) for d in dets_l], None if not M else [None if t is None else (cu(t[0]), cu(t[1])) for t in trk_l])
        if not (np.array_equal(gott["idx"].cpu().numpy(), want["idx"]) and np.array_equal(gott["boxes"].cpu().numpy(), want["boxes"])):
            bad["tubes"] += 1; print("TUBES MISMATCH", (F_, M), flush=True)
    print("second sweep (%d rounds):" % N, bad, flush=True)
    return bad




def run_ref(N=60, seed=0):
    """Random shapes, CPU oracle against the REFERENCE's own kernels (oracle/_ref, see oracle/build_ref.sh): forwards, index
    maps and keep lists bit for bit.  Returns a dict of mismatch counts per op; {} when oracle/_ref is not built."""
    from oracle import ref_kernels as RK
    if not RK.available():
        return {}
    rs = np.random.RandomState(seed)
    bad = dict(correlation=0, psroi=0, roi_align=0, roi_pool=0, roi_crop=0, nms=0)
    for it in range(N):
        B = rs.randint(1, 3); C = int(rs.choice([1, 7, 16, 33, 64, 100])); H = rs.randint(2, 30); W = rs.randint(2, 40)
        d = int(rs.choice([1, 2, 4, 8])); s2 = int(rs.choice([1, 1, 2])); s1 = int(rs.choice([1, 1, 2]))
        k = int(rs.choice([1, 1, 1, 3])); pad = d + (k - 1) // 2 if rs.rand() < 0.7 else rs.randint(0, d + 3)
        try:
            O.correlation_output_shape(C, H, W, pad, k, d, s1, s2)
        except ValueError:
            continue
        x1 = rs.normal(size=(B, C, H, W)).astype(np.float32); x2 = rs.normal(size=(B, C, H, W)).astype(np.float32)
        if not np.array_equal(O.correlation_forward(x1, x2, pad, k, d, s1, s2), RK.correlation_forward(x1, x2, pad, k, d, s1, s2)):
            bad["correlation"] += 1
            print("REF CORR MISMATCH", (B, C, H, W, pad, k, d, s1, s2), flush=True)
    for it in range(N // 2):
        B = rs.randint(1, 4); od = rs.randint(1, 6); gs = int(rs.choice([1, 3, 7])); H = rs.randint(3, 45); W = rs.randint(3, 70)
        R = rs.randint(1, 200); scale = float(rs.choice([1 / 16.0, 1 / 8.0, 0.1]))
        x1 = rs.uniform(-30, W / scale, R); y1 = rs.uniform(-30, H / scale, R)
        rois = np.stack([rs.randint(0, B, R), x1, y1, x1 + rs.uniform(0, W / scale * 0.8, R), y1 + rs.uniform(0, H / scale * 0.8, R)],
                        1).astype(np.float32)
        if rs.rand() < 0.3:
            rois[:, 1:] = np.round(rois[:, 1:])
        feat = rs.normal(size=(B, od * gs * gs, H, W)).astype(np.float32)
        a, am = O.psroi_pool_forward(feat, rois, gs, gs, scale, gs, od); b, bm = RK.psroi_pool_forward(feat, rois, gs, gs, scale, gs, od)
        if not (np.array_equal(a, b) and np.array_equal(am, bm)):
            bad["psroi"] += 1; print("REF PSROI MISMATCH", (B, od, gs, H, W, R, scale), flush=True)
        C = rs.randint(1, 9); feat = rs.normal(size=(B, C, H, W)).astype(np.float32); p = int(rs.choice([2, 7, 14]))
        if not np.array_equal(O.roi_align_forward(feat, rois, p, p, scale), RK.roi_align_forward(feat, rois, p, p, scale)):
            bad["roi_align"] += 1; print("REF ROI ALIGN MISMATCH", (B, C, H, W, R, p, scale), flush=True)
        a, aa = O.roi_pool_forward(feat, rois, p, p, scale); b, ba = RK.roi_pool_forward(feat, rois, p, p, scale)
        if not (np.array_equal(a, b) and np.array_equal(aa, ba)):
            bad["roi_pool"] += 1; print("REF ROI POOL MISMATCH", (B, C, H, W, R, p, scale), flush=True)
        grid = rs.uniform(-1.4, 1.4, size=(B * rs.randint(1, 5), p, p, 2)).astype(np.float32)
        if not np.array_equal(O.roi_crop_forward(feat, grid), RK.roi_crop_forward(feat, grid)):
            bad["roi_crop"] += 1; print("REF ROI CROP MISMATCH", (B, C, H, W, grid.shape), flush=True)
    for it in range(N // 2):
        n = rs.randint(1, 2500); thr = float(rs.choice([0.3, 0.5, 0.7]))
        c = rs.uniform(0, 300, size=(n, 2)); wh = rs.uniform(5, 150, size=(n, 2))
        dets = np.concatenate([c - wh / 2, c + wh / 2, np.sort(rs.uniform(0, 1, n))[::-1][:, None]], 1).astype(np.float32)
        if rs.rand() < 0.4:
            dets[:, :4] = np.round(dets[:, :4])
        if not np.array_equal(O.nms(dets, thr).reshape(-1), RK.nms(dets, thr)):
            bad["nms"] += 1; print("REF NMS MISMATCH", n, thr, flush=True)
    print("oracle vs reference kernels:", bad, flush=True)
    return bad


def run_tail(N=60, seed=0):
    """Round-2 kernels on random shapes: channels-last correlation (both output layouts) against the oracle, the head GEMM
    against a float64 product, position-major PSRoI pooling + vote against the oracle pooling of the same scores (bit for bit),
    the two-phase proposal layer against the one-call layer.  Returns {name: mismatches}."""
    from dtt.heads import PackedHeads, head_gemm, pm_to_nchw, psroi_pm
    from dtt.ops import correlation_forward_nhwc, correlation_output_shape
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(seed)
    bad = {"corr_nhwc": 0, "head_gemm": 0, "psroi_pm": 0}
    for it in range(N):
        B = rs.randint(1, 4); C = 16 * rs.randint(1, 9); s = int(rs.choice([1, 1, 1, 2]))
        R = int(rs.choice([1, 2, 3, 4, 5, 6, 7, 8, 8, 8, 9, 10, 11, 12, 13, 14, 15, 16])); d = R * s   # any radius up to 16 (window-split kernel)
        H = rs.randint(1, 45); W = rs.randint(1, 75)
        pad = d if rs.rand() < 0.7 else d + s * int(rs.choice([-1, 1, 2])) * (1 if R > 1 else 0)
        if pad < 0 or (H + 2 * pad - 2 * d + s - 1) // s < 1 or (W + 2 * pad - 2 * d + s - 1) // s < 1:
            continue
        x1 = np.maximum(rs.normal(size=(B, C, H, W)), 0).astype(np.float32)
        x2 = np.maximum(rs.normal(size=(B, C, H, W)), 0).astype(np.float32)
        ref = O.correlation_forward(x1, x2, pad, 1, d, s, s)
        t1 = torch.from_numpy(x1).to(dev).contiguous(memory_format=torch.channels_last)
        t2 = torch.from_numpy(x2).to(dev).contiguous(memory_format=torch.channels_last)
        out = correlation_forward_nhwc(t1, t2, pad, 1, d, s, s)
        budget = int(rs.choice([1, 7, 64, 240]))                       # another plan: the result may not depend on the partition
        same_plan = torch.equal(correlation_forward_nhwc(t1, t2, pad, 1, d, s, s, max_workgroups=budget), out)
        oc, oh, ow = correlation_output_shape(C, H, W, pad, 1, d, s, s)
        rows = torch.full((B * oh * ow, oc + 12), -3.0, device=dev)
        correlation_forward_nhwc(t1, t2, pad, 1, d, s, s, rows=rows, col=8)
        torch.cuda.synchronize()
        e = float(np.abs(out.cpu().numpy() - ref).max())
        r = rows.cpu().numpy()
        same = np.array_equal(r[:, 8:8 + oc].reshape(B, oh, ow, oc).transpose(0, 3, 1, 2), out.cpu().numpy())
        untouched = bool((r[:, :8] == -3).all() and (r[:, 8 + oc:] == -3).all())
        if e > 1e-5 or not same or not untouched or not same_plan:
            bad["corr_nhwc"] += 1
            print("CORR NHWC MISMATCH", (B, C, H, W, pad, d, s), e, same, untouched, flush=True)
    print("corr_nhwc: %d cases, %d bad" % (N, bad["corr_nhwc"]), flush=True)
    for it in range(N // 2):
        B = rs.randint(1, 4); H = rs.randint(2, 40); W = rs.randint(2, 70); K = 32 * rs.randint(1, 9)
        ods = [(31, 4), (31,), (4,), (21, 4), (3,), (17, 2), (32, 1), (5,), (9, 4)][rs.randint(0, 9)]
        torch.manual_seed(int(rs.randint(1 << 30)))
        convs = [torch.nn.Conv2d(K, od * 49, 1).to(dev) for od in ods]
        packed = PackedHeads(convs)
        x = torch.from_numpy(np.maximum(rs.normal(size=(B, K, H, W)), 0).astype(np.float32)).to(dev)
        pm = head_gemm(x.permute(0, 2, 3, 1).reshape(-1, K).contiguous(), packed)
        R = rs.randint(1, 500)
        x1 = rs.uniform(-40, W * 16, R); y1 = rs.uniform(-40, H * 16, R)
        rois = np.stack([rs.randint(0, B, R), x1, y1, x1 + rs.uniform(0, W * 14, R), y1 + rs.uniform(0, H * 14, R)], 1).astype(np.float32)
        if rs.rand() < 0.3:
            rois[:, 1:] = np.round(rois[:, 1:] / 16) * 16      # bin edges on pixel boundaries
        rt = torch.from_numpy(rois).to(dev)
        for conv, head in zip(convs, packed.heads):
            maps = pm_to_nchw(pm, head, B, H, W)
            want = torch.nn.functional.conv2d(x.double(), conv.weight.detach().double(), conv.bias.detach().double())
            rel = float((maps.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))
            if rel > 1e-5:
                bad["head_gemm"] += 1
                print("HEAD GEMM MISMATCH", (B, H, W, K, ods), rel, flush=True)
            vote, pooled = psroi_pm(pm, head, B, H, W, rt, 1 / 16.0, want_pooled=True)
            ref_p, _ = O.psroi_pool_forward(maps.cpu().numpy(), rois, 7, 7, 1 / 16.0, 7, head["od"])
            ref_v = ref_p.reshape(R, head["od"], 49)
            acc = np.zeros((R, head["od"]), np.float32)
            for k in range(49):                                  # the reference's AvgPool2d order: row-major sum, then / 49
                acc = (acc + ref_v[:, :, k]).astype(np.float32)
            ref_vote = (acc / np.float32(49)).astype(np.float32)
            if not np.array_equal(pooled.cpu().numpy(), ref_p) or not np.array_equal(vote.cpu().numpy(), ref_vote):
                bad["psroi_pm"] += 1
                print("PSROI PM MISMATCH", (B, H, W, K, ods, R), flush=True)
    print("head_gemm / psroi_pm: %d cases, %d / %d bad" % (N // 2, bad["head_gemm"], bad["psroi_pm"]), flush=True)
    return bad


if __name__ == "__main__":
    b = run(int(os.environ.get("N", 150)), int(os.environ.get("SEED", 0)))
    m = run_more(int(os.environ.get("N2", 60)), int(os.environ.get("SEED", 0)))
    t = run_tail(int(os.environ.get("N3", 80)), int(os.environ.get("SEED", 0)))
    sys.exit(1 if (any(b) or any(m.values()) or any(t.values())) else 0)
