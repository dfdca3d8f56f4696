"""R-FCN 1x1 heads + position-sensitive pooling in the position-major layout (csrc/heads.hip).

The reference runs `RFCN_cls_net` / `RFCN_bbox_net` / `corr_bbox_net` as cuDNN 1x1 convolutions into NCHW score maps
whose channel is (ctop*7 + ph)*7 + pw (faster_rcnn/rfcn.py:49-53, resnet.py:311-312) and pools them with
`_PSRoIPooling` + `AvgPool2d((7,7))` (rfcn.py:40-43, 62-64, 133-140).  Here the same arithmetic is laid out for the
hardware: the heads are one exact-fp32 MFMA GEMM over the channels-last trunk output (`dtt_head_gemm`) that writes, per
pixel, the 49 bins one after the other with the classes of a bin contiguous (padded to `cp`), and the pooling kernel
(`dtt_psroi_pm_forward`) reads one aligned run of classes per bin element.  Weights are only permuted / zero padded
(`pack_heads`); the parameters, the state_dict and the checkpoint layout are untouched.
"""
import torch

from . import _lib
from ._lib import check, ptr, require_f32_contig, require_gpu, stream_ptr


def _pow2_at_least(n):
    p = 1
    while p < n:
        p *= 2
    return p


class PackedHeads:
    """Row-permuted copy of one or more 1x1 head convolutions that share their input.

    heads[i] = dict(offset=first float of the head inside a pixel row, cp=classes-per-bin padding, od=output_dim,
    group=G); `w` is (rows16, K), `bias` (rows16,), `n_store` the floats of a pixel row that the GEMM writes,
    `stride` the floats between pixels (a multiple of 32: every bin run of the first head starts on a 128-byte line)."""

    def __init__(self, convs, group=7, k_pad=None, in_perm=None):
        """in_perm (optional, length K): input channel j of the packed heads is input channel in_perm[j] of the
        convolutions (callers whose activation rows are not in the reference's channel order)."""
        dev = convs[0].weight.device
        K = convs[0].weight.shape[1]
        Kp = K if k_pad is None else int(k_pad)
        rows, biases, self.heads = [], [], []
        off = 0
        for conv in convs:
            w = conv.weight.detach().reshape(conv.weight.shape[0], -1).float()
            if in_perm is not None:
                w = w[:, in_perm.to(dev)]
            assert w.shape[1] == K, "heads must share their input"
            od = w.shape[0] // (group * group)
            assert od * group * group == w.shape[0]
            cp = 4 if od <= 4 else _pow2_at_least(max(od, 32))   # the pooling kernel's instantiations: 4 or 32 classes per bin
            b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=dev)
            # emitted row bin*cp + c  <-  reference channel c*G*G + bin
            wp = torch.zeros(group * group, cp, Kp, device=dev)
            wp[:, :od, :K] = w.view(od, group * group, K).permute(1, 0, 2)
            bp = torch.zeros(group * group, cp, device=dev)
            bp[:, :od] = b.view(od, group * group).t()
            rows.append(wp.reshape(-1, Kp))
            biases.append(bp.reshape(-1))
            self.heads.append(dict(offset=off, cp=cp, od=od, group=group))
            off += group * group * cp
        self.n_store = off
        n16 = -(-off // 16) * 16
        w = torch.cat(rows, 0)
        b = torch.cat(biases, 0)
        if n16 > off:
            w = torch.cat([w, torch.zeros(n16 - off, Kp, device=dev)], 0)
            b = torch.cat([b, torch.zeros(n16 - off, device=dev)], 0)
        self.w = w.contiguous()
        self.bias = b.contiguous()
        self.K = Kp
        self.stride = -(-off // 32) * 32


def head_gemm(x_rows, packed, out=None, passes=0):
    """x_rows (M, K) fp32, row-major (channels-last pixels) -> out (M, packed.stride); columns < packed.n_store are
    written: out[m, n] = sum_k x[m, k] * w[n, k] + bias[n]."""
    require_gpu(x_rows)
    require_f32_contig("x_rows", x_rows)
    M, K = x_rows.shape
    if K != packed.K:
        raise ValueError("head_gemm: input has %d channels, the packed heads expect %d" % (K, packed.K))
    if out is None:
        out = torch.empty((M, packed.stride), dtype=torch.float32, device=x_rows.device)
    with torch.cuda.device(x_rows.device):
        check(_lib.lib().dtt_head_gemm(ptr(x_rows), K, M, K, ptr(packed.w), ptr(packed.bias), packed.w.shape[0], ptr(out),
                                       out.stride(0), packed.n_store, int(passes), stream_ptr(x_rows.device)), "head_gemm")
    return out


class PackedRPNHeads:
    """`RPN_cls_score` + `RPN_bbox_pred` (rpn.py:63-71) as ONE weight matrix for `dtt_rpn_head_gemm`: rows in the order
    [bg_0, fg_0, bg_1, fg_1, ..., box deltas 0 .. 4A-1, zero rows up to a multiple of 16] -- a lane of the GEMM's accumulator
    then holds whole (background, foreground) pairs and the reference's reshape(2) -> softmax -> reshape(2A) happens in
    registers.  Weights are only permuted; parameters, state_dict and checkpoint layout are untouched."""

    def __init__(self, cls_conv, bbox_conv):
        dev = cls_conv.weight.device
        wc = cls_conv.weight.detach().reshape(cls_conv.weight.shape[0], -1).float()
        wb = bbox_conv.weight.detach().reshape(bbox_conv.weight.shape[0], -1).float()
        self.A = wc.shape[0] // 2
        assert wc.shape[0] == 2 * self.A and wb.shape[0] == 4 * self.A and wc.shape[1] == wb.shape[1]
        if self.A % 2:
            raise ValueError("PackedRPNHeads: an even number of anchors is required (got %d)" % self.A)
        pair = torch.stack([torch.arange(self.A), self.A + torch.arange(self.A)], 1).reshape(-1).to(dev)   # bg_a, fg_a
        bc = cls_conv.bias.detach().float() if cls_conv.bias is not None else torch.zeros(2 * self.A, device=dev)
        bb = bbox_conv.bias.detach().float() if bbox_conv.bias is not None else torch.zeros(4 * self.A, device=dev)
        w = torch.cat([wc[pair], wb], 0)
        b = torch.cat([bc[pair], bb], 0)
        n16 = -(-w.shape[0] // 16) * 16
        if n16 > w.shape[0]:
            b = torch.cat([b, torch.zeros(n16 - w.shape[0], device=dev)], 0)
            w = torch.cat([w, torch.zeros(n16 - w.shape[0], w.shape[1], device=dev)], 0)
        self.w, self.bias, self.K = w.contiguous(), b.contiguous(), w.shape[1]


def rpn_head_gemm(x_rows, packed, batch, height, width):
    """x_rows (batch*height*width, K) channels-last rows of relu(RPN_Conv(.)) -> (rpn_cls_prob (batch, 2A, H, W),
    rpn_bbox_pred (batch, 4A, H, W)): both 1x1 heads and the pairwise softmax in one launch (`dtt_rpn_head_gemm`)."""
    require_gpu(x_rows)
    require_f32_contig("x_rows", x_rows)
    M, K = x_rows.shape
    if K != packed.K or M != batch * height * width:
        raise ValueError("rpn_head_gemm: rows %s do not match batch %d x %d x %d, K %d" % (tuple(x_rows.shape), batch, height, width, packed.K))
    prob = torch.empty((batch, 2 * packed.A, height, width), dtype=torch.float32, device=x_rows.device)
    bbox = torch.empty((batch, 4 * packed.A, height, width), dtype=torch.float32, device=x_rows.device)
    with torch.cuda.device(x_rows.device):
        check(_lib.lib().dtt_rpn_head_gemm(ptr(x_rows), K, batch, height * width, K, ptr(packed.w), ptr(packed.bias),
                                           packed.w.shape[0], packed.A, ptr(prob), ptr(bbox), stream_ptr(x_rows.device)),
              "rpn_head_gemm")
    return prob, bbox


def gather_column_blocks(dst, dst_col, src, src_col, rows, n_blocks, ncols):
    """dst[r, dst_col + k*ncols + c] = src[k*rows + r, src_col + c] (`dtt_gather_column_blocks`): the box-delta columns of the
    two legs side by side in the tracking head's input rows (rfcn.py:133-140's torch.cat on position-major rows)."""
    import ctypes
    require_gpu(dst, src)
    assert dst.dtype == src.dtype == torch.float32 and dst.stride(1) == 1 and src.stride(1) == 1
    if dst.shape[0] < rows or src.shape[0] < n_blocks * rows or dst_col + n_blocks * ncols > dst.shape[1] or src_col + ncols > src.shape[1]:
        raise ValueError("gather_column_blocks: block does not fit (dst %s, src %s)" % (tuple(dst.shape), tuple(src.shape)))
    with torch.cuda.device(dst.device):
        check(_lib.lib().dtt_gather_column_blocks(ctypes.c_void_p(dst.data_ptr() + 4 * dst_col), dst.stride(0),
                                                  ctypes.c_void_p(src.data_ptr() + 4 * src_col), src.stride(0), rows, n_blocks,
                                                  rows, ncols, stream_ptr(dst.device)), "gather_column_blocks")
    return dst


def psroi_pm(pm_map, head, batch, height, width, rois, spatial_scale, want_pooled=False):
    """Position-sensitive pooling + 7x7 vote over a position-major map.

    pm_map (batch*height*width, stride) as written by `head_gemm`; head = one entry of PackedHeads.heads; rois (R, 5).
    Returns vote (R, od) [and pooled (R, od, G, G) in the reference layout when want_pooled]."""
    require_gpu(pm_map, rois)
    require_f32_contig("rois", rois)
    if rois.dim() != 2 or rois.size(1) != 5:
        raise ValueError("rois must have shape (R, 5) [batch_idx, x1, y1, x2, y2], got %s" % (tuple(rois.shape),))
    assert pm_map.dtype == torch.float32 and pm_map.stride(1) == 1 and pm_map.shape[0] == batch * height * width
    R, od, G = rois.size(0), head["od"], head["group"]
    vote = torch.empty((R, od), dtype=torch.float32, device=pm_map.device)
    pooled = torch.empty((R, od, G, G), dtype=torch.float32, device=pm_map.device) if want_pooled else None
    base = pm_map.data_ptr() + 4 * head["offset"]
    import ctypes
    with torch.cuda.device(pm_map.device):
        check(_lib.lib().dtt_psroi_pm_forward(ctypes.c_void_p(base), pm_map.stride(0), head["cp"], batch, R, height, width, G,
                                              ptr(rois), float(spatial_scale), od, ptr(vote), ptr(pooled),
                                              stream_ptr(pm_map.device)), "psroi_pm")
    return (vote, pooled) if want_pooled else vote


def psroi_pm_det(pm_map, cls_head, loc_head, batch, height, width, rois, spatial_scale, want_scores=False):
    """The detection pooling of rfcn.py:133-140 in ONE launch (`dtt_psroi_pm_det_forward`): class scores and box deltas of every RoI
    pooled + voted over the same position-major map, the class softmax folded into the kernel's epilogue.
    Returns (cls_prob (R, n_cls), bbox_pred (R, 4)[, cls_score (R, n_cls)]).  Needs the layout `PackedHeads` produces for the
    (class, box) pair: 32 class slots per bin from float 0, 4 box deltas per bin behind them."""
    import ctypes
    require_gpu(pm_map, rois)
    require_f32_contig("rois", rois)
    if rois.dim() != 2 or rois.size(1) != 5:
        raise ValueError("rois must have shape (R, 5) [batch_idx, x1, y1, x2, y2], got %s" % (tuple(rois.shape),))
    if cls_head["cp"] != 32 or loc_head["cp"] != 4 or cls_head["offset"] != 0 or cls_head["group"] != loc_head["group"]:
        raise ValueError("psroi_pm_det: needs a 32-slot class head at float 0 and a 4-slot box head of the same group size")
    assert pm_map.dtype == torch.float32 and pm_map.stride(1) == 1 and pm_map.shape[0] == batch * height * width
    R, G = rois.size(0), cls_head["group"]
    dev = pm_map.device
    prob = torch.empty((R, cls_head["od"]), dtype=torch.float32, device=dev)
    pred = torch.empty((R, loc_head["od"]), dtype=torch.float32, device=dev)
    score = torch.empty((R, cls_head["od"]), dtype=torch.float32, device=dev) if want_scores else None
    with torch.cuda.device(dev):
        check(_lib.lib().dtt_psroi_pm_det_forward(ptr(pm_map), pm_map.stride(0), loc_head["offset"], batch, R, height, width, G, ptr(rois),
                                                  float(spatial_scale), cls_head["od"], loc_head["od"], ptr(score), ptr(prob), ptr(pred),
                                                  stream_ptr(dev)), "psroi_pm_det")
    return (prob, pred, score) if want_scores else (prob, pred)


# ------------------------------------------------------------------------------------------------ training (autograd)
def pack_heads_differentiable(convs, group=7, k_pad=None, in_perm=None):
    """PackedHeads' row permutation as differentiable tensor ops on the LIVE parameters (training: the weights change every
    step, and their gradients must flow back through the permutation): returns (w (rows16, K), bias (rows16,), heads, n_store,
    stride) with the same meaning as PackedHeads' fields.  in_perm / k_pad as in PackedHeads (input channels permuted, then
    zero-padded to k_pad)."""
    dev = convs[0].weight.device
    K0 = convs[0].weight.shape[1]
    K = K0 if k_pad is None else int(k_pad)
    rows, biases, heads, off = [], [], [], 0
    for conv in convs:
        w = conv.weight.reshape(conv.weight.shape[0], K0)
        if in_perm is not None:
            w = w[:, in_perm.to(dev)]
        if K > K0:
            w = torch.cat([w, w.new_zeros(w.shape[0], K - K0)], 1)
        od = w.shape[0] // (group * group)
        cp = 4 if od <= 4 else _pow2_at_least(max(od, 32))
        b = conv.bias if conv.bias is not None else torch.zeros(w.shape[0], device=dev)
        wp = w.view(od, group * group, K).permute(1, 0, 2)                       # (bin, class, K)
        bp = b.view(od, group * group).t()
        if cp > od:
            wp = torch.cat([wp, wp.new_zeros(group * group, cp - od, K)], 1)
            bp = torch.cat([bp, bp.new_zeros(group * group, cp - od)], 1)
        rows.append(wp.reshape(-1, K))
        biases.append(bp.reshape(-1))
        heads.append(dict(offset=off, cp=cp, od=od, group=group))
        off += group * group * cp
    n16 = -(-off // 16) * 16
    w, b = torch.cat(rows, 0), torch.cat(biases, 0)
    if n16 > off:
        w = torch.cat([w, w.new_zeros(n16 - off, K)], 0)
        b = torch.cat([b, b.new_zeros(n16 - off)], 0)
    return w.contiguous(), b.contiguous(), heads, off, -(-off // 32) * 32


class HeadGemmFn(torch.autograd.Function):
    """out = x_rows @ w.T + bias through `dtt_head_gemm`, with a backward on the same kernel (closes SURVEY 8 row A9 for the
    training graph, rfcn.py:49-53): dX = gOut @ w is the head GEMM over the gradient rows with the transposed weights as
    its "weight" operand; dW = gOut.T @ x is `dtt_head_gemm_dw` (both operands read as they lie, pixel rows split over
    workgroups, partial tiles added in a fixed order: no atomics); dBias is a column sum.
    The gradient rows must have finite values in their padding columns (they are multiplied by zero weights): the
    position-major PSRoI backward writes whole rows."""

    @staticmethod
    def forward(ctx, x_rows, w, bias, n_store, stride):
        require_gpu(x_rows)
        x_rows = x_rows.contiguous()
        M, K = x_rows.shape
        out = torch.empty((M, stride), dtype=torch.float32, device=x_rows.device)   # columns >= n_store are never read
        with torch.cuda.device(x_rows.device):
            check(_lib.lib().dtt_head_gemm(ptr(x_rows), K, M, K, ptr(w), ptr(bias), w.shape[0], ptr(out), stride, n_store, 0,
                                           stream_ptr(x_rows.device)), "head_gemm")
        ctx.save_for_backward(x_rows, w)
        ctx.n_store = n_store
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x_rows, w = ctx.saved_tensors
        gout = gout.contiguous()
        M, K = x_rows.shape
        N16, stride = w.shape[0], gout.shape[1]
        dev = x_rows.device
        L = _lib.lib()
        gx = gw = gb = None
        zeros = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[0]:
                # dX (M, K) = gout (M, stride) @ Wt.T with Wt (K, stride) = w.T zero-padded: the GEMM's K is the stride (a multiple of 32)
                wt = torch.zeros((K, stride), dtype=torch.float32, device=dev)
                wt[:, :N16] = w.t()
                gx = torch.empty((M, K), dtype=torch.float32, device=dev)
                check(L.dtt_head_gemm(ptr(gout), stride, M, stride, ptr(wt), ptr(zeros(K)), K, ptr(gx), K, K, 0, stream_ptr(dev)),
                      "head_gemm dX")
            if ctx.needs_input_grad[1]:
                # dW (N16, K) = gout[:, :N16].T @ x: its own kernel (dtt_head_gemm_dw) -- an MFMA step wants 16 outputs x 4 pixels
                # of gout and 4 pixels x 16 inputs of x, i.e. 64-byte runs of the rows as they lie: nothing is transposed
                gw = torch.empty((N16, K), dtype=torch.float32, device=dev)
                nb = L.dtt_head_gemm_dw_workspace_bytes(M, N16, K)
                ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
                check(L.dtt_head_gemm_dw(ptr(gout), stride, stride, ptr(x_rows), K, M, N16, K, ptr(gw), ptr(ws), nb, stream_ptr(dev)),
                      "head_gemm dW")
            if ctx.needs_input_grad[2]:
                gb = gout[:, :N16].sum(0)
        return gx, gw, gb, None, None


class PsroiPmFn(torch.autograd.Function):
    """Position-sensitive pooling + vote of SEVERAL heads of one position-major map in one autograd node (`dtt_psroi_pm_forward`
    per head; backward `dtt_psroi_pm_backward` per head into ONE gradient map, whose remaining columns are zeroed): the
    map-stationary backward needs no atomics and no pre-zeroed output (psroi_pooling_kernel.cu:109-170 scatters with atomicAdd)."""

    @staticmethod
    def forward(ctx, pm_map, rois, batch, height, width, spatial_scale, heads, extract=None):
        """extract = (first column, columns): additionally returns a compact copy of those columns of the map (the box deltas the
        tracking branch concatenates, rfcn.py:166-169) as one more output -- its gradient is ADDED into the one gradient map this
        node hands back, so the map has a single consumer under autograd (two consumers = two dense 73 MB gradients + an add)."""
        require_gpu(pm_map, rois)
        rois = rois.detach().float().contiguous()
        ctx.save_for_backward(rois)
        ctx.geom = (batch, height, width, float(spatial_scale), heads, pm_map.shape[0], pm_map.stride(0))
        ctx.extract = extract
        votes = tuple(psroi_pm(pm_map, h, batch, height, width, rois, spatial_scale) for h in heads)
        if extract is not None:
            return votes + (pm_map[:, extract[0]:extract[0] + extract[1]].contiguous(),)
        return votes

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gvotes):
        import ctypes
        (rois,) = ctx.saved_tensors
        batch, height, width, scale, heads, M, stride = ctx.geom
        dev = rois.device
        gmap = torch.empty((M, stride), dtype=torch.float32, device=dev)
        R = rois.shape[0]
        L = _lib.lib()
        add = gvotes[len(heads)] if ctx.extract is not None else None
        tiled = all(h["offset"] == sum(g["group"] ** 2 * g["cp"] for g in heads[:i]) for i, h in enumerate(heads))
        assert tiled, "heads must tile the row from column 0"
        import os
        per_head = os.environ.get("DTT_PSROI_BWD_OLD", "0") not in ("", "0")      # developer A/B: round 5's launches (one per head + zero_ + add)
        if not per_head and len(heads) <= 2 and all(h["group"] == 7 for h in heads) and sum(h["cp"] for h in heads) <= 64:
            # one launch for the heads of the map (one wave per pixel, the padding columns and the second consumer's gradient in the
            # same pass): csrc/psroi_bwd.hip
            gvs = [torch.zeros((R, h["od"]), dtype=torch.float32, device=dev) if gv is None else gv.contiguous()
                   for h, gv in zip(heads, gvotes[:len(heads)])]
            h0, h1 = heads[0], heads[1] if len(heads) == 2 else None
            if add is not None:
                add = add.contiguous()
            with torch.cuda.device(dev):
                check(L.dtt_psroi_pm_backward_heads(ptr(gvs[0]), h0["od"], h0["cp"], ptr(gvs[1]) if h1 else None, h1["od"] if h1 else 0,
                                                    h1["cp"] if h1 else 0, ptr(rois), R, batch, height, width, 7, scale, stride, stride,
                                                    ptr(add) if add is not None else None, ctx.extract[0] if add is not None else 0,
                                                    ctx.extract[1] if add is not None else 0, ptr(gmap), stream_ptr(dev)),
                      "psroi_pm backward (heads)")
            return gmap, None, None, None, None, None, None, None
        covered = 0
        with torch.cuda.device(dev):
            for h, gv in zip(heads, gvotes[:len(heads)]):
                G, cp, od = h["group"], h["cp"], h["od"]
                covered += G * G * cp
                gv = torch.zeros((R, od), dtype=torch.float32, device=dev) if gv is None else gv.contiguous()
                edges = torch.empty((max(R, 1) * (4 * G + 1) + 2 * batch,), dtype=torch.int32, device=dev)   # bin edges + per-image RoI runs
                check(L.dtt_psroi_pm_backward(ptr(gv), ptr(rois), R, batch, height, width, G, scale, od, cp, stride,
                                              ctypes.c_void_p(gmap.data_ptr() + 4 * h["offset"]), ptr(edges),
                                              stream_ptr(dev)), "psroi_pm backward")
        if covered < stride:
            gmap[:, covered:].zero_()
        if add is not None:
            c0, nc = ctx.extract
            gmap[:, c0:c0 + nc] += add
        return gmap, None, None, None, None, None, None, None


class TrackingRowsFn(torch.autograd.Function):
    """The tracking head's input under autograd, assembled in place as position-major rows (rfcn.py:166-174's torch.cat):
        rows[p] = [box deltas(t) | box deltas(t+tau) | corr3 | corr4 | corr5 | zero padding]      (B*H*W, k_pad)
    loc_cols: (2*B*H*W, n_box) compact box-delta columns of both legs (PsroiPmFn's `extract` output, position-major order);
    conv3 / conv4 / conv5: the whole channels-last (2*B, C, h, w) trunk maps, leg 0 = frame t.  The three correlations write their
    columns straight into the rows (window-split forward kernel); backward: the correlation gradient kernels read THEIR columns of
    the rows' gradient where they lie and write both legs of one channels-last gradient per map -- no 1051-channel NCHW tensor, no
    concat, no slices, no layout copies in either direction."""

    @staticmethod
    def forward(ctx, loc_cols, conv3, conv4, conv5, B, geoms, k_pad):
        from .ops import correlation_forward_nhwc, correlation_output_shape
        require_gpu(loc_cols, conv3, conv4, conv5)
        n_box = loc_cols.shape[1]
        maps = (conv3, conv4, conv5)
        oshape = [correlation_output_shape(m.size(1), m.size(2), m.size(3), *g) for m, g in zip(maps, geoms)]
        oh, ow = oshape[0][1], oshape[0][2]
        assert all(o[1:] == (oh, ow) for o in oshape) and loc_cols.shape[0] == 2 * B * oh * ow
        cols = [2 * n_box]
        for o in oshape:
            cols.append(cols[-1] + o[0])
        assert cols[-1] <= k_pad
        rows = torch.empty((B * oh * ow, k_pad), dtype=torch.float32, device=loc_cols.device)
        if cols[-1] < k_pad:
            rows[:, cols[-1]:].zero_()
        gather_column_blocks(rows, 0, loc_cols, 0, B * oh * ow, 2, n_box)
        for m, g, c0 in zip(maps, geoms, cols):
            correlation_forward_nhwc(m[:B], m[B:2 * B], *g, rows=rows, col=c0)
        ctx.save_for_backward(conv3, conv4, conv5)
        ctx.cfg = (B, geoms, cols, n_box, oh * ow)
        return rows

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grows):
        from .ops import correlation_backward_nhwc
        maps = ctx.saved_tensors
        B, geoms, cols, n_box, hw = ctx.cfg
        grows = grows.contiguous()
        g_loc = None
        if ctx.needs_input_grad[0]:
            g_loc = torch.cat([grows[:, :n_box], grows[:, n_box:2 * n_box]], 0)
        # The three gradient ops run one after the other; each starts with a small launch that lays out its band words (18 us for
        # conv4 / conv5: 680 workgroups that leave most of the chip idle).  The bands of the LATER ops only read `grows`: with
        # DTT_CORR_BWD_OVERLAP=1 they are laid out on a second stream beside the first op (round 6).  MEASURED AND LEFT OFF: the two
        # ops get 17 us shorter each (conv4 111.8 -> 94.1, conv5 183.5 -> 167.2 us, conv3 beside the bands 45.6 -> 55.2) but the training
        # step gets 0.85 ms LONGER (37.94 -> 38.80 ms, two runs each way): the second stream's fork / join costs more elsewhere in
        # the step than the 25 us it hides.
        import os
        from .ops import corr_bwd_stream_enabled
        todo = [(i, m, g, c0) for i, (m, g, c0) in enumerate(zip(maps, geoms, cols)) if ctx.needs_input_grad[1 + i]]
        gmaps = [None] * len(maps)
        dev = grows.device
        split = (len(todo) >= 2 and os.environ.get("DTT_CORR_BWD_OVERLAP", "0") == "1" and corr_bwd_stream_enabled()
                 and all(m.size(1) % 64 == 0 for _, m, _, _ in todo))
        wss = {}
        if split:
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ready)                       # (the rows' gradient is the main stream's work)
                grows.record_stream(side)
                for i, m, g, c0 in todo[1:]:
                    wss[i] = correlation_backward_nhwc(None, m[:B], m[B:2 * B], None, None, *g, rows=grows, col=c0, phase=1)
                    wss[i].record_stream(main)
                laid = torch.cuda.Event()
                laid.record(side)
        for k, (i, m, g, c0) in enumerate(todo):
            # (two legs: every image's gradient is written by one of the two kernels, which zero-fill where they have to)
            gm = torch.empty_like(m) if m.size(0) == 2 * B else torch.zeros_like(m)
            if split and k == 1:
                torch.cuda.current_stream(dev).wait_event(laid)
            if i in wss:
                correlation_backward_nhwc(None, m[:B], m[B:2 * B], gm[:B], gm[B:2 * B], *g, rows=grows, col=c0, phase=2, workspace=wss[i])
            else:
                correlation_backward_nhwc(None, m[:B], m[B:2 * B], gm[:B], gm[B:2 * B], *g, rows=grows, col=c0)
            gmaps[i] = gm
        return (g_loc, *gmaps, None, None, None)


_SIDE_STREAMS = {}


def _side_stream(dev):
    """One extra stream per device for work that only has to meet the main stream again later (the band words of the correlation gradients)."""
    key = str(dev)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


_DEVICE_CONSTANTS = {}


def _device_constant(key, build):
    """Small index tensors of the training graph that depend on shapes only: uploaded once per (shape, device) -- a per-step
    .to(dev) from host memory makes the host wait for the stream in the middle of the forward."""
    t = _DEVICE_CONSTANTS.get(key)
    if t is None:
        t = _DEVICE_CONSTANTS[key] = build()
    return t


def pack_rpn_heads_differentiable(cls_conv, bbox_conv):
    """PackedRPNHeads' row order as differentiable tensor ops on the live parameters: (w (rows16, K), bias (rows16,), A)."""
    K = cls_conv.weight.shape[1]
    wc = cls_conv.weight.reshape(cls_conv.weight.shape[0], K)
    wb = bbox_conv.weight.reshape(bbox_conv.weight.shape[0], K)
    A = wc.shape[0] // 2
    if A % 2 or wc.shape[0] != 2 * A or wb.shape[0] != 4 * A:
        raise ValueError("pack_rpn_heads_differentiable: an even number of anchors is required (got %d score channels)" % wc.shape[0])
    dev = wc.device
    pair = _device_constant(("rpn_pair", A, str(dev)),
                            lambda: torch.stack([torch.arange(A), A + torch.arange(A)], 1).reshape(-1).to(dev))   # bg_a, fg_a
    bc = cls_conv.bias if cls_conv.bias is not None else wc.new_zeros(2 * A)
    bb = bbox_conv.bias if bbox_conv.bias is not None else wb.new_zeros(4 * A)
    w = torch.cat([wc[pair], wb], 0)
    b = torch.cat([bc[pair], bb], 0)
    n16 = -(-w.shape[0] // 16) * 16
    if n16 > w.shape[0]:
        b = torch.cat([b, b.new_zeros(n16 - w.shape[0])], 0)
        w = torch.cat([w, w.new_zeros(n16 - w.shape[0], K)], 0)
    return w.contiguous(), b.contiguous(), A


class RpnHeadFn(torch.autograd.Function):
    """`RPN_cls_score` + reshape(2) -> softmax -> reshape(2A) + `RPN_bbox_pred` (rpn.py:63-71) as ONE launch of the hand-written
    GEMM (`dtt_rpn_head_gemm`) under autograd: x_rows (batch*H*W, K) channels-last rows of relu(RPN_Conv(.)), packed weights ->
    (cls_prob (batch, 2A, H, W), bbox_pred (batch, 4A, H, W)).  Backward: one small kernel turns the two NCHW gradients into the
    rows of the packed output's gradient, applying the adjoint of the pairwise softmax (`dtt_rpn_head_grad_rows`); dX is the head
    GEMM over those rows with the transposed weights, dW `dtt_head_gemm_dw`, dBias a column sum."""

    @staticmethod
    def forward(ctx, x_rows, w, bias, A, batch, height, width, logit_grads=False):
        """logit_grads: the gradient that comes back for cls_prob is the gradient with respect to the score LOGITS (RpnLossFn's:
        cls_prob's only differentiable consumer in the training graph), not with respect to the probabilities."""
        require_gpu(x_rows)
        x_rows = x_rows.contiguous()
        M, K = x_rows.shape
        if M != batch * height * width or K != w.shape[1]:
            raise ValueError("RpnHeadFn: rows %s do not match batch %d x %d x %d, K %d" % (tuple(x_rows.shape), batch, height, width, w.shape[1]))
        prob = torch.empty((batch, 2 * A, height, width), dtype=torch.float32, device=x_rows.device)
        bbox = torch.empty((batch, 4 * A, height, width), dtype=torch.float32, device=x_rows.device)
        with torch.cuda.device(x_rows.device):
            check(_lib.lib().dtt_rpn_head_gemm(ptr(x_rows), K, batch, height * width, K, ptr(w), ptr(bias), w.shape[0], A, ptr(prob),
                                               ptr(bbox), stream_ptr(x_rows.device)), "rpn_head_gemm")
        ctx.save_for_backward(x_rows, w, prob)
        ctx.cfg = (A, batch, height * width, bool(logit_grads))
        return prob, bbox

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_prob, g_bbox):
        x_rows, w, prob = ctx.saved_tensors
        A, batch, hw, logit_grads = ctx.cfg
        M, K = x_rows.shape
        N16 = w.shape[0]
        dev = x_rows.device
        L = _lib.lib()
        g_prob = None if g_prob is None else g_prob.contiguous()
        g_bbox = None if g_bbox is None else g_bbox.contiguous()
        gx = gw = gb = None
        LG = -(-N16 // 32) * 32                                          # the head GEMM's K granularity
        with torch.cuda.device(dev):
            grows = torch.empty((M, LG), dtype=torch.float32, device=dev)   # (columns >= 6 A are written as zeros)
            check(L.dtt_rpn_head_grad_rows(ptr(g_prob) if g_prob is not None else None, ptr(g_bbox) if g_bbox is not None else None,
                                           ptr(prob), batch, hw, A, ptr(grows), LG, 1 if logit_grads else 0, stream_ptr(dev)), "rpn_head_grad_rows")
            if ctx.needs_input_grad[0]:
                wt = torch.zeros((K, LG), dtype=torch.float32, device=dev)    # dX (M, K) = grows (M, LG) @ wt.T
                wt[:, :N16] = w.t()
                gx = torch.empty((M, K), dtype=torch.float32, device=dev)
                zero_bias = _device_constant(("zero_bias", K, str(dev)), lambda: torch.zeros(K, dtype=torch.float32, device=dev))   # (read only; lives on)
                check(L.dtt_head_gemm(ptr(grows), LG, M, LG, ptr(wt), ptr(zero_bias), K, ptr(gx),
                                      K, K, 0, stream_ptr(dev)), "rpn head dX")
            if ctx.needs_input_grad[1]:
                gw = torch.empty((N16, K), dtype=torch.float32, device=dev)
                nb = L.dtt_head_gemm_dw_workspace_bytes(M, N16, K)
                ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
                check(L.dtt_head_gemm_dw(ptr(grows), LG, LG, ptr(x_rows), K, M, N16, K, ptr(gw), ptr(ws), nb, stream_ptr(dev)),
                      "rpn head dW")
            if ctx.needs_input_grad[2]:
                gb = grows[:, :N16].sum(0)
        return gx, gw, gb, None, None, None, None, None


class RpnLossFn(torch.autograd.Function):
    """The two RPN losses of rpn/rpn.py:86-105 for `legs` legs in one launch (`dtt_rpn_loss_forward`): cls_prob (legs*B, 2A, H, W)
    and bbox_pred (legs*B, 4A, H, W) from `RpnHeadFn(..., logit_grads=True)`, the anchor-target layer's four outputs for the same
    images -> losses (2*legs,): [leg] = class loss (cross-entropy over the labelled anchors), [legs + leg] = box loss.
    THE GRADIENT RETURNED FOR cls_prob IS THE GRADIENT WITH RESPECT TO THE SCORE LOGITS, (p - y) * g / count
    (`dtt_rpn_loss_backward`): it bypasses the softmax adjoint, which multiplies by p and vanishes where p underflows -- the
    reference's cross_entropy on the logits keeps -g / count there.  Only `RpnHeadFn` with logit_grads=True may produce cls_prob."""

    @staticmethod
    def forward(ctx, cls_prob, bbox_pred, labels, targets, w_in, w_out, legs, sigma):
        require_gpu(cls_prob)
        dev = cls_prob.device
        B2, A2, H, W = cls_prob.shape
        A = A2 // 2
        ts = [t.contiguous() for t in (cls_prob, bbox_pred, labels, targets, w_in, w_out)]
        if ts[1].shape != (B2, 4 * A, H, W) or ts[2].numel() != B2 * A * H * W or any(t.shape != ts[1].shape for t in ts[3:]):
            raise ValueError("RpnLossFn: shapes %s do not belong to one RPN head" % [tuple(t.shape) for t in ts])
        if any(t.dtype != torch.float32 for t in ts) or B2 % legs:
            raise ValueError("RpnLossFn: float32 tensors and a batch that is a multiple of the legs are required")
        L = _lib.lib()
        loss = torch.empty((2 * legs,), dtype=torch.float32, device=dev)
        count = torch.empty((legs,), dtype=torch.float32, device=dev)
        nb = int(L.dtt_rpn_loss_workspace_bytes(B2, H * W))
        ws = torch.empty((nb // 4,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.dtt_rpn_loss_forward(*[ptr(t) for t in ts], B2, legs, A, H * W, float(sigma), ptr(loss), ptr(count), ptr(ws), nb,
                                         stream_ptr(dev)), "rpn_loss")
        ctx.save_for_backward(*ts, count)
        ctx.cfg = (B2, legs, A, H * W, float(sigma))
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss):
        *ts, count = ctx.saved_tensors
        B2, legs, A, hw, sigma = ctx.cfg
        dev = ts[0].device
        g_loss = g_loss.contiguous().float()
        g_logits, g_bbox = torch.empty_like(ts[0]), torch.empty_like(ts[1])
        with torch.cuda.device(dev):
            check(_lib.lib().dtt_rpn_loss_backward(*[ptr(t) for t in ts], ptr(g_loss), ptr(count), B2, legs, A, hw, sigma, ptr(g_logits),
                                                   ptr(g_bbox), stream_ptr(dev)), "rpn_loss backward")
        return g_logits, g_bbox, None, None, None, None, None, None


def pm_to_nchw(pm_map, head, batch, height, width):
    """The head's score map in the reference layout (batch, od*G*G, H, W) -- for tests and for callers that want the
    reference tensor."""
    G, od, cp, off = head["group"], head["od"], head["cp"], head["offset"]
    v = pm_map[:, off:off + G * G * cp].reshape(batch, height, width, G * G, cp)[..., :od]
    return v.permute(0, 4, 3, 1, 2).reshape(batch, od * G * G, height, width).contiguous()


class PositionMajorTail:
    """Inference tail of `_RFCN.forward` (rfcn.py:133-140, 166-196) in the position-major layout, built from the model's
    current weights (dtt.fuse.fuse_for_inference; rebuild after loading a checkpoint):

      det   RFCN_cls_net + RFCN_bbox_net as ONE GEMM over the channels-last `top` rows (both legs, all images)
      trk   corr_bbox_net over rows  [bbox_t | bbox_t+tau | corr3 | corr4 | corr5 | 0-pad]  -- the box-delta columns are
            copied out of the det map (position-major order, so the weight's input channels are permuted to match) and
            the three correlations write their columns directly (dtt.ops.correlation_forward_rows)
    """

    def __init__(self, model):
        self.rpn = None
        try:
            self.rpn = PackedRPNHeads(model.RFCN_rpn.RPN_cls_score, model.RFCN_rpn.RPN_bbox_pred)
        except (ValueError, AttributeError):
            pass   # odd anchor count: the RPN heads stay library convolutions
        self.det = PackedHeads([model.RFCN_cls_net, model.RFCN_bbox_net])
        self.cls_head, self.loc_head = self.det.heads
        self.trk = None
        conv = getattr(model, "corr_bbox_net", None)
        if conv is not None:
            G = self.loc_head["group"]
            nb = self.loc_head["od"] * G * G                     # 196 box-delta channels per leg
            K = conv.weight.shape[1]
            perm = torch.arange(K)
            # buffer column leg*nb + bin*od + k  <-  reference channel leg*nb + k*G*G + bin
            od = self.loc_head["od"]
            b, k = torch.meshgrid(torch.arange(G * G), torch.arange(od), indexing="ij")
            for leg in range(2):
                perm[leg * nb:(leg + 1) * nb] = (leg * nb + k * G * G + b).reshape(-1)
            self.trk_k = -(-K // 32) * 32
            self.trk = PackedHeads([conv], k_pad=self.trk_k, in_perm=perm)
            self.trk_head = self.trk.heads[0]
            self.n_box = nb
            self.trk_in = K
        self._rows = {}

    def tracking_rows(self, n_pixels, device):
        """(n_pixels, K padded) scratch whose padding columns are zero (written once)."""
        key = (n_pixels, device)
        buf = self._rows.get(key)
        if buf is None:
            buf = self._rows[key] = torch.zeros((n_pixels, self.trk_k), dtype=torch.float32, device=device)
        return buf
