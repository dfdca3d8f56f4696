"""No-GPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly
the symbols include/dtt_hip.h declares; the Python binding table matches the header; ops refuse CPU tensors
instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dtt_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dtt_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from dtt import _lib
    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libdtt_hip.so does not export %s" % n
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.dtt_abi_version() == 1


def test_header_is_plain_c():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)  # declarations only, comments stripped
    for forbidden in ("torch", "at::", "Tensor", "hip/hip_runtime", "#include <hip"):
        assert forbidden not in src, forbidden


def test_shape_and_workspace_queries_run_without_gpu():
    from dtt import _lib
    from dtt.ops import correlation_output_shape
    assert correlation_output_shape(512, 75, 134, 8, 1, 8, 2, 2) == (81, 38, 67)      # conv3 (rfcn.py:58)
    assert correlation_output_shape(2048, 38, 67, 8, 1, 8, 1, 1) == (289, 38, 67)     # conv5
    assert correlation_output_shape(1024, 36, 63, 16, 1, 16, 1, 1) == (1089, 36, 63)  # config 5
    assert correlation_output_shape(8, 20, 20, 3, 3, 4, 1, 2)[0] == 25
    with pytest.raises(RuntimeError):
        correlation_output_shape(8, 4, 4, 0, 1, 8, 1, 1)  # empty output -> error string, not exit()
    assert "empty output" in _lib.last_error()
    L = _lib.lib()
    # head weight gradient: slices x 7 padded 256-row tile rows x 512 inputs; no pixel rows -> nothing
    dw = L.dtt_head_gemm_dw_workspace_bytes(10184, 1776, 512)
    assert dw > 0 and dw % (7 * 256 * 512 * 4) == 0 and L.dtt_head_gemm_dw_workspace_bytes(0, 1776, 512) == 0
    assert L.dtt_nms_workspace_bytes(6000) == (6000 * 94 + 94 + 8 + 6000) * 8   # bit matrix + parked two-phase sweep state + one "lower" word per box
    assert L.dtt_proposal_workspace_bytes(2, 12, 38, 67, 6000) > 2 * 6000 * 94 * 8
    assert L.dtt_correlation_forward_workspace_bytes(2, 2048, 38, 67, 8, 1, 8, 1, 1) > 0


def test_ops_refuse_cpu_tensors():
    from dtt.ops import Correlation, _PSRoIPooling, _RoICrop, _RoIPooling, RoIAlignAvg, nms
    x = torch.zeros(1, 49 * 2, 8, 8)
    rois = torch.tensor([[0.0, 0, 0, 31, 31]])
    for fn in (lambda: _PSRoIPooling(7, 7, 1 / 16.0, 7, 2)(x, rois), lambda: Correlation(4, 1, 4, 1, 1)(x, x),
               lambda: RoIAlignAvg(7, 7, 1 / 16.0)(x, rois), lambda: _RoIPooling(7, 7, 1 / 16.0)(x, rois),
               lambda: _RoICrop()(x, torch.zeros(1, 7, 7, 2)), lambda: nms(torch.rand(4, 5), 0.5)):
        with pytest.raises(RuntimeError, match="GPU only"):
            fn()
    assert nms(torch.zeros(0, 5), 0.5) == []  # nms_wrapper.py:13-14


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dtt import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DttLibraryError, match="no CPU"):
        _lib.lib()


def test_correlation_window_split_plans_without_gpu():
    """dtt_correlation_nhwc_plan is host code: the launch plan of the window-split correlation at the benchmark shapes
    (B = 2, 38 x 67 outputs = 10 x 17 pixel blocks; radius 8 -> 25 window blocks per pixel block)."""
    from dtt import _lib
    L = _lib.lib()

    def plan(*a):
        v = [ctypes.c_int() for _ in range(4)]
        assert L.dtt_correlation_nhwc_plan(*a, *[ctypes.byref(x) for x in v]) == 1, a
        return tuple(x.value for x in v)
    assert plan(2, 38, 67, 8, 256)[:3] == (3, 9, 256)     # 2 x ((40 + 2) four-block tiles x 3 parts + one 2 x 1 tile x 2)
    assert plan(2, 38, 67, 8, 240)[:3] == (5, 5, 426)     # CUs left to other kernels: more, shorter workgroups
    assert plan(2, 38, 67, 4, 256)[:3] == (3, 3, 256)     # conv3 (stride 2: radius 4 on the lattice)
    parts, nacc, wgs, slots = plan(1, 36, 63, 16, 256)    # BASELINE configs[4]: 81 window blocks
    assert parts * nacc >= 81 and wgs >= 36 and slots >= 4
    for B, oh, ow, R in ((1, 1, 1, 1), (3, 9, 11, 4), (8, 38, 67, 8), (1, 75, 134, 16), (2, 5, 4, 13)):
        parts, nacc, wgs, slots = plan(B, oh, ow, R, 256)
        nblk = (1 + (R + 1) // 2) ** 2
        assert parts >= 1 and -(-nblk // parts) <= nacc and wgs >= 1 and 3 <= slots <= 8
    assert L.dtt_correlation_nhwc_plan(1, 8, 8, 17, 0, None, None, None, None) == 0   # radius > 16: not this kernel


def test_correlation_window_split_plans_cover_every_pair_exactly_once():
    """dtt_correlation_nhwc_plan_check replays every work item of a plan through the kernel's own item decode (the same
    magic-number divisions, compiled for the host) and counts the owners of every (image, pixel block, window block) triple:
    exactly one, for the benchmark shapes and a sweep of odd sizes, radii and CU budgets (pure host code)."""
    from dtt import _lib
    L = _lib.lib()
    cases = [(2, 38, 67, 8, 0), (2, 38, 67, 8, 240), (2, 38, 67, 4, 0), (2, 38, 67, 4, 240), (1, 36, 63, 16, 0), (8, 38, 67, 8, 0),
             (1, 14, 34, 16, 0), (1, 1, 1, 1, 0), (3, 9, 11, 4, 7), (1, 75, 134, 16, 0), (2, 5, 4, 13, 1), (1, 4, 33, 8, 0), (1, 20, 4, 8, 0)]
    import numpy as np
    rs = np.random.RandomState(0)
    for _ in range(120):
        cases.append((int(rs.randint(1, 5)), int(rs.randint(1, 80)), int(rs.randint(1, 140)), int(rs.randint(1, 17)),
                      int(rs.choice([0, 1, 17, 64, 240, 252]))))
    for c in cases:
        assert L.dtt_correlation_nhwc_plan_check(*c) == 1, c


def test_correlation_streamed_backward_plans_without_gpu():
    """dtt_correlation_backward_plan / _plan_check are host code: the launch plans of the band-stationary streamed correlation
    gradient kernels (csrc/correlation_bwd.hip).  Every (image, 4 x 4 target block, 64-channel group) must be owned by exactly one
    wave of exactly one work item, the dispatch table must be a permutation of the items and every tile shape must fit its LDS
    ring -- for the training shapes and a sweep of map sizes, radii, channel counts and CU counts."""
    import numpy as np
    from dtt import _lib
    L = _lib.lib()

    def plan(*a):
        v = [ctypes.c_int() for _ in range(4)]
        assert L.dtt_correlation_backward_plan(*a, *[ctypes.byref(x) for x in v]) == 1, a
        return tuple(x.value for x in v)
    # the 600 px training step, B = 2: 2 x (20 tiles of 2 x 4 blocks + the 17th block column as two 4 x 1 and one 2 x 1) = 46 tiles
    items, chunk, lds, table = plan(2, 38, 67, 8, 2048, 256)
    # (the channel groups are dealt evenly over items / 46 chunks; `chunk` = the longest chunk)
    assert items % 46 == 0 and chunk == -(-32 // (items // 46)) and lds <= 144 * 1024 and table == 1
    # one round of workgroups (longest first) must beat the naive one-workgroup-per-tile plan by far: <= 7 two-deep units of a CU
    assert 3 <= chunk <= 8, chunk
    items4, chunk4, _, _ = plan(2, 38, 67, 8, 1024, 256)
    assert items4 % 46 == 0 and chunk4 == -(-16 // (items4 // 46)) and 2 <= chunk4 <= 4
    # both directions as one grid (pad == displacement): the plan sees 2 x batch images
    itemsm, chunkm, _, tablem = plan(4, 38, 67, 8, 2048, 256)
    assert itemsm % 92 == 0 and chunkm == -(-32 // (itemsm // 92)) and tablem == 1
    assert plan(2, 38, 67, 4, 512, 256)[0] >= 46                 # conv3 (radius 4 on the stride-2 lattice)
    cases = [(2, 38, 67, 8, 2048, 256), (2, 38, 67, 8, 1024, 256), (2, 38, 67, 4, 512, 256), (8, 38, 67, 8, 2048, 256), (1, 36, 63, 8, 1024, 256),
             (1, 1, 1, 1, 64, 256), (3, 9, 11, 4, 64, 7), (1, 4, 33, 8, 128, 1), (1, 20, 4, 8, 64, 256), (2, 5, 4, 8, 64, 304), (16, 38, 67, 8, 2048, 256),
             (1, 75, 134, 8, 512, 256)]
    rs = np.random.RandomState(0)
    for _ in range(150):
        cases.append((int(rs.randint(1, 5)), int(rs.randint(1, 80)), int(rs.randint(1, 140)), int(rs.randint(1, 9)),
                      64 * int(rs.randint(1, 33)), int(rs.choice([1, 8, 64, 104, 240, 256, 304]))))
    for c in cases:
        assert L.dtt_correlation_backward_plan_check(*c) == 1, c
    assert L.dtt_correlation_backward_plan_check(1, 8, 8, 12, 64, 256) == 1     # radius 9 .. 16: the window's four quarters inside the launch
    assert L.dtt_correlation_backward_plan_check(2, 36, 63, 16, 2048, 256) == 1 and L.dtt_correlation_backward_plan_check(1, 36, 63, 16, 1024, 256) == 1
    assert L.dtt_correlation_backward_plan_check(1, 8, 8, 17, 64, 256) == 0     # radius > 16: not these kernels
    assert L.dtt_correlation_backward_plan_check(1, 8, 8, 4, 48, 256) == 0      # channels % 64 != 0: round 1's kernels
    assert L.dtt_correlation_backward_stream_supported(2048, 1, 8, 1, 1) == 1 and L.dtt_correlation_backward_stream_supported(512, 1, 8, 2, 2) == 1
    assert L.dtt_correlation_backward_stream_supported(80, 1, 8, 1, 1) == 0 and L.dtt_correlation_backward_stream_supported(64, 3, 8, 1, 1) == 0
    assert L.dtt_correlation_backward_stream_supported(64, 1, 16, 1, 1) == 1 and L.dtt_correlation_backward_stream_supported(64, 1, 17, 1, 1) == 0
    assert L.dtt_correlation_backward_workspace_bytes(1, 1024, 36, 63, 16, 1, 16, 1, 1) == 4 * 2 * 9 * 16 * 100 * 64 * 4   # d = 16: four window quarters, all resident
    # workspace: the band words of both directions, NBR^2 * 4 * 64 floats per 4 x 4 target block
    assert L.dtt_correlation_backward_workspace_bytes(2, 2048, 38, 67, 8, 1, 8, 1, 1) == 2 * 2 * 10 * 17 * 100 * 64 * 4
    assert L.dtt_correlation_backward_workspace_bytes(2, 80, 38, 67, 8, 1, 8, 1, 1) == 0


def test_no_inline_asm_statement_clobbers_m0():
    """The LDS-DMA statements (csrc/correlation_wsplit.hip, correlation_bwd.hip, heads.hip) set m0 and put it back inside the statement:
    m0 is a reserved register hipcc neither allocates nor saves around inline asm, so a statement that merely LISTS it as clobbered
    (-Winline-asm: "clobber list contains reserved registers ... undefined behaviour") relies on the compiler never keeping a value
    of its own there.  No source may name it in a clobber list, and every statement that writes it must also restore it."""
    import glob
    import re
    src = os.path.join(ROOT, "pytorch-detect-to-track_amd", "csrc")
    hits = 0
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h"))):
        text = open(f).read()
        assert not re.search(r':[^;]*"m0"\s*\)', text), f + ": m0 in a clobber list"
        for stmt in re.findall(r'asm volatile\((?:[^;]|\n)*?\);', text):
            if "s_mov_b32 m0" in stmt:
                hits += 1
                assert "s_mov_b32 %0, m0" in stmt and stmt.count("s_mov_b32 m0") == 2, f + ": m0 written but not saved and restored"
    assert hits == 3


def test_round6_entry_points_validate_their_arguments_without_a_gpu():
    """dtt_psroi_pm_backward_heads / dtt_correlation_backward_nhwc_phase / dtt_rpn_loss_forward: bad shapes, null pointers and
    unsupported geometries come back as status 0 + an error string before anything touches a device (the reference's launchers print
    and return 0: correlation_cuda_kernel.cu:362-368)."""
    from dtt import _lib
    L = _lib.lib()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    heads = lambda **kw: L.dtt_psroi_pm_backward_heads(*[kw.get(k, d) for k, d in (
        ("gv0", p), ("od0", 31), ("cp0", 32), ("gv1", p), ("od1", 4), ("cp1", 4), ("rois", p), ("R", 1), ("B", 1), ("H", 4), ("W", 4),
        ("pooled", 7), ("scale", 0.0625), ("stride", 1792), ("row", 1792), ("add", None), ("a0", 0), ("an", 0), ("gmap", p), ("stream", None))])
    assert heads(pooled=3) == 0 and "not instantiated" in _lib.last_error()
    assert heads(cp0=48, cp1=32) == 0 and "at most 64" in _lib.last_error()
    assert heads(row=1000) == 0 and "do not fit" in _lib.last_error()
    assert heads(gmap=None) == 0 and "null pointer" in _lib.last_error()
    assert heads(add=p, a0=1700, an=196) == 0 and "outside the row" in _lib.last_error()
    assert heads(H=40000) == 0 and "32767" in _lib.last_error()
    # correlation gradients in phases: the phase selector and the streamed kernels' geometry
    phase = lambda ph, ic=64: L.dtt_correlation_backward_nhwc_phase(p, 81 * 16, 16, 1, 1, 81, 4, 4, p, ic, 4, 4, p, p, p, 4, 1, 4, 1, 1, 3, ph, p, 1 << 20, None)
    assert phase(0) == 0 and "phase" in _lib.last_error()
    assert phase(4) == 0 and "phase" in _lib.last_error()
    assert phase(3, ic=48) == 0 and "channels" in _lib.last_error()
    # RPN losses: the workspace query runs anywhere; a workspace that is too small is refused
    nb = L.dtt_rpn_loss_workspace_bytes(4, 38 * 67)
    assert nb > 0
    assert L.dtt_rpn_loss_forward(p, p, p, p, p, p, 4, 2, 12, 38 * 67, 3.0, p, p, p, 8, None) == 0 and _lib.last_error()
