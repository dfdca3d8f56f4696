"""CPU oracle (oracle/dtt_oracle.c) against golden vectors produced by the REFERENCE's own kernels
(tests/golden/ref_kernels.npz, made on an MI355X by tests/golden/make_golden_ref_kernels.py from oracle/_ref):
forward values, channel / argmax maps and keep lists bit for bit; gradients (the reference accumulates them with
float atomics, so their summation order is free) to 1e-5.  No GPU, no /root/reference needed."""
import os

import numpy as np
import pytest

from oracle import oracle_lib as O

G = os.path.join(os.path.dirname(__file__), "golden", "ref_kernels.npz")


@pytest.fixture(scope="module")
def gold():
    assert os.path.exists(G), "tests/golden/ref_kernels.npz is missing"
    return np.load(G)


@pytest.mark.parametrize("name", ["a", "b", "c", "d", "e"])
def test_correlation_matches_reference_kernel_outputs(gold, name):
    pad, k, d, s1, s2 = [int(v) for v in gold["corr_%s_cfg" % name]]
    x1, x2 = gold["corr_%s_x1" % name], gold["corr_%s_x2" % name]
    np.testing.assert_array_equal(O.correlation_forward(x1, x2, pad, k, d, s1, s2), gold["corr_%s_out" % name])
    if "corr_%s_gout" % name in gold.files:
        g1, g2 = O.correlation_backward(gold["corr_%s_gout" % name], x1, x2, pad, k, d, s1, s2)
        np.testing.assert_allclose(g1, gold["corr_%s_g1" % name], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(g2, gold["corr_%s_g2" % name], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["a", "b"])
def test_psroi_matches_reference_kernel_outputs(gold, name):
    g, od = [int(v) for v in gold["psroi_%s_cfg" % name]]
    feat, rois = gold["psroi_%s_feat" % name], gold["psroi_%s_rois" % name]
    out, mapping = O.psroi_pool_forward(feat, rois, g, g, 1 / 16.0, g, od)
    np.testing.assert_array_equal(mapping, gold["psroi_%s_map" % name])
    np.testing.assert_array_equal(out, gold["psroi_%s_out" % name])
    np.testing.assert_allclose(O.psroi_pool_backward(gold["psroi_%s_top" % name], rois, feat.shape, g, g, 1 / 16.0, g, od),
                               gold["psroi_%s_grad" % name], rtol=1e-5, atol=1e-5)


def test_roi_align_pool_crop_match_reference_kernel_outputs(gold):
    feat, rois, top = gold["roi_feat"], gold["roi_rois"], gold["roi_top"]
    np.testing.assert_array_equal(O.roi_align_forward(feat, rois, 7, 7, 1 / 16.0), gold["align_out"])
    np.testing.assert_allclose(O.roi_align_backward(top, rois, feat.shape, 7, 7, 1 / 16.0), gold["align_grad"], rtol=1e-5,
                               atol=1e-5)
    out, arg = O.roi_pool_forward(feat, rois, 7, 7, 1 / 16.0)
    np.testing.assert_array_equal(out, gold["pool_out"])
    np.testing.assert_array_equal(arg, gold["pool_argmax"])
    np.testing.assert_allclose(O.roi_pool_backward(top, rois, arg, feat.shape, 7, 7, 1 / 16.0), gold["pool_grad"], rtol=1e-5,
                               atol=1e-5)
    np.testing.assert_array_equal(O.roi_crop_forward(feat, gold["crop_grid"]), gold["crop_out"])
    np.testing.assert_allclose(O.roi_crop_backward(feat, gold["crop_grid"], gold["crop_gout"]), gold["crop_grad"], rtol=1e-5,
                               atol=1e-5)


@pytest.mark.parametrize("name", ["a", "b"])
def test_nms_matches_reference_kernel_keep_list(gold, name):
    keep = O.nms(gold["nms_%s_dets" % name], float(gold["nms_%s_thresh" % name]))
    np.testing.assert_array_equal(keep, gold["nms_%s_keep" % name])
