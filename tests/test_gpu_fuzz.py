"""Randomized shape sweep of correlation (forward + backward), PSRoI pooling and NMS against the oracle: map sizes around
tile / piece boundaries, channel counts around chunk boundaries, strides, pad != displacement, integer boxes.  The sweep
(tests/fuzz_ops.py) found two out-of-bounds reads in this round that the fixed cases did not reach."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 11])
def test_random_shape_sweep(seed):
    import torch
    assert torch.cuda.is_available()
    import fuzz_ops
    assert fuzz_ops.run(N=120, seed=seed) == (0, 0, 0)


def test_random_shape_sweep_remaining_ops():
    """Proposal layer (incl. heavy score ties), anchor-target layer, RoI Align / Pool / Crop, per-class NMS and tube
    linking on random shapes against the oracle."""
    import fuzz_ops
    assert not any(fuzz_ops.run_more(N=40, seed=5).values())


@pytest.mark.parametrize("seed", [2, 23])
def test_random_shape_sweep_tail_kernels(seed):
    """Round-2 kernels (channels-last correlation in both output layouts, head GEMM, position-major PSRoI pooling + vote) on
    random shapes: window radii 1 ... 16, strides, pad != displacement, maps smaller than a tile, 32 ... 256 input channels,
    1 - 2 heads of 4 ... 31 classes, RoIs hanging over the image, bin edges on pixel boundaries."""
    import fuzz_ops
    assert not any(fuzz_ops.run_tail(N=60, seed=seed).values())


def test_oracle_against_reference_kernels_on_random_shapes():
    """The CPU oracle against the reference's own kernels (oracle/_ref) over random geometries: bit for bit."""
    import fuzz_ops
    from oracle import ref_kernels as RK
    if not RK.available():
        pytest.skip("oracle/_ref not built (oracle/build_ref.sh needs /root/reference + hipify-perl)")
    assert not any(fuzz_ops.run_ref(N=60, seed=3).values())
