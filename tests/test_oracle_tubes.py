"""oracle/tubes_oracle.py against the outputs of the reference's own zero-jump Viterbi tube linker
(tests/golden/tubes.npz, produced by tests/golden/make_golden_tubes.py running tracking_utils.py on CPU tensors)."""
import os

import numpy as np
import pytest

from oracle import tubes_oracle as to

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tubes.npz"))
CASES = sorted({k.split("/")[0] for k in G.files})


@pytest.mark.parametrize("case", CASES)
def test_tube_linking_oracle_matches_reference(case):
    g = lambda k: G[case + "/" + k]
    trk = g("trk") if int(g("has_tracks")[0]) else None
    got = to.make_tubes(g("dets"), g("n"), trk, g("m"))
    np.testing.assert_array_equal(got["idx"], g("idx"))                      # the paths themselves: exact
    np.testing.assert_array_equal(got["boxes"], g("boxes"))
    np.testing.assert_allclose(got["total_score"], g("total_score"), rtol=0, atol=1e-6)
    np.testing.assert_allclose(got["scores"], g("scores"), rtol=0, atol=1e-6)
    np.testing.assert_allclose(got["smooth_scores"], g("smooth_scores"), rtol=0, atol=1e-6)


def test_fixture_covers_the_interesting_paths():
    assert len(CASES) >= 8
    assert int(G["ragged_cut/idx"].max()) < 25 and int(G["ragged_cut/n"].max()) > 50   # the max_per_image cut applies
    # the tracking bonus actually fires: some path scores exceed what detection scores alone could reach
    b = G["tracks/boxes"]
    assert float(G["tracks/total_score"].max()) * (b.shape[1] + 1) > float(b[0, :, 4].sum()) * 1.5
    assert (G["missing_tracks/m"] < 0).sum() == 3 and G["two_frames/idx"].shape[1] == 1


def test_filter2d_reflect101():
    v = np.array([1, 2, 4, 8, 16, 32], np.float32)
    # borders: gfedcb|abcdefgh|gfedcba
    want = [(4 * 1 + 2 * 4 + 1 * 6 + 2 * 4 + 4) / 16.0, (2 * 1 + 1 * 4 + 2 * 6 + 4 * 4 + 8) / 16.0]
    got = to.filter2d_reflect101(v)
    assert abs(got[0] - want[0]) < 1e-6 and abs(got[1] - want[1]) < 1e-6
    assert to.filter2d_reflect101(np.array([3.0], np.float32))[0] == 3.0
