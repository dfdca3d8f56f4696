"""oracle/tubes_oracle.py against the outputs of the reference's own zero-jump Viterbi tube linker
(tests/golden/tubes.npz, produced by tests/golden/make_golden_tubes.py running tracking_utils.py on CPU tensors)."""
import os

import numpy as np
import pytest

from oracle import tubes_oracle as to

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tubes.npz"))
CASES = sorted({k.split("/")[0] for k in G.files} - {"video"})


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


def test_video_post_processor_oracle_matches_reference():
    """The whole VideoPostProcessor flow (frame-pair bookkeeping, dynamic class thresholds, tracklet selection, class
    paths) restated in oracle/tubes_oracle.py against the reference object run on the same predictions."""
    paths, aboxes, thresh = to.build_class_paths(G["video/pred_boxes"], G["video/scores"], G["video/pred_trk_boxes"])
    np.testing.assert_allclose(thresh, G["video/conf_thresh"], rtol=0, atol=0)
    C = G["video/scores"].shape[3]
    for c in range(1, C):
        kept = np.array([0 if b is None else len(b) for b in aboxes[c]], np.int32)
        np.testing.assert_array_equal(kept, G["video/n_kept_c%d" % c])
        np.testing.assert_array_equal(paths[c]["idx"], G["video/c%d/idx" % c])
        np.testing.assert_array_equal(paths[c]["boxes"], G["video/c%d/boxes" % c])
        for k in ("total_score", "scores", "smooth_scores"):
            np.testing.assert_allclose(paths[c][k], G["video/c%d/%s" % (c, k)], rtol=0, atol=1e-6)


def test_keep_top_k_reproduces_the_reference_index_error():
    frames = [np.array([[0, 0, 5, 5, 0.5, 0.5]], np.float32)] * 3
    with pytest.raises(IndexError):
        to.keep_top_k(frames, 160 * 3)
