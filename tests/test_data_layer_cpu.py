"""dtt.data (ImageNet VID / DET readers, frame-pair roidb, aspect-ratio grouping, crop / pad loader, VOC-style AP) against
golden vectors produced by RUNNING the reference's own data layer (tests/golden/make_golden_data.py) on the synthetic devkit
of tests/data_fixture.py.  Same driver function, same seeds: roidb entries, pair selection and shuffle, ratio lists, every
loader item (training crops included, i.e. the same numpy RNG consumption), parsed annotations and precision / recall / AP
must agree.  CPU only."""
import os
import types

import numpy as np
import pytest

import data_fixture as fx

G = os.path.join(os.path.dirname(__file__), "golden", "data_layer.npz")


@pytest.fixture()
def data_cfg(tmp_path):
    from dtt.config import cfg
    saved = (cfg.DATA_DIR, cfg.TRAIN.SCALES, cfg.TRAIN.USE_FLIPPED, cfg.MAX_NUM_GT_BOXES)
    gold = np.load(G)
    cfg.DATA_DIR = str(tmp_path / "data")
    cfg.TRAIN.SCALES = (int(gold["scale"]),)
    cfg.TRAIN.USE_FLIPPED = False
    cfg.MAX_NUM_GT_BOXES = 30
    fx.build_devkit(cfg.DATA_DIR)
    yield cfg, gold, tmp_path
    cfg.DATA_DIR, cfg.TRAIN.SCALES, cfg.TRAIN.USE_FLIPPED, cfg.MAX_NUM_GT_BOXES = saved


def _api():
    from dtt.data import combined_roidb, roibatchLoader
    from dtt.data.vid_eval import parse_vid_rec, vid_eval

    def write_results(imdb, all_boxes, pairs):
        imdb._roidb = pairs
        imdb._image_index = [os.path.splitext("/".join(p[0]["image"].split("/")[-3:]))[0] for p in pairs]
        imdb._write_results(all_boxes)
        text = []
        for cls in imdb.classes[1:]:
            with open(imdb._results_template().format(cls)) as f:
                text.append(f.read())
        return "\x1e".join(text)
    return types.SimpleNamespace(combined_roidb=combined_roidb, roibatchLoader=roibatchLoader, vid_eval=vid_eval,
                                 parse_vid_rec=parse_vid_rec, write_results=write_results)


def test_data_layer_matches_reference_run(data_cfg):
    cfg, gold, tmp = data_cfg
    out = fx.run_data_layer(_api(), cfg.DATA_DIR, str(tmp / "out"))
    mine_text = out.pop("results_text")
    assert set(gold.files) - {"scale", "results_text"} <= set(out), sorted(set(gold.files) - set(out))[:5]
    assert int(gold["train_n"]) == 13 and int(gold["det_n"]) == 2 and int(gold["test_n"]) == 3  # fixture sanity
    for key in gold.files:
        if key in ("scale", "results_text"):
            continue
        ref, got = gold[key], np.asarray(out[key])
        if ref.dtype.kind in "US":
            assert str(ref) == str(got), key
        elif ref.dtype.kind in "iub":
            np.testing.assert_array_equal(got, ref, err_msg=key)
        else:
            assert got.shape == ref.shape, key
            np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-5, err_msg=key)
    # the results files our writer produces are the ones the evaluation above consumed (written by the driver in the
    # reference's format, imagenet_detect.py:256-261)
    dets = []
    for cls in ("airplane", "bear"):
        with open(os.path.join(str(tmp / "out"), "det_test_%s.txt" % cls)) as f:
            dets.append(f.read())
    blocks = str(mine_text).split("\x1e")
    assert blocks[0] == dets[0] and blocks[2] == dets[1]


def test_evaluate_detections_end_to_end(data_cfg):
    """imdb.evaluate_detections: results files -> per-class AP -> mean AP, cleanup of the results files."""
    cfg, gold, tmp = data_cfg
    from dtt.data import combined_roidb
    imdb, pairs, _, _ = combined_roidb("imagenet_vid_test", False)
    aps = imdb.evaluate_detections(fx.synthetic_detections(pairs, imdb.num_classes), pairs, str(tmp / "eval"))
    assert len(aps) == 30
    np.testing.assert_allclose(np.mean(aps), float(gold["eval_map"]), rtol=1e-9)
    assert not os.listdir(os.path.join(cfg.DATA_DIR, "ILSVRC", "results"))
    assert os.path.exists(os.path.join(str(tmp / "eval"), "airplane_pr.pkl"))


def test_loader_batches_collate_through_torch_dataloader(data_cfg):
    """The training path of trainval_net.py: ratio-sorted pairs, batch-permuting sampler, default collate.  Every batch
    holds frame pairs padded / cropped to one size: data (B, 2, 3, H, W), im_info (B, 2, 3), gt (B, 2, 30, 6), num (B, 2, 1),
    and im_info carries the padded size."""
    import torch
    from dtt.data import combined_roidb, roibatchLoader, sampler
    cfg, gold, tmp = data_cfg
    np.random.seed(5)
    torch.manual_seed(5)
    imdb, pairs, ratio_list, ratio_index = combined_roidb("imagenet_vid_train")
    bs = 2
    ds = roibatchLoader(pairs, ratio_list, ratio_index, bs, imdb.num_classes, training=True)
    loader = torch.utils.data.DataLoader(ds, batch_size=bs, sampler=sampler(len(pairs), bs), num_workers=0)
    seen = 0
    for data, im_info, gt, num in loader:
        b = data.shape[0]
        seen += b
        assert data.shape[1:3] == (2, 3) and im_info.shape == (b, 2, 3) and gt.shape == (b, 2, 30, 6) and num.shape == (b, 2, 1)
        assert torch.all(im_info[..., 0] == data.shape[3]) and torch.all(im_info[..., 1] == data.shape[4])
        for i in range(b):
            for leg in range(2):
                n = int(num[i, leg, 0])
                boxes = gt[i, leg, :n]
                assert n >= 1 and torch.all(boxes[:, 2] > boxes[:, 0]) and torch.all(boxes[:, 3] > boxes[:, 1])
                assert float(boxes[:, 2].max()) <= data.shape[4] and float(boxes[:, 3].max()) <= data.shape[3]
                assert torch.all(gt[i, leg, n:] == 0)
    assert seen == len(pairs)


def test_sampler_keeps_batches_together():
    """trainval_net.py:125-150: whole batches are permuted, the remainder goes last."""
    import torch
    from dtt.data import sampler
    torch.manual_seed(0)
    order = torch.stack(list(iter(sampler(11, 3)))).tolist()
    assert len(order) == 11 and order[-2:] == [9, 10]
    heads = order[0:9:3]
    assert sorted(heads) == [0, 3, 6]
    for k, h in enumerate(heads):
        assert order[3 * k:3 * k + 3] == [h, h + 1, h + 2]


def test_resize_linear_matches_torch_bilinear():
    """The OpenCV INTER_LINEAR restatement (dtt/data/blob.py) against torch's half-pixel bilinear (same sampling rule
    when the scale is exact), and its output-size rule."""
    import torch
    import torch.nn.functional as F
    from dtt.data.blob import prep_im_for_blob, resize_linear
    rng = np.random.RandomState(0)
    im = rng.uniform(0, 255, size=(24, 36, 3)).astype(np.float32)
    out = resize_linear(im, 2.0)
    ref = F.interpolate(torch.from_numpy(im).permute(2, 0, 1)[None], scale_factor=2.0, mode="bilinear", align_corners=False)
    np.testing.assert_allclose(out, ref[0].permute(1, 2, 0).numpy(), rtol=1e-5, atol=1e-3)
    assert resize_linear(im, 600 / 24.0).shape == (600, 900, 3)
    assert resize_linear(np.zeros((50, 75, 3), np.float32), 0.9).shape[:2] == (45, 68)  # round half to even: 67.5 -> 68
    scaled, scale = prep_im_for_blob(np.full((30, 40, 3), 128, np.uint8), np.array([[[1.0, 2.0, 3.0]]]), 60, 1000)
    assert scale == 2.0 and scaled.shape == (60, 80, 3)
    np.testing.assert_allclose(scaled[5, 7], [127.0, 126.0, 125.0], rtol=1e-6)


def test_rank_shards_of_two_datasets_stay_separate():
    """trainval_net._Shard: with imagenet_vid+imagenet_det every loader owns its sampler and its length (the
    shard class used to close over the dataset loop's variables, so the VID loader drew DET-sized permutations)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dtt.data import sampler
    from trainval_net import _Shard
    bs, world = 2, 2
    sizes = (23, 9)                                                   # VID larger than DET
    shards = [[_Shard(sampler(n, bs * world, seed=7 + 7919 * k), n, bs, r, world) for r in range(world)]
              for k, n in enumerate(sizes)]
    for k, n in enumerate(sizes):
        full = n - n % (bs * world)
        per_rank = [list(iter(s)) for s in shards[k]]
        assert [len(p) for p in per_rank] == [len(shards[k][0])] * world == [full // world] * world
        union = sorted(i for p in per_rank for i in p)
        assert union == list(range(full)) and max(union) < n       # disjoint, complete, inside THIS dataset
        # slots [r*bs, (r+1)*bs) of every global batch: consecutive (ratio-sorted) indices stay together on a rank
        for p in per_rank:
            assert all(p[i + 1] == p[i] + 1 for i in range(0, len(p), bs))
    second_epoch = [list(iter(s)) for s in shards[0]]
    assert sorted(i for p in second_epoch for i in p) == list(range(20))
