"""Frame-pair dataset with aspect-ratio grouping, crop and pad (reference: lib/roi_data_layer/roibatchLoader.py:23-272)
and the batch-permuting sampler of the training driver (trainval_net.py:125-150).

Differences by design: `num_boxes` stays on the host (the reference calls .cuda() inside Dataset.__getitem__,
roibatchLoader.py:218,222,241 -- it breaks worker processes); everything else, including what is drawn from numpy's
global RNG and in which order, follows the reference so that a seeded run sees the same crops.
"""
import numpy as np
import torch
import torch.utils.data as data
from torch.utils.data.sampler import Sampler

from ..config import cfg
from .minibatch import get_minibatch


def _crop_start(box_lo, box_hi, trim, extent):
    """Start of a `trim`-long window along one axis that keeps the span [box_lo, box_hi] of the ground-truth boxes in
    view where it can (roibatchLoader.py:124-140 for y, :158-174 for x).  Draws at most one value from np.random."""
    if box_lo == 0:
        return 0
    region = box_hi - box_lo + 1
    if region - trim < 0:
        first, last = max(box_hi - trim, 0), min(box_lo, extent - trim)
        return first if first == last else np.random.choice(range(first, last))
    slack = int((region - trim) / 2)
    return box_lo if slack == 0 else np.random.choice(range(box_lo, box_lo + slack))


class roibatchLoader(data.Dataset):
    def __init__(self, roidb, ratio_list, ratio_index, batch_size, num_classes, training=True, normalize=None):
        self._roidb = roidb
        self._num_classes = num_classes
        self.trim_height, self.trim_width = cfg.TRAIN.TRIM_HEIGHT, cfg.TRAIN.TRIM_WIDTH
        self.max_num_box = cfg.MAX_NUM_GT_BOXES
        self.training, self.normalize, self.batch_size = training, normalize, batch_size
        self.ratio_list, self.ratio_index = ratio_list, ratio_index
        self.data_size = len(ratio_list)
        # one target aspect ratio per batch of consecutive (ratio-sorted) samples: the leftmost ratio when the batch
        # is all portrait, the rightmost when all landscape, 1 when it straddles (roibatchLoader.py:38-56)
        self.ratio_list_batch = torch.zeros(self.data_size)
        for left in range(0, len(ratio_index), batch_size):
            right = min(left + batch_size - 1, self.data_size - 1)
            if ratio_list[right] < 1:
                target = ratio_list[left]
            elif ratio_list[left] > 1:
                target = ratio_list[right]
            else:
                target = 1
            self.ratio_list_batch[left:right + 1] = float(target)

    def __len__(self):
        return len(self._roidb)

    def _train_frame(self, blob, ratio, need_crop):
        """One frame -> (1,3,H,W) image padded / cropped to the batch ratio, im_info (1,3), gt (1,max,6), count (1,1)."""
        img = torch.from_numpy(blob["data"])          # (1, H, W, 3)
        im_info = torch.from_numpy(blob["im_info"])
        gt = torch.from_numpy(blob["gt_boxes"])
        H, W = img.size(1), img.size(2)
        # the batch ratio is a float32 tensor element in the reference, so W / ratio and H * ratio round in float32
        w_over_r, h_times_r = float(W / ratio), float(H * ratio)
        ratio = float(ratio)
        if need_crop:
            if ratio < 1.0:   # tall image: crop rows
                trim = min(int(np.floor(w_over_r)), H)
                y_s = _crop_start(int(torch.min(gt[:, 1])), int(torch.max(gt[:, 3])), trim, H)
                img = img[:, y_s:y_s + trim, :, :]
                gt[:, 1] -= float(y_s); gt[:, 3] -= float(y_s)
                gt[:, 1].clamp_(0, trim - 1); gt[:, 3].clamp_(0, trim - 1)
            else:             # wide image: crop columns
                trim = min(int(np.ceil(h_times_r)), W)
                x_s = _crop_start(int(torch.min(gt[:, 0])), int(torch.max(gt[:, 2])), trim, W)
                img = img[:, :, x_s:x_s + trim, :]
                gt[:, 0] -= float(x_s); gt[:, 2] -= float(x_s)
                gt[:, 0].clamp_(0, trim - 1); gt[:, 2].clamp_(0, trim - 1)
        # pad to the batch ratio; H and W are the sizes BEFORE the crop, as in the reference (roibatchLoader.py:189-207)
        if ratio < 1:
            canvas = torch.zeros(int(np.ceil(w_over_r)), W, 3)
            canvas[:H, :, :] = img[0]
            im_info[0, 0] = canvas.size(0)
        elif ratio > 1:
            canvas = torch.zeros(H, int(np.ceil(h_times_r)), 3)
            canvas[:, :W, :] = img[0]
            im_info[0, 1] = canvas.size(1)
        else:
            side = min(H, W)
            canvas = img[0][:side, :side, :]
            gt[:, :4].clamp_(0, side)
            im_info[0, 0] = side
            im_info[0, 1] = side
        degenerate = (gt[:, 0] == gt[:, 2]) | (gt[:, 1] == gt[:, 3])
        keep = torch.nonzero(degenerate == 0).view(-1)
        padded = torch.zeros(self.max_num_box, gt.size(1))
        count = torch.zeros(1, dtype=torch.long)
        if keep.numel() != 0:
            gt = gt[keep]
            n = min(gt.size(0), self.max_num_box)
            count[0] = n
            padded[:n, :] = gt[:n]
        return canvas.permute(2, 0, 1).contiguous().unsqueeze(0), im_info, padded.unsqueeze(0), count.unsqueeze(0)

    def _test_frame(self, blob):
        img = torch.from_numpy(blob["data"])
        gt_np = blob["gt_boxes"] if blob["gt_boxes"].shape[0] else np.ones((1, 6), dtype=np.float32)
        gt = torch.from_numpy(gt_np)
        padded = torch.zeros(self.max_num_box, gt.size(1))
        n = min(gt.size(0), self.max_num_box)
        padded[:n, :] = gt[:n]
        return (img.permute(0, 3, 1, 2).contiguous(), torch.from_numpy(blob["im_info"]), padded.unsqueeze(0),
                torch.tensor([[n]], dtype=torch.long))

    def __getitem__(self, index):
        """-> data (2,3,H,W), im_info (2,3), gt_boxes (2, MAX_NUM_GT_BOXES, 6), num_boxes (2,1) for one frame pair."""
        pair = self._roidb[int(self.ratio_index[index]) if self.training else index]
        for entry in pair:
            assert len(entry["track_id"]) == len(np.unique(entry["track_id"])), \
                "Cannot have >1 track with same id in same frame."
        frames = []
        for entry in pair:
            blob = get_minibatch([entry], self._num_classes)
            if self.training:
                frames.append(self._train_frame(blob, self.ratio_list_batch[index], pair[0]["need_crop"]))
            else:
                frames.append(self._test_frame(blob))
        return tuple(torch.cat(parts, dim=0) for parts in zip(*frames))


class sampler(Sampler):
    """trainval_net.py:125-150: shuffles whole batches (consecutive, ratio-sorted indices stay together); the remainder
    that does not fill a batch goes last, in order."""

    def __init__(self, train_size, batch_size, seed=None):
        """seed: None reproduces the reference (the permutation comes from torch's global CPU generator); an int makes
        the permutation a function of (seed, epoch) only -- what several processes that shard the same epoch need, so
        that their shards stay disjoint whatever else each process has drawn from the global generator."""
        self.seed = seed
        self.epoch = 0
        self.num_data = train_size
        self.num_per_batch = int(train_size / batch_size)
        self.batch_size = batch_size
        self.range = torch.arange(0, batch_size).view(1, batch_size).long()
        self.leftover = torch.arange(self.num_per_batch * batch_size, train_size).long()

    def __iter__(self):
        if self.seed is None:
            perm = torch.randperm(self.num_per_batch)
        else:
            g = torch.Generator()
            g.manual_seed(int(self.seed) * 1000003 + self.epoch)
            perm = torch.randperm(self.num_per_batch, generator=g)
            self.epoch += 1
        starts = perm.view(-1, 1) * self.batch_size
        order = (starts.expand(self.num_per_batch, self.batch_size) + self.range).view(-1)
        if self.leftover.numel():
            order = torch.cat((order, self.leftover), 0)
        return iter(order)

    def __len__(self):
        return self.num_data
