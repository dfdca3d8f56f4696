#!/usr/bin/env python3
"""Training driver with the reference's command line (trainval_net.py:35-122) and checkpoint contract
(trainval_net.py:417-437): same flags and defaults, `cfgs/{net}[_ls].yml` selection, per-dataset `set_cfgs`,
per-parameter SGD groups, loss = sum of the five means, and `rfcn_detect_track_{session}_{epoch}_{step}.pth`
files holding {session, epoch, model, optimizer, pooling_mode, class_agnostic}.

What differs: one process per GPU instead of `nn.DataParallel` (`--mGPUs` = launch with torchrun; gradients are
all-reduced over RCCL, dtt/dist.py).  `--dataset imagenet_vid` / `imagenet_vid+imagenet_det` read an ILSVRC devkit under
cfg.DATA_DIR/ILSVRC through dtt/data (the reference's roidb / loader semantics); `--dataset synthetic` (the default: no
dataset ships with this repo) draws batches of the same layout from the seeded generator.

    python trainval_net.py --dataset synthetic --net res101 --bs 2 --cag --epochs 2 --iters_per_epoch 20   # epoch 1 only, as
                                                          # the reference's range(start_epoch, max_epochs) (trainval_net.py:317)
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 trainval_net.py --mGPUs --bs 2 --cag ...
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train a Detect-to-Track R-FCN network")
    p.add_argument("--dataset", dest="dataset", default="synthetic", type=str)
    p.add_argument("--net", dest="net", default="res101", type=str)
    p.add_argument("--start_epoch", dest="start_epoch", default=1, type=int)
    p.add_argument("--epochs", dest="max_epochs", default=20, type=int)
    p.add_argument("--disp_interval", dest="disp_interval", default=100, type=int)
    p.add_argument("--checkpoint_interval", dest="checkpoint_interval", default=10000, type=int)
    p.add_argument("--save_dir", dest="save_dir", default="output/models", type=str)
    p.add_argument("--nw", dest="num_workers", default=0, type=int)
    p.add_argument("--cuda", dest="cuda", action="store_true")
    p.add_argument("--ls", dest="large_scale", action="store_true")
    p.add_argument("--mGPUs", dest="mGPUs", action="store_true")
    p.add_argument("--bs", dest="batch_size", default=1, type=int)
    p.add_argument("--cag", dest="class_agnostic", action="store_true")
    p.add_argument("--use_det", dest="use_det", action="store_true")
    p.add_argument("--o", dest="optimizer", default="sgd", type=str)
    p.add_argument("--lr", dest="lr", default=0.001, type=float)
    p.add_argument("--lr_decay_step", dest="lr_decay_step", default=5, type=int)
    p.add_argument("--lr_decay_gamma", dest="lr_decay_gamma", default=0.1, type=float)
    p.add_argument("--s", dest="session", default=1, type=int)
    p.add_argument("--r", dest="resume", default=False, type=bool)
    p.add_argument("--checksession", dest="checksession", default=1, type=int)
    p.add_argument("--checkepoch", dest="checkepoch", default=1, type=int)
    p.add_argument("--checkpoint", dest="checkpoint", default=0, type=int)
    p.add_argument("--use_tfboard", dest="use_tfboard", default=False, type=bool)
    # additions for the synthetic loader
    p.add_argument("--iters_per_epoch", default=100, type=int)
    p.add_argument("--height", default=600, type=int)
    p.add_argument("--width", default=1067, type=int)
    p.add_argument("--init", default=None, choices=("reference", "random"),
                   help="initial weights: 'reference' = data/pretrained_model/{res101,rfcn_detect}.pth as the reference "
                        "driver (default for real datasets), 'random' = random init + BatchNorm statistics calibrated on the "
                        "first batch (default for --dataset synthetic; there are no weight files in this repo)")
    p.add_argument("--set", dest="set_cfgs", default=None, nargs=argparse.REMAINDER,
                   help="cfg overrides, KEY VALUE pairs (as test_net.py:50-52)")
    return p.parse_args(argv)


class _Shard(torch.utils.data.Sampler):
    """Rank r takes slots [r*bs, (r+1)*bs) of every global batch of one dataset's permuted epoch.  Each loader owns its
    sampler and pair count (instances, not a closure over the dataset loop: with imagenet_vid+imagenet_det both loaders
    would otherwise draw from the LAST dataset's sampler)."""

    def __init__(self, order, n_pairs, batch_size, rank, world):
        self.order, self.n_pairs, self.batch_size, self.rank, self.world = order, n_pairs, batch_size, rank, world

    def __iter__(self):
        idx = list(iter(self.order))
        full = len(idx) - len(idx) % (self.batch_size * self.world)
        g = torch.tensor(idx[:full], dtype=torch.long).view(-1, self.world, self.batch_size)[:, self.rank].reshape(-1)
        return iter(g.tolist())

    def __len__(self):
        return (self.n_pairs // (self.batch_size * self.world)) * self.batch_size


def _build_loaders(args, cfg, rank, world):
    """The reference's data path (trainval_net.py:186-238): frame-pair roidb(s), aspect-ratio-grouped loader(s) and the
    batch-permuting sampler.  With several processes every rank builds the same roidb and reads its own contiguous shard
    of each permuted epoch (per-snippet sharding, both frames of a pair on one GPU)."""
    from dtt.data import combined_roidb, roibatchLoader, sampler
    cfg.TRAIN.USE_FLIPPED = False  # trainval_net.py:191
    cfg.USE_GPU_NMS = True
    names = {"imagenet_vid": ["imagenet_vid_train"], "imagenet_vid+imagenet_det": ["imagenet_vid_train", "imagenet_det_train"]}
    if args.dataset not in names:
        raise KeyError("Unknown dataset: {} (imagenet_vid, imagenet_vid+imagenet_det or synthetic)".format(args.dataset))
    loaders = []
    for k, name in enumerate(names[args.dataset]):
        imdb, pairs, ratio_list, ratio_index = combined_roidb(name, duplicate_frames=(k == 1))
        if rank == 0:
            print("{:d} roidb frame pairs in {}".format(len(pairs), name))
        ds = roibatchLoader(pairs, ratio_list, ratio_index, args.batch_size * world, imdb.num_classes, training=True)
        # every rank must draw the SAME permutation each epoch for its slots to be disjoint: the sampler owns a generator
        # seeded from (RNG_SEED, dataset, epoch) instead of reading torch's global one (single process: the reference's)
        order = sampler(len(pairs), args.batch_size * world, seed=None if world == 1 else cfg.RNG_SEED + 7919 * k)

        loaders.append(torch.utils.data.DataLoader(ds, batch_size=args.batch_size, sampler=_Shard(order, len(pairs), args.batch_size, rank, world),
                                                   num_workers=args.num_workers, drop_last=True,
                                                   pin_memory=True))   # 53 GB/s host -> HBM, copies are non_blocking
    return loaders


def main(argv=None):
    args = parse_args(argv)
    from dtt.config import apply_dataset_defaults, cfg, cfg_from_file
    from dtt.dist import make_optimizer
    from dtt.synth import build_model, calibrate_batchnorm_, make_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        from dtt.dist import isolate_library_caches
        isolate_library_caches(local, world)   # every rank its own MIOpen find-db / kernel cache
        torch.cuda.set_device(local)  # before the process group exists: RCCL binds its communicator to this device
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if rank == 0:
        print("Called with args:")
        print(args)
    apply_dataset_defaults("imagenet_vid" if args.dataset == "synthetic" else args.dataset)  # trainval_net.py:162-172
    cfg_file = os.path.join(ROOT, "cfgs", "{}_ls.yml".format(args.net) if args.large_scale else "{}.yml".format(args.net))
    cfg_from_file(cfg_file)
    if args.set_cfgs:
        from dtt.config import cfg_from_list
        cfg_from_list(args.set_cfgs)
    loaders = None
    if args.dataset != "synthetic":
        loaders = _build_loaders(args, cfg, rank, world)  # ImageNet VID (+ DET) under cfg.DATA_DIR/ILSVRC (dtt/data)
    # after the roidb is built (combined_roidb re-seeds numpy with 123 for its shuffle, roidb.py:95):
    np.random.seed(cfg.RNG_SEED + rank)  # trainval_net.py:183 (+rank: each process samples its own anchors / RoIs)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    layers = {"res50": 50, "res101": 101, "res152": 152}[args.net]
    synthetic = args.dataset == "synthetic"
    # real datasets start from the reference's initial weights (trainval_net.py:264-269: res101 from rfcn_detect.pth,
    # res50 / res152 from the ImageNet trunk); the synthetic runs use random weights
    init = args.init or ("random" if synthetic else "reference")
    model = build_model(layers, class_agnostic=args.class_agnostic, cfg=cfg, pretrained=init == "reference" and args.net != "res101",
                        pretrained_rfcn=init == "reference" and args.net == "res101").to(dev)
    output_dir = os.path.join(args.save_dir, args.net, args.dataset)
    os.makedirs(output_dir, exist_ok=True)
    optimizer = make_optimizer(model, cfg, lr=args.lr, optimizer=args.optimizer)
    lr = args.lr * (0.1 if args.optimizer == "adam" else 1.0)
    if args.resume:
        load_name = os.path.join(output_dir, "rfcn_detect_track_{}_{}_{}.pth".format(args.checksession, args.checkepoch,
                                                                                     args.checkpoint))
        ck = torch.load(load_name, map_location=dev)
        args.session, args.start_epoch = ck["session"], ck["epoch"]
        model.load_state_dict(ck["model"])
        optimizer.load_state_dict(ck["optimizer"])
        lr = optimizer.param_groups[0]["lr"]
        if "pooling_mode" in ck:
            cfg.POOLING_MODE = ck["pooling_mode"]
        if rank == 0:
            print("loaded checkpoint %s" % load_name)
    # global batch = --bs snippets per process (per-snippet sharding; both frames of a pair stay on one GPU)
    if init == "random" and not args.resume:
        # random-init trunks only: give the frozen BatchNorm layers the statistics of the first batch.  Loaded weights
        # (pretrained / resumed) keep theirs -- the frozen BatchNorm IS those statistics (resnet.py:290-295).
        first = (tuple(t.to(dev) for t in next(iter(loaders[0]))) if loaders else
                 make_batch(args.batch_size, args.height, args.width, seed=1000, device=dev))
        calibrate_batchnorm_(model, first[0][:, 0])
    model.train()
    # broadcast rank 0's state, THEN fold the frozen constants, THEN lay out the gradient buckets (dtt.dist)
    from dtt.dist import prepare_replica
    runner = prepare_replica(model, world, channels_last=True)
    if loaders:   # trainval_net.py:312-315: train_size = the (smaller) roidb; VID and DET alternate when both are given
        sizes = [len(l.dataset) for l in loaders]
        per_epoch = int(min(sizes) / (args.batch_size * world))
        n_steps_data = 2 * per_epoch if len(loaders) > 1 else per_epoch
    if dev.type == "cuda":
        read_stream = torch.cuda.Stream(device=dev)           # the step's one host read (loss value + sampler status)
        host_pair = torch.empty(2, dtype=torch.float32).pin_memory()
    for epoch in range(args.start_epoch, args.max_epochs):   # trainval_net.py:317
        if epoch % (args.lr_decay_step + 1) == 0:
            for g in optimizer.param_groups:  # adjust_learning_rate (net_utils.py:63-66)
                g["lr"] *= args.lr_decay_gamma
            lr *= args.lr_decay_gamma
        loss_temp, start = 0.0, time.time()
        iters = [iter(l) for l in loaders] if loaders else None
        n_steps = args.iters_per_epoch if not loaders else min(n_steps_data, min(len(l) for l in loaders) * len(loaders))
        for step in range(n_steps):
            if loaders:  # VID and DET batches alternate when both are given (trainval_net.py:340-347)
                im, info, gt, nb = (t.to(dev, non_blocking=True) for t in next(iters[step % len(iters)]))
            else:
                im, info, gt, nb = make_batch(args.batch_size, args.height, args.width,
                                              seed=(epoch * 100003 + step) * world + rank, device=dev)
            runner.zero_grad(set_to_none=True)
            out = runner(im, info, gt, nb)
            rpn_cls, rpn_box, rcnn_cls, rcnn_box, trk = out[4], out[5], out[6], out[7], out[9]
            loss = rpn_cls.mean() + rpn_box.mean() + rcnn_cls.mean() + rcnn_box.mean() + trk.mean()  # :367-368
            # The reference raises "no fg and no bg RoIs" inside the forward (proposal_target_layer_cascade.py:186), i.e. before
            # the loss is used.  The device sampler flags the image instead: the flag is read here, with the loss value, in the
            # step's ONE host read -- before the weight update -- and all-reduced so that every rank aborts together
            # (a rank raising alone would leave the others waiting in the next gradient all-reduce).
            flag = model.RFCN_proposal_target.status_flag()
            flag = torch.zeros(1, device=dev) if flag is None else flag
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if dev.type == "cuda":
                # The read rides on a side stream that waits for the FORWARD only: backward is queued first, so the GPU keeps
                # working while the host waits for the two numbers (a plain .tolist() here drains the launch queue right in front
                # of the launch-bound start of backward).  A flagged step still raises before the weight update.
                packed = torch.cat([loss.detach().view(1), flag])
                fwd_done = torch.cuda.Event()
                fwd_done.record()
                loss.backward()
                runner.finish_gradients()
                with torch.cuda.stream(read_stream):
                    read_stream.wait_event(fwd_done)
                    host_pair.copy_(packed, non_blocking=True)
                packed.record_stream(read_stream)
                read_stream.synchronize()
                loss_value, bad = host_pair.tolist()
                if bad:
                    raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")
            else:
                loss_value, bad = torch.cat([loss.detach().view(1), flag]).tolist()
                if bad:
                    raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")
                loss.backward()
                runner.finish_gradients()
            optimizer.step()
            loss_temp += loss_value
            if (step + 1) % args.disp_interval == 0 and rank == 0:
                n = args.disp_interval
                fg = int((out[8] != 0).sum())
                print("[session %d][epoch %2d][iter %4d] loss: %.4f, lr: %.2e" % (args.session, epoch, step + 1,
                                                                                   loss_temp / n, lr))
                print("\t\t\tfg/bg=(%d/%d), time cost: %f" % (fg, out[8].numel() - fg, time.time() - start))
                print("\t\t\trpn_cls: %.4f, rpn_box: %.4f, rcnn_cls: %.4f, rcnn_box %.4f, tracking_box %.4f" %
                      (float(rpn_cls.mean().detach()), float(rpn_box.mean().detach()), float(rcnn_cls.mean().detach()), float(rcnn_box.mean().detach()),
                       float(trk.mean().detach())))
                loss_temp, start = 0.0, time.time()
        if rank == 0:
            save_name = os.path.join(output_dir, "rfcn_detect_track_{}_{}_{}.pth".format(args.session, epoch, step))
            torch.save({"session": args.session, "epoch": epoch + 1, "model": runner.state_dict(),
                        "optimizer": optimizer.state_dict(), "pooling_mode": cfg.POOLING_MODE,
                        "class_agnostic": args.class_agnostic}, save_name)
            print("save model: {}".format(save_name))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
