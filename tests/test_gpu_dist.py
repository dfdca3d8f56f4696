"""The N > 1 training path with the REAL model on a GPU: two ranks share cuda:0 and exchange gradients over gloo (the
box has one GPU; RCCL needs one device per rank), running the driver's start-up order (dtt.dist.prepare_replica) and
the bucketed all-reduce of DataParallelSnippets on `_RFCN` with the fused channels-last training trunk.  Their averaged
gradients must equal the gradients a single process gets by running the same two snippets one after the other."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE = (160, 224)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
    from dtt.config import apply_dataset_defaults, cfg, reset_cfg
    reset_cfg()   # (tests that ran before this one in the same process may have loaded a yml: the spawned ranks start from defaults)
    apply_dataset_defaults("imagenet_vid")
    return cfg


def _model(cfg, seed, dev):
    from dtt.synth import build_model, calibrate_batchnorm_, make_batch
    model = build_model(50, cfg=cfg, seed=seed).to(dev)
    im, _, _, _ = make_batch(1, SIZE[0], SIZE[1], seed=77, device=dev)
    calibrate_batchnorm_(model, im[:, 0])
    model.train()
    return model


def _loss(out):
    return out[4].mean() + out[5].mean() + out[6].mean() + out[7].mean() + out[9].mean()   # trainval_net.py:367-368


def _freeze_proposals(model, dev, SIZE=SIZE):
    """Replace the RPN's proposal step by boxes that do not depend on the network's outputs: a last-bit difference between
    two runs of the library convolutions can then no longer swap two near-tied proposals (and with them a sampled RoI), so
    the two sides of the comparison sample the SAME RoIs and their gradients can be compared tensor by tensor at 3e-3."""
    def fixed(cls_prob, bbox_pred, im_info):
        n = cls_prob.size(0)
        g = torch.Generator().manual_seed(4242)
        R = 300
        x1 = torch.rand(n, R, generator=g) * (SIZE[1] - 40)
        y1 = torch.rand(n, R, generator=g) * (SIZE[0] - 40)
        w = 16 + torch.rand(n, R, generator=g) * (SIZE[1] * 0.6)
        h = 16 + torch.rand(n, R, generator=g) * (SIZE[0] * 0.6)
        rois = torch.stack([torch.arange(n).float().view(n, 1).expand(n, R), x1, y1,
                            (x1 + w).clamp(max=SIZE[1] - 1), (y1 + h).clamp(max=SIZE[0] - 1)], 2)
        return rois.to(dev)
    model.RFCN_rpn.proposals = fixed


def _snippet(r, dev):
    from dtt.synth import make_batch
    return make_batch(1, SIZE[0], SIZE[1], seed=500 + r, device=dev)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = _setup()
    from dtt.dist import prepare_replica
    dev = torch.device("cuda:0")
    model = _model(cfg, seed=3 + 10 * rank, dev=dev)      # rank 1 starts from other weights: the broadcast must fix that
    _freeze_proposals(model, dev)
    runner = prepare_replica(model, world, channels_last=True)
    np.random.seed(1234 + rank)
    runner.zero_grad(set_to_none=True)
    loss = _loss(runner(*_snippet(rank, dev)))
    loss.backward()
    runner.finish_gradients()
    torch.cuda.synchronize()
    if rank == 0:
        q.put({n: p.grad.detach().float().cpu().numpy() for n, p in model.named_parameters()
               if p.requires_grad and p.grad is not None})
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_sequential_snippets():
    world = 2
    # Warm-up in this process first: on a fresh box the first training step makes MIOpen / hipBLASLt pick (and cache on disk)
    # their kernels; without it the workers run cold and this process warm, with different kernel choices between the two
    # sides of the comparison -- more last-bit differences, more swapped RoIs.
    cfg = _setup()
    from dtt.dist import prepare_replica
    dev = torch.device("cuda:0")
    warm = prepare_replica(_model(cfg, seed=3, dev=dev), 1, channels_last=True)
    np.random.seed(1)
    _loss(warm(*_snippet(0, dev))).backward()
    torch.cuda.synchronize()
    del warm
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    model = _model(cfg, seed=3, dev=dev)
    _freeze_proposals(model, dev)
    runner = prepare_replica(model, 1, channels_last=True)
    runner.zero_grad(set_to_none=True)
    for r in range(world):
        np.random.seed(1234 + r)
        (_loss(runner(*_snippet(r, dev))) / world).backward()
    torch.cuda.synchronize()
    checked, rels = 0, []
    for n, p in model.named_parameters():
        if not p.requires_grad or p.grad is None:
            continue
        assert n in got, n
        ref = p.grad.detach().float().cpu().numpy()
        # With the proposals frozen both sides sample the same RoIs (numpy's generator is seeded per snippet), so what is left
        # between them is the run-to-run rounding of the library convolutions' backward kernels: per tensor, the averaged
        # all-reduced gradient must equal the sequential one to 3e-3 of its norm (fp32 Winograd vs direct algorithm picks of the
        # two processes differ by up to 1e-3 on the 3x3 RPN convolution) -- one bucket left un-reduced, reduced twice
        # or averaged by the wrong count is off by 50 - 100 %.
        a, b = got[n].ravel().astype(np.float64), ref.ravel().astype(np.float64)
        rel = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
        # (trunk tensors many layers below the losses see that rounding amplified through ReLU gates: per cents at most)
        assert rel < (3e-3 if not n.startswith("RFCN_base") else 5e-2), (n, rel, float(np.linalg.norm(b)))
        rels.append(rel)
        checked += 1
    assert float(np.median(rels)) < 2e-2, sorted(rels)[-5:]   # most tensors are trunk tensors (see above); the head / RPN tensors carry the tight bound
    assert checked > 40 and set(got) == {n for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}


# ------------------------------------------------------------------------------------------------------------------
# RCCL on the hardware at hand: a ONE-rank `nccl` process group carrying the real gradient buckets (two ranks cannot share
# a GPU under RCCL, so this is the only form in which RCCL moves the training step's device buffers on a 1-GPU box).
def _nccl_one_rank_worker(port, q, full_size):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    cfg = _setup()
    from dtt.dist import prepare_replica
    from dtt.synth import build_model, calibrate_batchnorm_, make_batch
    dev = torch.device("cuda:0")
    layers, B, H, W = (101, 2, 600, 1067) if full_size else (50, 1, SIZE[0], SIZE[1])
    batch = make_batch(B, H, W, seed=3, device=dev)

    def replica(force):
        model = build_model(layers, cfg=cfg, seed=3).to(dev)
        calibrate_batchnorm_(model, batch[0][:, 0])
        model.train()
        _freeze_proposals(model, dev, (H, W))   # both replicas then sample the same RoIs whatever the last bits of the RPN outputs
        return model, prepare_replica(model, 1, channels_last=True, force_buckets=force)

    def step(runner):
        np.random.seed(1234)
        runner.zero_grad(set_to_none=True)
        _loss(runner(*batch)).backward()
        runner.finish_gradients()
        torch.cuda.synchronize()

    def grads(model):
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}

    m_plain, r_plain = replica(False)
    step(r_plain)                       # throw-away: the libraries pick / cache their kernels on the first step
    step(r_plain); g_a = grads(m_plain)
    step(r_plain); g_b = grads(m_plain)   # same step again: which tensors are run-to-run bit-identical at all?
    m_buck, r_buck = replica(True)
    assert r_buck.bucketed and len(r_buck._buckets) >= 2 and not r_plain.bucketed
    step(r_buck)
    # what autograd hands over for every parameter in THIS step (tensor hooks fire before the bucket hooks): the all-reduce over
    # one rank is the identity, so after finish_gradients every p.grad -- a view into a flat bucket -- must hold exactly these bits
    raw = {}
    handles = [p.register_hook(lambda g, n=n: raw.__setitem__(n, g.detach().clone())) for n, p in m_buck.named_parameters() if p.requires_grad]
    step(r_buck); g_c = grads(m_buck)
    for h in handles:
        h.remove()
    exact = {n: bool(torch.equal(raw[n].reshape(-1), g_c[n].reshape(-1))) if n in raw else None for n in g_c}
    ar_ms = r_buck.time_allreduce_ms(3)
    res = {"n": len(g_a), "keys_equal": set(g_a) == set(g_c), "bucket_bytes": r_buck.bucket_bytes_total(), "exact": exact,
           "n_buckets": len(r_buck._buckets), "allreduce_ms": ar_ms, "backend": dist.get_backend(),
           "in_bucket": all(any(p.grad.data_ptr() >= f.data_ptr() and p.grad.data_ptr() < f.data_ptr() + f.numel() * 4
                                for f, _ in r_buck._buckets)
                            for p in m_buck.parameters() if p.requires_grad and p.grad is not None),
           "rows": []}
    for n in g_a:
        det = bool(torch.equal(g_a[n], g_b[n]))
        same = bool(torch.equal(g_a[n], g_c[n]))
        nrm = float(g_a[n].double().norm())
        rel = float((g_a[n].double() - g_c[n].double()).norm() / max(nrm, 1e-30))
        rel_rr = float((g_a[n].double() - g_b[n].double()).norm() / max(nrm, 1e-30))
        res["rows"].append((n, det, same, rel, rel_rr))
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("full_size", [False, True], ids=["res50_small", "configs3_per_rank_step"])
def test_one_rank_nccl_group_carries_the_gradient_buckets(full_size):
    """force_buckets: hooks + flat buckets + asynchronous all-reduces over a 1-rank RCCL communicator, on the real `_RFCN`
    training step (full_size: BASELINE configs[3]'s per-rank workload, Res-101 600 x 1067, 2 frame pairs).  Every gradient
    tensor that the plain step reproduces bit for bit from run to run must come out of the bucketed step bit-identical
    (an all-reduce over one rank is the identity; a bucket reduced twice, skipped, mis-sliced or averaged is not); tensors the
    libraries' atomics make run-to-run noisy must agree as closely as two plain runs do."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_nccl_one_rank_worker, args=(_free_port(), q, full_size))
    p.start()
    res = q.get()
    p.join(300)
    assert p.exitcode == 0
    assert res["backend"] == "nccl" and res["keys_equal"] and res["in_bucket"] and res["n"] > 40
    assert res["n_buckets"] >= 2 and res["bucket_bytes"] > (150 << 20 if full_size else 50 << 20)
    assert res["allreduce_ms"] is not None and res["allreduce_ms"] > 0
    bad = [n for n, ok in res["exact"].items() if ok is not True]
    assert not bad, ("gradients changed on their way through the buckets / the 1-rank all-reduce", bad[:5])
    n_det = 0
    for n, det, same, rel, rel_rr in res["rows"]:
        if det:
            assert same, ("deterministic tensor differs between the plain and the bucketed step", n, rel)
            n_det += 1
        else:
            # (two replicas: MIOpen may pick another algorithm for the same layer -- fp32 Winograd against direct differ by up to
            #  1e-3 on the 3x3 RPN convolution's gradients; trunk tensors see that amplified through the ReLU gates below)
            assert rel <= max(10 * rel_rr, 3e-3 if not n.startswith("RFCN_base") else 5e-2), (n, rel, rel_rr)
    print("1-rank RCCL: %d tensors bit-identical through the buckets (%d of them run-to-run deterministic in the plain step), "
          "%d buckets / %.1f MB, all-reduce %.3f ms" % (res["n"], n_det, res["n_buckets"], res["bucket_bytes"] / 1e6, res["allreduce_ms"]))
