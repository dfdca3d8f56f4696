"""The N > 1 path on CPU: world_size-2 gloo processes running DataParallelSnippets (bucketed gradient
all-reduce, per-snippet sharding) must reproduce the single-process gradients of the full batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(3, 8, 3, padding=1)
        self.b = nn.Conv2d(8, 8, 1)
        self.shared = nn.Conv2d(8, 4, 1)
        self.alias = self.shared            # one module under two names, like RFCN_net
        self.unused = nn.Linear(4, 4)       # never receives a gradient
        self.frozen = nn.Conv2d(4, 4, 1)
        for p in self.frozen.parameters():
            p.requires_grad = False

    def forward(self, x):
        return self.frozen(self.alias(torch.relu(self.b(torch.relu(self.a(x)))))).mean(dim=(1, 2, 3))


def _worker(rank, world, port, q, n_snip=6):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "pytorch-detect-to-track_amd")]
    from dtt.dist import DataParallelSnippets, shard_snippets
    torch.manual_seed(100 + rank)  # different init per rank: the wrapper must broadcast rank 0's weights
    model = Toy()
    dp = DataParallelSnippets(model, world, bucket_bytes=1024)  # tiny buckets -> several all-reduces
    g = torch.Generator().manual_seed(0)
    data = torch.randn(n_snip, 3, 8, 8, generator=g)
    mine = list(shard_snippets(n_snip, rank, world))
    grads = []
    for step in range(2):  # two steps: buckets must be reusable
        dp.zero_grad(set_to_none=True)
        loss = dp(data[mine]).sum() / float(n_snip) * world   # per-rank mean convention: sum/N_global * world -> avg over ranks
        loss.backward()
        dp.finish_gradients()
        grads.append({n: p.grad.clone().numpy() for n, p in model.named_parameters()
                      if p.requires_grad and p.grad is not None})
    w0 = {n: p.detach().clone().numpy() for n, p in model.named_parameters()}
    if rank == 0:
        q.put((w0, grads))  # numpy: pickled by value (tensors travel as fds that die with this process)
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,n_snip", [(2, 6), (8, 16)], ids=["two_ranks", "eight_ranks_global_batch_16"])
def test_rank_gradients_match_single_process(world, n_snip):
    """world 8 / 16 snippets = BASELINE configs[3]'s layout (8 GPUs, global batch 16, two snippets per rank): what the 8-GPU node runs
    over RCCL, here over gloo on CPU."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_snip)) for r in range(world)]
    for p in procs:
        p.start()
    w0, grads = q.get()
    w0 = {n: torch.from_numpy(v) for n, v in w0.items()}
    grads = [{n: torch.from_numpy(v) for n, v in g.items()} for g in grads]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    torch.manual_seed(100)  # rank 0's initial weights
    ref = Toy()
    for n, p in ref.named_parameters():
        assert torch.equal(p.detach(), w0[n]), n
    g = torch.Generator().manual_seed(0)
    data = torch.randn(n_snip, 3, 8, 8, generator=g)
    (ref(data).sum() / float(n_snip)).backward()
    for step in range(2):
        for n, p in ref.named_parameters():
            if p.requires_grad and p.grad is not None and n in grads[step]:
                torch.testing.assert_close(grads[step][n], p.grad, rtol=1e-5, atol=1e-6)
    assert "unused.weight" in grads[0] and float(grads[0]["unused.weight"].abs().max()) == 0.0


PORT = _free_port()


def _fold_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "pytorch-detect-to-track_amd")]
    from dtt.config import cfg
    from dtt.dist import prepare_replica
    from dtt.synth import build_model
    model = build_model(50, cfg=cfg, seed=3 + rank)       # different weights per rank ...
    with torch.no_grad():                                 # ... and different "calibrated" BatchNorm statistics per rank
        gen = torch.Generator().manual_seed(50 + rank)
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen))
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
    model.train()
    prepare_replica(model, world, channels_last=True)
    ft = model._fused_train_trunk
    sig = [float(t.double().sum()) for t in ft.scales + ft.shifts]
    sig += [float(ft.stem.w.double().sum()), float(ft.stem.b.double().sum())]
    sig += [float(blk.c1.w.double().sum()) for stage in ft.frozen for blk in stage]
    sig += [float(p.detach().double().sum()) for p in model.parameters()]
    if rank == 0:
        ref = build_model(50, cfg=cfg, seed=3)
        q.put((sig, float(sum(p.detach().double().sum() for p in ref.parameters()))))
    else:
        q.put((sig, None))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_fold_identical_constants():
    """dtt.dist.prepare_replica: rank 0's weights and BatchNorm statistics reach every rank BEFORE fuse_for_training
    snapshots them (frozen-BatchNorm scales / shifts, folded stem and frozen-stage weights), so all replicas compute the
    same function; the per-rank statistics they started with are gone."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_fold_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get() for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    sigs = [g[0] for g in got]
    assert sigs[0] == sigs[1] and len(sigs[0]) > 100


def test_force_buckets_at_world_one_matches_the_plain_step():
    """DataParallelSnippets(force_buckets=True) at world size 1 (what bench.py's secondary.train_step and the 1-rank RCCL test on the
    GPU box run): the hooks and flat buckets of the N-rank path without a process group -- gradients bit-identical to the plain
    step over two steps (buckets are reused), views laid out in the buckets, unused parameters zero-filled, no averaging."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "pytorch-detect-to-track_amd")]
    from dtt.dist import DataParallelSnippets
    assert not dist.is_initialized()
    g = torch.Generator().manual_seed(0)
    data = torch.randn(6, 3, 8, 8, generator=g)
    torch.manual_seed(5)
    plain, bucketed = Toy(), Toy()
    bucketed.load_state_dict(plain.state_dict())
    dp0 = DataParallelSnippets(plain, 1)
    dp1 = DataParallelSnippets(bucketed, 1, bucket_bytes=1024, force_buckets=True)
    assert not dp0.bucketed and dp1.bucketed and len(dp1._buckets) >= 2 and dp1.bucket_bytes_total() > 0
    assert dp1.time_allreduce_ms() is None                     # no process group: nothing to time
    for step in range(2):
        for dp in (dp0, dp1):
            dp.zero_grad(set_to_none=True)
            dp(data).sum().backward()
            dp.finish_gradients()
        for (n, p), (_, q) in zip(plain.named_parameters(), bucketed.named_parameters()):
            if not p.requires_grad:
                continue
            if p.grad is None:                                 # never used: the bucketed path hands the optimizer zeros
                assert q.grad is not None and float(q.grad.abs().max()) == 0.0, n
            else:
                assert torch.equal(p.grad, q.grad), (n, step)
                assert any(q.grad.data_ptr() >= f.data_ptr() and q.grad.data_ptr() < f.data_ptr() + f.numel() * 4 for f, _ in dp1._buckets), n


def test_isolate_library_caches_gives_every_rank_its_own_miopen_paths(tmp_path):
    """N processes on N GPUs must not share MIOpen's per-user find-db / kernel cache during the first step's algorithm search."""
    from dtt.dist import isolate_library_caches
    envs = []
    for r in range(3):
        e = {"DTT_CACHE_ROOT": str(tmp_path)}
        base = isolate_library_caches(r, 3, env=e)
        assert base and os.path.isdir(e["MIOPEN_USER_DB_PATH"]) and os.path.isdir(e["MIOPEN_CUSTOM_CACHE_DIR"])
        envs.append(e)
    assert len({e["MIOPEN_USER_DB_PATH"] for e in envs}) == 3 and len({e["MIOPEN_CUSTOM_CACHE_DIR"] for e in envs}) == 3
    # private (0700, ours), per job: another MASTER_PORT gets other directories; a directory that is open to others is closed or refused
    import stat
    assert all(stat.S_IMODE(os.stat(e[k]).st_mode) & 0o077 == 0 for e in envs for k in ("MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR"))
    other = {"DTT_CACHE_ROOT": str(tmp_path), "MASTER_PORT": "29511"}
    isolate_library_caches(0, 3, env=other)
    assert other["MIOPEN_USER_DB_PATH"] not in {e["MIOPEN_USER_DB_PATH"] for e in envs} and "29511" in other["MIOPEN_USER_DB_PATH"]
    loose = tmp_path / "loose"
    loose.mkdir()
    os.chmod(loose, 0o777)
    isolate_library_caches(0, 2, env={"DTT_CACHE_ROOT": str(loose)})
    assert stat.S_IMODE(os.stat(loose).st_mode) & 0o022 == 0                                  # our own directory, left open: closed
    assert isolate_library_caches(0, 1, env={}) is None                                     # one rank: nothing to isolate
    mine = {"MIOPEN_USER_DB_PATH": "/x", "MIOPEN_CUSTOM_CACHE_DIR": "/y"}
    assert isolate_library_caches(1, 8, env=mine) is None and mine["MIOPEN_USER_DB_PATH"] == "/x"   # the user's setting wins
