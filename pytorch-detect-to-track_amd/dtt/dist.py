"""Per-snippet data parallelism: one process per GPU, gradients (only) all-reduced over RCCL / xGMI.

Replaces the reference's single-process `nn.DataParallel` (trainval_net.py:310-311), which re-broadcasts all
~55 M parameters GPU0 -> others every step and reduces gradients onto GPU 0.  Here every rank keeps its own
replica (identical after the one broadcast at construction), runs its shard of video snippets -- both frames
of a pair always stay on one GPU, exactly as DataParallel's dim-0 scatter did -- and the only collective is a
bucketed gradient all-reduce that overlaps with the rest of backward:
  * trainable gradients are gathered into a few flat fp32 buckets (one multi-tensor copy each; default 32 MiB:
    on 8 GPUs a ring all-reduce is bound by one xGMI link (~153 GB/s), so a 32 MiB bucket costs ~0.4 ms -- large enough to amortise the
    launch, small enough that the first bucket is on the wire while layer3/layer2 are still in backward);
  * buckets are filled in reverse registration order (the order autograd produces gradients) and each one
    is all-reduced asynchronously as soon as its last gradient has been accumulated;
  * `finish_gradients()` waits for the outstanding handles and divides by the world size.
Inference needs no communication at all (results stay per rank).
With world_size == 1 everything degenerates to a plain module call -- unless `force_buckets` is set: then the hooks, the flat
buckets and the asynchronous all-reduces run exactly as on N ranks (over a 1-rank process group when one is initialised).
That is how the RCCL path is exercised on a single-GPU box: two ranks cannot share one GPU under RCCL, a 1-rank `nccl`
group can carry the buckets (tests/test_gpu_dist.py, `bench.py --mode train` -> `allreduce_ms`).
"""
import torch
import torch.distributed as dist
from torch import nn


def _bucket_view(flat, off, p):
    """Slice of a flat bucket shaped and laid out like p (channels-last parameters get a channels-last view, so
    gradient accumulation and the multi-tensor optimizer never mix layouts)."""
    piece = flat[off:off + p.numel()]
    if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
        n, c, h, w = p.shape
        return piece.view(n, h, w, c).permute(0, 3, 1, 2)
    return piece.view_as(p)


class DataParallelSnippets(nn.Module):
    def __init__(self, module, world_size=None, bucket_bytes=32 << 20, process_group=None, force_buckets=False):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.bucketed = self.world > 1 or bool(force_buckets)
        self._handles = []
        self._buckets = []
        self._views = []
        self._pending = {}
        if self.world > 1:
            self._sync_initial_state()
        if self.bucketed:
            self._build_buckets(bucket_bytes)

    # ------------------------------------------------------------------ setup
    def _sync_initial_state(self):
        with torch.no_grad():
            seen = set()
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                if t.data_ptr() in seen or not t.is_floating_point():
                    continue
                seen.add(t.data_ptr())
                dist.broadcast(t.data, src=0, group=self.group)

    def _build_buckets(self, bucket_bytes):
        params, seen = [], set()
        for p in self.module.parameters():
            if p.requires_grad and id(p) not in seen:  # RFCN_net is registered under two names
                seen.add(id(p))
                params.append(p)
        params.reverse()  # gradients arrive roughly in reverse registration order
        cur, cur_bytes = [], 0
        groups = []
        for p in params:
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for bi, grp in enumerate(groups):
            flat = torch.zeros(sum(p.numel() for p in grp), dtype=grp[0].dtype, device=grp[0].device)
            for p in grp:
                p.grad = None
                p.register_post_accumulate_grad_hook(self._make_hook(bi))
            self._buckets.append((flat, grp))
            self._views.append(None)
        self._reset_pending()

    def _reset_pending(self):
        self._pending = {bi: len(grp) for bi, (_, grp) in enumerate(self._buckets)}

    def _bucket_views(self, bi):
        """Per-parameter views into bucket bi, laid out like the parameters (rebuilt if a layout changed)."""
        flat, grp = self._buckets[bi]
        cached = self._views[bi]
        if cached is None or any(v.stride() != p.stride() for v, p in zip(cached, grp)):
            cached, off = [], 0
            for p in grp:
                cached.append(_bucket_view(flat, off, p))
                off += p.numel()
            self._views[bi] = cached
        return cached

    def _fill_and_reduce(self, bi):
        """Gather the bucket's gradients (autograd hands each one over as a fresh tensor: no zeroing pass, no add
        per parameter) with one multi-tensor copy, re-point .grad at the bucket and put it on the wire."""
        flat, grp = self._buckets[bi]
        views = self._bucket_views(bi)
        dst, src = [], []
        for p, v in zip(grp, views):
            if p.grad is None:
                v.zero_()  # unused this step
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)
        if dist.is_initialized():   # (force_buckets without a process group: buckets and hooks only)
            self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _make_hook(self, bucket_index):
        def hook(param):
            self._pending[bucket_index] -= 1
            if self._pending[bucket_index] == 0:
                self._fill_and_reduce(bucket_index)
        return hook

    # ------------------------------------------------------------------ step API
    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def zero_grad(self, set_to_none=True):
        """Drop the gradients (default) so that backward hands fresh tensors over instead of running one add per
        parameter; set_to_none=False keeps the tensors and zeroes them."""
        if not self.bucketed:
            self.module.zero_grad(set_to_none=set_to_none)
            return
        for flat, grp in self._buckets:
            if set_to_none:
                for p in grp:
                    p.grad = None
            else:
                flat.zero_()

    def finish_gradients(self):
        """Wait for the in-flight bucket all-reduces and average.  Parameters that received no gradient this
        step (unused branches) still have their bucket reduced so all ranks stay in lockstep."""
        if not self.bucketed:
            return
        for bi, left in self._pending.items():
            if left > 0:
                self._fill_and_reduce(bi)
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.world > 1:
            inv = 1.0 / self.world
            for flat, _ in self._buckets:
                flat.mul_(inv)
        self._reset_pending()

    def bucket_bytes_total(self):
        return sum(flat.numel() * flat.element_size() for flat, _ in self._buckets)

    def time_allreduce_ms(self, repeats=5):
        """The bucket all-reduces alone (every flat bucket once, back to back, as finish_gradients waits for them), median of
        `repeats`; None without buckets or a process group.  Leaves the buckets' values untouched up to the sum over ranks
        being re-applied -- call it outside a training step."""
        if not self._buckets or not dist.is_initialized():
            return None
        import time
        dev = self._buckets[0][0].device
        saved = [flat.clone() for flat, _ in self._buckets]
        times = []
        for _ in range(repeats):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            hs = [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for flat, _ in self._buckets]
            for h in hs:
                h.wait()
            torch.cuda.synchronize(dev)
            times.append((time.perf_counter() - t0) * 1e3)
        for (flat, _), s in zip(self._buckets, saved):
            flat.copy_(s)
        return sorted(times)[len(times) // 2]

    def state_dict(self, *a, **k):  # checkpoints hold the bare module's keys (trainval_net.py:422 unwraps too)
        return self.module.state_dict(*a, **k)


def isolate_library_caches(local_rank, world, env=None):
    """One process per GPU means N processes that each let MIOpen search its convolution algorithms on the first step.  MIOpen keeps
    what it finds in a per-user sqlite find-db and a kernel cache under $HOME: N ranks writing the same files lock each other out for
    the whole warm-up (or read a half-written entry).  Give every rank its own MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR --
    must run before the first convolution of the process (the library reads the variables when its handle is created).  A user's
    own setting wins.  (hipBLASLt candidate timing, dtt_gemm_tune, keeps its picks in process memory: nothing is shared.)

    The directory holds code objects the library will LOAD, so it lives under the user's own cache directory
    ($DTT_CACHE_ROOT, else $XDG_CACHE_HOME/dtt, else ~/.cache/dtt -- not a world-writable /tmp), is created with mode 0700 and is
    refused when it (or its root) belongs to somebody else or is writable by group / others.  Its name carries the job
    (MASTER_PORT / TORCHELASTIC_RUN_ID: two jobs of one user on a node do not meet) and the rank the LAUNCHER gave -- pass
    LOCAL_RANK as it came, before any modulo onto shared devices.
    Returns the directory used, or None when nothing was changed (one rank, or both variables already set)."""
    import os
    import stat
    env = os.environ if env is None else env
    if world <= 1 or ("MIOPEN_USER_DB_PATH" in env and "MIOPEN_CUSTOM_CACHE_DIR" in env):
        return None
    root = env.get("DTT_CACHE_ROOT") or os.path.join(env.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "dtt")
    job = "".join(ch for ch in str(env.get("TORCHELASTIC_RUN_ID") or env.get("MASTER_PORT") or "job") if ch.isalnum() or ch in "-_")[:48] or "job"
    base = os.path.join(root, "miopen_%s_rank%d" % (job, local_rank))

    def private_dir(d):
        os.makedirs(d, mode=0o700, exist_ok=True)
        st = os.stat(d)
        if st.st_uid != os.getuid() or not stat.S_ISDIR(st.st_mode):
            raise PermissionError("isolate_library_caches: %s is not a directory owned by uid %d" % (d, os.getuid()))
        if st.st_mode & 0o022:
            os.chmod(d, st.st_mode & ~0o077 & 0o7777)      # ours, but left open by an earlier umask: close it
            if os.stat(d).st_mode & 0o022:
                raise PermissionError("isolate_library_caches: %s is writable by group / others" % d)
        return d
    private_dir(root)
    private_dir(base)
    for key, sub in (("MIOPEN_USER_DB_PATH", "db"), ("MIOPEN_CUSTOM_CACHE_DIR", "cache")):
        if key not in env:
            env[key] = private_dir(os.path.join(base, sub))
    return base


def broadcast_module_state(module, process_group=None, src=0):
    """Rank `src`'s parameters and buffers (BatchNorm statistics included) to every rank.  Must run BEFORE anything
    that snapshots them into constants (dtt.fuse.fuse_for_training folds the frozen-BatchNorm statistics and the frozen
    stem / stage weights): replicas that fold different statistics compute different functions for the whole run."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return module
    with torch.no_grad():
        seen = set()
        for t in list(module.parameters()) + list(module.buffers()):
            if t.data_ptr() in seen or not t.is_floating_point():
                continue
            seen.add(t.data_ptr())
            dist.broadcast(t.data, src=src, group=process_group)
    return module


def prepare_replica(model, world, channels_last=True, process_group=None, force_buckets=False):
    """The start-up order of a training replica: broadcast rank 0's state -> fold the frozen constants
    (fuse_for_training, which also moves the trainable filters to channels-last memory) -> lay out the gradient buckets
    (DataParallelSnippets) after that.  Returns the runner."""
    from .fuse import fuse_for_training
    if world > 1:
        broadcast_module_state(model, process_group)
    fuse_for_training(model, channels_last=channels_last)
    return DataParallelSnippets(model, world, process_group=process_group, force_buckets=force_buckets)


def shard_snippets(n_snippets, rank, world):
    """Contiguous, balanced split of snippet indices [0, n) across ranks (DataParallel scatter semantics)."""
    base, rem = divmod(n_snippets, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def make_optimizer(model, cfg, lr=None, optimizer="sgd"):
    """Per-parameter groups of trainval_net.py:280-294: biases get lr*(DOUBLE_BIAS+1) and no weight decay
    unless BIAS_DECAY."""
    T = cfg.TRAIN
    lr = T.LEARNING_RATE if lr is None else lr
    groups, seen = [], set()
    for name, p in model.named_parameters():
        if not p.requires_grad or id(p) in seen:
            continue
        seen.add(id(p))
        if "bias" in name:
            groups.append({"params": [p], "lr": lr * (T.DOUBLE_BIAS + 1),
                           "weight_decay": T.WEIGHT_DECAY if T.BIAS_DECAY else 0})
        else:
            groups.append({"params": [p], "lr": lr, "weight_decay": T.WEIGHT_DECAY})
    if optimizer == "adam":
        # the reference scales only the lr it PRINTS by 0.1 (trainval_net.py:290-292); the groups keep the full rate
        return torch.optim.Adam(groups)
    return GroupedSGD(groups, momentum=T.MOMENTUM)


def _grad_like_param(p):
    """p.grad with exactly p's strides.  The multi-tensor kernels take their fast path only when every (param, grad,
    buffer) triple has identical strides; MIOpen returns 1x1 filter gradients with channels-last flavoured strides
    on size-1 dimensions (same memory, different stride tuple), which would silently send the whole list down the
    one-kernel-per-tensor path."""
    g = p.grad
    if g.stride() == p.stride():
        return g
    if all(sz == 1 or a == b for sz, a, b in zip(p.shape, g.stride(), p.stride())):
        return g.as_strided(p.shape, p.stride(), g.storage_offset())  # same memory order: a relabelling
    return torch.empty_like(p).copy_(g)


class GroupedSGD(torch.optim.SGD):
    """torch.optim.SGD semantics and state_dict layout (one param group per parameter, as the reference builds them),
    but `step()` batches all parameters that share (lr, weight_decay, momentum) into the same multi-tensor launches:
    ~10 kernels per step instead of three per parameter (~950 for ResNet-101)."""

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        buckets = {}
        for g in self.param_groups:
            if g.get("nesterov") or g.get("dampening", 0) != 0 or g.get("maximize"):
                return super().step()  # not used by the D&T recipe
            key = (float(g["lr"]), float(g["weight_decay"]), float(g["momentum"]))
            for p in g["params"]:
                if p.grad is not None:
                    buckets.setdefault(key, []).append(p)
        for (lr, wd, mom), ps in buckets.items():
            grads = [_grad_like_param(p) for p in ps]
            if wd != 0:
                grads = torch._foreach_add(grads, ps, alpha=wd)
            if mom != 0:
                have, fresh = [], []
                for p, g in zip(ps, grads):
                    st = self.state[p]
                    if st.get("momentum_buffer") is None:
                        st["momentum_buffer"] = torch.clone(g).detach()  # first step: buf = g
                        fresh.append(p)
                    else:
                        have.append((st["momentum_buffer"], g))
                if have:
                    bufs = [b for b, _ in have]
                    torch._foreach_mul_(bufs, mom)
                    torch._foreach_add_(bufs, [g for _, g in have])
                grads = [self.state[p]["momentum_buffer"] for p in ps]
            torch._foreach_add_(ps, grads, alpha=-lr)
        return loss
