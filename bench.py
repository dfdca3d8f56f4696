#!/usr/bin/env python3
"""bench.py -- frame-pairs/sec of the Res-101 Detect-to-Track forward at 600 px on N MI355X (one process
per GPU), with the roofline of the dominant hot-path kernel and a CPU baseline.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # launches its own N ranks (below) when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one D&T inference forward over one synthetic batch of `--batch` frame pairs per GPU
(BASELINE.json configs[2]: Res-101 D&T siamese 2-frame, 600 px, correlation d=8 + PSRoI, bs=2, 1 x MI355X);
the batch is resident in HBM before the timed region.  Per GPU work is fixed as N grows (weak scaling); the
inference path shards by video snippet with no collective.  `--mode train` times a
full training step (forward + backward + SGD, gradients all-reduced over RCCL).

The one JSON line carries, besides the contract fields:
  roofline      the dominant hot-path op (conv5 cross-frame correlation: exact-f32 MFMA banded product + slice
                reduction, timed as ONE op): algorithmic FLOPs (and bytes) per op / its duration measured with HIP
                events on the launch stream inside the timed region (C-ABI hook dtt_profile_attach); both the
                MFMA and the HBM fraction
  secondary     the hand-written R-FCN heads (MFMA) and the class PSRoI pooling (HBM) of the same step
  cpu_baseline  the same forward for ONE frame pair through stock PyTorch CPU convolutions + the CPU
                oracle of every hot-path op (oracle/cpu_graph.py), on this box's host cores
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="frame pairs per GPU per step")
    ap.add_argument("--mode", choices=("infer", "train"), default="infer")
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=1067)
    ap.add_argument("--layers", type=int, default=101)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--disp", type=int, default=8, help="correlation max displacement (8 = reference; 16 = BASELINE config 5)")
    ap.add_argument("--nchw-trunk", action="store_true", help="inference: keep the fused trunk in NCHW (A/B against channels-last)")
    ap.add_argument("--pooling", choices=("psroi", "align", "pool", "crop"), default="psroi",
                    help="psroi = the reference's R-FCN graph; align / pool / crop additionally pool the 512-channel top map "
                         "per RoI (BASELINE config 5: --pooling align --disp 16 --height 563 --width 1000 --batch 1)")
    ap.add_argument("--cpu-passes", type=int, default=10, help="timed passes of the CPU baseline (median is reported)")
    ap.add_argument("--frames", type=int, choices=(1, 2), default=2,
                    help="2 = the D&T frame pair; 1 = single-frame R-FCN (BASELINE configs[1]: no correlations, no tracking head)")
    ap.add_argument("--no-train-step", action="store_true",
                    help="inference mode: skip the short training-step measurement reported as secondary.train_step")
    ap.add_argument("--train-steps", type=int, default=5, help="timed steps of secondary.train_step (after 3 warm-up steps)")
    return ap.parse_args()


class KernelTimer:
    """HIP events recorded by libdtt_hip.so around every launch of one kernel (dtt_profile_attach)."""

    def __init__(self, tag, n, device):
        from dtt import _lib
        self.lib = _lib.lib()
        self.tag = tag
        self.n = n
        self.begin = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        self.end = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        s = torch.cuda.current_stream(device)
        for e in self.begin + self.end:
            e.record(s)  # forces creation of the underlying hipEvent_t
        torch.cuda.synchronize(device)
        self._b = (ctypes.c_void_p * n)(*[e.cuda_event for e in self.begin])
        self._e = (ctypes.c_void_p * n)(*[e.cuda_event for e in self.end])

    def attach(self):
        self.lib.dtt_profile_attach(self.tag.encode(), self._b, self._e, self.n)

    def detach(self):
        used = self.lib.dtt_profile_count()
        self.lib.dtt_profile_attach(None, None, None, 0)
        return used

    def durations_us(self, used):
        return [self.begin[i].elapsed_time(self.end[i]) * 1e3 for i in range(used)]

    @staticmethod
    def event_pair_overhead_us(device, n=50):
        """What an empty begin/end event bracket measures on this stream (subtract from a bracketed kernel to compare
        with rocprofv3's kernel durations; `launch_us` itself is reported raw, i.e. conservatively)."""
        s = torch.cuda.current_stream(device)
        b = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        e = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        for i in range(n):
            b[i].record(s)
            e[i].record(s)
        torch.cuda.synchronize(device)
        d = sorted(b[i].elapsed_time(e[i]) * 1e3 for i in range(n))
        return d[n // 2]


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, cfg):
    """The same forward for ONE frame pair through the CPU graph (stock PyTorch fp32 convolutions + the oracle's OpenMP
    restatement of every hot-path op, oracle/cpu_graph.py) on this box's host cores -- SURVEY 8d's protocol on a bounded
    sample: the thread count is swept (one pass each), then three warm-up and `--cpu-passes` (10) timed passes at the best
    count; the median is reported.  Context, not the target: the reference has no CPU path for these ops at all."""
    from dtt.synth import build_model, calibrate_batchnorm_, make_batch
    from oracle import cpu_graph, oracle_lib
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    m = build_model(args.layers, cfg=cfg).eval()
    im, info, _, _ = make_batch(1, args.height, args.width, seed=3)
    calibrate_batchnorm_(m, im[:, 0])
    if args.frames == 1:
        im, info = im[:, :1].contiguous(), info[:, :1].contiguous()

    def one(threads):
        torch.set_num_threads(threads)
        oracle_lib.set_num_threads(threads)
        t0 = time.perf_counter()
        cpu_graph.rfcn_forward_test(m, im, info, cfg)
        return time.perf_counter() - t0

    one(min(avail, 32))                                        # untimed: page in weights, OpenMP teams, oneDNN primitives
    sweep = {}
    for t in sorted({t for t in (8, 16, 32, 64) if t <= avail}):   # beyond 64 the single-image convolutions only get slower
        sweep[t] = one(t)
    best = min(sweep, key=sweep.get)
    for _ in range(3):                                         # warm-up at the chosen count (3 passes)
        one(best)
    times = sorted(one(best) for _ in range(max(1, args.cpu_passes)))
    med = times[len(times) // 2]
    return {"value": round(1.0 / med, 4), "unit": "frame-pairs/s" if args.frames == 2 else "frames/s", "cores": int(best), "host_cores": int(avail),
            "cpu_model": _cpu_model(), "kind": "port",
            "passes": len(times), "pass_seconds": [round(t, 3) for t in times],
            "thread_sweep_seconds": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": "1 frame%s (B=1, %dx%d, Res-%d D&T test forward): torch CPU fp32 convs + oracle/ ops (OpenMP over "
                      "independent outputs); thread count swept, 3 warm-up + %d timed passes at %d threads, median %.2f s"
                      % (" pair" if args.frames == 2 else "", args.height, args.width, args.layers, len(times), best, med)}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: re-run this script as N ranks (one process per GPU) under
    torch.distributed.run on 127.0.0.1 -- what replaces the reference's single-process nn.DataParallel
    (trainval_net.py:310-311).  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL / cross-process device buffers
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def max_over_ranks(elapsed, world, dev, backend):
    """(max over ranks, [every rank's value]) of a rank-local wall time: the line reports the slowest rank (the contract) and shows
    the spread, so a straggler -- one rank still timing library candidates, a GPU with a neighbour's job on it -- is visible."""
    if world <= 1:
        return elapsed, [elapsed]
    t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    vals = [float(v.item()) for v in every]
    return max(vals), vals


def measured_traffic(stem):
    """The newest profiles/r*_<stem>.json (written by tools/profile_round.sh from separate rocprofv3 --pmc passes) whose
    `library_sha256` is the sha256 of the libdtt_hip.so loaded here -> (json, file name); (None, reason) when there is none.
    Counter-derived HBM bytes are quoted only for the binary they were measured on."""
    import glob
    import hashlib
    from dtt import _lib
    try:
        sha = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
    except OSError:
        return None, "libdtt_hip.so not readable"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_%s.json" % stem)), reverse=True)
    for f in files:
        try:
            pmc = json.load(open(f))
        except (OSError, ValueError):
            continue
        if pmc.get("library_sha256") == sha:
            return pmc, "profiles/" + os.path.basename(f)
    return None, ("%s was measured on another libdtt_hip.so build: not quoted (rerun tools/profile_round.sh)" % ("profiles/" + os.path.basename(files[0]))
                  if files else "no profiles/r*_%s.json" % stem)


def corr_bwd_secondary(op_us, args, model):
    """secondary.corr_bwd from the mean durations of a step's three correlation gradient ops (event tag corr_bwd_op), in launch
    order.  The hand-written training graph runs them inside one autograd node in map order (conv3, conv4, conv5:
    dtt.heads.TrackingRowsFn.backward); the library graph has three correlation nodes that autograd runs in reverse creation order."""
    H16, W16 = -(-args.height // 16), -(-args.width // 16)
    H8, W8 = -(-args.height // 8), -(-args.width // 8)
    ops = (("corr3_bwd", 512, args.disp // 2, H8 * W8), ("corr4_bwd", 1024, args.disp, H16 * W16), ("corr5_bwd", 2048, args.disp, H16 * W16))
    if not getattr(model, "_train_pm", False):
        ops = ops[::-1]
    pmc, pmc_src = measured_traffic("pmc_corr_bwd")
    shape = (args.batch, args.height, args.width, args.disp)
    # counters are taken at two shapes (tools/profile_round.sh): BASELINE configs[2] / [3]'s and, with the d16_ prefix, configs[4]'s per rank
    prefix = "" if shape == (2, 600, 1067, 8) else "d16_" if shape == (1, 563, 1000, 16) else None
    same_shape = prefix is not None
    bw = {}
    for us, (name, C, R, HWc) in zip(op_us, ops):
        D2 = (2 * R + 1) ** 2
        fl = 2.0 * 2.0 * C * D2 * H16 * W16 * args.batch      # both gradients: 2 x the forward's FLOPs
        # both gradient maps written once (whole maps: conv3's non-lattice pixels get their zeros from the same kernels), the LATTICE
        # pixels of both maps read once (conv3: every second pixel in both directions), gradOut read once
        by = (2 * C * HWc * 4 + 2 * C * H16 * W16 * 4 + D2 * H16 * W16 * 4) * args.batch
        mf, hb = fl / (us * 1e-6) / 1e12, by / (us * 1e-6) / 1e9
        # conv3 moves 166 MB for 1.1 GFLOP per gradient pair: its bound is HBM; conv4 / conv5 are bound by the fp32 MFMA rate
        bound = "hbm" if hb / HBM_PEAK_GBS > mf / FP32_MFMA_PEAK_TFLOPS else "mfma"
        e = {"op_us": round(us, 2), "bound": bound,
             "achieved": round(mf if bound == "mfma" else hb, 2), "peak": FP32_MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS,
             "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
             "frac": round(mf / FP32_MFMA_PEAK_TFLOPS if bound == "mfma" else hb / HBM_PEAK_GBS, 4),
             "mfma": {"achieved": round(mf, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(mf / FP32_MFMA_PEAK_TFLOPS, 4)},
             "hbm": {"achieved": round(hb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hb / HBM_PEAK_GBS, 4)},
             "algorithmic_flops_per_op": fl, "algorithmic_bytes_per_op": by, "traffic": None}
        m = (pmc or {}).get((prefix or "") + name[:5].replace("corr", "conv"))
        if m and same_shape:
            e["traffic"] = m["traffic_bytes_per_op"]
        bw[name] = e
    bw["traffic_source"] = (pmc_src + " (same libdtt_hip.so: sha256 checked; FETCH_SIZE x 2 + WRITE_SIZE over the op's launches)"
                            if pmc and same_shape else (pmc_src if not pmc else
                                                        "counters were taken at B=2 600x1067 d=8 and B=1 563x1000 d=16: not quoted for this shape"))
    return dict(bw, kernel="correlation gradient op = corr_bwd_band_kernel + ONE corr_bwd_stream_kernel launch (both gradients of one correlation "
                           "as one grid, the four window quarters of a radius 9 - 16 window inside it; channels-last, band-stationary / "
                           "halo-streamed: dtt_correlation_backward_nhwc_phase; event tag corr_bwd_op; DTT_CORR_BWD_OVERLAP=1 lays the bands of "
                           "the later ops out on a second stream: shorter ops, a longer step -- off)")


def rank_zero():
    return int(os.environ.get("RANK", "0")) == 0


def train_hot_path_secondary(step, args, dev, n=5):
    """secondary.train_step.hot_path: the hand-written NON-MFMA ops of the training step, each timed by the
    library's event hook over `n` more (untimed) steps -- one tag at a time, events on the launch stream -- with the algorithmic bytes
    of SURVEY 8(d) per step and the fraction of the 8 TB/s HBM peak they amount to.  All are far from it: these are latency- and
    dependency-bound ops on small data (the proposal layer's sort and greedy sweep are serial by construction); the figure says how
    far, the time says what it costs."""
    legs, B = args.frames, args.batch
    imgs = legs * B
    H16, W16 = -(-args.height // 16), -(-args.width // 16)
    K, A = H16 * W16, 12
    n_anchor = K * A
    N = 12000                                        # TRAIN.RPN_PRE_NMS_TOP_N (cfgs/res101.yml)
    mask = N * (-(-N // 64)) * 8
    od_cls, od_loc = 31, 4
    pm_floats = 49 * 32 + 49 * 4                      # position-major map floats per pixel: class bins (padded to 32) + box bins
    rois = 128 * imgs                                 # TRAIN.BATCH_SIZE RoIs per image
    specs = [
        # tag, algorithmic bytes per step, what they are
        ("proposal_op", imgs * ((24 + 48) * K * 4 + n_anchor * 5 * 4 + N * 16 + 2 * mask),
         "scores + deltas read, decoded boxes written (SURVEY 8d: 0.73 + 0.61 MB per image) + the NMS below"),
        ("nms_op", imgs * (N * 16 + 2 * mask), "N = 12000 boxes read, the 18.05 MB bit matrix written and read once per image (SURVEY 8d)"),
        ("anchor_target_op", imgs * (14 * n_anchor * 4), "keys read; labels, targets and the two weight maps written (13 floats per anchor)"),
        ("rpn_loss", imgs * (19 * n_anchor * 4), "cls_prob, labels, bbox_pred, targets and both weight maps read once"),
        ("psroi_pm_bwd", imgs * K * pm_floats * 4 + (B * K * 49 * 4 * 4 if legs == 2 else 0) + rois * (od_cls + od_loc) * 4,
         "the position-major gradient maps written once (class + box bins of every pixel of both legs; the tracking head's box bins) + "
         "the vote gradients read"),
    ]
    out = {}
    for tag, nbytes, what in specs:
        k = KernelTimer(tag, 64 * n, dev)
        k.attach()
        for _ in range(n):
            step()
        torch.cuda.synchronize(dev)
        d = k.durations_us(k.detach())
        if not d or len(d) % n:
            continue
        per = len(d) // n
        us = sorted(sum(d[i * per:(i + 1) * per]) for i in range(n))[n // 2]      # median step
        out[tag] = {"us_per_step": round(us, 2), "ops_per_step": round(len(d) / n, 2), "algorithmic_bytes_per_step": int(nbytes),
                    "bound": "hbm", "achieved": round(nbytes / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "bytes": what}
    out["note"] = ("event tags of libdtt_hip.so (dtt_profile_attach), %d untimed steps each; nms_op is part of proposal_op; psroi_pm_bwd sums the "
                   "detection and tracking heads' calls" % n)
    return out


def measure_train_step(args, cfg, dev, world, im, info, gt, nb):
    """BASELINE configs[3]'s per-rank workload beside the inference figure: `--train-steps` timed training steps (forward, five
    losses, backward, bucketed gradient all-reduce, SGD; trainval_net.py:310-368) of a second, identically built model on the same
    synthetic batch, after 3 warm-up steps -- outside the timed inference region.  The gradient buckets are forced on at one rank
    too (hooks + flat buckets + asynchronous all-reduce over the `nccl` = RCCL group when one is initialised), so the collective
    path runs on whatever hardware there is.  Also times the correlation gradient ops (event tag corr_bwd_op) of those steps."""
    from dtt.dist import make_optimizer, prepare_replica
    from dtt.synth import build_model, calibrate_batchnorm_
    one_rank_group = False
    if world == 1 and not dist.is_initialized() and os.environ.get("DTT_BENCH_BACKEND", "nccl") == "nccl":
        try:   # a 1-rank RCCL communicator: the only form in which RCCL carries the buckets on a single-GPU box
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            dist.init_process_group("nccl", rank=0, world_size=1)
            one_rank_group = True
        except Exception:   # noqa: BLE001  (no RCCL: buckets and hooks still run, without the collective)
            one_rank_group = False
    model = build_model(args.layers, cfg=cfg).to(dev)
    calibrate_batchnorm_(model, im[:, 0])
    model.train()
    runner = prepare_replica(model, world, channels_last=not args.nchw_trunk, force_buckets=True)
    opt = make_optimizer(model, cfg, lr=1e-4)

    def step():
        runner.zero_grad(set_to_none=True)
        out = runner(im, info, gt, nb)
        loss = out[4].mean() + out[5].mean() + out[6].mean() + out[7].mean() + out[9].mean()
        loss.backward()
        runner.finish_gradients()
        opt.step()

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    n_warm, n = 3, max(1, args.train_steps)
    for _ in range(n_warm):
        step()
    n_ops = 3 if args.frames == 2 else 0
    kt = KernelTimer("corr_bwd_op", max(1, n_ops * n), dev)
    kt.attach()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    host = time.perf_counter() - t0      # the host has QUEUED n steps (nothing in a step reads the device: dtt/rpn.py, SAMPLER_RNG "device")
    sync()
    elapsed = time.perf_counter() - t0
    used = kt.detach()
    elapsed, per_rank = max_over_ranks(elapsed, world, dev, os.environ.get("DTT_BENCH_BACKEND", "nccl"))
    res = {"ms_per_step": round(elapsed / n * 1e3, 3), "ms_per_step_ranks": [round(v / n * 1e3, 3) for v in per_rank],
           "steps": n, "warmup": n_warm, "host_queue_ms_per_step": round(host / n * 1e3, 3),
           "frame_pairs_per_s": round(args.batch * world * n / elapsed, 2),
           "workload": "BASELINE.json configs[%d] per-rank step: Res-%d D&T training, %dx%d, correlation d=%d%s, bs=%d per GPU (forward + 5 "
                       "losses + backward + bucketed all-reduce + SGD)" % (4 if args.disp == 16 else 3, args.layers, args.height, args.width, args.disp,
                                                                          "" if args.pooling == "psroi" else " + RoI-%s of the top map" % args.pooling, args.batch),
           "gradient_buckets": {"count": len(runner._buckets), "bytes": runner.bucket_bytes_total(),
                                "collective": ("rccl all_reduce over %d rank(s)" % dist.get_world_size()) if dist.is_initialized() else "none",
                                "allreduce_ms": None}}
    ar = runner.time_allreduce_ms(5)
    if ar is not None:
        res["gradient_buckets"]["allreduce_ms"] = round(ar, 3)
    durs = kt.durations_us(used)
    if n_ops and used == n_ops * n:
        res["corr_bwd"] = corr_bwd_secondary([sum(durs[i + pos] for i in range(0, len(durs), n_ops)) / n for pos in range(n_ops)], args, model)
    if world == 1 and rank_zero():
        res["hot_path"] = train_hot_path_secondary(step, args, dev)
    if one_rank_group:
        dist.destroy_process_group()
    return res


_JSON_OUT = sys.stdout


def emit_line(out):
    """The one JSON line, as the LAST thing the process writes anywhere: the C runtime's buffered streams are flushed first (RCCL
    printf()s its version banner when the communicator is created; with stdout a pipe it would otherwise sit in the stdio buffer
    until exit and come out behind the line, also for a caller that reads stdout and stderr through one pipe)."""
    flush_c_streams()
    print(json.dumps(out), file=_JSON_OUT, flush=True)


def flush_c_streams():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:   # noqa: BLE001
        pass
    sys.stdout.flush()
    sys.stderr.flush()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # stdout carries ONE line, the JSON.  Whatever the libraries write to file descriptor 1 -- RCCL prints its version banner there when
    # the 1-rank communicator of the training-step measurement is torn down, i.e. AFTER the JSON line -- goes to stderr instead.
    global _JSON_OUT
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus %d but the launcher started %d ranks" % (args.gpus, world)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    # DTT_BENCH_BACKEND=gloo: developer switch to exercise the multi-rank control flow on a box with fewer GPUs than ranks
    # (ranks then share devices; inference mode only -- the training path all-reduces device buffers over RCCL)
    backend = os.environ.get("DTT_BENCH_BACKEND", "nccl")
    from dtt.dist import isolate_library_caches
    isolate_library_caches(local, world)   # every rank its own MIOpen find-db / kernel cache (keyed on the launcher's LOCAL_RANK: ranks that share a device below still get their own)
    local = local if backend == "nccl" else local % torch.cuda.device_count()
    torch.cuda.set_device(local)  # before the process group exists: RCCL binds its communicator to the current device
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)

    from dtt.config import apply_dataset_defaults, cfg
    from dtt.synth import build_model, calibrate_batchnorm_, make_batch
    apply_dataset_defaults("imagenet_vid")
    cfg.CORR_MAX_DISPLACEMENT = args.disp
    cfg.RFCN_ROI_FEATURES = "" if args.pooling == "psroi" else args.pooling
    torch.backends.cudnn.benchmark = True

    # Developer mode DTT_BENCH_BACKEND=gloo with more ranks than GPUs: the ranks take turns through the start-up that needs no collective
    # (first touch of the device, code-object loads of the first launches).  Eight processes loading code objects on ONE device at the
    # same moment have aborted with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION inside a stock ATen kernel (round 5,
    # tests/test_gpu_z_bench.py); on the node every rank owns its GPU and nothing is serialised.
    shared_device = world > 1 and backend != "nccl" and world > torch.cuda.device_count()

    def in_turns(fn):
        if not shared_device:
            return fn()
        res = None
        for r in range(world):
            if r == rank:
                res = fn()
                torch.cuda.synchronize(dev)
            dist.barrier()
        return res

    def build():
        m = build_model(args.layers, cfg=cfg).to(dev)
        batch = make_batch(args.batch, args.height, args.width, seed=3 + rank, device=dev)
        calibrate_batchnorm_(m, batch[0][:, 0])
        return m, batch
    model, (im, info, gt, nb) = in_turns(build)
    if args.frames == 1:   # BASELINE configs[1]: plain R-FCN on one frame (dtt/model.py: n_legs == 1, no tracking branch)
        im, info, gt, nb = (t[:, :1].contiguous() for t in (im, info, gt, nb))
    if args.mode == "train":
        from dtt.dist import make_optimizer, prepare_replica
        model.train()
        # rank 0's state to every rank, THEN the frozen BatchNorm folded out of the activation path, THEN the buckets
        if shared_device:   # (the same three steps, the collective-free middle one rank at a time)
            from dtt.dist import DataParallelSnippets, broadcast_module_state
            from dtt.fuse import fuse_for_training
            broadcast_module_state(model)
            in_turns(lambda: fuse_for_training(model, channels_last=not args.nchw_trunk))
            runner = DataParallelSnippets(model, world)
        else:
            runner = prepare_replica(model, world, channels_last=not args.nchw_trunk)
        opt = make_optimizer(model, cfg, lr=1e-4)

        def step():
            runner.zero_grad(set_to_none=True)
            out = runner(im, info, gt, nb)
            loss = out[4].mean() + out[5].mean() + out[6].mean() + out[7].mean() + out[9].mean()
            loss.backward()
            runner.finish_gradients()
            opt.step()
    else:
        model.eval()
        from dtt.fuse import fuse_for_inference
        fuse_for_inference(model, channels_last=not args.nchw_trunk)  # frozen BatchNorm folded into the convolutions, fused bias/residual/ReLU epilogue

        def step():
            with torch.no_grad():
                return model(im, info, gt, nb)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    # The tag "corr_fwd_op" brackets a whole op.  Inference on the channels-last trunk: ONE launch of corr_wsplit_kernel per
    # op (window-split: no partial sums, d = 16 natively).  NCHW maps (training, --nchw-trunk): banded-product kernel +
    # slice-reduction kernel per op, and for d = 12 / 16 conv4 / conv5 are four R = 8 sub-window ops each.
    from dtt.ops import CorrelationNHWCFunction
    if args.mode == "infer":
        nhwc_corr = not args.nchw_trunk and getattr(model, "_pm_tail", None) is not None
    else:   # the channels-last training trunk runs the window-split forward under autograd (dtt.ops.Correlation.pair)
        nhwc_corr = not args.nchw_trunk and args.disp <= CorrelationNHWCFunction.MAX_RADIUS
    n_sub = 4 if (args.disp in (12, 16) and not nhwc_corr) else 1
    ops_per_step = (1 + 2 * n_sub) if args.frames == 2 else 0
    kt = KernelTimer("corr_fwd_op", max(1, ops_per_step * args.steps), dev)
    kt.attach()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_queue = time.perf_counter() - t0   # the host has queued the timed steps (it reads nothing back inside a step)
    sync()
    elapsed = time.perf_counter() - t0
    used = kt.detach()
    elapsed, per_rank = max_over_ranks(elapsed, world, dev, backend)

    def extra(tag, per_step, pick, n=8):
        """A few more (untimed) steps with another kernel's launches bracketed -- the library records one tag at a time.
        pick(durations of one step) -> the launch of interest; returns its mean in us (None if the tag never fired)."""
        k = KernelTimer(tag, per_step * n, dev)
        k.attach()
        for _ in range(n):
            step()
        torch.cuda.synchronize(dev)
        d = k.durations_us(k.detach())
        vals = [pick(d[i:i + per_step]) for i in range(0, len(d) - per_step + 1, per_step)]
        return sum(vals) / len(vals) if vals else None

    if rank == 0:
        durs = kt.durations_us(used)
        # which of a step's ops is conv5: channels-last inference issues conv5 first (ahead of the RPN heads, dtt/model.py),
        # then conv3, conv4; NCHW inference conv5, conv4, conv3; the training graph keeps the reference's conv3, conv4, conv5
        if nhwc_corr and args.mode == "train":
            c5 = lambda d: d[2]
        elif nhwc_corr:
            early = os.environ.get("DTT_CORR5_EARLY", "1") != "0"
            c5 = lambda d: d[0 if early else os.environ.get("DTT_CORR_ORDER", "021").index("2")]
        elif args.mode == "infer":
            c5 = lambda d: sum(d[:n_sub])
        else:
            c5 = lambda d: sum(d[1 + n_sub:])
        conv5 = [c5(durs[i:i + ops_per_step]) for i in range(0, len(durs) - ops_per_step + 1, ops_per_step)] if ops_per_step else []
        op_us = sum(conv5) / max(len(conv5), 1)
        B = args.batch
        H16, W16 = -(-args.height // 16), -(-args.width // 16)
        D2 = (2 * args.disp + 1) ** 2
        flops = 2.0 * 2048 * D2 * H16 * W16 * B            # 2*C*D^2*oH*oW per frame pair
        bytes_ = (2 * 2048 * H16 * W16 * 4 + D2 * H16 * W16 * 4) * B
        achieved = flops / (op_us * 1e-6) / 1e12 if op_us > 0 else 0.0
        hbm = bytes_ / (op_us * 1e-6) / 1e9 if op_us > 0 else 0.0
        main_us = red_us = head_us = psroi_us = rpn_us = None
        assert used == ops_per_step * args.steps, "expected %d correlation ops per step, saw %d in %d steps" % (
            ops_per_step, used, args.steps)
        # (extra() replays steps on rank 0 alone: not in train mode on several ranks, where a step contains collectives)
        if ops_per_step and (args.mode == "infer" or world == 1):
            if nhwc_corr:
                main_us = extra("corr_nhwc", ops_per_step, c5)
            elif args.mode == "infer":
                main_us = extra("corr_fwd_mfma", ops_per_step, c5)
                red_us = extra("corr_fwd_reduce", ops_per_step, c5)
        if args.mode == "infer":
            rpn_us = None
            if getattr(model, "_pm_tail", None) is not None:
                # launches per step: class + box heads (tag head_gemm), the tracking head (head_gemm; frame pairs only); the RPN's
                # two heads in one launch carry their own tag
                head_us = extra("head_gemm", 2 if args.frames == 2 else 1, lambda d: d[0])
                rpn_us = extra("rpn_head_gemm", 1, lambda d: d[0])
                fused_det = os.environ.get("DTT_PSROI_DET_FUSED", "1") != "0"   # class + box pooling + softmax of a RoI in one launch
                psroi_us = extra("psroi_pm", (2 if fused_det else 3) - (0 if args.frames == 2 else 1), lambda d: d[0])
        # HBM bytes of the op come from separate rocprofv3 --pmc passes over the same launch (tools/profile_round.sh ->
        # profiles/rNN_pmc_conv5.json, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes).  The json records the sha256 of
        # the libdtt_hip.so it was measured on: quoted only for that binary and for the shape the pass was taken on.
        traffic, traffic_src = None, None
        pmc, pmc_file = measured_traffic("pmc_conv5")
        if nhwc_corr and (args.batch, args.height, args.width, args.disp) == (2, 600, 1067, 8):
            if pmc is not None:
                traffic = pmc["traffic_bytes_per_op"]
                traffic_src = "static: " + pmc.get("source", pmc_file) + " [" + pmc_file + "] (same libdtt_hip.so: sha256 checked)"
            else:
                traffic_src = pmc_file
        corr_us = {}
        if nhwc_corr and args.mode == "infer" and ops_per_step and args.disp <= 8 and os.environ.get("DTT_CORR5_EARLY", "1") != "0":
            order = [int(c) for c in os.environ.get("DTT_CORR_ORDER", "021") if c != "2"]      # after conv5: conv3, conv4 by default
            for pos, which in enumerate(order):
                v = [durs[i + 1 + pos] for i in range(0, len(durs) - ops_per_step + 1, ops_per_step)]
                corr_us[which] = sum(v) / max(len(v), 1)
        pairs = args.batch * world * args.steps
        out = {
            "metric": "frame-pairs/sec (600px, Res101 D&T)" if args.frames == 2 else "frames/sec (600px, Res101 R-FCN, single frame)",
            "value": round(pairs / elapsed, 3),
            "unit": "frame-pairs/s" if args.frames == 2 else "frames/s",
            "n_gpus": dist.get_world_size() if world > 1 else 1,
            "backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if world > 1 else None,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_step_ranks": [round(v / args.steps * 1e3, 3) for v in per_rank],     # every rank's own clock; the line's value uses the max
            "host_queue_ms_per_step": round(host_queue / args.steps * 1e3, 3),   # (rank 0's host: how long it takes to QUEUE a step)
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("Res-%d D&T siamese 2-frame %s step, %dx%d, correlation d=%d + PSRoI%s, bs=%d per GPU "
                                    "(BASELINE.json configs[%d]); random-init weights, BN statistics calibrated on the "
                                    "synthetic input" % (args.layers, "inference" if args.mode == "infer" else "training",
                                                         args.height, args.width, args.disp,
                                                         "" if args.pooling == "psroi" else " + RoI-%s of the 512-ch top map" % args.pooling,
                                                         args.batch, 4 if args.disp == 16 else (3 if args.mode == "train" else 2)))
                       if args.frames == 2 else
                       ("Res-%d single-frame R-FCN %s step, %dx%d, PSRoI + NMS HIP kernels, bs=%d per GPU (BASELINE.json "
                        "configs[1]); random-init weights, BN statistics calibrated on the synthetic input"
                        % (args.layers, "inference" if args.mode == "infer" else "training", args.height, args.width, args.batch)),
                       "global_batch": args.batch * world, "parallelism": "dp%d (per-snippet sharding%s)" %
                       (world, ", RCCL gradient all-reduce" if args.mode == "train" else ", no collective")},
            # the dominant hot-path op: conv5 cross-frame correlation (exact-f32 MFMA banded product + slice reduction),
            # timed as ONE op with HIP events on its launch stream inside the timed region (other streams keep running
            # beside it, as in production)
            "roofline": {"kernel": ("conv5 correlation op = corr_wsplit_kernel (channels-last, 2048 ch, d=%d: exact-f32 MFMA banded "
                                    "product, displacement window split over workgroups, no partial sums; event tag corr_fwd_op)"
                                    % args.disp) if nhwc_corr else
                                   ("conv5 correlation op = %s + corr_fwd_reduce<5> (2048 ch, d=%d; event tag corr_fwd_op)"
                                    % ("corr_fwd_glds<5>" if args.disp <= 8 else "4 x corr_fwd_glds<5> sub-windows", args.disp)),
                         "bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                         "hbm": {"achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm / HBM_PEAK_GBS, 4)},
                         "traffic": traffic, "traffic_source": traffic_src,
                         "op_us": round(op_us, 2), "ops_timed": len(conv5),
                         "op_us_min_median_max": [round(v, 1) for v in (min(conv5), sorted(conv5)[len(conv5) // 2], max(conv5))] if conv5 else None,
                         "kernel_us": ({"corr_wsplit_kernel": None if main_us is None else round(main_us, 2)} if nhwc_corr else
                                       {"banded_product": None if main_us is None else round(main_us, 2),
                                        "slice_reduction": None if red_us is None else round(red_us, 2)}),
                         "event_bracket_overhead_us": round(KernelTimer.event_pair_overhead_us(dev), 2),
                         "algorithmic_flops_per_op": flops, "algorithmic_bytes_per_op": bytes_},
        }
        sec = {}
        # (conv3 correlates the stride-2 lattice of the /8 maps: the op reads H16 x W16 lattice pixels of each map, not the whole maps --
        #  counting the whole maps, as rounds 1 - 4 did, tripled its HBM fraction)
        for which, name, C, HWc in ((1, "corr4", 1024, H16 * W16), (0, "corr3", 512, H16 * W16)):
            if which in corr_us and corr_us[which] > 0:
                R = args.disp if which == 1 else args.disp // 2
                Dw = (2 * R + 1) ** 2
                fl = 2.0 * C * Dw * H16 * W16 * B
                by = (2 * C * HWc * 4 + Dw * H16 * W16 * 4) * B
                us = corr_us[which]
                sec[name] = {"kernel": "conv%d correlation op = corr_wsplit_kernel (%d ch, planned for 240 CUs: it runs beside the proposal "
                                       "layer; event tag corr_fwd_op)" % (4 if which == 1 else 3, C),
                             "bound": "mfma" if fl / FP32_MFMA_PEAK_TFLOPS / 1e12 >= by / HBM_PEAK_GBS / 1e9 else "hbm", "op_us": round(us, 2),
                             "mfma": {"achieved": round(fl / (us * 1e-6) / 1e12, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": round(fl / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)},
                             "hbm": {"achieved": round(by / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                             "algorithmic_flops_per_op": fl, "algorithmic_bytes_per_op": by}
        if head_us:
            n_img = args.frames * args.batch
            hf = 2.0 * n_img * H16 * W16 * 512 * (31 * 49 + 4 * 49)     # 3.96 + 0.51 GFLOP per image per leg
            sec["heads"] = {"kernel": "head_gemm_kernel (RFCN_cls_net + RFCN_bbox_net of %d images in one exact-f32 MFMA GEMM, "
                                      "position-major output)" % n_img, "bound": "mfma",
                            "achieved": round(hf / (head_us * 1e-6) / 1e12, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(hf / (head_us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4), "launch_us": round(head_us, 2),
                            "algorithmic_flops_per_launch": hf}
        if rpn_us:
            n_img = args.frames * args.batch
            rb = n_img * H16 * W16 * (512 + 72) * 4          # the RPN conv's rows in, 24 probabilities + 48 box deltas out
            sec["rpn_heads"] = {"kernel": "head_gemm_kernel, RPN epilogue (RPN_cls_score + pairwise softmax + RPN_bbox_pred of %d images in one "
                                          "launch, NCHW planes out)" % n_img, "bound": "hbm", "achieved": round(rb / (rpn_us * 1e-6) / 1e9, 1),
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(rb / (rpn_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                "launch_us": round(rpn_us, 2), "algorithmic_bytes_per_launch": rb}
        if psroi_us:
            fused_det = os.environ.get("DTT_PSROI_DET_FUSED", "1") != "0"
            n_img, od = args.frames * args.batch, (31 + (4 if fused_det else 0)) * 49
            ps_bytes = n_img * od * H16 * W16 * 4 + n_img * cfg.TEST.RPN_POST_NMS_TOP_N * (35 if fused_det else 31) * 4   # score maps in, votes out
            sec["psroi_cls"] = {"kernel": ("psroi_pm_det_kernel (R-FCN class scores + box deltas of a RoI in one launch, softmax in the epilogue: "
                                           "%d x %d x %d x %d position-major map, %d RoIs, pooling + 7x7 vote)" if fused_det else
                                           "psroi_pm_kernel (R-FCN class scores: %d x %d x %d x %d position-major map, %d RoIs, "
                                           "pooling + 7x7 vote)") % (n_img, od, H16, W16, n_img * cfg.TEST.RPN_POST_NMS_TOP_N),
                                "bound": "hbm", "achieved": round(ps_bytes / (psroi_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(ps_bytes / (psroi_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                "launch_us": round(psroi_us, 2), "algorithmic_bytes_per_launch": ps_bytes}
        if args.mode == "infer":
            # the proposal layer of the inference step (one call for every image of the batch) and the NMS inside it
            n_img = args.frames * args.batch
            N = cfg.TEST.RPN_PRE_NMS_TOP_N
            mask = N * (-(-N // 64)) * 8
            for tag, name, nbytes, what in (
                    ("proposal_op", "proposal", n_img * ((24 + 48) * H16 * W16 * 4 + 12 * H16 * W16 * 5 * 4 + N * 16 + 2 * mask),
                     "dtt_proposal_forward: ranking of 12 x %d x %d scores per image, decode + clip of the top %d, NMS (0.7), RoI tensor; "
                     "bytes = scores + deltas in, decoded boxes out (SURVEY 8d: 0.73 + 0.61 MB per image) + the NMS's" % (H16, W16, N)),
                    ("nms_op", "nms_test", n_img * (N * 16 + 2 * mask),
                     "the layer's NMS, both phases (nms_mask_kernel + nms_sweep_kernel): N = %d boxes read, the %.2f MB bit matrix written and "
                     "read once per image (SURVEY 8d); the two-phase split computes only the tiles the sweep visits" % (N, mask / 1e6))):
                us = extra(tag, 1, lambda d: d[0], n=4)
                if us:
                    sec[name] = {"kernel": what, "bound": "hbm", "achieved": round(nbytes / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "op_us": round(us, 2),
                                 "algorithmic_bytes_per_op": int(nbytes)}
        if args.mode == "train" and world == 1 and args.frames == 2:
            bw_us = [extra("corr_bwd_op", 3, lambda d, pos=pos: d[pos]) for pos in range(3)]
            if all(v for v in bw_us):
                sec["corr_bwd"] = corr_bwd_secondary(bw_us, args, model)
        if args.mode == "train" and world == 1:
            sec["hot_path"] = train_hot_path_secondary(step, args, dev)
        if args.frames == 1:
            # no correlation in the single-frame graph: the dominant hand-written kernel of the step is the class + box head GEMM
            h = sec.get("heads")
            out["roofline"] = ({"kernel": h["kernel"] + " (event tag head_gemm)", "bound": "mfma", "achieved": h["achieved"],
                                "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": h["frac"], "traffic": None,
                                "launch_us": h["launch_us"], "algorithmic_flops_per_launch": h["algorithmic_flops_per_launch"]}
                               if h else {"kernel": None, "bound": "mfma", "achieved": None, "peak": FP32_MFMA_PEAK_TFLOPS,
                                          "unit": "TFLOP/s", "frac": None, "traffic": None})
        if sec:
            out["secondary"] = sec
        if world == 1 and not args.no_cpu_baseline and args.mode == "infer":
            out["cpu_baseline"] = cpu_baseline(args, cfg)
    else:
        out = None
    if args.mode == "infer" and not args.no_train_step:
        # after (outside) the timed inference region, on every rank: the training step of configs[3].  A watchdog prints the
        # inference line without it should a collective hang -- the headline measurement must not depend on this extra.
        import threading

        def give_up():
            if rank == 0:
                out.setdefault("secondary", {})["train_step"] = {"error": "no result within 600 s"}
                emit_line(out)
            os._exit(0)
        dog = threading.Timer(600.0, give_up)
        dog.daemon = True
        dog.start()
        del model
        torch.cuda.empty_cache()
        try:
            ts = measure_train_step(args, cfg, dev, world, im, info, gt, nb)
        except Exception as e:   # noqa: BLE001
            ts = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        dog.cancel()
        if rank == 0:
            out.setdefault("secondary", {})["train_step"] = ts
    flush_c_streams()          # every rank: nothing buffered may come out behind rank 0's line
    if world > 1:
        dist.barrier()
    if rank == 0:
        emit_line(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
