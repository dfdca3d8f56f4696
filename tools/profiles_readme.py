#!/usr/bin/env python3
"""The table of a round's profile files -- file | made by | figures READ FROM THE FILE -- for profiles/README.md.

    tools/profiles_readme.py r05 gpurun_out/r05 > gpurun_out/r05/README_rows.md     (tools/profile_round.sh runs this last)
    tools/profiles_readme.py r05 profiles --prefix                                  (the same over profiles/r05_*)

Nothing in the third column is typed: every number is parsed out of the file named in the first column, so the table cannot
drift from the evidence (round 4's hand-edited row claimed a flag the script did not pass)."""
import json
import os
import re
import sys

R, D = sys.argv[1], sys.argv[2]
PREFIX = "--prefix" in sys.argv


def path(name):
    return os.path.join(D, ("%s_%s" % (R, name)) if PREFIX else name)


def read(name):
    try:
        return open(path(name)).read()
    except OSError:
        return None


def bench_line(name):
    txt = read(name)
    if not txt:
        return None
    for line in reversed(txt.splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                return None
    return None


def f(v, nd=1):
    return "n/a" if v is None else ("%." + str(nd) + "f") % v


def bench_figures(j):
    if j is None:
        return "no JSON line in the file"
    out = ["**%s %s** (%s ms per step, %d steps after %d warm-up; host queues a step in %s ms)" % (
        f(j["value"]), j["unit"], f(j["ms_per_step"], 3), j["steps"], j["warmup"], f(j.get("host_queue_ms_per_step"), 2))]
    r = j.get("roofline") or {}
    if r.get("frac") is not None:
        us = r.get("op_us", r.get("launch_us"))
        out.append("`roofline`: %s us in the event bracket = %s %s = **%s** of the %s peak%s" % (
            f(us), f(r.get("achieved"), 2), r.get("unit"), f(r["frac"], 3), r.get("bound"),
            "" if not r.get("traffic") else "; `traffic` %.1f MB = %.2fx the algorithmic %.1f MB" % (
                r["traffic"] / 1e6, r["traffic"] / r["algorithmic_bytes_per_op"], r["algorithmic_bytes_per_op"] / 1e6)))
        if r.get("hbm"):
            out.append("HBM view %s GB/s = %s" % (f(r["hbm"]["achieved"]), f(r["hbm"]["frac"], 3)))
    sec = j.get("secondary") or {}
    for k in ("corr4", "corr3", "heads", "rpn_heads", "psroi_cls"):
        e = sec.get(k)
        if e:
            fr = e.get("frac", (e.get("mfma") or {}).get("frac") if e.get("bound") == "mfma" else (e.get("hbm") or {}).get("frac"))
            out.append("`%s` %s us (%s of %s)" % (k, f(e.get("op_us", e.get("launch_us"))), f(fr, 3), e.get("bound")))

    def bwd(cb):
        parts = []
        for k in ("corr5_bwd", "corr4_bwd", "corr3_bwd"):
            e = cb.get(k)
            if e:
                parts.append("%s %s us (%s of %s%s)" % (k, f(e["op_us"]), f(e["frac"], 3), e.get("bound", "mfma"),
                                                      "" if not e.get("traffic") else ", traffic %.2fx" % (e["traffic"] / e["algorithmic_bytes_per_op"])))
        return ", ".join(parts)
    ts = sec.get("train_step")
    if ts:
        if "error" in ts:
            out.append("`train_step`: %s" % ts["error"])
        else:
            gb = ts.get("gradient_buckets", {})
            out.append("`train_step` **%s ms** (%s; host queue %s ms; %s buckets, %.1f MB, %s, all-reduce alone %s ms)" % (
                f(ts["ms_per_step"], 2), ts.get("workload", "")[:40], f(ts.get("host_queue_ms_per_step"), 1), gb.get("count"),
                (gb.get("bytes") or 0) / 1e6, gb.get("collective"), f(gb.get("allreduce_ms"), 3)))
            if ts.get("corr_bwd"):
                out.append("gradient ops: " + bwd(ts["corr_bwd"]))
    if sec.get("corr_bwd"):
        out.append("gradient ops: " + bwd(sec["corr_bwd"]))
    c = j.get("cpu_baseline")
    if c:
        out.append("`cpu_baseline` %s %s on %s threads (%s)" % (f(c["value"], 2), c["unit"], c["cores"], c["kind"]))
    return "; ".join(out)


rows = []


def row(name, made_by, figures):
    if read(name) is not None:
        rows.append("| `%s_%s` | %s | %s |" % (R, name, made_by, figures))


row("bench_stdout.log", "`python bench.py` (default flags), stdout only", bench_figures(bench_line("bench_stdout.log")))
t = read("bench_step_sequence.txt")
if t:
    lines = t.splitlines()
    c5 = [l for l in lines if "corr_wsplit_kernel<9>" in l and "grid 196608" in l]
    first = lines[0].split()
    last = lines[-1].split()
    seq = "%d launches in ONE inference step; last launch starts at %s us" % (len(lines), last[0])
    if c5:
        seq += "; conv5 correlation (grid 196608) %s us" % c5[0].split()[1]
    row("bench_step_sequence.txt", "`rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-train-step --steps 10 --warmup 5`; "
        "`tools/rocpd_sequence.py <db> psroi_pm_det_kernel`", seq)
t = read("bench_steady_state.txt")
if t:
    head = t.splitlines()[0]
    own = [l for l in t.splitlines() if "corr_wsplit_kernel" in l or "head_gemm_kernel" in l or "psroi_pm" in l]
    fig = head
    for l in own[:7]:
        m = re.match(r"\S*?(\w+_kernel<[^>]*>).*?(\d+x\d+x\d+/\d+)\s+([\d.]+) calls/step.*avg\s+([\d.]+)", l)
        if m:
            fig += "; `%s` grid %s: %s us avg" % (m.group(1), m.group(2), m.group(4))
    row("bench_steady_state.txt", "same trace; `tools/rocpd_steady.py <db> 5 psroi_pm_det_kernel 40 --expect <launches of the sequence file>` "
        "(REFUSES a window that is not five identical steps)", fig)
t = read("bench_kernel_stats.txt")
if t:
    own = [l for l in t.splitlines() if "corr_wsplit_kernel<9>" in l]
    fig = "whole run incl. warm-up; per (kernel, grid): " + "; ".join(
        "`corr_wsplit_kernel<9>` grid %s: %s calls, avg %s us (min %s)" % (l.split()[-7], l.split()[-6], l.split()[-4], l.split()[-3]) for l in own[:3])
    row("bench_kernel_stats.txt", "same trace; `tools/rocpd_stats.py <db>` (keyed by kernel AND grid)", fig)
t = read("bench_tail_overlap.txt")
if t:
    tails = [float(m) for m in re.findall(r"\| tail (\d+) us", t)]
    row("bench_tail_overlap.txt", "same trace; `tools/rocpd_tail_steps.py <db> 8` (REFUSES mixed windows)",
        t.splitlines()[0] if not tails else "conv5 start -> last pooling end over %d steps: %.0f - %.0f us (mean %.0f)" % (
            len(tails), min(tails), max(tails), sum(tails) / len(tails)))
t = read("pmc_tail.txt")
if t:
    def avg(kern, ctr):
        for l in t.splitlines():
            if kern in l and (" " + ctr + " ") in l:
                return float(l.split("avg")[1].split()[0])
        return None
    fig = []
    for kern, label in (("head_gemm_kernel<10", "head GEMM (class + box)"), ("psroi_pm_det_kernel", "detection pooling"), ("head_gemm_kernel<6", "tracking head")):
        w, fe = avg(kern, "WRITE_SIZE"), avg(kern, "FETCH_SIZE")
        if w is not None and fe is not None:
            fig.append("%s: WRITE_SIZE %.1f MB, FETCH_SIZE %.1f MB raw" % (label, w * 1024 / 1e6, fe * 1024 / 1e6))
    m, cyc = avg("corr_wsplit_kernel<9", "SQ_VALU_MFMA_BUSY_CYCLES"), avg("corr_wsplit_kernel<9", "SQ_LDS_BANK_CONFLICT")
    if m is not None:
        fig.append("`corr_wsplit_kernel<9>` (conv5 + conv4 averaged) SQ_VALU_MFMA_BUSY_CYCLES %.1f M, SQ_LDS_BANK_CONFLICT %s" % (m / 1e6, f(cyc, 0)))
    row("pmc_tail.txt", "one `rocprofv3 --kernel-trace --pmc <counter>` pass per counter over `tools/pmc_tail.py`; `tools/rocpd_pmc.py`", "; ".join(fig))
for name, keys in (("pmc_conv5.json", ("conv5", "conv4", "conv3")), ("pmc_corr_bwd.json", ("conv5", "conv4", "conv3"))):
    t = read(name)
    if t:
        try:
            j = json.loads(t)
            fig = "library sha256 %s...; " % j["library_sha256"][:8] + "; ".join(
                "%s %.1f MB = %.2fx the algorithmic %.1f MB" % (k, j[k]["traffic_bytes_per_op"] / 1e6, j[k]["ratio"], j[k]["algorithmic_bytes_per_op"] / 1e6)
                for k in keys if k in j)
        except (ValueError, KeyError) as e:
            fig = "unreadable: %s" % e
        row(name, "`tools/pmc_conv5_json.py`" if "conv5" in name else "`tools/pmc_corr_bwd_json.py` over `tools/time_corr_bwd.py ONLY=<map>` passes", fig)
t = read("pmc_corr_bwd.txt")
if t:
    row("pmc_corr_bwd.txt", "one `--pmc` pass per counter and map over `tools/time_corr_bwd.py`", "%d counter rows (FETCH / WRITE / MFMA busy / LDS conflicts / active / wave cycles x conv5, conv4, conv3)" % len(
        [l for l in t.splitlines() if "dispatches" in l]))
t = read("pmc_psroi_bwd.txt")
if t:
    fig = []
    for l in t.splitlines():
        m = re.search(r"(\[grid \d+\])?\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches\s+\d+\s+avg\s+([\d.]+)", l)
        if m:
            fig.append("%s %s %.1f MB raw" % (m.group(1) or "", m.group(2), float(m.group(3)) * 1024 / 1e6))
    row("pmc_psroi_bwd.txt", "`rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE` (separate passes) over `tools/time_psroi_bwd.py`; `tools/rocpd_pmc.py`", "; ".join(fig))
t = read("psroi_bwd_microbench.txt")
if t:
    row("psroi_bwd_microbench.txt", "`python tools/time_psroi_bwd.py`", " / ".join(l.split(":", 1)[1].split("  ")[0].strip() for l in t.splitlines() if l.startswith(("detection", "tracking"))))
for name, cmd in (("bench_config4_stdout.log", "`python bench.py --no-cpu-baseline --pooling align --disp 16 --height 563 --width 1000 --batch 1` (BASELINE configs[4], its training step rides along)"),
                  ("bench_frames1_stdout.log", "`python bench.py --frames 1 --no-train-step` (BASELINE configs[1])"),
                  ("bench_train_stdout.log", "`python bench.py --mode train --steps 8 --warmup 4` (BASELINE configs[3] per rank)"),
                  ("bench_train_config4_stdout.log", "`python bench.py --mode train --steps 8 --warmup 4 --pooling align --disp 16 --height 563 --width 1000 --batch 1` (BASELINE configs[4] per rank)")):
    row(name, cmd, bench_figures(bench_line(name)))
t = read("train_steady_state.txt")
if t:
    own = []
    for l in t.splitlines()[1:]:
        m = re.match(r"\S*?(\w+_kernel(?:<[^>]*>)?).*?\s([\d.]+) calls/step\s+([\d.]+) us/step.*avg\s+([\d.]+)", l)
        if m and any(k in l for k in ("corr_bwd", "nms_", "head_dw", "psroi_pm_bwd", "at_subsample")):
            own.append("`%s` %s x %s us" % (m.group(1), m.group(2), m.group(4)))
    row("train_steady_state.txt", "`rocprofv3 --kernel-trace -- python bench.py --mode train --steps 5 --warmup 3`; `tools/rocpd_steady.py <db> 3 \"psroi_pm_bwd_rows_kernel<7, 16, 2>\" 400`",
        t.splitlines()[0] + ("; " + "; ".join(own[:12]) if own else ""))
row("train_kernel_stats.txt", "same trace; `tools/rocpd_stats.py`", "whole run incl. warm-up")
for name, cmd in (("fetch_calib.txt", "`rocprofv3 --pmc FETCH_SIZE` over `tools/probes/fetch_calib.hip`"), ("write_calib.txt", "`rocprofv3 --pmc WRITE_SIZE` over `tools/probes/write_calib.hip`")):
    t = read(name)
    if t:
        req = dict(re.findall(r"(\w+) (\d+)", t.splitlines()[-1]))
        fig = []
        for l in t.splitlines():
            m = re.match(r"(\w+)\(.*avg\s+([\d.]+)", l)
            if m and m.group(1) in req:
                fig.append("%s: counted %.1f MB for %.1f MB requested = %.2f" % (m.group(1), float(m.group(2)) * 1024 / 1e6, int(req[m.group(1)]) / 1e6,
                                                                                   float(m.group(2)) * 1024 / int(req[m.group(1)])))
        row(name, cmd, "; ".join(fig))
row("corr_microbench.txt", "`tools/time_corr.py` at B = 2 and B = 8", "forward correlations alone on dense random maps")
row("proposal_microbench.txt", "`tools/time_proposal.py` (B = 2, 4) + its rocprofv3 per-kernel split", "the proposal layer alone")
row("corr_bwd_ablation.txt", "`tools/time_corr_bwd.py` under rocprofv3 with `DTT_CORR_BWD_ABLATE` = 0 / 1 / 2 / 3", "streamed gradient kernels with the DMA / MFMAs removed")
print("| file | made by | figures (parsed from the file by `tools/profiles_readme.py`) |\n|---|---|---|")
print("\n".join(rows))
