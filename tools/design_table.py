#!/usr/bin/env python3
"""the table of a round's figures, every one parsed from the profiles/ file named beside it.
    tools/design_table.py r05 > /tmp/table.md"""
import json, os, re, sys
R = sys.argv[1]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
f = lambda n: os.path.join(P, "%s_%s" % (R, n))


def line(n):
    for l in reversed(open(f(n)).read().splitlines()):
        if l.startswith("{"):
            return json.loads(l)


rows = []
add = lambda what, val, src: rows.append("| %s | %s | `%s_%s` |" % (what, val, R, src))
b = line("bench_stdout.log"); r = b["roofline"]; s = b["secondary"]
add("inference, configs[2] (Res-101 D&T, 600×1067, bs = 2)", "**%.1f frame-pairs/s**, %.3f ms per step" % (b["value"], b["ms_per_step"]), "bench_stdout.log")
t = open(f("bench_steady_state.txt")).read().splitlines()[0]
m = re.search(r"wall span ([\d.]+) us/step\s+kernel-busy ([\d.]+) us/step \((\d+) launches", t)
add("the same step under rocprofv3, five identical steps", "%s launches, %s µs wall, %s µs kernel-busy per step" % (m.group(3), m.group(1), m.group(2)), "bench_steady_state.txt")
ks = [l for l in open(f("bench_kernel_stats.txt")).read().splitlines() if "corr_wsplit_kernel<9>" in l and "196608" in l][0].split()
pj = json.load(open(f("pmc_conv5.json")))
add("conv5 forward correlation op (`roofline`)", "%.1f µs in the event bracket = %.1f TFLOP/s = **%.3f** of the fp32 MFMA peak (HBM view %.3f); %s µs average by rocprofv3 over %s launches; traffic %.1f MB = %.2f× the algorithmic %.1f MB" % (
    r["op_us"], r["achieved"], r["frac"], r["hbm"]["frac"], ks[-4], ks[-6], pj["conv5"]["traffic_bytes_per_op"] / 1e6, pj["conv5"]["ratio"], pj["conv5"]["algorithmic_bytes_per_op"] / 1e6),
    "bench_stdout.log`, `%s_bench_kernel_stats.txt`, `%s_pmc_conv5.json" % (R, R))
add("conv4 / conv3 forward ops (beside the proposal layer)", "%.1f µs = %.3f of MFMA / %.1f µs = %.3f of HBM" % (s["corr4"]["op_us"], s["corr4"]["mfma"]["frac"], s["corr3"]["op_us"], s["corr3"]["hbm"]["frac"]), "bench_stdout.log")
pt = open(f("pmc_tail.txt")).read()
w = [l for l in pt.splitlines() if "head_gemm_kernel<10" in l and " WRITE_SIZE " in l][0]
wmb = float(w.split("avg")[1].split()[0]) * 1024 / 1e6
add("class + box head GEMM", "%.1f µs = **%.3f** of MFMA; `WRITE_SIZE` %.1f MB for 71.9 MB stored (%.2f×; round 4: 92.1)" % (s["heads"]["launch_us"], s["heads"]["frac"], wmb, wmb / 71.86), "bench_stdout.log`, `%s_pmc_tail.txt" % R)
add("RPN heads (one launch) / detection pooling (one launch)", "%.1f µs = %.3f of HBM / %.1f µs = %.3f of HBM" % (s["rpn_heads"]["launch_us"], s["rpn_heads"]["frac"], s["psroi_cls"]["launch_us"], s["psroi_cls"]["frac"]), "bench_stdout.log")
tl = open(f("bench_tail_overlap.txt")).read()
tails = [float(x) for x in re.findall(r"\| tail (\d+) us", tl)]
wall = float(m.group(1))
add("hot path inside the step (conv5 start → last pooling end)", "%.0f – %.0f µs of %.0f = %.1f %% of the step; the rest is the fp32 ResNet-101 trunk in the vendor libraries" % (min(tails), max(tails), wall, 100 * sum(tails) / len(tails) / wall), "bench_tail_overlap.txt")
c = b["cpu_baseline"]
add("`cpu_baseline`", "%.2f frame-pairs/s on %d threads of %s (%s)" % (c["value"], c["cores"], c.get("cpu_model", "the host"), c["kind"]), "bench_stdout.log")
ts = s["train_step"]
tr = line("bench_train_stdout.log")
add("training step, configs[3] per rank (600×1067, bs = 2)", "**%.2f ms** (`--mode train`); %.2f ms as `secondary.train_step` with the buckets forced on (%d buckets, %.1f MB, all-reduce alone %.3f ms over a 1-rank RCCL group)" % (
    tr["ms_per_step"], ts["ms_per_step"], ts["gradient_buckets"]["count"], ts["gradient_buckets"]["bytes"] / 1e6, ts["gradient_buckets"]["allreduce_ms"]), "bench_train_stdout.log`, `%s_bench_stdout.log" % R)
bj = json.load(open(f("pmc_corr_bwd.json")))
cb = tr["secondary"]["corr_bwd"]
add("correlation gradient ops, radius 8 (band kernel + ONE launch for both directions)", "; ".join("%s %.1f µs = %.3f of %s, traffic %.1f MB = %.2f×" % (
    k[:5], cb[k]["op_us"], cb[k]["frac"], cb[k]["bound"].upper(), bj["conv" + k[4]]["traffic_bytes_per_op"] / 1e6, bj["conv" + k[4]]["ratio"]) for k in ("corr5_bwd", "corr4_bwd", "corr3_bwd")),
    "bench_train_stdout.log`, `%s_pmc_corr_bwd.json" % R)
c4 = line("bench_config4_stdout.log"); c4t = line("bench_train_config4_stdout.log")
add("configs[4] per rank (563×1000, d = 16, RoI-Align, bs = 1): inference", "**%.1f frame-pairs/s** (%.3f ms); conv5 op (33 × 33 window, one launch) %.1f µs = %.3f" % (c4["value"], c4["ms_per_step"], c4["roofline"]["op_us"], c4["roofline"]["frac"]), "bench_config4_stdout.log")
cb4 = c4t["secondary"]["corr_bwd"]
add("configs[4] per rank: training step", "**%.2f ms**; gradient ops at radius 16 (the four window quarters inside ONE launch; counter traffic: `r06_pmc_corr_bwd.json` `d16_*`): conv5 %.1f µs = %.3f, conv4 %.1f µs = %.3f, conv3 (radius 8) %.1f µs" % (
    c4t["ms_per_step"], cb4["corr5_bwd"]["op_us"], cb4["corr5_bwd"]["frac"], cb4["corr4_bwd"]["op_us"], cb4["corr4_bwd"]["frac"], cb4["corr3_bwd"]["op_us"]), "bench_train_config4_stdout.log")
f1 = line("bench_frames1_stdout.log")
add("configs[1] (single-frame R-FCN, bs = 2)", "**%.1f frames/s** (%.3f ms); head GEMM over 2 images %.1f µs = %.3f" % (f1["value"], f1["ms_per_step"], f1["roofline"]["launch_us"], f1["roofline"]["frac"]), "bench_frames1_stdout.log")
tt = open(f("train_steady_state.txt")).read().splitlines()
m2 = re.search(r"\((\d+) launches", tt[0])
def avg(pat):      # all grids of a kernel name together: launches per step x their mean duration
    calls = us = 0.0
    for l in tt[1:]:
        if pat in l:
            mm = re.search(r"([\d.]+) calls/step\s+([\d.]+) us/step", l)
            calls += float(mm.group(1)); us += float(mm.group(2))
    return "%g × %.1f µs" % (calls, us / calls) if calls else "n/a"
add("training step under rocprofv3 (%s launches per step)" % m2.group(1), "`nms_sweep_kernel` %s (round 4: 2 × 136), `nms_mask_kernel` %s, `corr_bwd_stream_kernel<5>` %s, `head_dw_kernel` %s, `psroi_pm_bwd_rows_kernel` (detection / tracking) %s" % (
    avg("nms_sweep_kernel"), avg("nms_mask_kernel"), avg("corr_bwd_stream_kernel<5"), avg("head_dw_kernel"), avg("psroi_pm_bwd_rows_kernel")), "train_steady_state.txt")
print("| what | figure | file(s) under `profiles/` |\n|---|---|---|")
print("\n".join(rows))
