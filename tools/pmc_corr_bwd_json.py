#!/usr/bin/env python3
"""HBM traffic of the correlation GRADIENT ops from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
tools/time_corr_bwd.py ONLY=<map> (tools/profile_round.sh) -> the json bench.py quotes in secondary.*.corr_bwd.*.traffic.
One op = both gradients of one correlation = corr_bwd_band_kernel + the corr_bwd_stream_kernel launches behind it (round 6: ONE
launch for both directions and, at window radius 9 - 16, for all four window quarters; rounds 4 - 5: one per direction and quarter --
the launches per band launch are counted, not assumed).
    tools/pmc_corr_bwd_json.py <out.json> conv5:<fetch.db>:<write.db> conv4:... conv3:... [d16_conv5:<fetch.db>:<write.db> ...]
A name with the d16_ prefix is BASELINE configs[4]'s per-rank shape (B = 1, 563 x 1000 frames, d = 16: ONLY=<map> B=1 D=16 SHAPE=563).
FETCH_SIZE x 2: the counter tallies one 64-B request per 128-B line for whole-line reads (MI355X_MICROARCH.md, HBM section);
confirmed on this kernel's 256-B-per-pixel LDS-DMA pattern by tools/probes/fetch_calib.hip (stream256 / stream256_lds rows of
profiles/rNN_fetch_calib.txt).  The json records the sha256 of the libdtt_hip.so it was measured on."""
import collections, hashlib, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("DTT_HIP_LIBRARY") or os.path.join(ROOT, "pytorch-detect-to-track_amd", "lib", "libdtt_hip.so")
# channels, lattice pixels, output pixels, window, batch
SHAPES = {"conv5": (2048, 38 * 67, 38 * 67, 289, 2), "conv4": (1024, 38 * 67, 38 * 67, 289, 2), "conv3": (512, 75 * 134, 38 * 67, 81, 2),
          "d16_conv5": (2048, 36 * 63, 36 * 63, 1089, 1), "d16_conv4": (1024, 36 * 63, 36 * 63, 1089, 1), "d16_conv3": (512, 71 * 125, 36 * 63, 289, 1)}


def per_kernel(db_path, counter):
    """{kernel family: (launches, mean KB per launch)} for the streamed gradient kernels."""
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for k, c, v, d in db.execute("select %s, counter_name, value, dispatch_id from counters_collection" % kcol):
        if c == counter:
            fam = "band" if "corr_bwd_band_kernel" in k else "stream" if "corr_bwd_stream_kernel" in k else None
            if fam:
                per[fam][d] += v
    return {f: (len(v), sum(v.values()) / len(v)) for f, v in per.items()}


out = {"library_sha256": hashlib.sha256(open(LIB, "rb").read()).hexdigest(),
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/time_corr_bwd.py ONLY=<map>, by tools/profile_round.sh",
       "fetch_correction": "x2 (tools/probes/fetch_calib.hip stream256 / stream256_lds: whole-line reads tally 64 B per 128-B line)",
       "write_correction": "x1 (tools/probes/write_calib.hip)", "shape": "B=2, 600x1067 frames, d=8; d16_*: B=1, 563x1000 frames, d=16"}
for spec in sys.argv[2:]:
    name, fdb, wdb = spec.split(":")
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    C, px, opx, win, batch = SHAPES[name]
    algo = (2 * C * px * 4 + 2 * C * opx * 4 + win * opx * 4) * batch      # gradients written whole, lattice pixels read, gradOut read
    assert "stream" in f and "band" in f and f["stream"][0] % f["band"][0] == 0, (name, f)
    per_op = f["stream"][0] // f["band"][0]                                  # stream launches behind one band launch
    fetch_kb = f["band"][1] + per_op * f["stream"][1]
    write_kb = w["band"][1] + per_op * w["stream"][1]
    traffic = int(fetch_kb * 1024 * 2 + write_kb * 1024)
    out[name] = {"stream_launches_per_op": per_op,
                 "FETCH_SIZE_KB_raw": {"band": round(f["band"][1], 1), "stream_per_launch": round(f["stream"][1], 1)},
                 "WRITE_SIZE_KB_raw": {"band": round(w["band"][1], 1), "stream_per_launch": round(w["stream"][1], 1)},
                 "traffic_bytes_per_op": traffic, "algorithmic_bytes_per_op": algo, "ratio": round(traffic / algo, 3)}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
