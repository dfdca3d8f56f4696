#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of the conv5 (and conv4 / conv3) channels-last correlation launches from two rocprofv3 rocpd
databases (separate --pmc passes over tools/pmc_tail.py) -> the json bench.py quotes as `roofline.traffic`.  The json records
the sha256 of the libdtt_hip.so it was measured on; bench.py refuses to quote it for any other binary.
    tools/pmc_conv5_json.py <fetch.db> <write.db> <out.json>"""
import collections, hashlib, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pytorch-detect-to-track_amd", "lib", "libdtt_hip.so")


def per_dispatch(db_path, counter):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
    agg = collections.defaultdict(float)
    names = {}
    for k, c, v, d in db.execute("select %s, counter_name, value, dispatch_id from counters_collection" % kcol):
        if c == counter and "corr_wsplit_kernel" in k:
            agg[d] += v
            names[d] = k
    return [(names[d], agg[d]) for d in sorted(agg)]


fetch, write = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")
# tools/pmc_tail.py launches conv5, conv4, conv3 in that order, five times; group the dispatches by position in the triple
assert len(fetch) == len(write) and len(fetch) % 3 == 0 and fetch, (len(fetch), len(write))
out = {"library_sha256": hashlib.sha256(open(LIB, "rb").read()).hexdigest(),
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_tail.py, by tools/profile_round.sh",
       "fetch_correction": "x2 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies 128-B requests at 64 B; confirmed on this 64-B-per-pixel DMA pattern by tools/probes/fetch_calib.hip, profiles/r02_fetch_calib.txt)",
       "write_correction": "x1 (tools/probes/write_calib.hip, profiles/r03_write_calib.txt (round 3; the probe is unchanged))"}
algo = {"conv5": (2 * 2048 * 38 * 67 * 4 + 289 * 38 * 67 * 4) * 2, "conv4": (2 * 1024 * 38 * 67 * 4 + 289 * 38 * 67 * 4) * 2,
        "conv3": (2 * 512 * 38 * 67 * 4 + 81 * 38 * 67 * 4) * 2}      # (the stride-2 lattice of the 75 x 134 maps: 38 x 67 pixels per map are read)
for i, name in enumerate(("conv5", "conv4", "conv3")):
    f = [v for k, (_, v) in enumerate(fetch) if k % 3 == i][1:]     # drop the first (cold) launch
    w = [v for k, (_, v) in enumerate(write) if k % 3 == i][1:]
    fkb, wkb = sum(f) / len(f), sum(w) / len(w)
    traffic = int(fkb * 1024 * 2 + wkb * 1024)
    out[name] = {"kernel": fetch[i][0][:60], "FETCH_SIZE_KB_raw": round(fkb, 1), "WRITE_SIZE_KB_raw": round(wkb, 1),
                 "traffic_bytes_per_op": traffic, "algorithmic_bytes_per_op": algo[name], "ratio": round(traffic / algo[name], 3)}
out["traffic_bytes_per_op"] = out["conv5"]["traffic_bytes_per_op"]
out["algorithmic_bytes_per_op"] = out["conv5"]["algorithmic_bytes_per_op"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
