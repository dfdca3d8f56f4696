"""The evidence pipeline's own checks: the rocpd summarisers refuse windows that are not identical steps, the
per-kernel table separates instantiations by grid, and bench.py quotes counter-derived traffic only from the newest json whose
library sha256 is the loaded library's."""
import hashlib
import importlib
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, inference_steps, foreign_steps):
    db = sqlite3.connect(path)
    db.execute("create table kernels(name, start, end, grid_x, grid_y, grid_z, workgroup_x)")
    t = [0]

    def k(n, d, g=256):
        db.execute("insert into kernels values(?,?,?,?,?,?,?)", (n, t[0], t[0] + d, g, 1, 1, 256))
        t[0] += d + 100
    for _ in range(inference_steps):
        k("head_gemm_kernel<10, 4, 4>", 150000)                      # issued by the trunk ahead of the step's conv5 correlation
        k("corr_wsplit_kernel<9>", 85000, 196608); k("head_gemm_kernel<3, 1, 2>", 18000); k("corr_wsplit_kernel<5>", 22000)
        k("corr_wsplit_kernel<9>", 62000, 327168); k("head_gemm_kernel<6, 1, 2>", 32000); k("psroi_pm_det_kernel<7>", 24000)
    for _ in range(foreign_steps):                                   # e.g. the training leg bench.py runs after the timed region
        k("corr_wsplit_kernel<9>", 50000, 327168); k("igemm_wrw", 100000); k("psroi_pm_det_kernel<7>", 24000)
    db.commit()
    db.close()


def _tool(name, *args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", name)] + [str(a) for a in args], capture_output=True, text=True)


def test_steady_state_and_tail_windows_refuse_mixed_steps(tmp_path):
    clean, mixed = str(tmp_path / "clean.db"), str(tmp_path / "mixed.db")
    _trace(clean, 9, 0)
    _trace(mixed, 8, 3)
    r = _tool("rocpd_steady.py", clean, 5, "psroi_pm_det_kernel", 40, "--expect", 7)
    assert r.returncode == 0 and "7 launches/step, every step" in r.stdout, r.stdout + r.stderr
    # conv5 and conv4 share an instantiation and get a row each (keyed by grid)
    rows = [l for l in r.stdout.splitlines() if "corr_wsplit_kernel<9>" in l]
    assert len(rows) == 2 and any("196608" in l for l in rows) and any("327168" in l for l in rows)
    assert _tool("rocpd_steady.py", clean, 5, "psroi_pm_det_kernel", 40, "--expect", 9).returncode == 3      # not the sequence's count
    r = _tool("rocpd_steady.py", mixed, 5, "psroi_pm_det_kernel")
    assert r.returncode == 3 and r.stdout.startswith("REFUSED"), r.stdout
    r = _tool("rocpd_tail_steps.py", clean, 6)
    assert r.returncode == 0 and r.stdout.count("| tail") == 6 and "head@" not in r.stdout, r.stdout   # the next step's early head GEMM is not this tail's
    r = _tool("rocpd_tail_steps.py", mixed, 6)
    assert r.returncode == 3 and r.stdout.startswith("REFUSED"), r.stdout
    r = _tool("rocpd_stats.py", clean)
    assert r.returncode == 0 and sum("corr_wsplit_kernel<9>" in l for l in r.stdout.splitlines()) == 2


def test_bench_quotes_traffic_only_for_the_loaded_library(tmp_path, monkeypatch):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
    bench = importlib.import_module("bench")
    from dtt import _lib
    lib = tmp_path / "libdtt_hip.so"
    lib.write_bytes(b"this build")
    sha = hashlib.sha256(b"this build").hexdigest()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(_lib, "LIB_PATH", str(lib))
    assert bench.measured_traffic("pmc_conv5")[0] is None                                             # nothing measured yet
    (prof / "r04_pmc_conv5.json").write_text(json.dumps({"library_sha256": sha, "traffic_bytes_per_op": 1}))
    (prof / "r05_pmc_conv5.json").write_text(json.dumps({"library_sha256": "another build", "traffic_bytes_per_op": 2}))
    got, src = bench.measured_traffic("pmc_conv5")
    assert got["traffic_bytes_per_op"] == 1 and src.endswith("r04_pmc_conv5.json")                    # the newest MATCHING file, not the newest
    (prof / "r06_pmc_conv5.json").write_text(json.dumps({"library_sha256": sha, "traffic_bytes_per_op": 3}))
    assert bench.measured_traffic("pmc_conv5")[0]["traffic_bytes_per_op"] == 3
    lib.write_bytes(b"rebuilt")
    got, why = bench.measured_traffic("pmc_conv5")
    assert got is None and "another libdtt_hip.so build" in why
