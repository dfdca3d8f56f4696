"""The randomised soak tools of tools/ as bounded `-m gpu` tests: each runs in its own process with a fixed
seed range and fails on the first mismatch the tool reports.
  tools/nms_soak.py       NMS keep lists against the oracle: sizes 1 .. 20000, every phase split, duplicates / degenerate boxes
  tools/corr_bwd_soak.py  streamed correlation gradients against the oracle at 1e-4 + run-to-run identity: window radius 1 .. 16
The case counts keep each under a minute on an MI355X box; the tools take SEEDS / S0 for longer runs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, seeds, s0):
    env = dict(os.environ, SEEDS=str(seeds), S0=str(s0))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)], env=env, capture_output=True, text=True, timeout=900)
    tail = (p.stdout + p.stderr)[-2000:]
    assert p.returncode == 0 and "MISMATCH" not in p.stdout, "%s SEEDS=%d S0=%d:\n%s" % (tool, seeds, s0, tail)
    return p.stdout


def test_nms_soak_bounded():
    out = _run("nms_soak.py", 120, 1000)
    assert "120 cases, 0 mismatches" in out, out[-500:]


def test_correlation_gradient_soak_bounded():
    out = _run("corr_bwd_soak.py", 48, 500)
    assert " 0 bad" in out, out[-500:]
