"""Bounded runs of the opt-in randomised suites, so that the driver's `pytest -m gpu` exercises them too (the long sweeps stay
opt-in: tests/fuzz_parity.py, fuzz_queries.py, fuzz_knn.py, soak.py take a case count and a seed):
  * fuzz_parity: random robots (1-6 tendons, polynomial routings, rotation / retraction, radii, step sizes) in random voxel
    environments under both state checkers -- verdicts, flags, tips, edges (verdict + reference FK count), voxel sets, the
    last_valid / discrete validators -- against the CPU oracle;
  * fuzz_queries: the batched lazy query loop against scipy's Dijkstra on the valid sub-graph of random geometric graphs;
  * fuzz_knn: neighbour tables against the oracle's distance rows;
  * soak: 50 create / use / destroy cycles with stable results and no device-memory drift.
Each runs as its own process (a fresh HIP context, as the opt-in form does)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script)] + [str(a) for a in args], capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    return p.stdout


def test_fuzz_parity_one_seed():
    out = _run("fuzz_parity.py", 5, 20261004)
    assert "mismatching checks: 0" in out and out.count("case ") == 5


def test_fuzz_queries_one_seed():
    out = _run("fuzz_queries.py", 4, 31)
    assert "mismatching cases: 0" in out


def test_fuzz_knn_one_seed():
    out = _run("fuzz_knn.py", 8, 5)
    assert "MISMATCH" not in out


def test_soak_50_cycles():
    out = _run("soak.py", 50)
    assert "soak ok: 50 iterations" in out
