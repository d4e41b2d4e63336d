"""Randomized differential run (tools/dev/fuzz.py): odd small clouds (slabs, clusters, surfaces, near-collinear,
lattice-like, exact duplicates; 5 .. 9000 points; three scales) -- k-NN distances, one-iteration ICP
correspondences (= the exact 1-NN, with and without max distance), voxel filter and normals validity against
the oracle.  Fixed seed, ~15 s."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_randomized_differential_run():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev", "fuzz.py"), "15", "7"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:]
    assert out.returncode == 0 and tail.startswith("fuzz:") and tail.endswith(" 0 problems"), out.stdout[-2000:] + out.stderr[-1000:]
