"""The second-neighbour certificate engages where it should and nowhere else (round 6).

It never changes a result (tests/test_gpu_fuzz.py::test_second_neighbour_certificate_and_dense_trips_never_change_a_result), so a gate that
stopped opening would cost 12 % of the TUM-shaped pair's throughput without failing anything.  TC_DEBUG=64 (a print-only bit: honoured by the
shipped library) prints the gate's state behind every registration: `searching lanes ... N (d_run R, wanted W)`.
* BASELINE configs[2]'s shape (depth-map surface, 1 mm noise on both scans: 85 % of the points search in every iteration without it): the gate
  opens, the certificate's instantiation runs for most of the 50 iterations (from the second registration of the context on it starts with the
  first chunk), and fewer than 2 % of the points still search.
* the uniform benchmark pair (clean once aligned): the gate stays shut -- the plain instantiation is the one the headline is measured on."""
import os
import re
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = """
    import numpy as np, torch, threecrate_amd as tc
    from threecrate_amd import synth
    ctx = tc.GpuContext(0)
    if "%s" == "tum":
        base = synth.tum_shaped_cloud(seed=1); n = len(base)
        src = (synth.apply_isometry(synth.yaw_isometry((-0.01, 0.004, 0.002), -np.deg2rad(0.3)), base) + synth.gaussian_noise(n, 100, 1e-3)).astype(np.float32)
        tgt = (base + synth.gaussian_noise(n, 200, 1e-3)).astype(np.float32)
    else:
        src, tgt, T = synth.registration_pair(1_000_000, seed=1, transform=synth.harness_transform(), noise_sigma=1e-4)
        n = len(src)
    dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
    t = tc.Cloud(ctx, dt); t.estimate_normals(16, out=False)
    for rep in range(2):
        s = tc.Cloud(ctx, ds)
        r = s.icp_point_to_plane(t, None, 50, None, 0.0)
        s.close()
        print("CALL", rep, n, r.iterations, flush=True)
"""


def _gate_lines(which):
    env = dict(os.environ, PYTHONPATH=ROOT, TC_DEBUG="64")
    env.pop("TC_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(CODE % which)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    n = int(re.search(r"CALL 0 (\d+) ", r.stdout).group(1))
    gates = [tuple(int(x) for x in m) for m in re.findall(r"searching lanes at the last evaluation of the certificate's gate (\d+) \(d_run (\d+), wanted (\d)\)", r.stderr)]
    assert len(gates) == 2, r.stderr[-2000:]
    return n, gates


def test_the_gate_opens_on_the_noisy_depth_map_pair_and_most_of_the_registration_runs_certified():
    n, gates = _gate_lines("tum")
    for searchers, d_run, wanted in gates:
        assert wanted == 1 and searchers < 0.02 * n, gates
    assert gates[0][1] >= 20 and gates[1][1] >= 30, gates          # certified passes in a row at the end of the call (of 50 iterations)
    assert gates[1][1] > gates[0][1], gates                         # the context's hint: the second registration starts with the instantiation


def test_the_gate_stays_shut_on_the_clean_benchmark_pair():
    n, gates = _gate_lines("uniform")
    for searchers, d_run, wanted in gates:
        assert wanted == 0 and d_run == 0, gates
