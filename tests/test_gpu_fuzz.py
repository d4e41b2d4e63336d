"""Randomized differential run (tools/dev/fuzz.py): odd small clouds (slabs, clusters, surfaces, near-collinear,
lattice-like, exact duplicates; 5 .. 9000 points; three scales) -- k-NN distances, one-iteration ICP
correspondences (= the exact 1-NN, with and without max distance), voxel filter and normals validity against
the oracle.  Fixed seed, ~15 s, in-process (no child process is started from a process that holds the GPU)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_randomized_differential_run(ctx):
    spec = importlib.util.spec_from_file_location("tc_fuzz", os.path.join(ROOT, "tools", "dev", "fuzz.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    lines = []
    cases, bad = fuzz.run(15.0, 7, ctx, fuzz.SMALL, log=lambda *a: lines.append(" ".join(map(str, a))), min_cases=101)
    assert cases > 100 and bad == 0, "\n".join(lines[-40:])


@pytest.mark.gpu
def test_second_neighbour_certificate_and_dense_trips_never_change_a_result():
    """Round 6: the main pass keeps a match whose distance is below (a lower bound of the distance to every OTHER target point, measured
    where the point last searched) - (how far the point has moved since), and packs the lanes that still search into dense trips.
    tools/dev/vor_fuzz.py "second": the same seeded registrations (odd clouds, 3 - 25 iterations, noisy pairs that switch the certificate
    on) with the certificate and with TC_DEBUG=4096 (off; the development build of the same objects): bit-identical transforms, mse,
    iteration counts and correspondences (registration.rs:87-107: the nearest neighbour is the nearest neighbour)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vor_fuzz", os.path.join(os.path.dirname(__file__), "..", "tools", "dev", "vor_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, bad = mod.compare(12.0, 7, min_cases=10, what="second")
    assert n >= 10 and bad == 0


@pytest.mark.gpu
def test_inscribed_ball_test_never_changes_a_result():
    """tools/dev/vor_fuzz.py: the same seeded sequence of registrations on odd clouds (slabs, surfaces, lattices, duplicates,
    non-finite points, handles and plain calls, 3-25 iterations) with the inscribed-ball bounds from the first iteration on and
    without them: bit-identical transforms, mse, iteration counts and correspondences."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vor_fuzz", os.path.join(os.path.dirname(__file__), "..", "tools", "dev", "vor_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, bad = mod.compare(12.0, 5, min_cases=10)          # (at least ten registrations however slow the box: up to eight budgets)
    assert n >= 10 and bad == 0


@pytest.mark.gpu
def test_loop_control_differential_run(ctx):
    """tools/dev/loop_fuzz.py: iteration counts 1..60, convergence thresholds from 0 to 1e-2 (stops after any number of iterations,
    in the middle of any chunk of the enqueue schedule), with / without a maximum correspondence distance, initial guesses,
    point-to-point and point-to-plane, plain calls and cloud handles, against the oracle.  Where a run parts from the oracle's,
    both sides are replayed iteration by iteration and the first step at which they part must be decided by rounding: pairs that
    are each the nearest under the side's own transform, a stop whose margin is within what the transforms' distance explains,
    a residual at the rounding floor of the coordinates."""
    spec = importlib.util.spec_from_file_location("tc_loop_fuzz", os.path.join(ROOT, "tools", "dev", "loop_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    cases, bad = mod.run(15.0, 11, ctx, log=lambda *a: lines.append(" ".join(map(str, a))), min_cases=301)
    assert cases > 300 and bad == 0, "\n".join(lines[-40:])


@pytest.mark.gpu
def test_normals_parameter_space_differential_run(ctx):
    """tools/dev/normals_fuzz.py: k from 1 to 128 (and k >= n), radius mode from far below to far above the point spacing with
    its k-NN fallback, viewpoints, orientation on / off, plain calls / device tensors / handles, clouds of 2 .. 6000 points incl.
    lattices with ties everywhere and exact duplicates.  Every normal beyond 1e-4 cosine of the oracle's must be explained by
    the input: a tie at the neighbourhood boundary, a degenerate smallest eigen-pair, or a covariance at which the reference's
    own eigen-solver is discontinuous (tests/h1.py reference_solver_spread)."""
    spec = importlib.util.spec_from_file_location("tc_normals_fuzz", os.path.join(ROOT, "tools", "dev", "normals_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    cases, bad = mod.run(15.0, 13, ctx, log=lambda *a: lines.append(" ".join(map(str, a))), min_cases=301)
    assert cases > 300 and bad == 0, "\n".join(lines[-40:])
