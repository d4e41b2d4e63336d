"""The shipped library honours only the print / count bits of TC_DEBUG.

The bits that change a result or the road to it -- 1 no warm start, 4 no inscribed-ball test, 16 no sums, 32 transform frozen,
512 no edge adaptation, 2048 exact box only, 8192 shells only -- are timing experiments; they exist in the development build
(`make dev`: api.hip compiled -DTC_DEV) and nowhere else, so an environment variable inherited by a host process cannot turn the
product into a wrong-answer build.  registration.rs:508-602 is what the call must keep computing."""
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_LIB = os.path.join(ROOT, "threecrate_amd", "variants", "libthreecrate_hip_dev.so")

CODE = """
    import numpy as np, threecrate_amd as tc
    from threecrate_amd import synth, _lib
    ctx = tc.GpuContext(0)
    src, tgt, T = synth.registration_pair(20000, seed=1)
    nrm = ctx.estimate_normals(tgt, 16)
    g = ctx.icp_point_to_plane_detailed(src, tgt, nrm, None, 12, None, 0.0)
    t = g.transformation
    print("LIB", _lib.LIB_PATH)
    print("RES", np.asarray(nrm, np.float32).tobytes().hex()[:64], np.asarray(t, np.float32).tobytes().hex(), np.float32(g.mse).tobytes().hex(), g.iterations)
"""


def _run(debug, lib=None, may_fail=False):
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("TC_DEBUG", None); env.pop("TC_HIP_LIB", None)
    if debug is not None:
        env["TC_DEBUG"] = str(debug)
    if lib:
        env["TC_HIP_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(CODE)], env=env, capture_output=True, text=True, timeout=600)
    if may_fail and r.returncode != 0:
        return "FAILED " + r.stderr.strip().splitlines()[-1][:200]
    assert r.returncode == 0, r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l.startswith("RES")][0]


def test_the_mask_is_in_the_source_and_only_the_dev_object_defines_tc_dev():
    api = open(os.path.join(ROOT, "threecrate_amd", "csrc", "api.hip")).read()
    m = re.search(r"constexpr int kDebugPrintOnlyBits = ([^;]+);", api)
    assert m and eval(m.group(1)) == (8 | 64 | 256 | 1024)
    assert "return f & kDebugPrintOnlyBits;" in api
    mk = open(os.path.join(ROOT, "threecrate_amd", "csrc", "Makefile")).read()
    rules = "\n".join(l for l in mk.split("api_dev.o:")[0].splitlines() if not l.startswith("#"))
    assert "-DTC_DEV" not in rules                                          # the shipped objects are never built with it
    assert "-DTC_DEV -c api.hip" in mk


@pytest.mark.gpu
def test_result_altering_bits_do_nothing_in_the_shipped_library():
    base = _run(None)
    for bits in (16, 1, 4, 32, 512, 2048, 8192, 16 | 1 | 4 | 32 | 512 | 2048 | 8192):
        assert _run(bits) == base, f"TC_DEBUG={bits} changed the shipped library's answer"


@pytest.mark.gpu
def test_the_development_build_still_has_the_switches():
    """(so that the A/B scripts of tools/dev keep working, and so that the test above cannot pass because the bits went away)"""
    assert os.path.exists(DEV_LIB)
    base = _run(None, DEV_LIB)
    assert base == _run(None)                      # without TC_DEBUG the two builds are the same library
    assert _run(16, DEV_LIB, may_fail=True) != base    # no sums -> "insufficient correspondences" (or another transform): the switch is live there
    assert _run(4, DEV_LIB) == base                # no inscribed-ball test: another road, the same bits
