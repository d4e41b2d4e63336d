"""No C++ exception crosses the C ABI (threecrate-core/src/error.rs:7-28: every failure is an Error VALUE -- a Rust,
C or Python host cannot unwind through an extern "C" frame, and an exception escaping a thread body is std::terminate).

Every extern "C" entry point whose body can allocate is a function-try-block closed by TC_CATCH_* (tc_internal.h), the thread
bodies (frame streamer, tc_batch_icp workers) catch for themselves.  TC_FAULT=<site> makes a named site throw std::bad_alloc:
the call must come back with a status and the process must live."""
import ctypes as C
import glob
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from threecrate_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TC_GPU = 3


def _sources():
    return {f: open(f).read() for f in glob.glob(os.path.join(ROOT, "threecrate_amd", "csrc", "*.hip"))}


# entry points whose whole body is one expression that cannot throw (plain member reads / constants / delete of a POD holder)
TRIVIAL = {"tc_abi_version", "tc_last_error_message", "tc_icp_shard_sums", "tc_icp_shard_destroy", "tc_cloud_size",
           "tc_cloud_points_device", "tc_comm_rank", "tc_comm_size", "tc_search_index_size", "tc_profile_enable"}


def test_every_export_is_a_function_try_block():
    src = "\n".join(_sources().values())
    unguarded = []
    for name in _lib.EXPORTS:
        m = re.search(r"^[^\n/]*\b" + name + r"\([^;{]*\)\s*(try )?\{", src, re.M)
        assert m, f"definition of {name} not found"
        if name == "tc_comm_create_local":        # delegates to a guarded entry point
            continue
        if not m.group(1) and name not in TRIVIAL:
            unguarded.append(name)
    assert not unguarded, f"extern \"C\" entry points without a function-try-block: {unguarded}"
    # and every `try {` that opens an entry point is closed by one of the handler macros
    assert src.count(") try {") == len(re.findall(r"^\} TC_CATCH_(STATUS|VOID|VALUE)|\} TC_CATCH_STATUS\(", src, re.M))


def test_thread_bodies_catch_for_themselves():
    s = _sources()
    stream = next(v for k, v in s.items() if k.endswith("stream.hip"))
    body = stream[stream.index("void worker_main(tc_frame_stream *s) {"):]
    body = body[:body.index("\n}\n")]
    assert "try {" in body and "catch (...)" in body and "worker_status = TC_GPU" in body
    api = next(v for k, v in s.items() if k.endswith("api.hip"))
    batch = api[api.index("tc_status tc_batch_icp("):]
    batch = batch[:batch.index("TC_CATCH_STATUS")]
    # a failed spawn joins what was started and serves the rest on the caller's thread
    assert "catch (...)" in batch and "for (size_t c = started; c < n_ctx; ++c) worker(c);" in batch and "t.join()" in batch


# TC_FAULT exists only in the development build (api.hip compiled -DTC_DEV, every other object shared with the shipped library:
# csrc/Makefile `dev`); the shipped library ignores the variable (test_shipped_library_ignores_fault_injection)
DEV_LIB = os.path.join(ROOT, "threecrate_amd", "variants", "libthreecrate_hip_dev.so")


def _run(code, fault, lib=DEV_LIB):
    env = dict(os.environ, TC_FAULT=fault, PYTHONPATH=ROOT)
    if lib:
        assert os.path.exists(lib), "run `make -C threecrate_amd/csrc` (or __graft_entry__.build()) first"
        env["TC_HIP_LIB"] = lib
    else:
        env.pop("TC_HIP_LIB", None)
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=300)


def test_shipped_library_ignores_fault_injection():
    """the hook must not be live in the product -- an inherited TC_FAULT would turn every error return into a throw."""
    r = _run("""
        import ctypes as C, os, tempfile
        import numpy as np
        from threecrate_amd import _lib
        L = _lib.load()
        assert _lib.LIB_PATH.endswith(os.path.join("threecrate_amd", "libthreecrate_hip.so")), _lib.LIB_PATH
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
            np.arange(16, dtype=np.float32).tofile(f)
        n = C.c_size_t(77)
        L.tc_read_kitti_bin.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        rc = L.tc_read_kitti_bin(f.name.encode(), None, 0, C.byref(n))        # site "kitti": not armed in this build
        os.unlink(f.name)
        assert rc == 0 and n.value == 4, (rc, n.value)
        rc = L.tc_read_kitti_bin(b"/nonexistent/x.bin", None, 0, C.byref(n))   # an error return (site "fail"): its own status, no throw
        assert rc != 0
        print("alive")
    """, "fail,context,kitti", lib=None)
    assert r.returncode == 0 and "alive" in r.stdout, r.stdout + r.stderr


def test_injected_bad_alloc_in_device_free_entry_points_returns_a_status():
    """tc_context_create and tc_read_kitti_bin run without a GPU: with the fault injected the call returns TC_GPU (the handler's
    status), writes nothing, and the process goes on to make further calls."""
    r = _run("""
        import ctypes as C, os, tempfile
        import numpy as np
        from threecrate_amd import _lib
        L = _lib.load()
        h = C.c_void_p(1234)
        rc = L.tc_context_create(0, C.byref(h))
        assert rc == 3 and not h.value, (rc, h.value)
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
            np.arange(16, dtype=np.float32).tofile(f)
        n = C.c_size_t(77)
        L.tc_read_kitti_bin.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        rc = L.tc_read_kitti_bin(f.name.encode(), None, 0, C.byref(n))
        assert rc == 3 and n.value == 0, (rc, n.value)
        os.environ["TC_FAULT"] = ""             # read per call: the same process now works
        rc = L.tc_read_kitti_bin(f.name.encode(), None, 0, C.byref(n))
        assert rc == 0 and n.value == 4, (rc, n.value)
        os.unlink(f.name)
        print("alive")
        """, "context,kitti")
    assert r.returncode == 0 and "alive" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_bad_alloc_inside_an_error_return_becomes_a_status():
    """fail() builds a std::string: with the allocation failing, the entry point returns TC_GPU "out of host memory" instead of
    unwinding into the caller."""
    r = _run("""
        import ctypes as C, os
        import numpy as np
        from threecrate_amd import _lib
        L = _lib.load()
        os.environ["TC_FAULT"] = ""
        h = C.c_void_p()
        assert L.tc_context_create(0, C.byref(h)) == 0
        pts = np.random.default_rng(0).random((100, 3), dtype=np.float32)
        out = np.zeros((100, 6), np.float32)
        cfg = _lib.NormalConfig()
        L.tc_normal_config_default(C.byref(cfg))
        cfg.k_neighbors = 2                      # normals.rs:266-270: InvalidData -> fail()
        rc = L.tc_estimate_normals(h, pts.ctypes.data, 100, C.byref(cfg), out.ctypes.data)
        assert rc == 1, rc
        os.environ["TC_FAULT"] = "fail"
        rc = L.tc_estimate_normals(h, pts.ctypes.data, 100, C.byref(cfg), out.ctypes.data)
        assert rc == 3 and L.tc_last_error_message(h) == b"out of host memory", (rc, L.tc_last_error_message(h))
        os.environ["TC_FAULT"] = ""
        cfg.k_neighbors = 8
        assert L.tc_estimate_normals(h, pts.ctypes.data, 100, C.byref(cfg), out.ctypes.data) == 0
        assert np.all(np.abs(np.linalg.norm(out[:, 3:], axis=1) - 1) < 1e-4)
        L.tc_context_destroy(h)
        print("alive")
        """, "")
    assert r.returncode == 0 and "alive" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_batch_icp_survives_a_thread_that_cannot_start():
    """The second worker's spawn throws: the first thread is joined, the second context's jobs run on the caller's thread, every
    job has its result (same as an undisturbed run)."""
    r = _run("""
        import os
        import numpy as np
        import threecrate_amd as tc
        from threecrate_amd import synth
        os.environ["TC_FAULT"] = ""
        ctxs = [tc.GpuContext(0), tc.GpuContext(0)]
        jobs = []
        for s in range(4):
            src, tgt, T = synth.registration_pair(4000, seed=s + 1)
            jobs.append(tc.BatchICPJob(src, tgt, 10, 1e-6, 1.0))
        good = tc.gpu_batch_icp(ctxs, jobs)
        os.environ["TC_FAULT"] = "batch_thread"
        got = tc.gpu_batch_icp(ctxs, jobs)
        for a, b in zip(good, got):
            assert a.status == 0 and b.status == 0
            assert np.array_equal(np.asarray(a.transformation), np.asarray(b.transformation)) and a.iterations == b.iterations
        print("alive")
        """, "")
    assert r.returncode == 0 and "alive" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_frame_stream_worker_failure_is_a_status_not_a_terminate():
    r = _run("""
        import os
        import numpy as np
        import threecrate_amd as tc
        os.environ["TC_FAULT"] = "stream_worker"
        ctx = tc.GpuContext(0)
        fs = tc.FrameStream(ctx, 2000, voxel_size=0.0, k_neighbors=8, max_iterations=5, backpressure=tc.BackpressureConfig(2))
        rng = np.random.default_rng(1)
        sent = 0
        try:
            for i in range(6):                  # more frames than slots: a sender blocked on a dead worker must wake up
                fs.send(rng.random((1000, 3), dtype=np.float32))
                sent += 1
        except tc.Error:
            pass
        try:
            fs.finish()
            raise SystemExit("finish() of a failed stream must report the failure")
        except tc.Error:
            pass
        os.environ["TC_FAULT"] = ""
        n = ctx.estimate_normals(rng.random((500, 3), dtype=np.float32), 8)      # the context is still usable
        assert n.shape == (500, 6)
        print("alive")
        """, "")
    assert r.returncode == 0 and "alive" in r.stdout, r.stdout + r.stderr
