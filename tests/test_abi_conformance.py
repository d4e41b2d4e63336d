"""A COMPILED consumer of include/threecrate_hip.h: tests/abi/abi_conformance.c is built with gcc -std=c11
and again as C++ (-Wall -Wextra -Werror), linked against libthreecrate_hip.so.

  CPU:  the struct layouts (sizeof / offsetof) and constants the compiler derives from the header must equal the hand-written
        ctypes mirror (threecrate_amd/_lib.py) and the #[repr(C)] structs of the Rust shim (bindings/rust, uncompiled: no Rust
        toolchain in this image -- its layouts are computed here with the C layout rules).
  GPU:  the program runs the reference's call shapes through the HOST entry points (tc_estimate_normals,
        tc_icp_point_to_plane_detailed, tc_icp_detailed, tc_icp) on the 10 k-point golden case and its raw output is compared
        with the golden fixtures (tests/golden/) and with the ctypes path, bit for bit.
"""
import ctypes as C
import json
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from threecrate_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "abi", "abi_conformance.c")
LIBDIR = os.path.join(ROOT, "threecrate_amd")

MIRROR = {"tc_normal_config": _lib.NormalConfig, "tc_icp_result": _lib.IcpResultC, "tc_batch_icp_job": _lib.BatchJobC,
          "tc_batch_icp_result": _lib.BatchResultC, "tc_kernel_stat": _lib.KernelStatC, "tc_icp_scale_level": _lib.ScaleLevelC,
          "tc_multiscale_icp_config": _lib.MultiScaleConfigC, "tc_gicp_config": _lib.GicpConfigC, "tc_kiss_icp_config": _lib.KissIcpConfigC,
          "tc_frame_stream_config": _lib.FrameStreamConfigC, "tc_frame_result": _lib.FrameResultC,
          "tc_frame_stream_metrics": _lib.FrameStreamMetricsC}


def _build(tmp, lang):
    exe = os.path.join(tmp, f"abi_{lang}")
    cc = ["gcc", "-std=c11"] if lang == "c" else ["g++", "-std=c++17", "-x", "c++"]
    subprocess.check_call(cc + ["-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-L", LIBDIR,
                                "-lthreecrate_hip", f"-Wl,-rpath,{LIBDIR}", "-o", exe])
    return exe


def _layout(exe):
    out = subprocess.check_output([exe, "layout"], text=True)
    return {k: int(v) for k, v in (line.split() for line in out.strip().splitlines())}


@pytest.fixture(scope="module")
def exes(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("abi"))
    return {lang: _build(tmp, lang) for lang in ("c", "cpp")}


def test_c_and_cpp_compilers_agree_on_the_header(exes):
    assert _layout(exes["c"]) == _layout(exes["cpp"])


def test_compiled_layouts_equal_the_ctypes_mirror(exes):
    lay = _layout(exes["c"])
    seen = set()
    for name, cls in MIRROR.items():
        assert lay[f"sizeof.{name}"] == C.sizeof(cls), name
        for fname, _ in cls._fields_:
            assert lay[f"offsetof.{name}.{fname}"] == getattr(cls, fname).offset, (name, fname)
            seen.add(f"offsetof.{name}.{fname}")
    # every field the header declares is mirrored (no field missing on the Python side)
    assert {k for k in lay if k.startswith("offsetof.")} == seen
    for cname, val in (("TC_OK", _lib.TC_OK), ("TC_INVALID_DATA", _lib.TC_INVALID_DATA), ("TC_ALGORITHM", _lib.TC_ALGORITHM),
                       ("TC_GPU", _lib.TC_GPU), ("TC_UNSUPPORTED", _lib.TC_UNSUPPORTED), ("TC_COMM_ID_BYTES", _lib.TC_COMM_ID_BYTES),
                       ("TC_ICP_SUMS_P2PLANE", _lib.SUMS_P2PLANE), ("TC_ICP_SUMS_P2P", _lib.SUMS_P2P), ("TC_ICP_SUMS_STRIDE", _lib.SUMS_STRIDE),
                       ("TC_COLL_SUM_F64", _lib.TC_COLL_SUM_F64), ("TC_COLL_SUM_U32", _lib.TC_COLL_SUM_U32),
                       ("TC_COLL_ALLGATHER_U8", _lib.TC_COLL_ALLGATHER_U8), ("TC_SHARD_SPATIAL", _lib.TC_SHARD_SPATIAL),
                       ("TC_SHARD_LOCAL", _lib.TC_SHARD_LOCAL), ("TC_SHARD_INDEX", _lib.TC_SHARD_INDEX),
                       ("TC_COUNTER_INDEXED_POINTS", _lib.TC_COUNTER_INDEXED_POINTS), ("TC_COUNTER_INDEX_BUILDS", _lib.TC_COUNTER_INDEX_BUILDS),
                       ("TC_COUNTER_ICP_ITERATIONS", _lib.TC_COUNTER_ICP_ITERATIONS), ("TC_COUNTER_ICP_TRIPS", _lib.TC_COUNTER_ICP_TRIPS),
                       ("TC_COUNTER_ICP_TRIPS_WITHOUT_SEARCH", _lib.TC_COUNTER_ICP_TRIPS_WITHOUT_SEARCH),
                       ("TC_COUNTER_ICP_SEARCHES", _lib.TC_COUNTER_ICP_SEARCHES), ("TC_COUNTER_ICP_STEPS_NEEDED", _lib.TC_COUNTER_ICP_STEPS_NEEDED),
                       ("TC_COUNTER_ICP_STEPS_TAKEN", _lib.TC_COUNTER_ICP_STEPS_TAKEN)):
        assert lay[f"const.{cname}"] == val, cname
    assert lay["call.tc_abi_version"] == lay["const.TC_ABI_VERSION"] == _lib.load().tc_abi_version()


def _rust_structs():
    """#[repr(C)] structs of the Rust shim -> {name: [(field, size, align)]} with the C layout rules for the types it uses"""
    text = open(os.path.join(ROOT, "bindings", "rust", "threecrate-hip", "src", "ffi.rs")).read()
    text = re.sub(r"//[^\n]*", "", text)
    prim = {"u64": (8, 8), "usize": (8, 8), "i32": (4, 4), "f32": (4, 4), "u32": (4, 4), "f64": (8, 8), "u8": (1, 1), "c_char": (1, 1)}
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\][^\n]*\n?\s*pub struct (\w+)\s*\{([^}]*)\}", text):
        fields = []
        for fm in re.finditer(r"pub (\w+)\s*:\s*([^,]+?)\s*(?:,|$)", m.group(2).strip()):
            ty = fm.group(2).strip()
            am = re.fullmatch(r"\[(\w+);\s*(\d+)\]", ty)
            if am: size, align = prim[am.group(1)][0] * int(am.group(2)), prim[am.group(1)][1]
            elif ty.startswith("*"): size, align = 8, 8
            else: size, align = prim[ty]
            fields.append((fm.group(1), size, align))
        if fields: out[m.group(1)] = fields
    return out


def test_rust_shim_structs_have_the_compiled_layout(exes):
    lay = _layout(exes["c"])
    rs = _rust_structs()
    assert {"tc_normal_config", "tc_icp_result", "tc_frame_result", "tc_multiscale_icp_config"} <= set(rs)
    for name, fields in rs.items():
        off, amax = 0, 1
        for fname, size, align in fields:
            off = (off + align - 1) // align * align
            assert lay[f"offsetof.{name}.{fname}"] == off, (name, fname)
            off += size
            amax = max(amax, align)
        assert lay[f"sizeof.{name}"] == (off + amax - 1) // amax * amax, name
        assert len(fields) == sum(1 for k in lay if k.startswith(f"offsetof.{name}.")), name


def test_rust_shim_covers_the_header_and_the_reference_names():
    """Text level (no rustc in this image): ffi.rs declares EVERY tc_* export of the header -- with as many parameters --, and
    lib.rs has a `pub fn` for every function name of threecrate-algorithms / the threecrate-gpu facade that this path replaces
    (the shim used to stop short of gpu_icp_point_to_plane, gpu_batch_icp, multiscale_icp_point_to_point ...)."""
    rs_dir = os.path.join(ROOT, "bindings", "rust", "threecrate-hip", "src")
    ffi = re.sub(r"//[^\n]*", "", open(os.path.join(rs_dir, "ffi.rs")).read())
    lib = open(os.path.join(rs_dir, "lib.rs")).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "threecrate_hip.h")).read(), flags=re.S)

    def nparams(args):
        args = args.strip()
        return 0 if args in ("", "void") else args.count(",") + 1
    c_decl = {m.group(1): nparams(m.group(2)) for m in re.finditer(r"\b(tc_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", hdr)}
    r_decl = {m.group(1): nparams(m.group(2)) for m in re.finditer(r"pub fn (tc_[a-z0-9_]+)\s*\(([^()]*)\)", ffi)}
    assert sorted(c_decl) == sorted(_lib.EXPORTS)
    assert sorted(r_decl) == sorted(c_decl), sorted(set(c_decl) ^ set(r_decl))
    assert r_decl == c_decl, {k: (c_decl[k], r_decl[k]) for k in c_decl if c_decl[k] != r_decl[k]}
    # ... and with the same TYPES, parameter by parameter and for the return value: what a `bindgen` run + a Rust compile would check
    # (a C type maps to exactly one Rust spelling below; pointers to the opaque / repr(C) structs keep their names)
    hdr_nc = re.sub(r"//[^\n]*", "", hdr)

    def c2r(t):
        toks = t.replace("*", " * ").split()
        base_const = toks[0] == "const"
        toks = toks[1:] if base_const else toks
        cut = toks.index("*") if "*" in toks else len(toks)
        base, ptrs = " ".join(toks[:cut]), toks[cut:]
        prim = {"float": "f32", "double": "f64", "int": "c_int", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8",
                "size_t": "usize", "char": "c_char", "void": "c_void", "unsigned long long": "u64", "tc_status": "c_int",
                "tc_host_collective_fn": "tc_host_collective_fn"}
        r = prim.get(base, base)                     # tc_* struct names are spelled alike on both sides
        pointee_const = base_const
        i = 0
        while i < len(ptrs):                         # `*` [const]: a pointer whose Rust mutability is its POINTEE's constness
            assert ptrs[i] == "*", t
            r = ("*const " if pointee_const else "*mut ") + r
            pointee_const = i + 1 < len(ptrs) and ptrs[i + 1] == "const"
            i += 2 if pointee_const else 1
        return r

    def c_sig(ret, args):
        out = []
        args = args.strip()
        for prm in ([] if args in ("", "void") else args.split(",")):
            prm = " ".join(prm.split())
            m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)(\[[A-Za-z0-9_]*\])?$", prm)
            ty = m.group(1).strip() + (" *" if m.group(3) else "")       # an array parameter is a pointer
            out.append(c2r(ty))
        ret = " ".join(ret.split())
        return out, ("" if ret == "void" else c2r(ret))

    def r_sig(args, ret):
        norm = lambda t: " ".join(t.split()).replace("std::os::raw::", "")
        prms = [norm(x.split(":", 1)[1]) for x in args.split(",") if x.strip()]
        return prms, norm((ret or "").replace("->", ""))

    c_full = {m.group(2): c_sig(m.group(1), m.group(3))
              for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(tc_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", hdr_nc)}
    r_full = {m.group(1): r_sig(m.group(2), m.group(3))
              for m in re.finditer(r"pub fn (tc_[a-z0-9_]+)\s*\(([^()]*)\)\s*(->\s*[^;]+)?;", ffi)}
    assert sorted(c_full) == sorted(r_full) == sorted(c_decl)
    bad = {k: (c_full[k], r_full[k]) for k in c_full if c_full[k] != r_full[k]}
    assert not bad, bad
    # the reference's names on this path: threecrate-algorithms normals.rs:238-380, registration.rs:232-789, gicp.rs:100, kiss_icp.rs:183,
    # filtering.rs:38; threecrate-gpu lib.rs re-exports (normals.rs:443, icp.rs:977-1036, filtering.rs:908, nearest_neighbor.rs:332-367)
    wanted = ["estimate_normals", "estimate_normals_with_config", "estimate_normals_radius", "icp", "icp_detailed", "icp_point_to_point",
              "icp_point_to_point_default", "icp_point_to_plane", "icp_point_to_plane_detailed", "multiscale_icp_point_to_point", "gicp",
              "kiss_icp", "voxel_grid_filter", "gpu_estimate_normals", "gpu_icp", "gpu_icp_point_to_plane", "gpu_batch_icp",
              "gpu_voxel_grid_filter", "gpu_find_k_nearest", "gpu_find_k_nearest_batch", "gpu_find_radius_neighbors"]
    missing = [n for n in wanted if not re.search(r"^pub fn " + n + r"\(", lib, re.M)]
    assert not missing, missing
    for s in ("BatchICPJob", "BatchICPResult", "GpuPointToPlaneICPResult"):
        assert re.search(r"^pub struct " + s + r"\b", lib, re.M), s
    assert "k.min(self.len()).min(2048)" not in lib          # (the silent clamp: the ABI's limit is 2047 and an error)
    # every ffi:: call in lib.rs names a declared function
    used = set(re.findall(r"ffi::(tc_[a-z0-9_]+)\(", lib))
    assert used <= set(r_decl), used - set(r_decl)


def _read_result(buf, pos, ns):
    t = np.frombuffer(buf, np.float32, 7, pos); pos += 28
    mse = struct.unpack_from("<f", buf, pos)[0]; pos += 4
    it, conv, ncorr = struct.unpack_from("<3Q", buf, pos); pos += 24
    corr = np.frombuffer(buf, np.uint32, ns, pos); pos += 4 * ns
    return (t, mse, it, bool(conv), ncorr, corr), pos


@pytest.mark.gpu
def test_compiled_consumer_reproduces_the_golden_case(exes, tmp_path, ctx):
    """the 10 k-point golden case (tests/golden/make_golden.py) through the C program's host-buffer calls"""
    from tests.golden.make_golden import corr_digest
    n, k, iters = 10000, 16, 20
    src, tgt, T = synth.registration_pair(n, seed=1)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<4Q", n, n, k, iters)); f.write(tgt.tobytes()); f.write(src.tobytes())
    for lang in ("c", "cpp"):
        subprocess.check_call([exes[lang], "run", inp, outp])
        buf = open(outp, "rb").read()
        np6 = np.frombuffer(buf, np.float32, n * 6, 0).reshape(n, 6); pos = n * 24
        pl, pos = _read_result(buf, pos, n)
        pp, pos = _read_result(buf, pos, n)
        t_icp = np.frombuffer(buf, np.float32, 7, pos); pos += 28
        codes = struct.unpack_from("<3i", buf, pos); pos += 12
        assert pos == len(buf)
        assert codes == (_lib.TC_INVALID_DATA, _lib.TC_OK, _lib.TC_INVALID_DATA)
        # normals: the golden fixture (oracle) within the budget, the ctypes path bit for bit
        G = os.path.join(ROOT, "tests", "golden")
        ref = np.load(os.path.join(G, "normals_u10k_k16.npy"))
        cs = np.abs((np6[:, 3:].astype(np.float64) * ref.astype(np.float64)).sum(1))
        assert np.array_equal(np6[:, :3], tgt) and cs.min() >= 1 - 1e-4
        assert np.array_equal(np6, ctx.estimate_normals(tgt, k))
        # registrations: golden transform / mse / correspondences
        gold = json.load(open(os.path.join(G, "icp_u10k.json")))
        from oracle import oracle as O
        mat = lambda t: O.isometry_to_matrix(np.asarray(t, np.float32)).astype(np.float64)
        gp = gold["icp_p2p_u10k_20it"]
        assert pp[2] == gp["iterations"] == iters and pp[3] == gp["converged"] and pp[4] == gp["n_correspondences"]
        assert np.linalg.norm(mat(pp[0]) - mat(gp["transformation"])) <= 1e-5 and abs(pp[1] - gp["mse"]) <= 1e-9 + 1e-3 * abs(gp["mse"])
        pairs = np.stack([np.arange(n, dtype=np.int64), pp[5].astype(np.int64)], 1)[pp[5] != 0xFFFFFFFF]
        assert corr_digest(pairs) == gp["correspondences_sha256"]
        # point-to-plane with the program's own normals == the same call through ctypes, bit for bit
        r = ctx.icp_point_to_plane_detailed(src, tgt, np6[:, 3:], None, iters, None, 0.0)
        assert np.array_equal(pl[0], r.transformation) and pl[1] == r.mse and pl[2] == r.iterations == iters
        assert np.array_equal(pl[5].astype(np.int64)[r.correspondences[:, 0]], r.correspondences[:, 1])
        gl = gold["icp_p2pl_u10k_20it"]
        assert np.linalg.norm(mat(pl[0]) - mat(gl["transformation"])) <= 1e-5 and pl[4] == gl["n_correspondences"]
        # icp(): threshold 1e-6, errors swallowed -> a transform near the truth
        assert np.linalg.norm(mat(t_icp) - synth.isometry_matrix(T)) <= 1e-4
