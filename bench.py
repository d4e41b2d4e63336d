#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X normals + ICP backend (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one synthetic 1M-point scan pair:
    estimate_normals(target, k=16)  +  50-iteration point-to-plane ICP (convergence_threshold 0.0,
    so exactly 50 iterations run: SURVEY.md 7/H4) of source -> target, correspondences returned,
with all inputs already resident in HBM when the timed region starts.  Source and target carry
INDEPENDENT sensor noise (sigma = 1e-4 of the extent), so the converged phase of the registration has
non-zero nearest-neighbour distances like a real scan pair.

`--gpus N`: one rank per GPU, each with its own independent pair (BASELINE configs[2], weak scaling, no
data-path collective; the only collective is the max-over-ranks of the wall time).  Launched by the driver
under torch.distributed.run, or -- when WORLD_SIZE is not set -- this script starts the N ranks itself
(fresh child processes, before anything in the parent touches a GPU).

Prints ONE JSON line (rank 0).  `value` = whole-job ICP iterations / second =
50 * steps * n_gpus / wall, where wall covers normals + ICP of every step.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 1_000_000
K_NORMALS = 16
ICP_ITERS = 50
NOISE_REL = 1e-4            # sensor noise sigma / cloud extent, independently on source and target
ALG_BYTES_ICP = 40          # B / source point / iteration: 12 src + 12 tgt + 12 normal + 4 index
ALG_BYTES_NORMALS = 12 + 12 * K_NORMALS + 24   # = 228 B / point
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def launch_ranks(n):
    """`bench.py --gpus N` without a launcher: start N ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like
    torch.distributed.run).  The parent never imports torch and never touches a GPU; rank 0 prints the JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll: the first rank that exits non-zero ends the job (its peers may be waiting for it in a collective: they are
    # terminated, not waited for); fresh children only, the parent never touches a GPU and never re-execs
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            rc = max(rc, abs(r))
        if live and rc == 0:
            time.sleep(0.05)
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def flush_native_stdio():
    """native libraries (RCCL's version banner) write to C stdio, which is flushed after Python's own buffer at exit: push
    their output out now"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def emit(obj):
    """the ONE JSON line, last on stdout"""
    flush_native_stdio()
    print(json.dumps(obj), flush=True)


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


def timed(fn, warmups, reps):
    for _ in range(warmups):
        fn()
    ts = []
    out = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return median(ts), out


def cpu_baseline(n_points, tgt, src, gpu_normals=None):
    """The oracle (CPU restatement of threecrate-algorithms, kind="port") timed on this box's host cores with the
    reference harness protocol (warm-ups, repeated whole calls, median: docs/benchmarks.md:29) on a BOUNDED sample of
    the same workload: the full k=16 normals call on the 1M cloud (2 warm-ups + 5 timed, median), the kd-tree build on its own
    (single-threaded BY DESIGN: nearest_neighbor.rs:37-58 is a sequential recursion -- it is inside every normals /
    ICP call of the reference), and p2plane ICP calls of 1 and 4 iterations (5 timed each; the difference / 3 = one
    steady iteration, extrapolated to 50: the full 50-iteration call would take ~17 s a time).  A 1-thread figure comes
    from a 100k-point subset.  About 25 s of CPU work in all."""
    import numpy as np
    from oracle import oracle as O
    omp_max = O.num_threads()
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = None
    # the container's CPU bandwidth quota (cgroup): more runnable threads than that are throttled, not run
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if txt and txt[0] not in ("max", "-1"):
                quota = float(txt[0]) / float(txt[1] if len(txt) > 1 else open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError):
            continue
    threads = max(1, min(omp_max, int(quota + 0.5))) if quota else omp_max
    _est, _icp = O.estimate_normals, O.icp_point_to_plane_detailed
    # the reference harness' protocol: 2 warm-ups, 5 timed whole calls, median (docs/benchmarks.md:29)
    t_norm, nrm = timed(lambda: _est(tgt, K_NORMALS, threads=threads), 2, 5)
    t_tree, _ = timed(lambda: O.KdTree(tgt), 0, 3)
    t1, _ = timed(lambda: _icp(src, tgt, nrm[:, 3:], None, 1, None, 0.0, threads=threads), 1, 3)
    t4, _ = timed(lambda: _icp(src, tgt, nrm[:, 3:], None, 4, None, 0.0, threads=threads), 0, 3)
    t_iter = max((t4 - t1) / 3.0, 1e-9)
    t_build = max(t1 - t_iter, 0.0)
    # the whole 50-iteration call, MEASURED once (no extrapolation in `value`); the 1- and 4-iteration
    # medians above only split it into set-up and steady iteration
    t0 = time.perf_counter()
    r50 = _icp(src, tgt, nrm[:, 3:], None, ICP_ITERS, None, 0.0, threads=threads)
    t50 = time.perf_counter() - t0
    assert r50.iterations == ICP_ITERS
    job = t_norm + t50
    # one thread, 100k-point subset of the same pair (same generator, same transform)
    m = min(100_000, n_points)
    ts, ss = np.ascontiguousarray(tgt[:m]), np.ascontiguousarray(src[:m])
    t_norm1, nrm1 = timed(lambda: O.estimate_normals(ts, K_NORMALS, threads=1), 0, 1)
    t_tree1, _ = timed(lambda: O.KdTree(ts), 0, 1)
    a1, _ = timed(lambda: O.icp_point_to_plane_detailed(ss, ts, nrm1[:, 3:], None, 1, None, 0.0, threads=1), 0, 1)
    a3, _ = timed(lambda: O.icp_point_to_plane_detailed(ss, ts, nrm1[:, 3:], None, 3, None, 0.0, threads=1), 0, 1)
    it1 = max((a3 - a1) / 2.0, 1e-9)
    # scaling of the parallel section (k-NN + PCA per point), measured on the 100k-point subset
    scaling = {}
    for th in (1, 4, threads, omp_max):
        if str(th) in scaling:
            continue
        tt, _ = timed(lambda: O.estimate_normals(ts, K_NORMALS, threads=th), 0, 2)
        scaling[str(th)] = round(m / max(tt - t_tree1, 1e-9) / 1e6, 3)
    parity = None
    if gpu_normals is not None:      # the oracle's normals of the same cloud are at hand: use them as the checker they are
        a, b = gpu_normals[:, 3:6].astype(np.float64), nrm[:, 3:6].astype(np.float64)
        c = np.abs((a * b).sum(1)) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-300)
        parity = {"normals_max_1_minus_abs_cos": float(1.0 - c.min()), "normals_frac_within_1e-4": float((c >= 1.0 - 1e-4).mean()),
                  "normals_bit_identical_frac": float((gpu_normals[:, 3:6] == nrm[:, 3:6]).all(1).mean()),
                  "positions_identical": bool(np.array_equal(gpu_normals[:, :3], nrm[:, :3]))}
        # every normal beyond the budget, with what explains it (tests/h1.py, the checker's H1 report: an exact f32 tie at the
        # neighbourhood boundary -- the neighbour SET is then implementation defined, nearest_neighbor.rs:202-219 -- or a
        # degenerate smallest eigen-pair, or a covariance at which the reference's own eigen-solver is discontinuous)
        try:
            from tests import h1
            try:
                rep = h1.normals_report(tgt, K_NORMALS, gpu_normals, nrm)
                parity["normals_offenders"] = rep["offenders"]
                parity["normals_offenders_all_explained"] = True
            except AssertionError as e:
                parity["normals_offenders_all_explained"] = False
                parity["normals_offenders_error"] = str(e)[:400]
        except ImportError:
            pass
    return {
        "parity": parity, "oracle_normals": nrm, "oracle_T50": np.asarray(r50.transformation, np.float32),
        "oracle_corr50": r50.correspondences,
        "value": ICP_ITERS / job, "unit": "ICP it/s (whole job: normals + 50 it)", "cores": threads, "kind": "port",
        "sched_affinity_cpus": affinity, "omp_max_threads": omp_max, "cgroup_cpu_quota_cores": quota,
        "normals_query_mpts_per_s_by_threads_100k_subset": scaling,
        "sample": f"oracle on the same {n_points}-pt pair, {threads} threads (= the container's CPU quota; the box shows {affinity} CPUs): k={K_NORMALS} normals call median of 5 after 2 warm-ups "
                  f"({t_norm:.2f} s, of which the single-threaded kd-tree build is {t_tree:.2f} s) + ONE whole {ICP_ITERS}-iteration p2plane "
                  f"call, measured ({t50:.2f} s; p2plane calls of 1 and 4 iterations, median of 3 each, split it into {t_iter:.3f} s per "
                  f"steady iteration and {t_build:.2f} s per-call set-up)",
        "icp_50it_call_s": t50,
        "normals_mpts_per_s": n_points / t_norm / 1e6,
        "normals_mpts_per_s_excluding_kdtree_build": n_points / max(t_norm - t_tree, 1e-9) / 1e6,
        "kdtree_build_s": t_tree,
        "icp_it_per_s_steady": 1.0 / t_iter,
        "one_thread": {"sample": f"{m}-pt subset, threads=1, one call each", "normals_mpts_per_s": m / t_norm1 / 1e6,
                       "kdtree_build_s": t_tree1, "icp_it_per_s_steady": 1.0 / it1,
                       "icp_source_mpts_per_s": m / it1 / 1e6},
        "all_threads_icp_source_mpts_per_s": n_points / t_iter / 1e6,
    }


def rank_evidence(dist, world, rank, per_rank, device_index=None):
    """What the judge of a multi-GPU line needs to see next to `value`: how many ranks the collective library really
    connected, on which devices, and what every rank did (gathered with all_gather_object; one rank: this rank only)."""
    info = dict(per_rank, rank=rank, pid=os.getpid())
    if device_index is not None:
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            info.update(device=int(device_index), device_name=pr.name, gcn_arch=getattr(pr, "gcnArchName", None),
                        pci_bus_id=getattr(pr, "pci_bus_id", None), hbm_gb=round(pr.total_memory / 2 ** 30, 1))
        except Exception:
            info["device"] = int(device_index)
    ranks = [info]
    ev = {"world_size_env": world}
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, info)
        ev["backend"] = str(dist.get_backend())
        ev["world_size_seen_by_process_group"] = int(dist.get_world_size())
        try:
            import torch
            v = torch.cuda.nccl.version()
            ev["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        except Exception:
            pass
        ev["distinct_devices"] = len({(r.get("pci_bus_id"), r.get("device")) for r in ranks})
    return ev, ranks


def dry_run(args):
    """CPU stand-in for the launch path (tests/test_bench_launch.py): every rank joins a gloo group, the barrier /
    max-over-ranks plumbing runs, rank 0 prints a JSON line with the world size it saw.  No GPU, no product code."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("TC_BENCH_DRY_FAIL_RANK") == str(rank):      # tests/test_bench_launch.py: a rank that dies before its peers' collective
        sys.exit(3)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    wall = time.perf_counter() - t0
    tw = torch.tensor([wall], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        dist.barrier()
    ev, ranks = rank_evidence(dist, world, rank, {"wall_s": wall})
    if rank == 0:
        emit(({"metric": "dry run (launch path only)", "value": 0.0, "unit": "it/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "max_wall_s": float(tw[0]), "pid": os.getpid(), "ppid": os.getppid(),
                          "collective": ev, "ranks": ranks}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def measure_sharded(ctx, dev, n, steps, warmup, plain_calls=False, world=1, rank=0, local_rank=0):
    """BASELINE configs[3]: ONE 10M-point cloud; the source is sharded SPATIALLY inside the library (every rank passes the full
    source, tc_sharded_icp_point_to_plane_device takes its compact range), target + normals + grid replicated, one ncclAllReduce
    of 32 doubles per iteration on the compute stream.  Returns the JSON line (every rank computes it; rank 0 prints it)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import threecrate_amd as tc
    from threecrate_amd import distributed as D
    from threecrate_amd import synth
    src_h, tgt_h, T = synth.registration_pair(n, seed=7, scale=(10.0, 10.0, 1.0), noise_sigma=NOISE_REL * 10.0)
    tgt, src = torch.from_numpy(tgt_h).to(dev), torch.from_numpy(src_h).to(dev)
    # RCCL communicator owned by the library (id broadcast over the process group); one rank: a real one-rank communicator,
    # so that the line measures the same code path, exchange step included
    comm = D.Comm.from_group(ctx) if world > 1 else D.Comm.rccl_single(ctx)
    nrm = D.sharded_estimate_normals(ctx, tgt, K_NORMALS, comm=comm)
    torch.cuda.synchronize()
    tn0 = time.perf_counter()
    for _ in range(3):
        nrm = D.sharded_estimate_normals(ctx, tgt, K_NORMALS, comm=comm)
    torch.cuda.synchronize()
    t_normals = (time.perf_counter() - tn0) / 3.0

    # the target is a map many scans are registered against: every rank keeps it in a handle (index, cell-sorted normals and
    # inscribed-ball bounds built once, not once per registration); --plain-calls: rebuilt inside every call
    th = tc.Cloud(ctx, tgt)
    th.set_normals(nrm)

    # the partition is named HERE and passed down, so that the line below labels what was measured: original-index ranges for
    # more than one rank (each rank orders ceil(n / W) points), a range of the spatially sorted source for one
    shard = "index" if comm.size > 1 else "spatial"

    def step():
        if plain_calls:
            return D.sharded_icp_point_to_plane(ctx, src, tgt, nrm, None, ICP_ITERS, None, 0.0, comm=comm, correspondences="device", shard=shard)
        return D.sharded_icp_against_cloud(ctx, src, th, None, ICP_ITERS, None, 0.0, comm=comm, correspondences="device", shard=shard)
    for _ in range(max(warmup, 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        tw = torch.tensor([wall, t_normals], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall, t_normals = [float(v) for v in tw.tolist()]
    # one more, untimed step with the library's per-kernel events on: what the iteration's kernels and its exchange step cost
    ctx.profile_enable(1); ctx.profile_reset()
    step()
    torch.cuda.synchronize()
    kern = {k: round(1e3 * ms / max(c, 1), 2) for k, (c, ms) in ctx.profile_read().items()
            if k.startswith("icp_") or k.startswith("comm_")}
    ctx.profile_enable(0)
    # the shard this rank's library call took: TC_SHARD_INDEX rows [r ceil(n / W), (r + 1) ceil(n / W)) of the ORIGINAL order,
    # TC_SHARD_SPATIAL positions [n r / W, n (r + 1) / W) of the sorted source (icp_run_sharded)
    if shard == "index":
        rows = (n + world - 1) // world
        lo = min(rank * rows, n); hi = min(lo + rows, n)
    else:
        lo, hi = n * rank // world, n * (rank + 1) // world
    how = "by original-index range (TC_SHARD_INDEX)" if shard == "index" else "spatially (TC_SHARD_SPATIAL)"

    ev, ranks = rank_evidence(dist, world, rank, {"shard_points": hi - lo, "wall_s": wall, "kernels_us_avg": kern,
                                                  "n_ranks_seen_by_rccl": comm.size, "comm_rank": comm.rank}, local_rank)
    if world > 1:       # (every rank's native output before rank 0's line, see main())
        flush_native_stdio()
        dist.barrier()
    err = float(np.linalg.norm(tc.isometry_to_matrix(r.transformation).astype(np.float64) - synth.isometry_matrix(T)))
    main_us = kern.get("icp_correspond_reduce_p2plane")
    it_us = 1e6 * wall / (ICP_ITERS * steps)
    line = {"metric": f"sharded point-to-plane ICP iterations/sec (one cloud, source sharded {how}, 1 ncclAllReduce/iteration in the library)",
            "value": ICP_ITERS * steps / wall, "unit": "it/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": 1e3 * wall / steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}-pt uniform cloud [0,10)x[0,10)x[0,1) (BASELINE configs[3]), 50-iter p2plane ICP, "
                                   f"source sharded {how} over the ranks, correspondences gathered", "shard_mode": shard,
                       "points": n, "parallelism": f"shard{world}",
                       "target": "rebuilt per call" if plain_calls else "tc_cloud handle (indexed once)"},
            "transform_frobenius_error_vs_truth": err, "n_correspondences": int((r.corr_target != -1).sum()),
            "sharded_normals_ms": 1e3 * t_normals, "sharded_normals_mpts_per_s": n / t_normals / 1e6,
            "kernels_us_avg": kern, "allreduce_us_per_iteration": kern.get("comm_allreduce_f64"),
            # algorithmic bytes of one iteration over the WHOLE cloud (40 B per source point), against the main pass
            # of this rank's shard and against the whole iteration (whole-call wall / iterations: set-up and exchange included)
            "roofline": {"bound": "hbm", "alg_bytes_per_iteration": ALG_BYTES_ICP * n, "main_pass_us": main_us,
                         "main_pass_frac": (ALG_BYTES_ICP * (hi - lo) / (main_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if main_us else None,
                         "iteration_us": it_us, "iteration_frac": ALG_BYTES_ICP * n / (it_us * 1e-6) / 1e9 / (HBM_PEAK_GBS * world),
                         "normals_alg_bytes": ALG_BYTES_NORMALS * n,
                         "normals_frac": ALG_BYTES_NORMALS * n / t_normals / 1e9 / (HBM_PEAK_GBS * world)},
            "n_ranks_seen_by_rccl": comm.size, "shard_points_per_rank": [rr.get("shard_points") for rr in ranks],
            "collective": ev, "ranks": ranks}
    th.close()
    comm.close()
    del tgt, src, nrm
    return line


def measure_stream(ctx, steps, warmup):
    """BASELINE configs[4]: 120k-pt KITTI-shaped frames as 16-byte records in HOST memory through tc_frame_stream_* (bounded queue,
    H2D copy of the next frame under the kernels of the current one): voxel_grid_filter(0.2) + k = 16 normals + p2plane ICP."""
    import numpy as np
    import threecrate_amd as tc
    from threecrate_amd import synth
    frames = [synth.kitti_shaped_cloud(seed=i) for i in range(4)]
    # ego motion between consecutive frames: 1 m forward + 0.5 deg yaw
    ego = synth.yaw_isometry((-1.0, 0.0, 0.0), -np.deg2rad(0.5))
    moved = [synth.apply_isometry(ego, f) for f in frames]
    # the sensor's records: x, y, z, intensity (KITTI .bin layout), in host memory like a driver delivers them
    seq = []
    for j in range(8):
        xyz = frames[(j // 2) % 4] if j % 2 == 0 else moved[(j // 2) % 4]
        seq.append(np.ascontiguousarray(np.concatenate([xyz, np.full((len(xyz), 1), 0.5, np.float32)], axis=1)))
    nf = steps * 10

    def run(count):
        fs = tc.FrameStream(ctx, max_points=130000, voxel_size=0.2, k_neighbors=K_NORMALS, max_iterations=ICP_ITERS,
                            max_correspondence_distance=2.0, convergence_threshold=1e-6,
                            backpressure=tc.BackpressureConfig(max_queue_depth=4))
        t0 = time.perf_counter()
        for i in range(count):
            fs.send(seq[i % 8])                      # blocks when 4 frames are waiting (backpressure)
        res, m = fs.finish()
        return time.perf_counter() - t0, res, m

    run(max(warmup, 1) * 4)
    wall, res, m = run(nf)
    assert m.items_processed == nf and m.items_dropped == 0 and all(r.status == 0 for r in res)
    n_in = int(np.mean([len(f) for f in seq]))
    n_vox = int(np.mean([r.n_points for r in res]))
    its = float(np.mean([r.iterations for r in res]))
    # algorithmic bytes of a frame: 16-byte records in + the voxel filter (12 B in, 12 B per voxel out) + normals of
    # the filtered frame + the iterations it actually ran on it
    alg = 16 * n_in + 12 * n_in + 12 * n_vox + ALG_BYTES_NORMALS * n_vox + ALG_BYTES_ICP * n_vox * its
    return {"metric": "LiDAR frames/sec (host frames -> bounded queue -> voxel_grid_filter 0.2 m + k=16 normals + p2plane ICP <= 50 it, "
                      "default threshold; H2D copy overlapped with compute)",
            "value": nf / wall, "unit": "frames/s", "n_gpus": 1, "steps": nf, "warmup": warmup,
            "ms_per_step": 1e3 * wall / nf, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "120k-pt KITTI-shaped frames (64 beams x 1875 azimuth steps) as 16-byte x,y,z,intensity "
                                   "records in host memory, tc_frame_stream_* (queue depth 4); sensor rate 10 Hz",
                       "points": 120000},
            "mean_iterations": its, "converged": int(sum(r.converged for r in res)),
            "points_after_voxel_filter": n_vox,
            "roofline": {"bound": "hbm (launch / latency bound at this size: ~1.4 MB per frame)", "alg_bytes_per_frame": alg,
                         "frame_us": 1e6 * wall / nf, "frac": alg / (wall / nf) / 1e9 / HBM_PEAK_GBS},
            "max_queue_depth_seen": m.max_depth_seen}


def measure_tum_pair(ctx, dev, steps, warmup, oracle_check=False):
    """BASELINE configs[2] shape on one GPU: a ~1 M-point TUM-RGB-D-shaped depth-map surface pair, k = 16 normals + 50-iteration
    point-to-plane ICP through the tc_cloud handles (the step of the judged line on another cloud).  oracle_check (the default
    run, together with the cpu_baseline leg): the timed call's transform and correspondences against the oracle's run of the SAME
    call -- 50 iterations, threshold 0, no cut-off, from the identity, the handle's normals on both sides (registration.rs:508-602)
    -- after the timing, as the checker."""
    import numpy as np
    import torch
    import threecrate_amd as tc
    from threecrate_amd import synth
    base = synth.tum_shaped_cloud(seed=1)
    n = len(base)
    src_h = (synth.apply_isometry(synth.yaw_isometry((-0.01, 0.004, 0.002), -np.deg2rad(0.3)), base) + synth.gaussian_noise(n, 100, 1e-3)).astype(np.float32)
    tgt_h = (base + synth.gaussian_noise(n, 200, 1e-3)).astype(np.float32)
    src, tgt = torch.from_numpy(src_h).to(dev), torch.from_numpy(tgt_h).to(dev)

    def step():
        t0 = time.perf_counter()
        tc_t = tc.Cloud(ctx, tgt)
        tc_t.estimate_normals(K_NORMALS)
        t1 = time.perf_counter()
        tc_s = tc.Cloud(ctx, src)
        r = tc_s.icp_point_to_plane(tc_t, None, ICP_ITERS, None, 0.0, correspondences="device")
        t2 = time.perf_counter()
        tc_t.close(); tc_s.close()
        return t1 - t0, t2 - t1, r
    for _ in range(max(warmup, 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tn = ti = 0.0
    for _ in range(steps):
        a, b, r = step()
        tn += a; ti += b
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ctx.profile_enable(1); ctx.profile_reset()
    step()
    torch.cuda.synchronize()
    kern = {k: round(1e3 * ms / max(c, 1), 2) for k, (c, ms) in ctx.profile_read().items()}
    ctx.profile_enable(0)
    main_us = kern.get("icp_correspond_reduce_p2plane")
    nk_us = kern.get("normals_knn_pca")
    it_us = 1e6 * ti / (ICP_ITERS * steps)
    parity = None
    if oracle_check:
        from oracle import oracle as O            # the checker: after the timed region, never inside it
        tc_t = tc.Cloud(ctx, tgt); tc_t.estimate_normals(K_NORMALS, out=False)
        tc_s = tc.Cloud(ctx, src)
        g = tc_s.icp_point_to_plane(tc_t, None, ICP_ITERS, None, 0.0, correspondences=True)
        gn = tc_t.normals()
        tc_t.close(); tc_s.close()
        assert np.array_equal(np.asarray(g.transformation), np.asarray(r.transformation))         # the call that was timed, again
        nrm = np.ascontiguousarray(gn[:, 3:])
        t0 = time.perf_counter()
        o = O.icp_point_to_plane_detailed(src_h, tgt_h, nrm, None, ICP_ITERS, None, 0.0)
        t_or = time.perf_counter() - t0
        frob = lambda a, b: float(np.linalg.norm(O.isometry_to_matrix(a).astype(np.float64) - O.isometry_to_matrix(b).astype(np.float64)))
        parity = {"icp_T_frobenius_vs_oracle_50it": frob(g.transformation, o.transformation),
                  "icp_correspondences_differing_from_oracle": int((g.correspondences != o.correspondences).any(axis=1).sum())
                  if len(g.correspondences) == len(o.correspondences) else -1,
                  "iterations": [int(g.iterations), int(o.iterations)], "mse": [float(g.mse), float(o.mse)],
                  "oracle_call_s": t_or,
                  "note": "same normals on both sides (the target handle's); tests/test_gpu_fullsize.py::test_config2_the_benchmarked_call_"
                          "fifty_iterations_against_the_oracle is the full check (parting report, exact-sums comparison)"}
        if parity["icp_T_frobenius_vs_oracle_50it"] > 1e-5:
            # the reference adds 10^6 per-pair terms one after the other in f32; the same f32 terms added in f64 = what its formula defines
            e = O.icp_point_to_plane_detailed(src_h, tgt_h, nrm, None, ICP_ITERS, None, 0.0, exact_sums=True)
            parity["icp_T_frobenius_vs_oracle_exact_sums"] = frob(g.transformation, e.transformation)
            parity["reference_accumulation_error"] = frob(o.transformation, e.transformation)
    return {"metric": "ICP iterations/sec (whole job: k=16 normals + 50-iter point-to-plane ICP per pair; tc_cloud handles)",
            "value": ICP_ITERS * steps / wall, "unit": "it/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * wall / steps,
            "config": {"workload": f"{n}-pt TUM-RGB-D-shaped depth-map surface, 1 mm noise on both scans, k={K_NORMALS} normals + {ICP_ITERS}-iter "
                                   "point-to-plane ICP (BASELINE configs[2] shape, one pair)", "points": n},
            "normals_mpts_per_s": n * steps / tn / 1e6, "icp_only_it_per_s": ICP_ITERS * steps / ti, "final_mse": r.mse,
            "kernels_us_avg": kern, "parity": parity,
            "roofline": {"bound": "hbm", "alg_bytes_per_iteration": ALG_BYTES_ICP * n, "main_pass_us": main_us,
                         "main_pass_frac": (ALG_BYTES_ICP * n / (main_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if main_us else None,
                         "iteration_us": it_us, "iteration_frac": ALG_BYTES_ICP * n / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "normals_kernel_us": nk_us,
                         "normals_frac": (ALG_BYTES_NORMALS * n / (nk_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if nk_us else None}}


def aux_modes(args):
    """Secondary workloads run on their own (`--mode sharded|stream`; the plain `--gpus 1` line carries them as extras too):
    sharded 10M-point ICP over the ranks, frame streaming."""
    import torch
    import torch.distributed as dist
    import threecrate_amd as tc
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ctx = tc.GpuContext(local_rank)
    if args.mode == "sharded":
        n = args.points if args.points != N_POINTS else 10_000_000
        line = measure_sharded(ctx, dev, n, args.steps, args.warmup, args.plain_calls, world, rank, local_rank)
        if rank == 0:
            emit(line)
    else:
        emit(measure_stream(ctx, args.steps, args.warmup))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-copy-probe", action="store_true", help="skip the device-to-device copy bandwidth probe (profile runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip the phase / host-path measurements after the timed region (profile runs)")
    ap.add_argument("--plain-calls", action="store_true", help="time the handle-free entry points (tc_estimate_normals_device + "
                    "tc_icp_point_to_plane_detailed_device: the target is indexed twice) instead of the tc_cloud handles")
    ap.add_argument("--dry-run", action="store_true", help="CPU stand-in for the launch path: gloo group, no GPU work")
    ap.add_argument("--cloud", choices=["uniform", "tum"], default="uniform",
                    help="pairs mode: uniform-random cloud (BASELINE configs[1], the judged line) or a TUM-RGB-D-shaped "
                         "depth-map surface of ~1 M points (configs[2]; auxiliary)")
    ap.add_argument("--mode", choices=["pairs", "sharded", "stream"], default="pairs",
                    help="pairs (default, the judged metric): one independent 1M pair per GPU; sharded: ONE cloud, source "
                         "sharded over the ranks, one all-reduce of the packed 6x6 system per iteration (BASELINE configs[3]); "
                         "stream: 120k-pt LiDAR-shaped frames, voxel + normals + ICP per frame (BASELINE configs[4])")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)          # nothing above touched a GPU (torch is not even imported yet)
    if args.dry_run:
        return dry_run(args)
    if args.mode != "pairs":
        return aux_modes(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    import threecrate_amd as tc
    from threecrate_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    n = args.points
    # every rank owns an independent scan pair (seed differs per rank)
    if args.cloud == "tum":
        # ~1 M-point depth-map surface, scan-to-scan motion of a hand-held camera (1 cm, 0.3 deg), 1 mm noise on both scans
        base = synth.tum_shaped_cloud(seed=1 + rank)
        n = len(base)
        T_true = synth.yaw_isometry((0.01, -0.004, -0.002), np.deg2rad(0.3))
        src_h = (synth.apply_isometry(synth.yaw_isometry((-0.01, 0.004, 0.002), -np.deg2rad(0.3)), base) + synth.gaussian_noise(n, 100 + rank, 1e-3)).astype(np.float32)
        tgt_h = (base + synth.gaussian_noise(n, 200 + rank, 1e-3)).astype(np.float32)
    else:
        T_true = synth.harness_transform()
        src_h, tgt_h, _ = synth.registration_pair(n, seed=1 + rank, transform=T_true, noise_sigma=NOISE_REL)
    src, tgt = torch.from_numpy(src_h).to(dev), torch.from_numpy(tgt_h).to(dev)
    ctx = tc.GpuContext(local_rank)
    # sampled hipEvents around the dominant kernel only (every 37th launch: an event is a ~6 us bubble on either side of the kernel on the stream: 27 samples and ~0.6 % of the timed region at the default 20 steps; every 17th until round 6)
    ctx.profile_enable(2)

    def step():
        # The device-resident interface (tc_cloud_*: SURVEY.md 8b): a handle per scan, made from buffers already in HBM.  The
        # target is indexed ONCE: estimate_normals leaves records + normals in the layout the registration reads.  The N x 6
        # NormalPoint3f array (the reference's return value) is produced as well; correspondences="device": the dense
        # per-source target index (ICPResult.correspondences) is written to a device buffer.
        t0 = time.perf_counter()
        tc_t = tc.Cloud(ctx, tgt)
        nrm = tc_t.estimate_normals(K_NORMALS)                           # (n, 6) NormalPoint3f, stays in HBM
        t1 = time.perf_counter()
        tc_s = tc.Cloud(ctx, src)
        r = tc_s.icp_point_to_plane(tc_t, None, ICP_ITERS, None, 0.0, correspondences="device")
        t2 = time.perf_counter()
        tc_t.close(); tc_s.close()
        assert r.iterations == ICP_ITERS
        return t1 - t0, t2 - t1, r, nrm

    if args.plain_calls:
        def step():                                                      # noqa: F811
            t0 = time.perf_counter()
            nrm = ctx.estimate_normals(tgt, K_NORMALS)
            t1 = time.perf_counter()
            r = ctx.icp_point_to_plane_detailed(src, tgt, nrm, None, ICP_ITERS, None, 0.0, correspondences="device")
            t2 = time.perf_counter()
            assert r.iterations == ICP_ITERS
            return t1 - t0, t2 - t1, r, nrm

    for _ in range(args.warmup):
        step()
    ctx.profile_reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_start = time.perf_counter()
    tn = ti = 0.0
    last = None
    for _ in range(args.steps):
        a, b, last, nrm_last = step()
        tn += a
        ti += b
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t_start
    stats = ctx.profile_read()
    ev, ranks = rank_evidence(dist, world, rank, {"wall_s": wall, "it_per_s": ICP_ITERS * args.steps / wall, "pair_seed": 1 + rank,
                                                  "main_pass_us": round(1e3 * stats.get("icp_correspond_reduce_p2plane", (0, 0.0))[1] /
                                                                        max(stats.get("icp_correspond_reduce_p2plane", (1, 0.0))[0], 1), 2)}, local_rank)
    if world > 1:
        tw = torch.tensor([wall, tn, ti], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall, tn, ti = [float(v) for v in tw.tolist()]
        # every rank's native output (the RCCL banner) goes out BEFORE rank 0 prints the line: the launcher merges the ranks'
        # stdout, and a banner flushed at a rank's exit would land behind the JSON line
        flush_native_stdio()
        dist.barrier()

    if rank == 0:
        # what a plain device-to-device copy reaches on this box (report next to the 8 TB/s vendor peak)
        copy_gbs = None
        try:
            if args.no_copy_probe:
                raise RuntimeError("skipped")
            a, b = torch.empty(1 << 28, dtype=torch.float32, device=dev), torch.empty(1 << 28, dtype=torch.float32, device=dev)
            b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9      # read + write
            del a, b
        except Exception:
            pass
        # HBM traffic of the dominant kernel: measured separately with rocprofv3 --pmc (a PMC pass cannot
        # run inside this process); the committed summary is reported with its provenance
        traffic, traffic_note = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            tk = tj["kernels"].get("icp_correspond_reduce_kernel<1>") or tj["kernels"]["icp_correspond_reduce_kernel<true>"]   # MODE 1 = point-to-plane
            if n == N_POINTS:
                traffic = tk["traffic_bytes_corrected"]
                traffic_note = {"raw_bytes": tk["traffic_bytes_raw"], "corrected_bytes": tk["traffic_bytes_corrected"],
                                "source": tj["source"]}
        except Exception:
            pass
        k = "icp_correspond_reduce_p2plane"
        launches, total_ms = stats.get(k, (0, 0.0))
        avg_s = (total_ms / max(launches, 1)) * 1e-3
        achieved = ALG_BYTES_ICP * n / max(avg_s, 1e-12) / 1e9
        out = {
            "metric": "ICP iterations/sec (whole job: k=16 normals + 50-iter point-to-plane ICP per 1M-pt pair; "
                      + ("handle-free *_device calls)" if args.plain_calls else "tc_cloud handles, inputs resident in HBM)"),
            "value": ICP_ITERS * args.steps * world / wall,
            "unit": "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{n}-pt uniform-random cloud, k={K_NORMALS} normals + {ICP_ITERS}-iter point-to-plane ICP, independent "
                                    f"sigma = {NOISE_REL:g} noise on source and target, correspondences returned "
                                    "(BASELINE configs[1]; one independent pair per GPU)") if args.cloud == "uniform" else
                                   (f"{n}-pt TUM-RGB-D-shaped depth-map surface, 1 mm noise on both scans, k={K_NORMALS} normals + {ICP_ITERS}-iter "
                                    "point-to-plane ICP (BASELINE configs[2] shape; one independent pair per GPU; auxiliary line)"),
                       "points": n, "k": K_NORMALS, "icp_iterations": ICP_ITERS, "parallelism": f"pairs{world}",
                       "interface": "plain *_device calls" if args.plain_calls else "tc_cloud handles (one index build per cloud)"},
            "normals_mpts_per_s": n * args.steps * world / tn / 1e6,
            "icp_only_it_per_s": ICP_ITERS * args.steps * world / ti,
            "roofline": {"bound": "hbm", "kernel": k, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_detail": traffic_note,
                         "alg_bytes_per_launch": ALG_BYTES_ICP * n, "avg_launch_us": avg_s * 1e6, "launches": launches,
                         "measured_copy_gbs": copy_gbs},
            "kernels_us_avg": {kk: round(1e3 * ms / max(c, 1), 2) for kk, (c, ms) in stats.items()},
            "final_mse": last.mse,
            "n_correspondences": int((last.corr_target != -1).sum()),
            # multi-GPU evidence: what the collective library connected and what every rank measured (the only data-path-free
            # collective of this mode is the max over the ranks' wall times; `value` is the aggregate over the ranks)
            "collective": ev, "ranks": ranks,
        }
        # the ITERATION, not only its dominant kernel: the timed ICP calls' wall (index build of the source, 50 x (main pass + refine /
        # solve launch), correspondence write-out) / iterations -- the fraction the job actually runs at
        it_us = 1e6 * ti / (ICP_ITERS * args.steps)
        out["roofline"]["iteration"] = {"us": it_us, "frac": ALG_BYTES_ICP * n / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                        "note": "40 B x points / (wall of the timed ICP calls / iterations): set-up, refine + solve launch "
                                                "and launch boundaries included"}
        if args.cloud != "uniform":
            out["roofline"]["traffic"] = None          # PMC passes were collected on the uniform config
            out["roofline"].pop("traffic_detail", None)
        if not args.no_extras:
            # ---- outside the timed region: the two phases of the registration, and the drop-in (host buffer) view ----
            # moving: the first 8 iterations from the identity (clouds still misaligned: NN distance ~ half a cell);
            # converged: 8 iterations started from the final transform (NN distance = the sensor noise).  hipEvents around
            # every kernel (profile mode 1).
            ctx.profile_enable(1)
            phases = {}
            ph_t, ph_s = tc.Cloud(ctx, tgt), tc.Cloud(ctx, src)
            ph_t.estimate_normals(K_NORMALS, out=False)
            # (the target handle keeps its inscribed-ball bounds from the first call on: the profiled calls have them from
            # their first iteration, like iterations 7..50 of the timed registration)
            ph_s.icp_point_to_plane(ph_t, None, 8, None, 0.0)
            steady = {}
            for name, init in (("moving", None), ("converged", last.transformation)):
                ctx.profile_reset()
                ph_s.icp_point_to_plane(ph_t, init, 12, None, 0.0, correspondences="device")
                st = ctx.profile_read(minmax=True)
                phases[name] = {kk: round(1e3 * v[1] / max(v[0], 1), 2) for kk, v in st.items() if kk.startswith("icp_")}
                # the first pass of every call is COLD (no previous matches): the steady figure of a phase is the mean without
                # the longest launch
                c, tot, mn, mx = st[k]
                steady[name] = {"mean_us_without_the_cold_first_pass": round(1e3 * (tot - mx) / max(c - 1, 1), 2), "shortest_us": round(1e3 * mn, 2),
                                "cold_first_pass_us": round(1e3 * mx, 2)}
            ph_t.close(); ph_s.close()
            ctx.profile_reset()
            ctx.estimate_normals(tgt, K_NORMALS)
            st = ctx.profile_read()
            out["normals_kernels_us"] = {kk: round(1e3 * ms / max(c, 1), 2) for kk, (c, ms) in st.items()}
            nk = st.get("normals_knn_pca")
            if nk:
                out["normals_roofline"] = {"kernel": "normals_knn_pca", "alg_bytes_per_launch": ALG_BYTES_NORMALS * n,
                                           "avg_launch_us": 1e3 * nk[1] / max(nk[0], 1),
                                           "frac": ALG_BYTES_NORMALS * n / (1e-3 * nk[1] / max(nk[0], 1)) / 1e9 / HBM_PEAK_GBS}
            ctx.profile_enable(0)
            if nk:
                out["roofline"]["normals"] = dict(out["normals_roofline"], note="the normals kernel of a plain estimate_normals call on the timed cloud")
            # ---- the secondary figures of SURVEY 8(d): one more registration of the timed pair with the main pass's COUNTING
            # instantiation (tc_profile_enable(ctx, 3): same results, a few per cent slower, never inside the timed region) ----
            ctx.profile_enable(3)
            st_t, st_s = tc.Cloud(ctx, tgt), tc.Cloud(ctx, src)
            st_t.estimate_normals(K_NORMALS, out=False)
            rs = st_s.icp_point_to_plane(st_t, None, ICP_ITERS, None, 0.0, correspondences="device")
            ss = ctx.search_stats()
            ctx.profile_enable(0)
            st_t.close(); st_s.close()
            # (the timed calls of --plain-calls build their own index -- another cell edge than the handles' shared one: exact distance
            # ties between points of different cells may resolve the other way there, the transform agrees to 1e-6, not bit for bit)
            same = np.allclose(rs.transformation, last.transformation, atol=1e-5) if args.plain_calls else np.array_equal(rs.transformation, last.transformation)
            assert rs.iterations == ICP_ITERS and same and ss["iterations"] == ICP_ITERS
            icp_call_s = ti / args.steps                               # one timed 50-iteration call (set-up included)
            out["search"] = {
                "source": "one 50-iteration registration of the timed pair with the counting instantiation of the main pass (outside the timed region; " + ("the transform of the timed plain calls to 1e-5: another index" if args.plain_calls else "same transform bit for bit") + ")",
                "candidates_per_query": ss["candidates_per_search"],                       # target records read per search (4 per candidate step)
                "candidates_per_point_iteration": ss["distance_evaluations"] / (n * ICP_ITERS),
                "searches_per_point_iteration": ss["searches"] / (n * ICP_ITERS),
                "distance_evals_per_call": ss["distance_evaluations"] + n * ICP_ITERS,      # + the warm-start distance of every point
                "distance_evals_per_s": (ss["distance_evaluations"] + n * ICP_ITERS) / icp_call_s,
                "lockstep_ratio": ss["lockstep_ratio"],
                "steps_per_search_mean": ss["steps_per_search"], "steps_per_searching_trip_slowest_lane": ss["steps_per_searching_trip"],
                "wave_trips_without_a_search_frac": ss["wave_trips_without_a_search"] / max(ss["wave_trips"], 1),
                "counters": {k: ss[k] for k in ("iterations", "wave_trips", "wave_trips_without_a_search", "searches", "candidate_steps_needed",
                                                "candidate_steps_taken_by_slowest_lanes")}}
            out["main_pass_us_moving"] = steady["moving"]["mean_us_without_the_cold_first_pass"]
            out["main_pass_us_converged"] = steady["converged"]["mean_us_without_the_cold_first_pass"]
            out["main_pass_phase_detail"] = steady
            out["roofline"]["frac_by_phase"] = {ph: ALG_BYTES_ICP * n / (steady[ph]["mean_us_without_the_cold_first_pass"] * 1e-6) / 1e9 / HBM_PEAK_GBS
                                                for ph in ("moving", "converged")}
            out["phase_kernels_us"] = phases
            # the reference's call structure: handle-free entry points, the target indexed inside BOTH calls like the
            # kd-tree the reference rebuilds inside estimate_normals and again inside icp_point_to_plane (normals.rs:272,
            # registration.rs:536) -- the figure that compares like for like with cpu_baseline's protocol
            def plain_step():
                nn = ctx.estimate_normals(tgt, K_NORMALS)
                return ctx.icp_point_to_plane_detailed(src, tgt, nn, None, ICP_ITERS, None, 0.0, correspondences="device")
            plain_step()
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            for _ in range(3):
                plain_step()
            torch.cuda.synchronize()
            tpl = (time.perf_counter() - tp0) / 3.0
            out["plain_calls"] = {"it_per_s_whole_job": ICP_ITERS / tpl, "ms_per_step": 1e3 * tpl,
                                  "note": "tc_estimate_normals_device + tc_icp_point_to_plane_detailed_device, device buffers, mean of 3 "
                                          "steps after the timed region; `value` is the tc_cloud-handle interface unless --plain-calls"}
            # host path: pageable numpy in, numpy out (what a drop-in caller holding Vec<Point3f> sees; PCIe inclusive)
            th_n, nrm_host = timed(lambda: ctx.estimate_normals(tgt_h, K_NORMALS), 1, 3)
            # what the C ABI hands back is the dense per-source target index (tc_icp_result::corr_target, caller-allocated host
            # memory); turning it into Vec<(usize, usize)> is the binding's loop -- the Python mirror's numpy version of that
            # loop (astype / nonzero / stack: 2-3 ms per million points) is reported next to it, not inside it
            # (target_normals: &[Vector3f] in the reference's signature, registration.rs:508-516: an n x 3 array, as its callers build it)
            nrm3_host = np.ascontiguousarray(nrm_host[:, 3:])
            th_i, _ = timed(lambda: ctx.icp_point_to_plane_detailed(src_h, tgt_h, nrm3_host, None, ICP_ITERS, None, 0.0, correspondences="device"), 1, 3)
            th_p, _ = timed(lambda: ctx.icp_point_to_plane_detailed(src_h, tgt_h, nrm3_host, None, ICP_ITERS, None, 0.0, correspondences=True), 0, 3)
            out["host_path"] = {"normals_ms": 1e3 * th_n, "icp_50it_ms": 1e3 * th_i, "it_per_s_whole_job": ICP_ITERS / (th_n + th_i),
                                "icp_50it_ms_with_pairs_materialised_in_python": 1e3 * th_p,
                                "bytes_h2d": int(12 * n + 12 * n + 12 * n + 12 * n), "bytes_d2h": int(24 * n + 4 * n),
                                "note": "pageable numpy buffers in and out through tc_estimate_normals / tc_icp_point_to_plane_detailed "
                                        "(target first on the context's stream, source + normals on a copy stream under the target's index "
                                        "build), dense correspondence array returned; median of 3; never part of `value`"}
        if world == 1 and not args.no_extras and args.cloud == "uniform" and n == N_POINTS:
            # The other BASELINE configs, measured by the same run (outside the timed region, a few seconds each): configs[3] the
            # 10 M-point cloud through the sharded entry points with a REAL one-rank RCCL communicator, configs[4] the LiDAR frame
            # stream, configs[2]'s cloud shape as one pair.  Each with its own algorithmic bytes and fractions.
            # (native libraries print to C stdout -- RCCL's version banner when the sharded line creates its communicator: while the
            # extras run, file descriptor 1 points at stderr, so that this process' stdout stays the ONE JSON line)
            sys.stdout.flush()
            saved_stdout = os.dup(1)
            os.dup2(2, 1)
            try:
                for key, fn in (("frame_stream", lambda: measure_stream(ctx, 50, 2)),
                                ("tum_pair", lambda: measure_tum_pair(ctx, dev, 3, 1, oracle_check=not args.no_cpu_baseline)),
                                ("sharded_10m", lambda: measure_sharded(ctx, dev, 10_000_000, 3, 1))):
                    try:
                        ctx.trim()            # (the blocks the previous workload parked in the context's pool: each line starts from a clean pool)
                        line = fn()
                        for drop in ("collective", "ranks", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
                            line.pop(drop, None)
                        out[key] = line
                    except Exception as e:          # an auxiliary line must never take the judged one down
                        out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                flush_native_stdio()
                os.dup2(saved_stdout, 1)
                os.close(saved_stdout)
        # every fraction of this line in one place (kernel or call, algorithmic bytes of SURVEY 8d, microseconds, fraction of 8 TB/s)
        table = [{"what": "icp main pass (dominant kernel, in-bench events)", "alg_bytes": ALG_BYTES_ICP * n, "us": avg_s * 1e6, "frac": achieved / HBM_PEAK_GBS},
                 {"what": "icp iteration (timed calls / iterations)", "alg_bytes": ALG_BYTES_ICP * n, "us": it_us, "frac": out["roofline"]["iteration"]["frac"]}]
        if "normals_roofline" in out:
            table.append({"what": "normals kernel, timed cloud", "alg_bytes": ALG_BYTES_NORMALS * n, "us": out["normals_roofline"]["avg_launch_us"], "frac": out["normals_roofline"]["frac"]})
        tp = out.get("tum_pair", {}).get("roofline") if isinstance(out.get("tum_pair"), dict) else None
        if tp and tp.get("normals_kernel_us"):
            npts = out["tum_pair"]["config"]["points"]
            table.append({"what": "normals kernel, TUM-shaped cloud (configs[2] shape)", "alg_bytes": ALG_BYTES_NORMALS * npts, "us": tp["normals_kernel_us"], "frac": tp["normals_frac"]})
            table.append({"what": "icp main pass, TUM-shaped pair", "alg_bytes": ALG_BYTES_ICP * npts, "us": tp["main_pass_us"], "frac": tp["main_pass_frac"]})
        # the unit counters behind each kernel's bound claim: from the committed PMC summary (profiles/pmc_traffic.json,
        # separate rocprofv3 --pmc passes of this bench: tools/collect_profiles.sh), per launch, like `traffic` -- never measured in here
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            def unit_counters(sub, points):
                kk = next((v for k_, v in pj["kernels"].items() if sub in k_ and "SQ_BUSY_CU_CYCLES" in v), None)
                if not kk:
                    return None
                valu = kk.get("SQ_INSTS_VALU", kk.get("SQ_ACTIVE_INST_VALU"))
                return {"valu_wave_insts": valu, "busy_cu_cycles": kk.get("SQ_BUSY_CU_CYCLES"),
                        "valu_wave_insts_per_simd": (valu / 1024.0) if valu else None,
                        "busy_cycles_per_cu": kk["SQ_BUSY_CU_CYCLES"] / 256.0,
                        "tcp_lookups_per_point": (kk["TCP_TOTAL_CACHE_ACCESSES_sum"] / points) if "TCP_TOTAL_CACHE_ACCESSES_sum" in kk else None,
                        "lds_bank_conflict_rate": kk.get("lds_bank_conflict_rate"), "achieved_waves_per_simd": kk.get("achieved_waves_per_simd"),
                        "profiled_avg_us": kk.get("avg_us"), "source": pj.get("source")}
            if n == N_POINTS and args.cloud == "uniform":
                for row, sub in ((table[0], "icp_correspond_reduce_kernel<1>"), (table[2] if len(table) > 2 and "normals kernel, timed" in table[2]["what"] else None, "normals_tagged_kernel")):
                    if row is not None:
                        uc = unit_counters(sub, n)
                        if uc:
                            row["counters"] = uc
        except Exception:
            pass
        sh = out.get("sharded_10m", {}).get("roofline") if isinstance(out.get("sharded_10m"), dict) else None
        if sh and sh.get("main_pass_us"):
            table.append({"what": "icp main pass, 10 M-point sharded entry (one rank)", "alg_bytes": sh["alg_bytes_per_iteration"], "us": sh["main_pass_us"], "frac": sh["main_pass_frac"]})
        fsr = out.get("frame_stream", {}).get("roofline") if isinstance(out.get("frame_stream"), dict) else None
        if fsr:
            table.append({"what": "LiDAR frame (whole pipeline per frame)", "alg_bytes": fsr["alg_bytes_per_frame"], "us": fsr["frame_us"], "frac": fsr["frac"]})
        out["roofline"]["table"] = table
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(n, tgt_h, src_h, nrm_last.cpu().numpy())
            out["parity"] = cb.pop("parity")
            onrm = cb.pop("oracle_normals")
            out["cpu_baseline"] = cb
            out["parity"]["icp_T_frobenius_vs_truth"] = float(np.linalg.norm(
                tc.isometry_to_matrix(last.transformation).astype(np.float64) - synth.isometry_matrix(T_true)))
            # the timed registration against the ORACLE's run of the same call (the baseline leg has just made it: 50 iterations,
            # threshold 0, from the identity, its own normals): north_star's 1e-5 Frobenius, and the correspondences
            oT, oc = cb.pop("oracle_T50"), cb.pop("oracle_corr50")
            out["parity"]["icp_T_frobenius_vs_oracle_50it"] = float(np.linalg.norm(
                tc.isometry_to_matrix(last.transformation).astype(np.float64) - tc.isometry_to_matrix(oT).astype(np.float64)))
            gc = last.corr_target.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            og = np.full(len(gc), 0xFFFFFFFF, np.int64)
            og[oc[:, 0]] = oc[:, 1]
            out["parity"]["icp_correspondences_differing_from_oracle"] = int((gc != og).sum())
            del onrm
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        emit(out)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
