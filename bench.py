#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X normals + ICP backend (BASELINE.json config [1]).

One "step" = one pass of the hot path over one synthetic 1M-point scan pair:
    estimate_normals(target, k=16)  +  50-iteration point-to-plane ICP (convergence_threshold 0.0,
    so exactly 50 iterations run: SURVEY.md 7/H4) of source -> target,
with all inputs already resident in HBM when the timed region starts.  With --gpus N every rank
runs its own independent pair (BASELINE config [2], weak scaling, no data-path collective);
the only collective is the max-over-ranks of the wall time.

Prints ONE JSON line (rank 0).  `value` = whole-job ICP iterations / second =
50 * steps * n_gpus / wall, where wall covers normals + ICP of every step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 1_000_000
K_NORMALS = 16
ICP_ITERS = 50
ALG_BYTES_ICP = 40          # B / source point / iteration: 12 src + 12 tgt + 12 normal + 4 index
ALG_BYTES_NORMALS = 12 + 12 * K_NORMALS + 24   # = 228 B / point
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(n_points, tgt, src, gpu_normals=None):
    """The oracle (CPU restatement of threecrate-algorithms, kind="port") timed on this box's host
    cores on a bounded sample of the same workload: full k=16 normals on the 1M cloud + the kd-tree
    build + 3 of the 50 p2plane iterations, extrapolated to the 50-iteration job."""
    from oracle import oracle as O
    cores = O.num_threads()
    t0 = time.perf_counter()
    nrm = O.estimate_normals(tgt, K_NORMALS)
    t_norm = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.icp_point_to_plane_detailed(src, tgt, nrm[:, 3:], None, 1, None, 0.0)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.icp_point_to_plane_detailed(src, tgt, nrm[:, 3:], None, 4, None, 0.0)
    t4 = time.perf_counter() - t0
    t_iter = max((t4 - t1) / 3.0, 1e-9)
    t_build = max(t1 - t_iter, 0.0)
    job = t_norm + t_build + ICP_ITERS * t_iter
    parity = None
    if gpu_normals is not None:      # the oracle's normals of the same cloud are at hand: use them as the checker they are
        a, b = gpu_normals[:, 3:6].astype(np.float64), nrm[:, 3:6].astype(np.float64)
        c = np.abs((a * b).sum(1)) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-300)
        parity = {"normals_max_1_minus_abs_cos": float(1.0 - c.min()), "normals_frac_within_1e-4": float((c >= 1.0 - 1e-4).mean()),
                  "positions_identical": bool(np.array_equal(gpu_normals[:, :3], nrm[:, :3]))}
    return {
        "parity": parity,
        "value": ICP_ITERS / job, "unit": "ICP it/s (whole job: normals + 50 it)", "cores": cores, "kind": "port",
        "sample": f"oracle on {n_points} pts: full k={K_NORMALS} normals ({t_norm:.2f} s) + kd-tree build ({t_build:.2f} s) + "
                  f"3 timed p2plane iterations ({t_iter:.3f} s/it) extrapolated to {ICP_ITERS}",
        "normals_mpts_per_s": n_points / t_norm / 1e6,
        "icp_it_per_s_steady": 1.0 / t_iter,
    }


def aux_modes(args):
    """Secondary workloads (not the judged line): sharded 10M-point ICP and frame streaming."""
    import torch
    import torch.distributed as dist
    import threecrate_amd as tc
    from threecrate_amd import distributed as D
    from threecrate_amd import synth
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # on torch's current stream: the sharded loop (kernels, copies, RCCL all-reduce) is then stream ordered
    ctx = tc.GpuContext(local_rank, stream=torch.cuda.current_stream(dev).cuda_stream) if args.mode == "sharded" else tc.GpuContext(local_rank)
    if args.mode == "sharded":
        n = args.points if args.points != N_POINTS else 10_000_000
        tgt_h = synth.uniform_cloud(n, seed=7, scale=(10.0, 10.0, 1.0))
        T = synth.small_transform(n)
        Minv = synth.invert_isometry(T)
        src_h = (tgt_h.astype(np.float64) @ Minv[:3, :3].T + Minv[:3, 3]).astype(np.float32)
        tgt = torch.from_numpy(tgt_h).to(dev)
        lo, hi = D.shard_range(n, rank, world)
        src = torch.from_numpy(src_h[lo:hi]).to(dev)
        # the normals of the replicated target are sharded too: every rank computes its slice, one all-gather
        nrm = D.sharded_estimate_normals(ctx, tgt, K_NORMALS)
        torch.cuda.synchronize()
        tn0 = time.perf_counter()
        for _ in range(3):
            nrm = D.sharded_estimate_normals(ctx, tgt, K_NORMALS)
        torch.cuda.synchronize()
        t_normals = (time.perf_counter() - tn0) / 3.0
        def step():
            return D.sharded_icp_point_to_plane(ctx, src, tgt, nrm, None, ICP_ITERS, None, 0.0, source_is_local_slice=True)
        for _ in range(max(args.warmup, 1)):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        if rank == 0:
            err = float(np.linalg.norm(tc.isometry_to_matrix(r.transformation).astype(np.float64) - synth.isometry_matrix(T)))
            print(json.dumps({"metric": "sharded point-to-plane ICP iterations/sec (one cloud, source sharded, 1 all-reduce/iteration)",
                              "value": ICP_ITERS * args.steps / wall, "unit": "it/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True,
                              "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"{n}-pt uniform cloud [0,10)x[0,10)x[0,1), 50-iter p2plane ICP, source sharded over ranks",
                                         "points": n, "parallelism": f"shard{world}"},
                              "transform_frobenius_error": err,
                              "sharded_normals_ms": 1e3 * t_normals, "sharded_normals_mpts_per_s": n / t_normals / 1e6}))
    else:
        frames = [synth.kitti_shaped_cloud(seed=i) for i in range(4)]
        # ego motion between consecutive frames: 1 m forward + 0.5 deg yaw
        ego = synth.yaw_isometry((-1.0, 0.0, 0.0), -np.deg2rad(0.5))
        moved = [synth.apply_isometry(ego, f) for f in frames]
        # the sensor's records: x, y, z, intensity (KITTI .bin layout), in host memory like a driver delivers them
        seq = []
        for j in range(8):
            xyz = frames[(j // 2) % 4] if j % 2 == 0 else moved[(j // 2) % 4]
            seq.append(np.ascontiguousarray(np.concatenate([xyz, np.full((len(xyz), 1), 0.5, np.float32)], axis=1)))
        nf = args.steps * 10

        def run(count):
            fs = tc.FrameStream(ctx, max_points=130000, voxel_size=0.2, k_neighbors=K_NORMALS, max_iterations=ICP_ITERS,
                                max_correspondence_distance=2.0, convergence_threshold=1e-6,
                                backpressure=tc.BackpressureConfig(max_queue_depth=4))
            t0 = time.perf_counter()
            for i in range(count):
                fs.send(seq[i % 8])                      # blocks when 4 frames are waiting (backpressure)
            res, m = fs.finish()
            return time.perf_counter() - t0, res, m

        run(max(args.warmup, 1) * 4)
        wall, res, m = run(nf)
        assert m.items_processed == nf and m.items_dropped == 0 and all(r.status == 0 for r in res)
        print(json.dumps({"metric": "LiDAR frames/sec (host frames -> bounded queue -> voxel_grid_filter 0.2 m + k=16 normals + p2plane ICP <= 50 it, "
                                    "default threshold; H2D copy overlapped with compute)",
                          "value": nf / wall, "unit": "frames/s", "n_gpus": 1, "steps": nf, "warmup": args.warmup,
                          "ms_per_step": 1e3 * wall / nf, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "120k-pt KITTI-shaped frames (64 beams x 1875 azimuth steps) as 16-byte x,y,z,intensity "
                                                 "records in host memory, tc_frame_stream_* (queue depth 4); sensor rate 10 Hz",
                                     "points": 120000},
                          "mean_iterations": float(np.mean([r.iterations for r in res])), "converged": int(sum(r.converged for r in res)),
                          "points_after_voxel_filter": int(np.mean([r.n_points for r in res])),
                          "max_queue_depth_seen": m.max_depth_seen}))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-copy-probe", action="store_true", help="skip the device-to-device copy bandwidth probe (profile runs)")
    ap.add_argument("--cloud", choices=["uniform", "tum"], default="uniform",
                    help="pairs mode: uniform-random cloud (BASELINE configs[1], the judged line) or a TUM-RGB-D-shaped "
                         "depth-map surface of ~1 M points (configs[2]; auxiliary)")
    ap.add_argument("--mode", choices=["pairs", "sharded", "stream"], default="pairs",
                    help="pairs (default, the judged metric): one independent 1M pair per GPU; sharded: ONE cloud, source "
                         "sharded over the ranks, one all-reduce of the packed 6x6 system per iteration (BASELINE config [3]); "
                         "stream: 120k-pt LiDAR-shaped frames, voxel + normals + ICP per frame (BASELINE config [4])")
    args = ap.parse_args()
    if args.mode != "pairs":
        return aux_modes(args)

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
        # ~1 M-point depth-map surface, scan-to-scan motion of a hand-held camera (1 cm, 0.3 deg)
        tgt_h = synth.tum_shaped_cloud(seed=1 + rank)
        rng = np.random.default_rng(100 + rank)
        tgt_h = (tgt_h + rng.normal(0, 1e-4, tgt_h.shape)).astype(np.float32)
        src_h = synth.apply_isometry(synth.yaw_isometry((-0.01, 0.004, 0.002), -np.deg2rad(0.3)), tgt_h)
        n = len(tgt_h)
    else:
        src_h, tgt_h, _ = synth.registration_pair(n, seed=1 + rank, transform=synth.harness_transform())
    src, tgt = torch.from_numpy(src_h).to(dev), torch.from_numpy(tgt_h).to(dev)
    ctx = tc.GpuContext(local_rank)
    # sampled hipEvents around the dominant kernel only (every 4th launch): ~1 % overhead in the timed region
    ctx.profile_enable(2)

    def step():
        t0 = time.perf_counter()
        nrm = ctx.estimate_normals(tgt, K_NORMALS)                       # (n, 6) NormalPoint3f, stays in HBM
        t1 = time.perf_counter()
        r = ctx.icp_point_to_plane_detailed(src, tgt, nrm, None, ICP_ITERS, None, 0.0, correspondences=False)
        t2 = time.perf_counter()
        assert r.iterations == ICP_ITERS
        return t1 - t0, t2 - t1, r

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
        a, b, last = step()
        tn += a
        ti += b
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t_start
    stats = ctx.profile_read()
    if world > 1:
        tw = torch.tensor([wall, tn, ti], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall, tn, ti = [float(v) for v in tw.tolist()]

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
            "metric": "ICP iterations/sec (whole job: k=16 normals + 50-iter point-to-plane ICP per 1M-pt pair)",
            "value": ICP_ITERS * args.steps * world / wall,
            "unit": "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{n}-pt uniform-random cloud, k={K_NORMALS} normals + {ICP_ITERS}-iter point-to-plane ICP "
                                    "(BASELINE configs[1]; one independent pair per GPU)") if args.cloud == "uniform" else
                                   (f"{n}-pt TUM-RGB-D-shaped depth-map surface, k={K_NORMALS} normals + {ICP_ITERS}-iter point-to-plane ICP "
                                    "(BASELINE configs[2] shape; one independent pair per GPU; auxiliary line)"),
                       "points": n, "k": K_NORMALS, "icp_iterations": ICP_ITERS, "parallelism": f"pairs{world}"},
            "normals_mpts_per_s": n * args.steps * world / tn / 1e6,
            "icp_only_it_per_s": ICP_ITERS * args.steps * world / ti,
            "roofline": {"bound": "hbm", "kernel": k, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_detail": traffic_note,
                         "alg_bytes_per_launch": ALG_BYTES_ICP * n, "avg_launch_us": avg_s * 1e6, "launches": launches,
                         "measured_copy_gbs": copy_gbs},
            "kernels_us_avg": {kk: round(1e3 * ms / max(c, 1), 2) for kk, (c, ms) in stats.items()},
            "final_mse": last.mse,
        }
        if args.cloud != "uniform":
            out["roofline"]["traffic"] = None          # PMC passes were collected on the uniform config
            out["roofline"].pop("traffic_detail", None)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, tgt_h, src_h, ctx.estimate_normals(tgt, K_NORMALS).cpu().numpy())
            out["parity"] = out["cpu_baseline"].pop("parity")
            if args.cloud == "uniform":   # source = T^-1 target: the registration must return the harness transform
                out["parity"]["icp_T_frobenius_vs_truth"] = float(np.linalg.norm(
                    tc.isometry_to_matrix(last.transformation).astype(np.float64) - synth.isometry_matrix(synth.harness_transform())))
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
