#!/usr/bin/env python3
"""Dataset bench with the reference harness' command line and CSV row, on the HIP backend.

Mirrors examples/threecrate_dataset_bench.rs (flags :388-415, defaults :37-50, tasks :129-275, protocol :72-91,
CSV :93-113): `--warmups` untimed whole calls, `--iterations` timed whole calls on HOST buffers (the drop-in
view: uploads and read-backs are inside the timed call, like the reference's gpu_* tasks), median / min / mean ms,
one CSV row with the same columns, so scripts/bench_cross_library.py-style merges keep working.

--source is a KITTI velodyne `.bin` file (threecrate-io/src/lidar.rs:310-343 records) or one of the synthetic
stand-ins `synthetic:tum`, `synthetic:kitti`, `synthetic:nuscenes` (same sizes and shapes as the frames of
docs/benchmarks.md:41-49; there is no network for the datasets).  Without --target the target is the source moved
by the harness transform (:281-287).

    python tools/dataset_bench.py --task normals --dataset kitti --source synthetic:kitti --max-points all
"""
import argparse
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (first: see tests/conftest.py)
import threecrate_amd as tc  # noqa: E402
from threecrate_amd import synth  # noqa: E402

HEADER = "library,task,dataset,source_points,target_points,output_points,iterations,median_ms,min_ms,mean_ms,detail"
TASKS = ("read", "voxel", "normals", "icp", "multiscale_icp", "icp_point_to_plane", "knn")


def load_cloud(spec):
    if spec.startswith("synthetic:"):
        kind = spec.split(":", 1)[1]
        rng = np.random.default_rng(0)
        if kind == "tum":
            pts = synth.tum_shaped_cloud(seed=1, step=2.085)          # ~230 k valid pixels
        elif kind == "kitti":
            pts = synth.kitti_shaped_cloud(seed=1)                     # 120 k returns
        elif kind == "nuscenes":
            pts = synth.kitti_shaped_cloud(beams=32, azimuth_steps=1090, seed=1)   # ~35 k returns
        else:
            raise SystemExit(f"unknown synthetic source: {kind}")
        # sub-millimetre jitter: exact lattices / planes are the reference kd-tree's quadratic case
        return (pts + rng.normal(0.0, 1e-4, pts.shape)).astype(np.float32)
    return tc.read_kitti_bin(spec)


def csv_escape(v):
    return '"' + v.replace('"', '""') + '"' if any(c in v for c in ',"\n') else v


def run_task(a, ctx, source, target):
    """-> (output_points, detail); argument meanings as in the reference's run_task"""
    if a.task == "read":
        cloud = load_cloud(a.source)
        return len(cloud), "read_point_cloud"
    if a.task == "voxel":
        out = ctx.voxel_grid_filter(source, a.voxel_size)
        return len(out), f"voxel_size={a.voxel_size}"
    if a.task == "normals":
        out = ctx.estimate_normals(source, 10)
        return len(out), "k=10"
    if a.task == "icp":
        r = ctx.icp_point_to_point(source, target, None, a.max_icp_iters, a.convergence, None)
        return len(r.correspondences), f"icp_iters={r.iterations},converged={str(r.converged).lower()},mse={r.mse:.6f}"
    if a.task == "icp_point_to_plane":      # not a task of the reference harness: the headline path of this backend
        nrm = ctx.estimate_normals(target, 10)[:, 3:6]
        r = ctx.icp_point_to_plane_detailed(source, target, nrm, None, a.max_icp_iters, None, a.convergence)
        return len(r.correspondences), f"k=10,icp_iters={r.iterations},converged={str(r.converged).lower()},mse={r.mse:.6f}"
    if a.task == "multiscale_icp":
        cfg = tc.MultiScaleIcpConfig(
            levels=[tc.IcpScaleLevel(0.20, min(a.max_icp_iters, 10), 0.50), tc.IcpScaleLevel(0.10, min(a.max_icp_iters, 10), 0.25),
                    tc.IcpScaleLevel(0.05, a.max_icp_iters, 0.15)],
            final_refinement_iterations=a.max_icp_iters, final_max_correspondence_distance=0.10,
            convergence_threshold=a.convergence)
        r = ctx.multiscale_icp_point_to_point(source, target, None, cfg)
        return len(r.correspondences), f"levels=3,total_iters={r.iterations},converged={str(r.converged).lower()},mse={r.mse:.6f}"
    if a.task == "knn":
        nq = min(len(source), 256)
        _, _, cnt = ctx.find_k_nearest_batch(source, source[:nq], 8)
        return int(cnt.sum()), f"queries={nq},k=8"
    raise SystemExit(f"unsupported task: {a.task}; expected one of {', '.join(TASKS)}")


def parse_max_points(v):
    return None if v.lower() == "all" or v == "0" else int(v)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--task", default="icp")
    ap.add_argument("--dataset", default="dataset")
    ap.add_argument("--source", required=True)
    ap.add_argument("--target", default=None)
    ap.add_argument("--iterations", type=int, default=5)
    ap.add_argument("--warmups", type=int, default=1)
    ap.add_argument("--max-points", type=parse_max_points, default=20000)
    ap.add_argument("--voxel-size", type=float, default=0.2)
    ap.add_argument("--max-icp-iters", type=int, default=20)
    ap.add_argument("--convergence", type=float, default=1e-5)
    ap.add_argument("--no-header", action="store_true")
    a = ap.parse_args(argv)
    if a.iterations <= 0:
        raise SystemExit("--iterations must be greater than zero")

    source = load_cloud(a.source)[:a.max_points]
    target = load_cloud(a.target) if a.target else synth.apply_isometry(synth.harness_transform(), source)
    target = np.ascontiguousarray(target[:a.max_points], np.float32)
    source = np.ascontiguousarray(source, np.float32)
    ctx = tc.GpuContext(0)

    for _ in range(a.warmups):
        run_task(a, ctx, source, target)
    times, last = [], None
    for _ in range(a.iterations):
        t0 = time.perf_counter()
        last = run_task(a, ctx, source, target)
        times.append((time.perf_counter() - t0) * 1e3)
    if not a.no_header:
        print(HEADER)
    print(",".join(["threecrate-hip", a.task, csv_escape(a.dataset), str(len(source)), str(len(target)), str(last[0]), str(a.iterations),
                    f"{statistics.median(times):.3f}", f"{min(times):.3f}", f"{statistics.fmean(times):.3f}", csv_escape(last[1])]))


if __name__ == "__main__":
    main()
