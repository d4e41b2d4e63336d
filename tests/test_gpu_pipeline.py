"""BASELINE configs[4] END TO END against the oracle: KITTI-shaped 120k-point frames through

    voxel_grid_filter(0.2)  ->  estimate_normals(k = 16) on the previous frame  ->  icp_point_to_plane(current -> previous,
    max 50 iterations, default threshold 1e-6, no maximum distance)

(filtering.rs:38-133, normals.rs:257-357, registration.rs:488-496; the frame loop is RealtimePipeline's, streaming.rs:540-646),
once as the oracle's CPU calls chained frame by frame and once through tc_frame_stream_* (host frames in, bounded queue).  The
stages are compared where they hand over, so that a difference at the end is attributed, not waved through:

  1. voxel filter: bit-exact (same centroids in the same order) -> both chains register THE SAME down-sampled clouds;
  2. normals of the previous frame: every normal beyond 1e-4 cosine is explained per point (tests/h1.py: exact boundary tie,
     degenerate smallest eigen-pair, discontinuous reference solver); the counts go into a tracked report;
  3. registration: the streamed result against the oracle's registration fed (a) the oracle's own normals -- the chain -- and
     (b) the stream's normals -- the ICP stage alone.  (b) must meet north_star's budget (h1.transform_budget: 1e-5 Frobenius x
     the coordinate scale, or the reference's own accumulation error) with equal iteration counts, or its first parting
     iteration must be decided by rounding (tools/dev/loop_fuzz.py explain_by_replay); (a) may exceed (b) only by what the
     explained normals of stage 2 move the REFERENCE's own result (oracle with its normals vs oracle with the stream's).
"""
import importlib.util
import json
import os

import numpy as np
import pytest

import threecrate_amd as tc
from oracle import oracle as O
from tests import h1
from threecrate_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VOXEL, K, MAX_IT, THR = 0.2, 16, 50, 1e-6


def _frob(a, b):
    return float(np.linalg.norm(O.isometry_to_matrix(np.asarray(a, np.float32)).astype(np.float64) -
                                O.isometry_to_matrix(np.asarray(b, np.float32)).astype(np.float64)))


def _loop_fuzz():
    spec = importlib.util.spec_from_file_location("tc_loop_fuzz", os.path.join(ROOT, "tools", "dev", "loop_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _frames(n):
    """frame i = a fresh KITTI-shaped sweep (its own range noise) seen from the pose after i ego-motion steps of 1 m + 0.5 deg"""
    ego = synth.yaw_isometry((-1.0, 0.0, 0.0), -np.deg2rad(0.5))
    out = []
    for i in range(n):
        f = synth.kitti_shaped_cloud(seed=11 + i)
        for _ in range(i):
            f = synth.apply_isometry(ego, f)
        out.append(np.ascontiguousarray(f, np.float32))
    return out


@pytest.mark.gpu
def test_config4_pipeline_end_to_end_against_the_oracle_chain(ctx):
    frames = _frames(5)
    # ---- the product: host frames (KITTI records x, y, z, intensity) through the bounded-queue stream
    fs = tc.FrameStream(ctx, max_points=130000, voxel_size=VOXEL, k_neighbors=K, max_iterations=MAX_IT,
                        max_correspondence_distance=None, convergence_threshold=THR,
                        backpressure=tc.BackpressureConfig(max_queue_depth=2))
    for fr in frames:
        fs.send(np.concatenate([fr, np.full((len(fr), 1), 0.5, np.float32)], axis=1))
    res, m = fs.finish()
    assert m.items_processed == 5 and m.items_dropped == 0 and len(res) == 4 and all(r.status == 0 for r in res)

    # ---- the oracle's chain, stage by stage
    lf = _loop_fuzz()
    vox = [O.voxel_grid_filter(f, VOXEL) for f in frames]
    report = {"voxel": VOXEL, "k": K, "max_iterations": MAX_IT, "threshold": THR, "frames": []}
    for i in range(1, 5):
        prev, cur = vox[i - 1], vox[i]
        # 1. voxel filter, bit for bit (output order: ascending (kx, ky, kz) on both sides; the reference's own order is its HashMap's)
        gv = ctx.voxel_grid_filter(frames[i], VOXEL)
        assert gv.shape == cur.shape and np.array_equal(gv, cur)
        g = res[i - 1]
        assert g.n_points_in == len(frames[i]) and g.n_points == len(cur)
        # 2. normals of the previous (down-sampled) frame
        ref_n = O.estimate_normals(prev, K)
        gpu_n = ctx.estimate_normals(prev, K)
        nrep = h1.normals_report(prev, K, gpu_n, ref_n, max_offenders=max(200, len(prev) // 20))
        reasons = h1.offender_reasons(nrep)
        # 3. registration current -> previous
        gn = np.ascontiguousarray(gpu_n[:, 3:])
        on = np.ascontiguousarray(ref_n[:, 3:])
        chain = O.icp_point_to_plane_detailed(cur, prev, on, None, MAX_IT, None, THR)             # (a) the oracle's chain
        stage = O.icp_point_to_plane_detailed(cur, prev, gn, None, MAX_IT, None, THR)             # (b) the ICP stage alone
        scale = max(1.0, float(np.abs(prev).max()))
        fr = {"frame": i, "points_in": len(frames[i]), "points_after_voxel_filter": len(cur),
              "normals": {"n": nrep["n"], "beyond_1e-4": nrep["n_beyond"], "bit_identical": nrep["n_bit_identical"], "worst_1_minus_abs_cos": nrep["worst"],
                          "offenders_by_reason": reasons},
              "iterations": {"stream": g.iterations, "oracle_chain": chain.iterations, "oracle_with_stream_normals": stage.iterations},
              "converged": {"stream": bool(g.converged), "oracle_chain": bool(chain.converged), "oracle_with_stream_normals": bool(stage.converged)},
              "frobenius": {"stream_vs_oracle_chain": _frob(g.transformation, chain.transformation),
                            "stream_vs_oracle_with_stream_normals": _frob(g.transformation, stage.transformation),
                            "oracle_chain_vs_oracle_with_stream_normals": _frob(chain.transformation, stage.transformation)},
              "coordinate_scale": scale}
        report["frames"].append(fr)
        # (b): the ICP stage on identical inputs
        if (g.iterations, bool(g.converged)) == (stage.iterations, bool(stage.converged)):
            fr["stage_budget"] = h1.transform_budget(g.transformation, lambda: stage,
                                                     lambda: O.icp_point_to_plane_detailed(cur, prev, gn, None, MAX_IT, None, THR, exact_sums=True),
                                                     1e-5, scale=scale)
        else:
            grun = lambda k: ctx.icp_point_to_plane_detailed(cur, prev, gn, None, k, None, THR)
            orun = lambda k: O.icp_point_to_plane_detailed(cur, prev, gn, None, k, None, THR)
            ok, text = lf.explain_by_replay(grun, orun, cur, prev, None, MAX_IT, THR, scale, nrm_t=gn)
            fr["stage_parting"] = text
            assert ok, (fr, text)
        # the streamed registration IS the plain call on the same inputs (frame streaming changes scheduling, not arithmetic)
        plain = ctx.icp_point_to_plane_detailed(cur, prev, gn, None, MAX_IT, None, THR)
        assert plain.iterations == g.iterations and bool(plain.converged) == bool(g.converged)
        assert _frob(plain.transformation, g.transformation) <= 1e-5 * scale
        # (a): the chain may be farther off only by what the explained normals move the reference's own result
        f = fr["frobenius"]
        if nrep["n_beyond"] == 0 and (g.iterations, bool(g.converged)) == (chain.iterations, bool(chain.converged)):
            h1.transform_budget(g.transformation, lambda: chain,
                                lambda: O.icp_point_to_plane_detailed(cur, prev, on, None, MAX_IT, None, THR, exact_sums=True), 1e-5, scale=scale)
        else:
            assert f["stream_vs_oracle_chain"] <= f["oracle_chain_vs_oracle_with_stream_normals"] + max(f["stream_vs_oracle_with_stream_normals"], 1e-5 * scale) * 1.0001, fr
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "h1_config4_pipeline.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    # the ego motion is recovered (current -> previous = the inverse of one ego step of 1 m + 0.5 deg yaw; independent range
    # noise on every sweep: centimetres)
    T_true = synth.invert_isometry(synth.yaw_isometry((-1.0, 0.0, 0.0), -np.deg2rad(0.5)))
    for g in res:
        M = O.isometry_to_matrix(g.transformation).astype(np.float64)
        assert np.linalg.norm(M - T_true) < 0.05, (M, T_true)
