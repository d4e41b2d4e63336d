#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ with the CPU oracle (run in the build container).

The reference is Rust and cannot be imported or built here, and its own tests hold no golden
vectors, so these fixtures pin OUR oracle's outputs on the
BASELINE config-[0] plumbing case (10k points, k=10 normals, 20-iteration ICP); inputs are
regenerated from the counter-based RNG (threecrate_amd.synth), only outputs are stored.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O  # noqa: E402
from tests.helpers import sphere_cloud  # noqa: E402
from threecrate_amd import synth  # noqa: E402


def corr_digest(c):
    return hashlib.sha256(np.ascontiguousarray(c, dtype=np.int64).tobytes()).hexdigest()


def res_dict(r):
    return {"transformation": [float(np.float32(v)) for v in r.transformation], "mse": r.mse,
            "iterations": r.iterations, "converged": r.converged, "n_correspondences": int(len(r.correspondences)),
            "correspondences_sha256": corr_digest(r.correspondences)}


def main():
    n = 10000
    pts = synth.uniform_cloud(n, seed=1)
    nrm10 = O.estimate_normals(pts, 10)
    np.save(os.path.join(HERE, "normals_u10k_k10.npy"), nrm10[:, 3:].astype(np.float32))
    nrm16 = O.estimate_normals(pts, 16)
    np.save(os.path.join(HERE, "normals_u10k_k16.npy"), nrm16[:, 3:].astype(np.float32))
    idx, dist, cnt = O.knn_batch(pts, pts[:1000], 17)
    np.save(os.path.join(HERE, "knn_u10k_k17_q1k_idx.npy"), np.sort(idx, axis=1).astype(np.uint32))
    np.save(os.path.join(HERE, "knn_u10k_k17_q1k_dist.npy"), dist.astype(np.float32))

    src, tgt, T = synth.registration_pair(n, seed=1)
    out = {"T_true": [float(v) for v in T]}
    out["icp_p2p_u10k_20it"] = res_dict(O.icp_detailed(src, tgt, None, 20, None, 0.0))
    out["icp_p2p_u10k_default"] = res_dict(O.icp_detailed(src, tgt, None, 50, None, 1e-6))
    n16 = O.estimate_normals(tgt, 16)[:, 3:]
    out["icp_p2pl_u10k_20it"] = res_dict(O.icp_point_to_plane_detailed(src, tgt, n16, None, 20, None, 0.0))
    out["icp_p2pl_u10k_default"] = res_dict(O.icp_point_to_plane(src, tgt, n16, None, 50))
    out["icp_p2p_u10k_maxdist"] = res_dict(O.icp_detailed(src, tgt, None, 10, 0.02, 1e-9))
    s, nn = sphere_cloud(100)
    out["icp_p2pl_sphere100_shift"] = res_dict(O.icp_point_to_plane(s, s + np.array([0.15, 0, 0], np.float32), nn, None, 50))
    with open(os.path.join(HERE, "icp_u10k.json"), "w") as f:
        json.dump(out, f, indent=1)
    vox = O.voxel_grid_filter(pts, 0.1)
    np.save(os.path.join(HERE, "voxel_u10k_0p1.npy"), vox)
    print("golden fixtures written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
