"""threecrate_amd.compat: the pyo3 module's call shapes (threecrate-python/threecrate.pyi) on the HIP backend."""
import numpy as np
import pytest
import torch  # noqa: F401  (first: see conftest.py)

import threecrate_amd.compat as threecrate
from threecrate_amd import api, synth


def test_point_cloud_container_behaviour():
    """PointCloud / NormalPointCloud semantics of threecrate-python/src/lib.rs (PyPointCloud, PyNormalPointCloud)"""
    a = np.arange(12, dtype=np.float64).reshape(4, 3)
    pc = threecrate.PointCloud(a)
    assert len(pc) == 4 and not pc.is_empty and pc.points.dtype == np.float32 and repr(pc) == "PointCloud(4 points)"
    assert np.array_equal(pc[-1], a[3].astype(np.float32)) and np.array_equal(np.asarray(pc), a.astype(np.float32))
    assert len(pc + threecrate.PointCloud.from_numpy(a.astype(np.float32))) == 8
    assert threecrate.PointCloud().is_empty and len(threecrate.concatenate([pc, pc, pc])) == 12
    with pytest.raises(IndexError):
        pc[4]
    for bad in (np.zeros((3, 2)), np.zeros(3), np.zeros((2, 3), np.int32), [[0, 0, 0]]):
        with pytest.raises(ValueError):
            threecrate.PointCloud(bad)
    npc = threecrate.NormalPointCloud.from_numpy(a, a[::-1].copy())
    assert len(npc) == 4 and np.array_equal(npc.normals(), a[::-1].astype(np.float32)) and repr(npc) == "NormalPointCloud(4 points)"
    with pytest.raises(ValueError):
        threecrate.NormalPointCloud.from_numpy(a, a[:2])


def test_init_transform_conversion_round_trips():
    """numpy_to_isometry (lib.rs:76-137) / to_homogeneous (lib.rs:48-61)"""
    rng = np.random.default_rng(3)
    for _ in range(50):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        T = np.concatenate([q, rng.normal(size=3) * 5]).astype(np.float32)
        back = threecrate._isometry(api.isometry_to_matrix(T))
        if np.dot(back[:4], T[:4]) < 0:
            back[:4] = -back[:4]
        assert np.abs(back - T).max() < 2e-6
    assert threecrate._isometry(None) is None
    for bad in (np.eye(3), np.eye(4, dtype=np.int64), [[1, 0, 0, 0]] * 4):
        with pytest.raises(ValueError):
            threecrate._isometry(bad)
    m = api.isometry_to_matrix(synth.yaw_isometry((1.0, -2.0, 0.5), 0.3))
    pts = synth.uniform_cloud(100, 2)
    moved = threecrate.transform_point_cloud(threecrate.PointCloud(pts), m.astype(np.float64)).to_numpy()
    assert np.abs(moved - synth.apply_isometry(synth.yaw_isometry((1.0, -2.0, 0.5), 0.3), pts)).max() < 1e-6


@pytest.mark.gpu
def test_module_functions_match_the_oracle():
    from oracle import oracle as O
    from tests.helpers import cos_abs
    pts = synth.uniform_cloud(4000, 11)
    cloud = threecrate.PointCloud(pts)
    nc = threecrate.estimate_normals(cloud)                       # k_neighbors = 10
    ref = O.estimate_normals(pts, 10)
    assert np.array_equal(nc.positions(), pts) and cos_abs(nc.normals(), ref[:, 3:6]).min() >= 1 - 1e-4
    src_pts, _, _ = synth.registration_pair(len(pts), seed=11)
    src = threecrate.PointCloud(src_pts)
    r = threecrate.icp(src, cloud, max_iterations=15)
    o = O.icp_point_to_point(src.to_numpy(), pts, None, 15, 1e-6, None)
    assert r.iterations == o.iterations and r.converged == o.converged and abs(r.mse - o.mse) <= 1e-6 + 1e-3 * o.mse
    assert np.abs(r.transformation() - O.isometry_to_matrix(o.transformation)).max() <= 1e-5 and r.transformation().dtype == np.float32
    assert repr(r).startswith("IcpResult(converged=")
    r2 = threecrate.icp_point_to_plane(src, nc, max_iterations=10, init_transform=np.eye(4))
    o2 = O.icp_point_to_plane(src.to_numpy(), pts, ref[:, 3:6].copy(), None, 10)
    assert np.abs(r2.transformation() - O.isometry_to_matrix(o2.transformation)).max() <= 1e-4
    g = threecrate.gicp(src, cloud, max_iterations=5)
    k = threecrate.kiss_icp(threecrate.PointCloud(pts * 20), threecrate.PointCloud(pts * 20), voxel_size=1.0, max_iterations=5)
    assert g.transformation().shape == (4, 4) and k.transformation().shape == (4, 4)
    down = threecrate.voxel_downsample(cloud, 0.1)
    assert np.array_equal(down.to_numpy(), O.voxel_grid_filter(pts, 0.1))
    tree = threecrate.KdTree(cloud)
    idx, dist = tree.knn(pts[7].astype(np.float64), 5)
    d = np.linalg.norm(pts - pts[7], axis=1)
    assert idx[0] == 7 and len(idx) == 5 and np.allclose(dist, np.sort(d)[:5], atol=1e-6)
    ridx, rdist = tree.radius_search(pts[7], 0.08)
    assert set(ridx) == set(np.nonzero(d <= 0.08)[0].tolist()) and len(rdist) == len(ridx)
    with pytest.raises(RuntimeError):
        threecrate.voxel_downsample(cloud, 0.0)
    with pytest.raises(RuntimeError):
        threecrate.icp(threecrate.PointCloud(), cloud)
    assert threecrate.KdTree(threecrate.PointCloud()).knn(np.zeros(3), 3) == ([], [])
    with pytest.raises(ValueError):
        tree.knn(np.zeros(2), 3)
