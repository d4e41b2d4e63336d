"""Pins the CPU oracle against every known-answer assertion the reference's own tests hold for the
normals + ICP path (normals.rs:394-625, registration.rs:797-1267, nearest_neighbor.rs:389-728)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import kats
from tests.backends import OracleBackend


@pytest.fixture(scope="module")
def backend():
    return OracleBackend()


@pytest.mark.parametrize("kat", kats.ALL_KATS, ids=lambda f: f.__name__)
def test_reference_kat(backend, kat):
    kat(backend)


CUBE = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1], [1, 1, 1]], np.float32)


def test_kdtree_empty():
    """nearest_neighbor.rs:418-426"""
    t = O.KdTree(np.zeros((0, 3), np.float32))
    idx, d = t.find_k_nearest([0, 0, 0], 5)
    assert len(idx) == 0


def test_kdtree_knn_consistency_cube():
    """nearest_neighbor.rs:429-483, :685-727"""
    t = O.KdTree(CUBE)
    idx, d = t.find_k_nearest([0.5, 0.5, 0.5], 3)
    bi, bd = O.brute_knn(CUBE, [0.5, 0.5, 0.5], 3)
    assert len(idx) == len(bi) == 3
    assert np.all(np.diff(d) >= 0) and np.all(np.diff(bd) >= 0)
    assert np.all(np.abs(np.sort(d) - np.sort(bd)) < 1e-6)


def test_kdtree_radius_consistency_cube():
    """nearest_neighbor.rs:486-538"""
    t = O.KdTree(CUBE)
    idx, d = t.find_radius_neighbors([0.5, 0.5, 0.5], 1.5)
    assert len(idx) == 8 and np.all(d <= 1.5) and np.all(np.diff(d) >= 0)


def test_kdtree_edge_cases():
    """nearest_neighbor.rs:541-563"""
    t = O.KdTree(CUBE)
    assert len(t.find_k_nearest([0, 0, 0], 0)[0]) == 0
    assert len(t.find_k_nearest([0, 0, 0], 20)[0]) == 8
    assert len(t.find_radius_neighbors([0, 0, 0], 0.0)[0]) == 0
    assert len(t.find_radius_neighbors([0, 0, 0], -1.0)[0]) == 0


def test_kdtree_random_points_vs_bruteforce():
    """nearest_neighbor.rs:566-641 (seeded instead of thread_rng)"""
    rng = np.random.default_rng(7)
    pts = rng.uniform(-10, 10, (100, 3)).astype(np.float32)
    t = O.KdTree(pts)
    for _ in range(10):
        q = rng.uniform(-5, 5, 3).astype(np.float32)
        k = int(rng.integers(1, 11))
        radius = float(rng.uniform(1.0, 5.0))
        idx, d = t.find_k_nearest(q, k)
        bi, bd = O.brute_knn(pts, q, k)
        assert len(idx) == len(bi) == min(k, 100)
        assert np.all(np.abs(d - bd) < 1e-6)
        ri, rd = t.find_radius_neighbors(q, radius)
        dd = np.sqrt(((pts - q) ** 2).sum(1, dtype=np.float32))
        assert len(ri) == int((dd * dd <= np.float32(radius) * np.float32(radius)).sum()) or abs(len(ri) - (dd <= radius).sum()) <= 1


def test_find_neighbors_counts():
    """normals.rs:595-624 test_find_neighbors (k-NN = 2 neighbours, radius 1.5 -> 2 neighbours)"""
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 0, 0]], np.float32)
    t = O.KdTree(pts)
    idx, _ = t.find_k_nearest(pts[0], 3)
    assert len([i for i in idx if i != 0][:2]) == 2
    ri, _ = t.find_radius_neighbors(pts[0], 1.5)
    assert len([i for i in ri if i != 0]) == 2


def test_voxel_grid_filter_kats():
    """filtering.rs:537-576 (empty, single point, duplicates -> 3, invalid size -> Err)"""
    assert len(O.voxel_grid_filter(np.zeros((0, 3), np.float32), 0.1)) == 0
    assert len(O.voxel_grid_filter(np.array([[0, 0, 0]], np.float32), 0.1)) == 1
    pts = np.array([[0, 0, 0], [0, 0, 0], [0.1, 0, 0], [0.1, 0, 0], [0, 0.1, 0]], np.float32)
    assert len(O.voxel_grid_filter(pts, 0.05)) == 3
    one = np.array([[0, 0, 0]], np.float32)
    with pytest.raises(O.OracleError):
        O.voxel_grid_filter(one, 0.0)
    with pytest.raises(O.OracleError):
        O.voxel_grid_filter(one, -1.0)
