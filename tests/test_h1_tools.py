"""The explanation tools of tests/h1.py are test infrastructure themselves: a case they once missed stays a test.
Fuzz seed 112 case 5754 (round 3): a 17-point cross on a lattice -- tie groups of 5 / 4 / 7 neighbours at three distances -- whose
covariance, summed in f32 in one legal order of the tied neighbours, makes the REFERENCE's eigen-solver (nalgebra symmetric_eigen as
restated in oracle/) pair the smallest eigenvalue with the other eigenvector of a 2 x 2 block whose diagonal entries agree to 1e-7.
reference_solver_spread must see that (it samples orders inside the tie groups when given the query)."""
import numpy as np

from oracle import oracle as O
from tests import h1


def _case():
    rng = np.random.default_rng([112, 5754])
    n = int(rng.choice([2, 3, 7, 40, 300, 1500, 6000])); kind = int(rng.integers(0, 6))
    assert (n, kind) == (1500, 4)
    p = np.round(rng.random((n, 3)) * 8) / 8
    return (p * rng.choice([1e-2, 1.0, 50.0])).astype(np.float32)


def test_tie_group_orders_expose_the_reference_solvers_flip():
    p = _case()
    i, k = 904, 16
    idx, _ = O.KdTree(p).find_k_nearest(p[i], k + 1)
    nbh = [int(a) for a in idx if int(a) != i][:k] + [i]
    d2 = h1.d2_f32(p[nbh[:-1]], p[i])
    assert len(np.unique(d2)) == 3                      # three tie groups: 5 + 4 + 7 neighbours
    assert h1.reference_solver_spread(p[nbh], query=p[i]) > 0.5          # some legal order flips the normal by ~90 degrees
    # the two orders that met in the fuzz run, through the oracle's own solver
    def cov(order):
        q = p[order]; cen = np.zeros(3, np.float32)
        for v in q: cen = (cen + v).astype(np.float32)
        cen = (cen / np.float32(len(q))).astype(np.float32)
        c = np.zeros((3, 3), np.float32)
        for v in q:
            d = (v - cen).astype(np.float32); c = (c + np.outer(d, d).astype(np.float32)).astype(np.float32)
        return (c / np.float32(len(q))).astype(np.float32)
    kd_order = [1302, 509, 113, 165, 897, 991, 393, 692, 1177, 514, 79, 567, 97, 576, 214, 1201, 904]
    pos_order = [113, 165, 509, 897, 1302, 692, 991, 1177, 393, 79, 514, 1201, 97, 214, 567, 576, 904]
    assert sorted(kd_order) == sorted(pos_order) == sorted(nbh)
    na, nb = h1.reference_normal_of_cov(cov(kd_order)), h1.reference_normal_of_cov(cov(pos_order))
    assert abs(float(np.dot(na, nb))) < 1e-3            # orthogonal answers from the same set, same solver
