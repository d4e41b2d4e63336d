"""world_size-2 gloo tests (CPU) of the N>1 host path: shard ranges, the per-iteration
all-reduce of the packed 29-word system, identical state on every rank, independent-job
partitioning.  The per-shard kernel work is stood in for by the CPU oracle (checker backend)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from threecrate_amd import distributed as D
from threecrate_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleShardBackend:
    """Checker backend: per-shard packed sums from the oracle (tco_p2plane_partial), the solve /
    compose / convergence bookkeeping restated with numpy f64 + the oracle's isometry algebra."""

    def __init__(self, src_slice, tgt, nrm, init, max_dist, thr):
        from oracle import oracle as O
        self.O, self.src, self.nrm = O, np.ascontiguousarray(src_slice, np.float32), np.ascontiguousarray(nrm, np.float32)
        self.tree = O.KdTree(tgt)
        self.T = np.asarray(init, np.float32).copy()
        self.max_dist, self.thr = max_dist, np.float32(thr)
        self.prev, self.mse, self.iters, self.converged, self.failed = np.float32(np.inf), np.float32(0), 0, False, False

    def reduce(self):
        out, _ = self.O.p2plane_partial(self.src, 0, len(self.src), self.tree, self.nrm, self.T, self.max_dist)
        s = torch.zeros(D.SUMS, dtype=torch.float64)
        s[:29] = torch.from_numpy(out)
        return s

    def apply(self, sums):
        if self.converged or self.failed:
            return
        S = sums.numpy()
        cnt = S[28]
        if cnt < 6:
            self.failed = True
            return
        A = np.zeros((6, 6))
        A[np.triu_indices(6)] = S[:21]
        A = A + A.T - np.diag(np.diag(A))
        x = np.linalg.solve(A, S[21:27]).astype(np.float32)
        h = x[:3] / np.float32(2)
        qx = np.array([np.sin(h[0]), 0, 0, np.cos(h[0])], np.float32)
        qy = np.array([0, np.sin(h[1]), 0, np.cos(h[1])], np.float32)
        qz = np.array([0, 0, np.sin(h[2]), np.cos(h[2])], np.float32)
        ident = np.zeros(3, np.float32)
        rot = self.O.isometry_mul(self.O.isometry_mul(np.r_[qz, ident], np.r_[qy, ident]), np.r_[qx, ident])[:4]
        self.T = self.O.isometry_mul(np.r_[rot, x[3:6]].astype(np.float32), self.T)
        mse = np.float32(S[27] / cnt)
        self.iters += 1
        self.mse = mse
        if abs(self.prev - mse) < self.thr:
            self.converged = True
            return
        self.prev = mse

    def done(self):
        return self.converged or self.failed

    def finish(self, max_iters):
        return dict(T=self.T, mse=float(self.mse if self.converged else self.prev),
                    iterations=self.iters if self.converged else max_iters, converged=self.converged)


class OracleNormalsBackend:
    """Checker backend of sharded_normals: a fixed "cell-sorted" order (a permutation every rank derives the same way),
    records of a slice from the oracle's normals, unsort = the inverse permutation."""

    def __init__(self, pts, k):
        from oracle import oracle as O
        self.n = len(pts)
        self.perm = np.random.default_rng(99).permutation(self.n)          # sorted position -> original index
        self.full = O.estimate_normals(pts, k, threads=2)                  # (a real rank computes only its slice)
        self.calls = []

    def slice(self, begin, end):
        self.calls.append((begin, end))
        return torch.from_numpy(self.full[self.perm[begin:end]].copy())

    def unsort(self, sorted_all):
        out = np.empty((self.n, 6), np.float32)
        out[self.perm] = sorted_all.numpy()
        return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        src, tgt, T = synth.registration_pair(6000, seed=11)
        nrm = O.estimate_normals(tgt, 12, threads=2)[:, 3:]
        lo, hi = D.shard_range(len(src), rank, world)
        be = OracleShardBackend(src[lo:hi], tgt, nrm, O.IDENTITY, None, 1e-6)
        res = D.sharded_icp_loop(be, 30, poll_every=1)
        be2 = OracleShardBackend(src[lo:hi], tgt, nrm, O.IDENTITY, None, 0.0)
        res2 = D.sharded_icp_loop(be2, 6)
        jobs = list(range(5))
        got = D.run_independent_jobs(jobs, lambda j: (j * j, rank))
        nb = OracleNormalsBackend(tgt[:5001], 10)                           # odd size: unequal slices, padded gather
        normals = D.sharded_normals(nb)
        q.put((rank, res, res2, got, (normals, nb.calls, nb.full)))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_everything():
    for n in [0, 1, 7, 8, 1000003]:
        for w in [1, 2, 3, 8]:
            r = [D.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_icp_world2_matches_single_process():
    from oracle import oracle as O
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, a0, b0, j0, n0), (r1, a1, b1, j1, n1) = outs
    # sharded normals: each rank computed only its slice, both hold the full array in input order
    assert n0[1] == [(0, 2501)] and n1[1] == [(2501, 5001)]
    assert np.array_equal(n0[0], n0[2]) and np.array_equal(n1[0], n0[2])
    # every rank ends with the bit-identical state (same reduced buffer, same solve)
    assert np.array_equal(a0["T"], a1["T"]) and a0["iterations"] == a1["iterations"] and a0["converged"] == a1["converged"]
    assert np.array_equal(b0["T"], b1["T"])
    # and it equals the single-process reference run within the parity budget
    src, tgt, T = synth.registration_pair(6000, seed=11)
    nrm = O.estimate_normals(tgt, 12)[:, 3:]
    ref = O.icp_point_to_plane(src, tgt, nrm, None, 30)
    assert (a0["converged"], a0["iterations"]) == (ref.converged, ref.iterations)
    assert np.linalg.norm(O.isometry_to_matrix(a0["T"]).astype(np.float64) - O.isometry_to_matrix(ref.transformation)) <= 1e-5
    ref2 = O.icp_point_to_plane_detailed(src, tgt, nrm, None, 6, None, 0.0)
    assert b0["iterations"] == 6 and not b0["converged"]
    assert np.linalg.norm(O.isometry_to_matrix(b0["T"]).astype(np.float64) - O.isometry_to_matrix(ref2.transformation)) <= 1e-5
    # independent jobs: rank r ran jobs r, r+2, ...; every rank sees all results in job order
    assert j0 == j1 == [(0, 0), (1, 1), (4, 0), (9, 1), (16, 0)]
