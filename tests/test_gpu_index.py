"""-m gpu: the two placements of the index build give the SAME layout.

Round 4 added a binned placement (grid.hip: bin_count / bin_scatter / bin_place -- LDS stages instead of one global atomic per point)
for large clouds on dense-ish grids; the atomic counting sort of rounds 1-3 stays for everything else and as the fallback when a
bin overflows a block's LDS.  Both must produce the cell-sorted order with ascending original index inside a cell, bit for bit:
every search result, tie-break and sum order downstream depends on it.  TC_INDEX_BINNED is read per call."""
import numpy as np
import pytest

import threecrate_amd as tc
from threecrate_amd import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _clouds():
    rng = np.random.default_rng(11)
    u = synth.uniform_cloud(400_000, seed=3)
    yield "uniform 400k", u
    # a lattice with exact duplicates: distance ties everywhere, cells with several points -> the in-cell order decides tie-breaks
    g = np.stack(np.meshgrid(np.arange(70), np.arange(70), np.arange(60), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(0.01)
    lat = np.concatenate([g, g[rng.integers(0, len(g), 40_000)]]).astype(np.float32)
    yield "lattice + duplicates 334k", lat[rng.permutation(len(lat))]
    nf = u.copy()
    nf[rng.integers(0, len(nf), 50)] = np.nan
    nf[rng.integers(0, len(nf), 20), 1] = np.inf
    yield "uniform with non-finite points", nf
    # strongly non-uniform: most bins nearly empty, a few far beyond a block's LDS -> the build must fall back, same answers
    clu = np.concatenate([synth.uniform_cloud(200_000, seed=5), (0.5 + 0.002 * rng.standard_normal((120_000, 3))).astype(np.float32)])
    yield "uniform + dense cluster", clu.astype(np.float32)


@pytest.mark.parametrize("name,pts", list(_clouds()), ids=lambda v: v if isinstance(v, str) else "")
def test_binned_and_atomic_index_builds_agree_bit_for_bit(ctx, monkeypatch, name, pts):
    d = torch.from_numpy(pts).cuda()
    qh = np.ascontiguousarray(pts[::97][:3000])
    qh = qh[np.isfinite(qh).all(1)]
    src = torch.from_numpy((pts[::3] + np.float32(0.003)).astype(np.float32)).cuda()          # (the ICP source: ordered by tile-major target cell)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TC_INDEX_BINNED", mode)
        nrm = ctx.estimate_normals(d, 12)
        idx, dist, cnt = ctx.find_k_nearest_batch(pts, qh, 9)
        fin = torch.isfinite(d).all(1)
        r = ctx.icp_point_to_plane_detailed(src[torch.isfinite(src).all(1)], d[fin], nrm[fin], None, 4, None, 0.0, correspondences=True)
        h = tc.Cloud(ctx, d)
        hn = h.estimate_normals(12)
        h.close()
        out[mode] = (nrm.cpu().numpy(), np.asarray(idx), np.asarray(dist), np.asarray(cnt), r.transformation, r.mse, r.correspondences, hn.cpu().numpy())
    monkeypatch.delenv("TC_INDEX_BINNED")
    a, b = out["1"], out["0"]
    for x, y, what in zip(a, b, ("normals", "knn idx", "knn dist", "knn count", "icp T", "icp mse", "icp pairs", "handle normals")):
        assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True), (name, what)


def test_a_cell_of_a_hundred_thousand_points_is_ordered_deterministically(ctx, monkeypatch):
    """A cluster far below the cell edge: ~120 k points of a 400 k cloud in ONE cell.  Until round 4 a cell of more than 65 536
    points kept the atomic arrival order of its records (the only documented source of run-to-run differences): about one run in
    ten matched a different member of an equidistant neighbour pair somewhere in the cluster (tools/dev/index_stress.py).  The
    cut-off is 2^20 now: four builds -- both placements, twice each -- give the same bits; with the cut-off lowered below the
    cell's population the build still works (arrival order inside that cell: results valid, not compared)."""
    rng = np.random.default_rng(3)
    u = synth.uniform_cloud(280_000, seed=8)
    pts = np.concatenate([u, (np.float32(0.4137) + 0.002 * rng.standard_normal((120_000, 3))).astype(np.float32)]).astype(np.float32)
    d = torch.from_numpy(pts).cuda()
    src = torch.from_numpy((pts[::3] + np.float32(0.003)).astype(np.float32)).cuda()
    runs = []
    for mode in ("1", "0", "1", "0"):
        monkeypatch.setenv("TC_INDEX_BINNED", mode)
        nrm = ctx.estimate_normals(d, 12)
        r = ctx.icp_point_to_plane_detailed(src, d, nrm, None, 3, None, 0.0, correspondences=True)
        runs.append((nrm.cpu().numpy(), r.transformation, np.float32(r.mse), np.asarray(r.correspondences)))
    monkeypatch.delenv("TC_INDEX_BINNED")
    for other in runs[1:]:
        for x, y, what in zip(runs[0], other, ("normals", "icp T", "icp mse", "icp pairs")):
            assert np.array_equal(np.asarray(x), np.asarray(y)), what
    monkeypatch.setenv("TC_RANK_QUADRATIC_MAX", "4096")
    nrm = ctx.estimate_normals(d, 12)
    monkeypatch.delenv("TC_RANK_QUADRATIC_MAX")
    dev = 1.0 - np.abs((nrm.cpu().numpy()[:, 3:] * runs[0][0][:, 3:]).sum(1))
    assert (dev > 1e-4).mean() < 1e-4          # (an equidistant pair here and there may be resolved the other way)


def test_a_cloud_that_does_not_start_on_a_16_byte_boundary(ctx):
    """The bounding-box pass reads four points as three 16-byte loads when the cloud starts on a 16-byte boundary, point by point
    otherwise (a view into a larger tensor: row 1 of an (n, 3) array starts 12 bytes in); both give the same box and the same
    sample boxes, hence the same grid and the same answers."""
    pts = synth.uniform_cloud(300_001, seed=12)
    pts[[5, 77_000, 299_999]] = [[-3.0, 0.5, 0.5], [0.5, 4.0, 0.5], [0.5, 0.5, -2.5]]       # (outliers: the robust box's sample logic runs)
    big = torch.from_numpy(np.concatenate([np.zeros((1, 3), np.float32), pts])).cuda()
    view = big[1:]                                       # contiguous, 12 bytes into its allocation
    assert view.data_ptr() % 16 != 0 and view.is_contiguous()
    own = torch.from_numpy(pts).cuda()
    assert own.data_ptr() % 16 == 0
    a, b = ctx.estimate_normals(view, 10), ctx.estimate_normals(own, 10)
    assert torch.equal(a, b)
    src = torch.from_numpy((pts[::4] + np.float32(0.002)).astype(np.float32)).cuda()
    ra = ctx.icp_point_to_plane_detailed(src, view, a, None, 4, None, 0.0, correspondences=True)
    rb = ctx.icp_point_to_plane_detailed(src, own, b, None, 4, None, 0.0, correspondences=True)
    assert np.array_equal(ra.transformation, rb.transformation) and ra.mse == rb.mse and np.array_equal(ra.correspondences, rb.correspondences)
    va, vb = ctx.voxel_grid_filter(view, 0.05), ctx.voxel_grid_filter(own, 0.05)
    assert torch.equal(va, vb)
