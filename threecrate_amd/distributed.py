"""Multi-GPU host logic, one process per GPU over torch.distributed.

Two ways the path shards:
  * independent scan pairs / batch ICP (BASELINE config [2]): rank r runs jobs r, r+W, ... on its
    own GPU; there is NO data-path collective (run_independent_jobs only gathers the results).
  * one big cloud (BASELINE config [3]): the SOURCE points are sharded, the target (+ normals +
    grid) is replicated; every iteration each rank reduces its shard to the packed normal
    equations (29 f64 words for point-to-plane) and ONE all-reduce(sum) of that 232-byte vector
    (RCCL over xGMI when the backend is "nccl") makes them global.  Every rank then applies the
    identical reduced buffer, so all ranks hold the same transform without a broadcast.

The product path is the C ABI: `Comm` wraps a tc_comm (an RCCL communicator bound to the context's stream, created
from a 128-byte id that this module distributes over the torch.distributed group -- any bootstrap would do) and
sharded_icp_point_to_plane / sharded_icp_detailed / sharded_estimate_normals call tc_sharded_*_device: the whole
loop, the per-iteration ncclAllReduce included, runs inside the library on the context's stream.  A Rust host binds
the same entry points.

The older step-wise building blocks (tc_icp_shard_* driven by sharded_icp_loop) remain for hosts that own the
collective themselves; the CPU tests (gloo, world_size 2) plug a checker backend into that loop, so that the
sharding / collective / control flow is exercised without a GPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from .api import ICPResult, Unsupported, _ERR, Error

SUMS = _lib.SUMS_STRIDE


def shard_range(n, rank, world):
    """Contiguous, balanced source range of `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HipShardBackend:
    """Per-rank shard session on one GPU (tc_icp_shard_*).  `sums` is a torch CUDA f64 tensor that
    the caller all-reduces in place."""

    def __init__(self, ctx, source_slice, target, target_normals, init, max_correspondence_distance, convergence_threshold):
        import torch
        self.torch, self.ctx, self.L = torch, ctx, _lib.load()
        self.s = source_slice.detach().to(torch.float32).contiguous().reshape(-1, 3)
        self.t = target.detach().to(torch.float32).contiguous().reshape(-1, 3)
        n = target_normals.detach().to(torch.float32).contiguous()
        self.stride = 6 if (n.dim() == 2 and n.shape[1] == 6) else 3
        self.n = n
        if (n.shape[0] if n.dim() == 2 else n.numel() // 3) != self.t.shape[0]:
            from .api import InvalidData
            raise InvalidData("target_normals length must equal the number of target points")
        i7 = np.ascontiguousarray(np.asarray(init, np.float32).reshape(7))
        md = ctx._max_dist(max_correspondence_distance)
        if md is None:
            ctx._reject_all(self.s.shape[0], self.t.shape[0], 1, n.shape[0] if n.dim() == 2 else n.numel() // 3)
        h = C.c_void_p()
        ctx._order(self.s.device)
        rc = self.L.tc_icp_shard_create(ctx._h, 1, self.s.data_ptr(), self.s.shape[0], self.t.data_ptr(), self.t.shape[0],
                                        n.data_ptr() + (12 if self.stride == 6 else 0), self.stride, i7.ctypes.data, md,
                                        convergence_threshold, C.byref(h))
        ctx._check(rc)
        self.h = h
        self.sums = torch.zeros(SUMS, dtype=torch.float64, device=self.s.device)
        # a context created on torch's current stream (GpuContext(device, stream=torch.cuda.current_stream().cuda_stream))
        # shares the stream the collective is enqueued on: the whole loop is stream ordered, no host waits
        self.same_stream = ctx.stream is not None and ctx.stream == torch.cuda.current_stream(self.s.device).cuda_stream

    def reduce(self):
        self.ctx._check(self.L.tc_icp_shard_reduce(self.h))
        self.ctx._check(self.L.tc_icp_shard_get_sums(self.h, self.sums.data_ptr()))
        if not self.same_stream:
            self.ctx._check(self.L.tc_synchronize(self.ctx._h))     # the collective runs on torch's stream
        return self.sums

    def apply(self, sums):
        if not self.same_stream:
            self.torch.cuda.current_stream().synchronize()
        self.ctx._check(self.L.tc_icp_shard_set_sums(self.h, sums.data_ptr()))
        self.ctx._check(self.L.tc_icp_shard_apply(self.h))

    def done(self):
        d = C.c_int(0)
        self.ctx._check(self.L.tc_icp_shard_done(self.h, C.byref(d)))
        return bool(d.value)

    def finish(self, max_iters):
        r = _lib.IcpResultC()
        rc = self.L.tc_icp_shard_finish(self.h, max_iters, C.byref(r))
        self.L.tc_icp_shard_destroy(self.h)
        self.h = None
        self.ctx._check(rc)
        return ICPResult(np.array(list(r.transformation), np.float32), float(r.mse), int(r.iterations), bool(r.converged))


def sharded_icp_loop(backend, max_iters, group=None, poll_every=4):
    """Drives `max_iters` iterations of a sharded ICP: reduce -> all_reduce(sum) -> apply.
    Returns the backend's result (identical on every rank)."""
    import torch.distributed as dist
    use = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    it = 0
    while it < max_iters:
        sums = backend.reduce()
        if use:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        backend.apply(sums)
        it += 1
        if it % poll_every == 0 and it < max_iters and backend.done():
            break
    return backend.finish(max_iters)


class Comm:
    """tc_comm of this rank (include/threecrate_hip.h, "communicator").

    Comm.from_group(ctx, group): over a torch.distributed process group.  With the "nccl" backend (RCCL) rank 0 draws
    a unique id (tc_comm_unique_id), the group broadcasts its 128 bytes and every rank calls tc_comm_create =
    ncclCommInitRank: the library then owns its own communicator and enqueues its collectives on the context's
    stream.  With a CPU backend ("gloo": several ranks sharing one GPU in the tests, or hosts without RCCL between
    the ranks) the collectives go through the tc_comm_create_host callback and this module runs them over the group.
    Comm.local(ctx): one rank, collectives are no-ops."""

    def __init__(self, ctx, handle, rank, size, keep=None):
        self.ctx, self._h, self.rank, self.size, self._keep = ctx, handle, rank, size, keep
        self._L = _lib.load()

    @classmethod
    def local(cls, ctx):
        L = _lib.load()
        h = C.c_void_p()
        ctx._check(L.tc_comm_create_local(ctx._h, C.byref(h)))
        return cls(ctx, h, 0, 1)

    @classmethod
    def rccl_single(cls, ctx):
        """a real one-rank RCCL communicator (tc_comm_unique_id + tc_comm_create): the code path of N ranks, exchange step
        included, on one GPU"""
        L = _lib.load()
        ident = (C.c_uint8 * _lib.TC_COMM_ID_BYTES)()
        if L.tc_comm_unique_id(ident) != _lib.TC_OK:
            raise Unsupported("RCCL is not available to libthreecrate_hip (tc_comm_unique_id)")
        h = C.c_void_p()
        ctx._check(L.tc_comm_create(ctx._h, 1, 0, ident, C.byref(h)))
        return cls(ctx, h, 0, 1)

    @classmethod
    def from_group(cls, ctx, group=None, force_host=False):
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return cls.local(ctx)
        L = _lib.load()
        rank, size = dist.get_rank(group), dist.get_world_size(group)
        h = C.c_void_p()
        backend = str(dist.get_backend(group))
        if "nccl" in backend and not force_host:
            ident = (C.c_uint8 * _lib.TC_COMM_ID_BYTES)()
            if rank == 0:
                rc = L.tc_comm_unique_id(ident)
                if rc != _lib.TC_OK:
                    raise Unsupported("RCCL is not available to libthreecrate_hip (tc_comm_unique_id)")
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = (C.c_uint8 * _lib.TC_COMM_ID_BYTES).from_buffer_copy(box[0])
            ctx._check(L.tc_comm_create(ctx._h, size, rank, ident, C.byref(h)))
            return cls(ctx, h, rank, size)

        def host_collective(_user, op, buf, count):
            try:
                if op == _lib.TC_COLL_SUM_F64:
                    a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(count,))
                    t = torch.from_numpy(a)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                elif op == _lib.TC_COLL_SUM_U32:
                    a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint32)), shape=(count,))
                    t = torch.from_numpy(a.astype(np.int64))            # gloo has no uint32 reduction
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                    a[:] = t.numpy().astype(np.uint32)
                elif op == _lib.TC_COLL_ALLGATHER_U8:
                    a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(size * count,))
                    mine = torch.from_numpy(a[rank * count:(rank + 1) * count].copy())
                    parts = [torch.empty(count, dtype=torch.uint8) for _ in range(size)]
                    dist.all_gather(parts, mine, group=group)
                    a[:] = torch.cat(parts).numpy()
                else:
                    return 2
                return 0
            except Exception:       # never raise through the C frame
                return 1

        cb = _lib.HOST_COLLECTIVE_FN(host_collective)
        ctx._check(L.tc_comm_create_host(ctx._h, size, rank, cb, None, C.byref(h)))
        return cls(ctx, h, rank, size, keep=cb)

    def close(self):
        if getattr(self, "_h", None):
            self._L.tc_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _icp_args(ctx, source, target, init, max_correspondence_distance):
    import torch
    from .api import IDENTITY
    s = source.detach().to(torch.float32).contiguous().reshape(-1, 3)
    t = target.detach().to(torch.float32).contiguous().reshape(-1, 3)
    i7 = np.ascontiguousarray(np.asarray(IDENTITY if init is None else init, np.float32).reshape(7))
    md = ctx._max_dist(max_correspondence_distance)
    return s, t, i7, md


def _shard_mode(comm, source_is_local_slice, shard):
    """TC_SHARD_LOCAL for a caller-partitioned source; otherwise the full source is replicated and the library takes this rank's
    part: by ORIGINAL-index range (TC_SHARD_INDEX: each rank orders ns / W points -- the default for more than one rank, the
    per-call set-up shrinks with the rank count) or as a spatially compact range of the source ordered by target cell
    (TC_SHARD_SPATIAL: every rank orders the whole source)."""
    if source_is_local_slice:
        return _lib.TC_SHARD_LOCAL
    if shard is None:
        shard = "index" if comm.size > 1 else "spatial"
    if shard not in ("index", "spatial"):
        raise ValueError("shard: 'index' or 'spatial'")
    return _lib.TC_SHARD_INDEX if shard == "index" else _lib.TC_SHARD_SPATIAL


def _finish(ctx, r, corr, correspondences):
    import torch
    if corr is not None and correspondences != "device":      # "device": the dense per-source index stays as written (int32 bits, -1 = none)
        corr = corr.to(torch.int64) & 0xFFFFFFFF
    return ctx._result(r, 0, corr, correspondences)


def sharded_icp_point_to_plane(ctx, source, target, target_normals, init=None, max_iters=50,
                               max_correspondence_distance=None, convergence_threshold=1e-6, group=None,
                               source_is_local_slice=False, comm=None, correspondences=False, shard=None):
    """icp_point_to_plane_detailed (registration.rs:508-602) over all ranks of `comm` (default: Comm.from_group(ctx,
    group)) through tc_sharded_icp_point_to_plane_device: `source` is the full source cloud, replicated (the library
    takes a spatially compact range of it per rank: TC_SHARD_SPATIAL), or -- source_is_local_slice=True -- this rank's
    own part (TC_SHARD_LOCAL); target / normals are replicated torch CUDA tensors.  One ncclAllReduce of the packed
    6x6 system per iteration on the context's stream; every rank returns the same ICPResult."""
    import torch
    own = comm is None
    comm = comm or Comm.from_group(ctx, group)
    try:
        s, t, i7, md = _icp_args(ctx, source, target, init, max_correspondence_distance)
        n = target_normals.detach().to(torch.float32).contiguous()
        stride = 6 if (n.dim() == 2 and n.shape[1] == 6) else 3
        nn = n.shape[0] if n.dim() == 2 else n.numel() // 3
        if md is None:
            ctx._reject_all(max(s.shape[0], 1 if source_is_local_slice else 0), t.shape[0], max_iters, nn)
        r = _lib.IcpResultC()
        corr = torch.empty(max(1, s.shape[0]), dtype=torch.int32, device=s.device) if correspondences else None
        r.corr_target = corr.data_ptr() if corr is not None else None
        ctx._order(t.device)
        ctx._check(_lib.load().tc_sharded_icp_point_to_plane_device(
            ctx._h, comm._h, _shard_mode(comm, source_is_local_slice, shard), s.data_ptr(), s.shape[0],
            t.data_ptr(), t.shape[0], n.data_ptr() + (12 if stride == 6 else 0), nn, stride, i7.ctypes.data, max_iters, md,
            convergence_threshold, C.byref(r)))
        return _finish(ctx, r, None if corr is None else corr[: s.shape[0]], correspondences)
    finally:
        if own:
            comm.close()


def sharded_icp_against_cloud(ctx, source, target_cloud, init=None, max_iters=50, max_correspondence_distance=None,
                              convergence_threshold=1e-6, point_to_plane=True, group=None, source_is_local_slice=False, comm=None,
                              correspondences=False, shard=None):
    """The same registration against a TARGET HANDLE (tc.Cloud, the same cloud on every rank): tc_cloud_sharded_icp.  Index, normals
    and inscribed-ball bounds of the target are built once per handle instead of once per call."""
    import torch
    own = comm is None
    comm = comm or Comm.from_group(ctx, group)
    try:
        s = source.detach().to(torch.float32).contiguous().reshape(-1, 3)
        from .api import IDENTITY
        i7 = np.ascontiguousarray(np.asarray(IDENTITY if init is None else init, np.float32).reshape(7))
        md = ctx._max_dist(max_correspondence_distance)
        if md is None:
            ctx._reject_all(max(s.shape[0], 1 if source_is_local_slice else 0), len(target_cloud), max_iters)
        r = _lib.IcpResultC()
        corr = torch.empty(max(1, s.shape[0]), dtype=torch.int32, device=s.device) if correspondences else None
        r.corr_target = corr.data_ptr() if corr is not None else None
        ctx._order(s.device)
        ctx._check(_lib.load().tc_cloud_sharded_icp(comm._h, _shard_mode(comm, source_is_local_slice, shard),
                                                    1 if point_to_plane else 0, s.data_ptr(), s.shape[0], target_cloud._h, i7.ctypes.data,
                                                    max_iters, md, convergence_threshold, C.byref(r)))
        return _finish(ctx, r, None if corr is None else corr[: s.shape[0]], correspondences)
    finally:
        if own:
            comm.close()


def sharded_icp_detailed(ctx, source, target, init=None, max_iters=50, max_correspondence_distance=None,
                         convergence_threshold=1e-6, group=None, source_is_local_slice=False, comm=None, correspondences=False, shard=None):
    """icp_detailed (registration.rs:258-370, point-to-point) over all ranks: tc_sharded_icp_detailed_device."""
    import torch
    own = comm is None
    comm = comm or Comm.from_group(ctx, group)
    try:
        s, t, i7, md = _icp_args(ctx, source, target, init, max_correspondence_distance)
        if md is None:
            ctx._reject_all(max(s.shape[0], 1 if source_is_local_slice else 0), t.shape[0], max_iters)
        r = _lib.IcpResultC()
        corr = torch.empty(max(1, s.shape[0]), dtype=torch.int32, device=s.device) if correspondences else None
        r.corr_target = corr.data_ptr() if corr is not None else None
        ctx._order(t.device)
        ctx._check(_lib.load().tc_sharded_icp_detailed_device(
            ctx._h, comm._h, _shard_mode(comm, source_is_local_slice, shard), s.data_ptr(), s.shape[0],
            t.data_ptr(), t.shape[0], i7.ctypes.data, max_iters, md, convergence_threshold, C.byref(r)))
        return _finish(ctx, r, None if corr is None else corr[: s.shape[0]], correspondences)
    finally:
        if own:
            comm.close()


def sharded_estimate_normals(ctx, cloud, k=10, config=None, group=None, comm=None):
    """estimate_normals(cloud, k) (normals.rs:238-241) of a device-resident cloud replicated on every rank:
    tc_sharded_estimate_normals_device (every rank computes its range of cell-sorted positions, ONE ncclAllGather of
    n x 24 bytes in total on the context's stream, a local kernel restores the input order)."""
    import torch
    from .api import NormalEstimationConfig
    own = comm is None
    comm = comm or Comm.from_group(ctx, group)
    try:
        c = ctx._cfg(config or NormalEstimationConfig(k_neighbors=k))
        x = cloud.detach().to(torch.float32).contiguous().reshape(-1, 3)
        out = torch.empty((x.shape[0], 6), dtype=torch.float32, device=x.device)
        ctx._order(x.device)
        ctx._check(_lib.load().tc_sharded_estimate_normals_device(ctx._h, comm._h, x.data_ptr(), x.shape[0], C.byref(c), out.data_ptr()))
        return out
    finally:
        if own:
            comm.close()


def sharded_estimate_normals_local(ctx, cloud, k=10, config=None, group=None, comm=None):
    """This rank's part of the normals of a replicated cloud WITHOUT the all-gather (tc_sharded_estimate_normals_local_device): the
    records (count, 6) of the cell-sorted positions [first, first + count) and the input index of each.  -> (records, orig_index,
    first).  For callers that keep the normals sharded (240 MB at 10 M points is what bounds the gathered call on 8 GPUs)."""
    import torch
    from .api import NormalEstimationConfig
    own = comm is None
    comm = comm or Comm.from_group(ctx, group)
    try:
        c = ctx._cfg(config or NormalEstimationConfig(k_neighbors=k))
        x = cloud.detach().to(torch.float32).contiguous().reshape(-1, 3)
        rows = -(-x.shape[0] // comm.size) if x.shape[0] else 0
        out = torch.empty((max(rows, 1), 6), dtype=torch.float32, device=x.device)
        idx = torch.empty(max(rows, 1), dtype=torch.int32, device=x.device)
        first, count = C.c_size_t(0), C.c_size_t(0)
        ctx._order(x.device)
        ctx._check(_lib.load().tc_sharded_estimate_normals_local_device(ctx._h, comm._h, x.data_ptr(), x.shape[0], C.byref(c), out.data_ptr(),
                                                                        idx.data_ptr(), C.byref(first), C.byref(count)))
        return out[: count.value], idx[: count.value], int(first.value)
    finally:
        if own:
            comm.close()


def stepwise_sharded_icp_point_to_plane(ctx, source, target, target_normals, init=None, max_iters=50,
                               max_correspondence_distance=None, convergence_threshold=1e-6, group=None,
                               source_is_local_slice=False):
    """The same registration driven step by step from the host (tc_icp_shard_* + torch.distributed.all_reduce): for
    hosts that own the collective themselves.  `source` is the full source cloud (every rank takes its index
    shard_range) or, with source_is_local_slice=True, already this rank's slice."""
    import torch.distributed as dist
    from .api import IDENTITY, InvalidData
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if source.shape[0] == 0 or target.shape[0] == 0:
        raise InvalidData("Source or target point cloud is empty")
    if max_iters == 0:
        raise InvalidData("Max iterations must be positive")
    sl = source
    if not source_is_local_slice:
        # every rank sees the same n: fewer source points than ranks would leave a rank with an empty slice, which the step-wise
        # session rejects -- on THAT rank only, while the others wait in the all-reduce.  Refuse on all ranks together
        # (the in-library path, sharded_icp_point_to_plane, accepts empty shards).
        if source.shape[0] < world:
            raise InvalidData(f"the step-wise sharded loop needs at least one source point per rank ({source.shape[0]} points, {world} ranks)")
        lo, hi = shard_range(source.shape[0], rank, world)
        sl = source[lo:hi]
    be = HipShardBackend(ctx, sl, target, target_normals, IDENTITY if init is None else init,
                         max_correspondence_distance, convergence_threshold)
    return sharded_icp_loop(be, max_iters, group)


class HipNormalsBackend:
    """One rank's part of the normals of a replicated cloud: slice(begin, end) -> (end - begin, 6) records of the
    cell-sorted positions [begin, end), unsort(all) -> (n, 6) in input order (tc_estimate_normals_slice_device /
    tc_normals_unsort_device)."""

    def __init__(self, ctx, cloud, config):
        self.ctx, self.cloud, self.config = ctx, cloud, config
        self.n = int(cloud.shape[0])

    def slice(self, begin, end):
        return self.ctx.estimate_normals_slice(self.cloud, self.config, begin, end)

    def unsort(self, sorted_all):
        return self.ctx.normals_unsort(sorted_all)


def sharded_normals(backend, group=None):
    """estimate_normals of ONE cloud over all ranks of `group` (BASELINE config [3]): the cloud and its
    index are replicated, rank r computes the records of the cell-sorted positions shard_range(n, r, W), ONE all-gather
    (n x 24 bytes in total; RCCL over xGMI with the "nccl" backend) hands every rank all of them, a local kernel puts
    them back into input order.  Every rank returns the full (n, 6) array."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    n = backend.n
    lo, hi = shard_range(n, rank, world)
    mine = backend.slice(lo, hi)
    if world == 1:
        return backend.unsort(mine)
    rows = -(-n // world)                                   # equal-sized parts for all_gather_into_tensor
    part = torch.zeros((rows, 6), dtype=mine.dtype, device=mine.device)
    part[: hi - lo] = mine
    gathered = torch.empty((world * rows, 6), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(gathered, part, group=group)
    pieces = []
    for r in range(world):
        a, b = shard_range(n, r, world)
        pieces.append(gathered[r * rows: r * rows + (b - a)])
    return backend.unsort(torch.cat(pieces))


def stepwise_sharded_estimate_normals(ctx, cloud, k=10, config=None, group=None):
    """sharded normals with the all-gather done by the host (torch.distributed) around tc_estimate_normals_slice_device /
    tc_normals_unsort_device; the product path is sharded_estimate_normals."""
    from .api import NormalEstimationConfig
    return sharded_normals(HipNormalsBackend(ctx, cloud, config or NormalEstimationConfig(k_neighbors=k)), group)


def run_independent_jobs(jobs, run_one, group=None):
    """Independent scan pairs: rank r runs jobs[r::world] with `run_one(job)`; results are
    gathered to every rank in job order.  No data-path collective."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    mine = [(i, run_one(jobs[i])) for i in range(rank, len(jobs), world)]
    if world == 1:
        return [r for _, r in mine]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine, group=group)
    out = [None] * len(jobs)
    for part in gathered:
        for i, r in part:
            out[i] = r
    return out
